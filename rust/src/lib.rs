//! libstacker's public API (reference: `/root/reference/src/lib.rs`) on the MI355X-native engine.
//!
//! Same names, argument meaning and error variants as the reference: `keypoint_match` (lib.rs:129-137), `ecc_match`
//! (lib.rs:702-710), the four `sharpness_*` metrics (lib.rs:1032-1166), `KeyPointMatchParameters` (lib.rs:48-73),
//! `EccMatchParameters` / `MotionType` (lib.rs:603-623), `StackerError` (lib.rs:27-45), `prelude` (lib.rs:1168-1173).
//! Everything behind the signatures is `amd.rs`: one shared engine context, the `*_files` entry points of
//! `include/stacker.h`, OpenCV's `imread` only for file types the engine does not decode itself.
pub use opencv;
use opencv::core;
use opencv::prelude::*;
use std::path::PathBuf;
use thiserror::Error;

#[cfg(feature = "amd")]
mod amd;
#[cfg(feature = "amd")]
#[allow(clippy::all)]
pub mod amd_ffi;
pub mod utils;

/// lib.rs:27-45 (the PoisonError variant of the reference wraps a Rayon-side lock the engine does not have; it is kept so
/// that `match` arms written against the reference still compile).
#[derive(Error, Debug)]
pub enum StackerError {
    #[error(transparent)]
    OpenCvError(#[from] opencv::Error),
    #[error("Not enough files")]
    NotEnoughFiles,
    #[error("Not implemented")]
    NotImplemented,
    #[error(transparent)]
    IoError(#[from] std::io::Error),
    #[error(transparent)]
    PoisonError(#[from] std::sync::PoisonError<core::MatExprResult<core::MatExpr>>),
    #[error("Invalid path encoding {0}")]
    InvalidPathEncoding(PathBuf),
    #[error("Invalid parameter(s) {0}")]
    InvalidParams(String),
    #[error("Internal error {0}")]
    ProcessingError(String),
}

/// lib.rs:48-73.
#[derive(Debug, Clone, Copy)]
pub struct KeyPointMatchParameters {
    /// `calib3d::find_homography` method: 0 least squares, 4 LMEDS, 8 RANSAC (16 RHO: `NotImplemented`).
    pub method: i32,
    pub ransac_reproj_threshold: f64,
    pub match_keep_ratio: f32,
    pub match_ratio: f32,
    pub border_mode: i32,
    pub border_value: core::Scalar,
}

/// utils.rs:250-261.
impl Default for KeyPointMatchParameters {
    fn default() -> Self {
        Self {
            method: 8, // opencv::calib3d::RANSAC
            ransac_reproj_threshold: 3.0,
            match_keep_ratio: 0.75,
            match_ratio: 0.8,
            border_mode: core::BORDER_CONSTANT,
            border_value: core::Scalar::default(),
        }
    }
}

/// lib.rs:603-609 (the discriminants are OpenCV's MOTION_* values, which are the engine's STK_MOTION_* too).
#[derive(Debug, Copy, Clone, PartialEq, Eq)]
pub enum MotionType {
    Homography = 3,
    Affine = 2,
    Euclidean = 1,
    Translation = 0,
}

/// lib.rs:611-623.
#[derive(Debug, Copy, Clone)]
pub struct EccMatchParameters {
    pub motion_type: MotionType,
    pub max_count: Option<i32>,
    pub epsilon: Option<f64>,
    pub gauss_filt_size: i32,
}

/// lib.rs:129-137: aligns every frame to the first by ORB + brute-force Hamming + `findHomography`, warps and averages.
/// Returns (number of frames that could not be matched and were left out, averaged CV_32FC3 image).
#[cfg(feature = "amd")]
pub fn keypoint_match<I, P>(
    files: I,
    params: KeyPointMatchParameters,
    scale_down_width: Option<f32>,
) -> Result<(i32, Mat), StackerError>
where
    I: IntoIterator<Item = P>,
    P: AsRef<std::path::Path>,
{
    let files: Vec<PathBuf> = files.into_iter().map(|p| p.as_ref().to_path_buf()).collect();
    amd::keypoint_match(&files, params, scale_down_width)
}

/// lib.rs:702-710: aligns every frame to the first by `findTransformECC`, warps and averages. Returns the CV_32FC3 image.
#[cfg(feature = "amd")]
pub fn ecc_match<I, P>(
    files: I,
    params: EccMatchParameters,
    scale_down_width: Option<f32>,
) -> Result<Mat, StackerError>
where
    I: IntoIterator<Item = P>,
    P: AsRef<std::path::Path>,
{
    let files: Vec<PathBuf> = files.into_iter().map(|p| p.as_ref().to_path_buf()).collect();
    amd::ecc_match(&files, params, scale_down_width)
}

/// lib.rs:1032 — 'LAPM' (Nayar89). Single-channel 8-bit or f32 image.
#[cfg(feature = "amd")]
pub fn sharpness_modified_laplacian(src_mat: &Mat) -> Result<f64, StackerError> {
    amd::sharpness(src_mat, amd_ffi::STK_SHARPNESS_LAPM, 0)
}

/// lib.rs:1074 — 'LAPV' (Pech2000).
#[cfg(feature = "amd")]
pub fn sharpness_variance_of_laplacian(src_mat: &Mat) -> Result<f64, StackerError> {
    amd::sharpness(src_mat, amd_ffi::STK_SHARPNESS_LAPV, 0)
}

/// lib.rs:1101 — 'TENG' (Krotkov86); `k_size` must be 1, 3, 5 or 7 (`InvalidParams` otherwise, lib.rs:1105).
#[cfg(feature = "amd")]
pub fn sharpness_tenengrad(src_grey_mat: &Mat, k_size: i32) -> Result<f64, StackerError> {
    amd::sharpness(src_grey_mat, amd_ffi::STK_SHARPNESS_TENG, k_size)
}

/// lib.rs:1151 — 'GLVN' (Santos97).
#[cfg(feature = "amd")]
pub fn sharpness_normalized_gray_level_variance(src_mat: &Mat) -> Result<f64, StackerError> {
    amd::sharpness(src_mat, amd_ffi::STK_SHARPNESS_GLVN, 0)
}

/// lib.rs:1168-1173.
pub mod prelude {
    pub use super::{
        EccMatchParameters, KeyPointMatchParameters, MotionType, StackerError, ecc_match,
        keypoint_match,
    };
}
