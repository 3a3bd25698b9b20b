//! The engine behind the public functions: `include/stacker.h` through `amd_ffi.rs` (generated).
//!
//! * ONE context per process, made on first use over `STACKER_AMD_GPUS` GPUs of the node (default 1; `stk_create_multi`
//!   shards the moving frames over them and reduces the accumulators with RCCL inside the library) and kept behind a
//!   `Mutex`: a `stk_ctx` takes one call at a time, and its HBM workspaces, pinned staging and communicators are grow-only.
//! * Files go to `stk_*_match_files`: the engine decodes them on a pool of host threads straight into page-locked memory
//!   while it already uploads and aligns the frames that have arrived (what the reference does inside its Rayon fold,
//!   lib.rs:200, 756). A file type its decoders do not read comes back as `STK_NOT_IMPLEMENTED`: the whole stack is then
//!   decoded by OpenCV's `imread(IMREAD_UNCHANGED)` exactly as `read_grey_and_f32` does (utils.rs:132) and handed over
//!   as host frames.
use crate::amd_ffi::*;
use crate::{utils, EccMatchParameters, KeyPointMatchParameters, StackerError};
use opencv::core;
use opencv::imgcodecs;
use opencv::prelude::*;
use std::ffi::{c_void, CStr, CString};
use std::path::PathBuf;
use std::sync::{Mutex, MutexGuard, OnceLock};

struct Ctx(*mut stk_ctx);
// the pointer is only ever used under the Mutex below
unsafe impl Send for Ctx {}

fn n_gpus() -> i32 {
    std::env::var("STACKER_AMD_GPUS").ok().and_then(|v| v.parse().ok()).filter(|&n| n >= 1).unwrap_or(1)
}

fn shared_ctx() -> Result<MutexGuard<'static, Ctx>, StackerError> {
    static CTX: OnceLock<Result<Mutex<Ctx>, String>> = OnceLock::new();
    let cell = CTX.get_or_init(|| {
        let ids: Vec<i32> = (0..n_gpus()).collect();
        let mut ctx: *mut stk_ctx = std::ptr::null_mut();
        let st = unsafe { stk_create_multi(ids.len() as i32, ids.as_ptr(), &mut ctx) };
        if st != STK_OK || ctx.is_null() {
            Err(format!("stk_create_multi({} GPUs) failed with status {}", ids.len(), st))
        } else {
            Ok(Mutex::new(Ctx(ctx)))
        }
    });
    match cell {
        Ok(m) => Ok(m.lock().unwrap_or_else(|poisoned| poisoned.into_inner())),
        Err(msg) => Err(StackerError::ProcessingError(msg.clone())),
    }
}

/// stk_status -> StackerError (include/stacker.h:33-42 lists the correspondence).
fn to_err(ctx: *mut stk_ctx, st: stk_status) -> StackerError {
    let msg = unsafe {
        let p = stk_last_error(ctx);
        if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    };
    match st {
        STK_NOT_ENOUGH_FILES => StackerError::NotEnoughFiles,
        STK_INVALID_PARAMS => StackerError::InvalidParams(msg),
        STK_PROCESSING_ERROR | STK_HIP_ERROR => StackerError::ProcessingError(msg),
        STK_BACKEND_ERROR => StackerError::OpenCvError(opencv::Error::new(core::StsError, msg)),
        STK_IO_ERROR => StackerError::IoError(std::io::Error::other(msg)),
        _ => StackerError::NotImplemented,
    }
}

fn c_paths(files: &[PathBuf]) -> Result<Vec<CString>, StackerError> {
    files
        .iter()
        .map(|p| {
            let s = p.to_str().ok_or_else(|| StackerError::InvalidPathEncoding(p.clone()))?;
            CString::new(s).map_err(|_| StackerError::InvalidPathEncoding(p.clone()))
        })
        .collect()
}

fn kp_params(p: &KeyPointMatchParameters) -> stk_keypoint_params {
    let bv = p.border_value;
    stk_keypoint_params {
        method: p.method, // handed through untouched, as lib.rs:51,271 hand it to findHomography
        ransac_reproj_threshold: p.ransac_reproj_threshold,
        match_keep_ratio: p.match_keep_ratio,
        match_ratio: p.match_ratio,
        border_mode: p.border_mode,
        border_value: [bv[0], bv[1], bv[2], bv[3]],
    }
}

fn ecc_params(p: &EccMatchParameters) -> stk_ecc_params {
    stk_ecc_params {
        motion_type: p.motion_type as i32,
        has_max_count: p.max_count.is_some() as i32,
        max_count: p.max_count.unwrap_or(0),
        has_epsilon: p.epsilon.is_some() as i32,
        epsilon: p.epsilon.unwrap_or(0.0),
        gauss_filt_size: p.gauss_filt_size,
    }
}

/// Geometry of the stack = geometry of its first file (the output has the reference frame's size, lib.rs:166, 290-299).
fn first_geometry(ctx: *mut stk_ctx, first: &CString) -> Result<(i32, i32, i32), stk_status> {
    let (mut w, mut h, mut cn, mut depth) = (0i32, 0i32, 0i32, 0i32);
    let st = unsafe { stk_imread(ctx, first.as_ptr(), std::ptr::null_mut(), 0, &mut w, &mut h, &mut cn, &mut depth) };
    if st == STK_OK { Ok((w, h, cn)) } else { Err(st) }
}

/// The result Mat: CV_32FC3, or CV_32FC4 for a stack of BGRA frames (the reference's `&acc + &warped` keeps whatever
/// channel count imread(IMREAD_UNCHANGED) delivered, utils.rs:132).
fn new_output(w: i32, h: i32, cn: i32) -> Result<(Mat, stk_image_f32), StackerError> {
    let cn = if cn == 4 { 4 } else { 3 };
    let mut out = unsafe { Mat::new_rows_cols(h, w, if cn == 4 { core::CV_32FC4 } else { core::CV_32FC3 })? };
    let img = stk_image_f32 {
        data: out.data_mut() as *mut f32,
        width: w,
        height: h,
        channels: cn,
        location: STK_HOST,
        row_stride_bytes: 0,
    };
    Ok((out, img))
}

/// The frames of a stack decoded by OpenCV (the fallback for file types the engine does not read). `stk_frames` carries ONE
/// geometry for the whole stack, so `frames()` insists on it (the ECC path: the reference fails on frames of differing
/// size anyway, cv::add at lib.rs:809); the keypoint path hands every Mat's own geometry to `stk_keypoint_match_mixed`,
/// which treats such a stack as the reference does (ORB per frame size, every frame warped into the first frame's size:
/// lib.rs:166, 200-204, 290-299). Type (channels, depth) must match in both.
struct DecodedStack {
    mats: Vec<Mat>,
    ptrs: Vec<*const c_void>,
    geometry: Vec<stk_frame_geometry>,
    uniform: bool,
}

impl DecodedStack {
    fn read(files: &[PathBuf]) -> Result<Self, StackerError> {
        let mats: Vec<Mat> = files
            .iter()
            .map(|p| utils::imread(p, imgcodecs::IMREAD_UNCHANGED))
            .collect::<Result<_, _>>()?;
        let first = mats.first().ok_or(StackerError::NotEnoughFiles)?;
        let mut uniform = true;
        for (m, p) in mats.iter().zip(files) {
            if m.empty() {
                return Err(StackerError::OpenCvError(opencv::Error::new(
                    core::StsError,
                    format!("{}: not an image", p.display()),
                )));
            }
            if m.typ() != first.typ() {
                return Err(StackerError::OpenCvError(opencv::Error::new(
                    core::StsError,
                    format!("{}: type differs from the first frame's", p.display()),
                )));
            }
            if !m.is_continuous() {
                return Err(StackerError::ProcessingError("imread returned a non-continuous Mat".into()));
            }
            uniform &= m.size()? == first.size()?;
        }
        let ptrs = mats.iter().map(|m| m.data() as *const c_void).collect();
        let geometry = mats
            .iter()
            .map(|m| stk_frame_geometry { width: m.cols(), height: m.rows(), row_stride_bytes: 0 })
            .collect();
        Ok(Self { mats, ptrs, geometry, uniform })
    }

    fn depth(&self) -> Result<i32, StackerError> {
        Ok(match self.mats[0].depth() {
            core::CV_8U => STK_DEPTH_U8,
            core::CV_16U => STK_DEPTH_U16,
            core::CV_32F => STK_DEPTH_F32,
            _ => return Err(StackerError::NotImplemented),
        })
    }

    /// One geometry for the whole stack, or the error the reference ends in on such a stack (cv::add, lib.rs:809).
    fn frames(&self) -> Result<stk_frames, StackerError> {
        if !self.uniform {
            return Err(StackerError::OpenCvError(opencv::Error::new(
                core::StsUnmatchedSizes,
                "the frames differ in size".to_string(),
            )));
        }
        self.frames_of_first()
    }

    /// `stk_frames` with the FIRST frame's geometry (what `stk_keypoint_match_mixed` reads channels / depth / location from).
    fn frames_of_first(&self) -> Result<stk_frames, StackerError> {
        let m = &self.mats[0];
        Ok(stk_frames {
            data: self.ptrs.as_ptr(),
            n: self.ptrs.len() as i32,
            width: m.cols(),
            height: m.rows(),
            channels: m.channels(),
            depth: self.depth()?,
            location: STK_HOST,
            row_stride_bytes: 0,
        })
    }
}

pub(crate) fn ecc_match(
    files: &[PathBuf],
    params: EccMatchParameters,
    scale_down_width: Option<f32>,
) -> Result<Mat, StackerError> {
    if files.is_empty() {
        return Err(StackerError::NotEnoughFiles); // lib.rs:725
    }
    let paths = c_paths(files)?;
    let guard = shared_ctx()?;
    let ctx = guard.0;
    let p = ecc_params(&params);
    let sdw = scale_down_width.unwrap_or(0.0);
    match first_geometry(ctx, &paths[0]) {
        Ok((w, h, cn)) => {
            let raw: Vec<*const std::os::raw::c_char> = paths.iter().map(|c| c.as_ptr()).collect();
            let (out, mut img) = new_output(w, h, cn)?;
            let st = unsafe {
                stk_ecc_match_files(ctx, raw.as_ptr(), raw.len() as i32, &p, sdw, &mut img, std::ptr::null_mut())
            };
            match st {
                STK_OK => return Ok(out),
                STK_NOT_IMPLEMENTED => {} // a later file of a type the engine does not decode: OpenCV reads the stack
                _ => return Err(to_err(ctx, st)),
            }
        }
        Err(STK_NOT_IMPLEMENTED) => {}
        Err(st) => return Err(to_err(ctx, st)),
    }
    let stack = DecodedStack::read(files)?;
    let frames = stack.frames()?;
    let (out, mut img) = new_output(frames.width, frames.height, frames.channels)?;
    let st = unsafe { stk_ecc_match(ctx, &frames, &p, sdw, &mut img, std::ptr::null_mut()) };
    if st == STK_OK { Ok(out) } else { Err(to_err(ctx, st)) }
}

pub(crate) fn keypoint_match(
    files: &[PathBuf],
    params: KeyPointMatchParameters,
    scale_down_width: Option<f32>,
) -> Result<(i32, Mat), StackerError> {
    if files.is_empty() {
        return Err(StackerError::NotEnoughFiles); // lib.rs:155
    }
    let paths = c_paths(files)?;
    let guard = shared_ctx()?;
    let ctx = guard.0;
    let p = kp_params(&params);
    let sdw = scale_down_width.unwrap_or(0.0);
    let mut dropped: i32 = 0;
    // Frames whose homography cannot be estimated are skipped and COUNTED (the documented contract, lib.rs:98); every
    // frame dropped -> InvalidParams, as lib.rs:324. The tuple is (dropped, image / (n - dropped)), lib.rs:339-345.
    match first_geometry(ctx, &paths[0]) {
        Ok((w, h, cn)) => {
            let raw: Vec<*const std::os::raw::c_char> = paths.iter().map(|c| c.as_ptr()).collect();
            let (out, mut img) = new_output(w, h, cn)?;
            let st = unsafe {
                stk_keypoint_match_files(ctx, raw.as_ptr(), raw.len() as i32, &p, sdw, &mut img, &mut dropped, std::ptr::null_mut())
            };
            match st {
                STK_OK => return Ok((dropped, out)),
                STK_NOT_IMPLEMENTED => {}
                _ => return Err(to_err(ctx, st)),
            }
        }
        Err(STK_NOT_IMPLEMENTED) => {}
        Err(st) => return Err(to_err(ctx, st)),
    }
    let stack = DecodedStack::read(files)?;
    let frames = stack.frames_of_first()?;
    let (out, mut img) = new_output(frames.width, frames.height, frames.channels)?;
    let st = if stack.uniform {
        unsafe { stk_keypoint_match(ctx, &frames, &p, sdw, &mut img, &mut dropped, std::ptr::null_mut()) }
    } else {
        // frames of differing size, with or without scale_down_width (lib.rs:355-601 scales every grey by ITS OWN factor)
        unsafe {
            stk_keypoint_match_mixed(ctx, &frames, stack.geometry.as_ptr(), &p, sdw, &mut img, &mut dropped, std::ptr::null_mut())
        }
    };
    if st == STK_OK { Ok((dropped, out)) } else { Err(to_err(ctx, st)) }
}

/// lib.rs:1030-1166 on a single-channel CV_8U or CV_32F Mat (what examples/main.rs:40-47 feeds them).
pub(crate) fn sharpness(src: &Mat, metric: i32, ksize: i32) -> Result<f64, StackerError> {
    if src.channels() != 1 {
        return Err(StackerError::OpenCvError(opencv::Error::new(core::StsError, "sharpness: single-channel image expected".to_string())));
    }
    let depth = match src.depth() {
        core::CV_8U => STK_DEPTH_U8,
        core::CV_32F => STK_DEPTH_F32,
        _ => return Err(StackerError::NotImplemented),
    };
    // the engine reads a tightly packed image: a ROI or padded Mat is copied first
    let owned;
    let m = if src.is_continuous() { src } else { owned = src.try_clone()?; &owned };
    let guard = shared_ctx()?;
    let ctx = guard.0;
    let mut out = 0f64;
    let st = unsafe { stk_sharpness(ctx, m.data() as *const c_void, depth, m.cols(), m.rows(), STK_HOST, metric, ksize, &mut out) };
    if st == STK_OK { Ok(out) } else { Err(to_err(ctx, st)) }
}
