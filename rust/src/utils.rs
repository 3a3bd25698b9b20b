//! The part of the reference's `utils.rs` a caller of the public API can reach: `imread` (utils.rs:110-117).
use crate::StackerError;
use opencv::imgcodecs;
use opencv::prelude::*;

/// utils.rs:110-117: `imgcodecs::imread` on a `Path`, `InvalidPathEncoding` for a path that is not UTF-8.
pub fn imread<P: AsRef<std::path::Path>>(path: P, imread_flags: i32) -> Result<Mat, StackerError> {
    let p = path.as_ref();
    let s = p
        .to_str()
        .ok_or_else(|| StackerError::InvalidPathEncoding(p.to_path_buf()))?;
    Ok(imgcodecs::imread(s, imread_flags)?)
}
