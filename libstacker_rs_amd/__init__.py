"""libstacker_rs_amd — MI355X-native drop-in for the hot path of eadf/libstacker.rs.

`keypoint_match()` / `ecc_match()` (reference: src/lib.rs:129-144, 702-717) implemented as
hand-written HIP kernels for gfx950 behind a C ABI (include/stacker.h, libstacker_amd.so).
The directory is also reachable under the dotted name `libstacker.rs_amd/` (a symlink): a dot
cannot appear in an importable Python package name.

Importing the package is cheap and never touches the GPU; the shared library is loaded on first
use (api.Stacker / _ffi.load) and there is no CPU fallback.
"""
from .api import (BORDER_CONSTANT, BORDER_REFLECT, BORDER_REFLECT_101, BORDER_REPLICATE, BORDER_WRAP,  # noqa: F401
                  LEAST_SQUARES, LMEDS, RANSAC, RHO, EccMatchParameters, HipError, InvalidParams, IoError,
                  KeyPointMatchParameters, MotionType, NotEnoughFiles, NotImplementedYet, OpenCvError,
                  ProcessingError, Stacker, StackerError, default_stacker, ecc_match, keypoint_match)

__all__ = ["keypoint_match", "ecc_match", "KeyPointMatchParameters", "EccMatchParameters", "MotionType",
           "StackerError", "Stacker"]
