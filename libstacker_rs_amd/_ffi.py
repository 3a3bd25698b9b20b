"""ctypes declarations for include/stacker.h (the C ABI of libstacker_amd.so).

This is plumbing for tests and bench.py; a Rust `libstacker` shim binds the same symbols
(INTEGRATION.md). There is no fallback: if the shared library is missing or fails to load,
importing it raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STACKER_AMD_LIB") or os.path.join(_HERE, "libstacker_amd.so")   # override: A/B builds

c_status = C.c_int


class KeypointParams(C.Structure):
    _fields_ = [("method", C.c_int32), ("ransac_reproj_threshold", C.c_double),
                ("match_keep_ratio", C.c_float), ("match_ratio", C.c_float),
                ("border_mode", C.c_int32), ("border_value", C.c_double * 4)]


class EccParams(C.Structure):
    _fields_ = [("motion_type", C.c_int32), ("has_max_count", C.c_int32), ("max_count", C.c_int32),
                ("has_epsilon", C.c_int32), ("epsilon", C.c_double), ("gauss_filt_size", C.c_int32)]


class Frames(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_void_p)), ("n", C.c_int32), ("width", C.c_int32),
                ("height", C.c_int32), ("channels", C.c_int32), ("depth", C.c_int32),
                ("location", C.c_int32), ("row_stride_bytes", C.c_size_t)]


class FrameGeometry(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("row_stride_bytes", C.c_size_t)]


class ImageF32(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
                ("channels", C.c_int32), ("location", C.c_int32), ("row_stride_bytes", C.c_size_t)]


class FrameStats(C.Structure):
    _fields_ = [("status", C.c_int32), ("iterations", C.c_int32), ("rho", C.c_double),
                ("n_keypoints", C.c_int32), ("n_matches", C.c_int32), ("n_inliers", C.c_int32),
                ("reserved", C.c_int32), ("warp", C.c_double * 9)]


class Timing(C.Structure):
    _fields_ = [("prep_ms", C.c_double), ("align_ms", C.c_double), ("warp_ms", C.c_double),
                ("finalize_ms", C.c_double), ("ecc_iter_launches", C.c_int64),
                ("ecc_slot_iterations", C.c_int64), ("warp_launches", C.c_int64),
                ("warp_frames", C.c_int64), ("ecc_iter_ms", C.c_double), ("ecc_iter_timed", C.c_int64),
                ("h2d_ms", C.c_double), ("h2d_bytes", C.c_int64), ("fast_ms", C.c_double), ("fast_launches", C.c_int64),
                ("fast_pixels", C.c_int64), ("ecc_ring_fallbacks", C.c_int64)]


# every symbol include/stacker.h declares, with its signature
SIGNATURES = {
    "stk_version": (C.c_char_p, []),
    "stk_create": (c_status, [C.c_int32, C.POINTER(C.c_void_p)]),
    "stk_create_multi": (c_status, [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]),
    "stk_shard_moving_frames": (c_status, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "stk_rccl_selftest": (c_status, [C.c_void_p, C.c_int64]),
    "stk_host_alloc": (c_status, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "stk_host_free": (None, [C.c_void_p]),
    "stk_destroy": (None, [C.c_void_p]),
    "stk_last_error": (C.c_char_p, [C.c_void_p]),
    "stk_set_stream": (c_status, [C.c_void_p, C.c_void_p]),
    "stk_get_timing": (c_status, [C.c_void_p, C.POINTER(Timing)]),
    "stk_set_option": (c_status, [C.c_void_p, C.c_char_p, C.c_int64]),
    "stk_keypoint_match": (c_status, [C.c_void_p, C.POINTER(Frames), C.POINTER(KeypointParams), C.c_float,
                                      C.POINTER(ImageF32), C.POINTER(C.c_int32), C.POINTER(FrameStats)]),
    "stk_keypoint_match_mixed": (c_status, [C.c_void_p, C.POINTER(Frames), C.POINTER(FrameGeometry), C.POINTER(KeypointParams),
                                            C.c_float, C.POINTER(ImageF32), C.POINTER(C.c_int32), C.POINTER(FrameStats)]),
    "stk_ecc_match": (c_status, [C.c_void_p, C.POINTER(Frames), C.POINTER(EccParams), C.c_float,
                                 C.POINTER(ImageF32), C.POINTER(FrameStats)]),
    "stk_ecc_match_shard": (c_status, [C.c_void_p, C.POINTER(Frames), C.POINTER(EccParams), C.c_float, C.c_int32,
                                       C.POINTER(ImageF32), C.POINTER(C.c_int32), C.POINTER(FrameStats)]),
    "stk_keypoint_match_shard": (c_status, [C.c_void_p, C.POINTER(Frames), C.POINTER(KeypointParams), C.c_float,
                                            C.c_int32, C.POINTER(ImageF32), C.POINTER(C.c_int32),
                                            C.POINTER(C.c_int32), C.POINTER(FrameStats)]),
    "stk_finalize_mean": (c_status, [C.c_void_p, C.POINTER(ImageF32), C.c_int64, C.POINTER(ImageF32)]),
    "stk_grey": (c_status, [C.c_void_p, C.POINTER(Frames), C.c_void_p]),
    "stk_convert_f32": (c_status, [C.c_void_p, C.POINTER(Frames), C.c_double, C.c_void_p]),
    "stk_hybrid_match": (c_status, [C.c_void_p, C.POINTER(Frames), C.POINTER(KeypointParams), C.POINTER(EccParams),
                                    C.POINTER(ImageF32), C.POINTER(FrameStats)]),
    "stk_hybrid_match_shard": (c_status, [C.c_void_p, C.POINTER(Frames), C.POINTER(KeypointParams), C.POINTER(EccParams), C.c_int32,
                                          C.POINTER(ImageF32), C.POINTER(C.c_int32), C.POINTER(FrameStats)]),
    "stk_imread": (c_status, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                              C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "stk_keypoint_match_files": (c_status, [C.c_void_p, C.POINTER(C.c_char_p), C.c_int32, C.POINTER(KeypointParams), C.c_float,
                                            C.POINTER(ImageF32), C.POINTER(C.c_int32), C.POINTER(FrameStats)]),
    "stk_ecc_match_files": (c_status, [C.c_void_p, C.POINTER(C.c_char_p), C.c_int32, C.POINTER(EccParams), C.c_float,
                                       C.POINTER(ImageF32), C.POINTER(FrameStats)]),
    "stk_hybrid_match_files": (c_status, [C.c_void_p, C.POINTER(C.c_char_p), C.c_int32, C.POINTER(KeypointParams),
                                          C.POINTER(EccParams), C.POINTER(ImageF32), C.POINTER(FrameStats)]),
    "stk_sharpness": (c_status, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                 C.POINTER(C.c_double)]),
    "stk_grey_blur_f32": (c_status, [C.c_void_p, C.POINTER(Frames), C.c_int32, C.c_void_p]),
    "stk_gaussian_blur_f32": (c_status, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_void_p]),
    "stk_find_transform_ecc": (c_status, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_int32, C.POINTER(EccParams), C.c_void_p, C.POINTER(C.c_double),
                                          C.POINTER(C.c_int32)]),
    "stk_warp_accumulate": (c_status, [C.c_void_p, C.POINTER(Frames), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_double, C.c_int32, C.POINTER(ImageF32)]),
    "stk_scale_image_grey": (c_status, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                                        C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "stk_scale_image_grey_f32": (c_status, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                                            C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "stk_orb_detect_and_compute": (c_status, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                              C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "stk_bf_knn2_hamming": (c_status, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "stk_find_homography": (c_status, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double,
                                       C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
}

_lib = None


def load() -> C.CDLL:
    """Load libstacker_amd.so (built by __graft_entry__.build() / csrc/Makefile). Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C libstacker_rs_amd/csrc)")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64
        # and a second copy (from /opt/rocm) in the same process leaves whichever initialises last
        # without a GPU. Importing torch first makes the loader resolve our NEEDED libamdhip64.so.7 to
        # the copy torch already mapped, so tensors, streams and our kernels share one runtime.
        # (Without torch installed the library binds to /opt/rocm through its RUNPATH.)
        try:
            import torch  # noqa: F401
        except ImportError:  # pragma: no cover - torch is part of the target image
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
