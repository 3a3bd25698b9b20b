"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY. Only tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg may import this package; the product (libstacker_rs_amd)
never does. PARITY UNPINNED: see oracle/oracle_common.h for what the oracle is
anchored on (no OpenCV and no reference golden vectors exist in this container).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MOTION_TRANSLATION, MOTION_EUCLIDEAN, MOTION_AFFINE, MOTION_HOMOGRAPHY = 0, 1, 2, 3
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101 = 0, 1, 2, 3, 4


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h"))]
    stale = not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_ecc_prepare_input.restype = C.c_void_p
    return _LIB


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _depth(a: np.ndarray) -> int:
    return {np.dtype(np.uint8): 8, np.dtype(np.uint16): 16, np.dtype(np.float32): 32}[a.dtype]


def grey(bgr: np.ndarray) -> np.ndarray:
    bgr = np.ascontiguousarray(bgr)
    h, w, _ = bgr.shape
    out = np.empty((h, w), bgr.dtype)
    rc = lib().orc_grey(_p(bgr), _depth(bgr), w, h, C.c_size_t(0), _p(out))
    assert rc == 0
    return out


def convert_f32(img: np.ndarray, alpha: float = 1.0 / 255.0) -> np.ndarray:
    img = np.ascontiguousarray(img)
    out = np.empty(img.shape, np.float32)
    rc = lib().orc_convert_f32(_p(img), _depth(img), C.c_size_t(img.size), C.c_double(alpha), _p(out))
    assert rc == 0
    return out


def gaussian_blur_f32(grey_img: np.ndarray, ksize: int) -> np.ndarray:
    g = np.ascontiguousarray(grey_img)
    h, w = g.shape
    out = np.empty((h, w), np.float32)
    rc = lib().orc_gaussian_blur_f32(_p(g), _depth(g), w, h, ksize, _p(out))
    assert rc == 0
    return out


def gradients(img: np.ndarray):
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape
    gx = np.empty_like(img)
    gy = np.empty_like(img)
    lib().orc_gradients(_p(img), w, h, _p(gx), _p(gy))
    return gx, gy


def warp_frame(src: np.ndarray, M, *, is_affine=False, border_mode=BORDER_CONSTANT,
               border_value=(0, 0, 0, 0), alpha=1.0 / 255.0, subpixel_bits=0, acc=None) -> np.ndarray:
    """warp_perspective/warp_affine(convert(src, alpha), M) [+ acc]; M is the forward matrix."""
    src = np.ascontiguousarray(src)
    if src.ndim == 2:
        src = src[:, :, None]
    h, w, cn = src.shape
    Md = np.ascontiguousarray(np.asarray(M, np.float64).reshape(-1))
    if Md.size == 6:
        Md = np.concatenate([Md, [0, 0, 1]])
    bv = np.asarray(list(border_value) + [0] * 4, np.float64)[:4].copy()
    accumulate = acc is not None
    dst = acc if accumulate else np.empty((h, w, cn), np.float32)
    assert dst.dtype == np.float32 and dst.flags.c_contiguous
    rc = lib().orc_warp_frame(_p(src), _depth(src), w, h, cn, C.c_size_t(0), _p(Md), int(is_affine),
                              int(border_mode), _p(bv), C.c_double(alpha), int(subpixel_bits), _p(dst),
                              int(accumulate))
    if rc:
        raise ValueError("orc_warp_frame rc=%d" % rc)
    return dst


def scale(img: np.ndarray, divisor: float) -> np.ndarray:
    img = np.ascontiguousarray(img, np.float32)
    out = np.empty_like(img)
    lib().orc_scale(_p(img), C.c_size_t(img.size), C.c_double(divisor), _p(out))
    return out


def find_transform_ecc(templ: np.ndarray, inp: np.ndarray, warp: np.ndarray, motion: int,
                       max_count=None, epsilon=None, gauss_filt_size=5):
    """Returns (rc, warp 3x3 f32, rho, iterations). rc: 0 ok, 1 NaN, 2 no-convergence, 3 bad args."""
    t = np.ascontiguousarray(templ)
    i = np.ascontiguousarray(inp)
    wm = np.eye(3, dtype=np.float32)
    wv = np.asarray(warp, np.float32)
    wm[: wv.shape[0], :] = wv
    rho = C.c_double(0)
    its = C.c_int(0)
    rc = lib().orc_find_transform_ecc(_p(t), t.shape[1], t.shape[0], _p(i), i.shape[1], i.shape[0],
                                      _depth(t), _p(wm), int(motion), int(max_count is not None),
                                      int(max_count or 0), int(epsilon is not None),
                                      C.c_double(epsilon or 0.0), int(gauss_filt_size), C.byref(rho),
                                      C.byref(its))
    return rc, wm, rho.value, its.value


def ecc_match(frames, motion=MOTION_HOMOGRAPHY, max_count=5000, epsilon=1e-5, gauss_filt_size=5,
              n_threads=0):
    """ecc_match_no_scaling (lib.rs:719-847) on decoded BGR frames. Returns (image, warps, iters)."""
    frames = [np.ascontiguousarray(f) for f in frames]
    n = len(frames)
    h, w, _ = frames[0].shape
    ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
    out = np.empty((h, w, 3), np.float32)
    warps = np.zeros((n, 3, 3), np.float32)
    iters = np.zeros(n, np.int32)
    rc = lib().orc_ecc_match(ptrs, n, w, h, _depth(frames[0]), int(motion), int(max_count is not None),
                             int(max_count or 0), int(epsilon is not None), C.c_double(epsilon or 0.0),
                             int(gauss_filt_size), _p(out), _p(warps), _p(iters), int(n_threads))
    if rc:
        raise RuntimeError("orc_ecc_match rc=%d" % rc)
    return out, warps, iters
