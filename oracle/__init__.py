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
        _LIB = C.CDLL(os.environ.get("STACKER_ORACLE_LIB") or build())     # override: the sanitizer build
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
              n_threads=0, scale_down_width=None):
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
                             int(gauss_filt_size), C.c_float(scale_down_width or 0.0), _p(out), _p(warps), _p(iters),
                             int(n_threads))
    if rc:
        raise RuntimeError("orc_ecc_match rc=%d" % rc)
    return out, warps, iters


# ---- keypoint path -------------------------------------------------------------------------------
def orb_level_sizes(w: int, h: int, nlevels: int = 8):
    ws = (C.c_int * nlevels)()
    hs = (C.c_int * nlevels)()
    sc = (C.c_float * nlevels)()
    lib().orc_orb_level_sizes(w, h, nlevels, ws, hs, sc)
    return list(ws), list(hs), list(sc)


def resize_linear_exact(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    s = np.ascontiguousarray(src, np.uint8)
    out = np.empty((dh, dw), np.uint8)
    lib().orc_resize_linear_exact(_p(s), s.shape[1], s.shape[0], _p(out), dw, dh)
    return out


def fast_score_map(img: np.ndarray, threshold: int = 20) -> np.ndarray:
    s = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(s)
    lib().orc_fast_score_map(_p(s), s.shape[1], s.shape[0], int(threshold), _p(out))
    return out


def gauss7_u8(img: np.ndarray) -> np.ndarray:
    s = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(s)
    lib().orc_gauss7_u8(_p(s), s.shape[1], s.shape[0], _p(out))
    return out


def orb_detect_and_compute(grey_img: np.ndarray, max_keypoints: int = 4096):
    """ORB::create_def().detect_and_compute (utils.rs:174-183): (keypoints [n,7] f32, descriptors [n,32] u8)."""
    g = np.ascontiguousarray(grey_img, np.uint8)
    kps = np.zeros((max_keypoints, 7), np.float32)
    des = np.zeros((max_keypoints, 32), np.uint8)
    n = C.c_int(0)
    lib().orc_orb_detect_and_compute(_p(g), g.shape[1], g.shape[0], max_keypoints, _p(kps), _p(des), C.byref(n))
    return kps[: n.value].copy(), des[: n.value].copy()


def bf_knn2_hamming(query: np.ndarray, train: np.ndarray) -> np.ndarray:
    q = np.ascontiguousarray(query, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(train, np.uint8).reshape(-1, 32)
    out = np.full((q.shape[0], 4), -1, np.int32)
    lib().orc_bf_knn2_hamming(_p(q), q.shape[0], _p(t), t.shape[0], _p(out))
    return out


def rng_sequence(n: int, state: int = 0xFFFFFFFFFFFFFFFF):
    st = C.c_uint64(state)
    f = lib().orc_rng_next
    f.restype = C.c_uint
    return [f(C.byref(st)) for _ in range(n)]


def find_homography(src_pts, dst_pts, method: int = 8, ransac_reproj_threshold: float = 3.0):
    """calib3d::find_homography(src, dst, method, thr) -> (H 3x3 f64 or None, inlier mask). Raises on bad input."""
    s = np.ascontiguousarray(src_pts, np.float32).reshape(-1, 2)
    d = np.ascontiguousarray(dst_pts, np.float32).reshape(-1, 2)
    H = np.zeros(9, np.float64)
    mask = np.zeros(max(s.shape[0], 1), np.uint8)
    found = C.c_int(0)
    rc = lib().orc_find_homography(_p(s), _p(d), s.shape[0], int(method), C.c_double(ransac_reproj_threshold), _p(H),
                                   _p(mask), C.byref(found))
    if rc:
        raise ValueError("orc_find_homography rc=%d" % rc)
    return (H.reshape(3, 3) if found.value else None), mask[: s.shape[0]]


def keypoint_match(frames, method: int = 8, ransac_reproj_threshold: float = 5.0, match_keep_ratio: float = 0.80,
                   match_ratio: float = 0.9, border_mode: int = BORDER_CONSTANT, border_value=(0, 0, 0, 0),
                   n_threads: int = 0, details: bool = False, scale_down_width=None):
    """keypoint_match_no_scale (lib.rs:146-353) with the documented drop semantics -> (dropped, image).
    Frames may differ in size: each is described at its own size (lib.rs:200-204) and warped into the first frame's
    (lib.rs:290-299); with scale_down_width every grey is scaled to ITS OWN smaller dimension = scale_down_width (lib.rs:429)."""
    frames = [np.ascontiguousarray(f, np.uint8) for f in frames]
    n = len(frames)
    h, w, _ = frames[0].shape
    ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
    out = np.empty((h, w, 3), np.float32)
    bv = np.asarray((list(border_value) + [0] * 4)[:4], np.float64)
    Hs = np.zeros((n, 3, 3), np.float64)
    status = np.zeros(n, np.int32)
    dropped = C.c_int(0)
    if len({f.shape for f in frames}) > 1:
        ws = np.array([f.shape[1] for f in frames], np.int32)
        hs = np.array([f.shape[0] for f in frames], np.int32)
        rc = lib().orc_keypoint_match_sized(ptrs, n, _p(ws), _p(hs), int(method), C.c_double(ransac_reproj_threshold),
                                            C.c_float(match_keep_ratio), C.c_float(match_ratio), int(border_mode), _p(bv),
                                            C.c_float(scale_down_width or 0.0), _p(out), C.byref(dropped), _p(Hs), _p(status), int(n_threads))
        if rc:
            raise RuntimeError("orc_keypoint_match_sized rc=%d" % rc)
        return (dropped.value, out, Hs, status) if details else (dropped.value, out)
    rc = lib().orc_keypoint_match(ptrs, n, w, h, int(method), C.c_double(ransac_reproj_threshold),
                                  C.c_float(match_keep_ratio), C.c_float(match_ratio), int(border_mode), _p(bv),
                                  C.c_float(scale_down_width or 0.0), _p(out), C.byref(dropped), _p(Hs), _p(status),
                                  int(n_threads))
    if rc:
        raise RuntimeError("orc_keypoint_match rc=%d" % rc)
    return (dropped.value, out, Hs, status) if details else (dropped.value, out)


def scaled_size(w: int, h: int, scale_down: float):
    nw, nh = C.c_int(0), C.c_int(0)
    lib().orc_scaled_size(w, h, C.c_float(scale_down), C.byref(nw), C.byref(nh))
    return nw.value, nh.value


def resize_area_u8(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    s = np.ascontiguousarray(src, np.uint8)
    out = np.empty((dh, dw), np.uint8)
    rc = lib().orc_resize_area_u8(_p(s), s.shape[1], s.shape[0], _p(out), dw, dh)
    if rc:
        raise ValueError("orc_resize_area_u8 rc=%d" % rc)
    return out


def resize_area_f32(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    s = np.ascontiguousarray(src, np.float32)
    out = np.empty((dh, dw), np.float32)
    rc = lib().orc_resize_area_f32(_p(s), s.shape[1], s.shape[0], _p(out), dw, dh)
    if rc:
        raise ValueError("orc_resize_area_f32 rc=%d" % rc)
    return out


def sharpness(grey, metric: int, ksize: int = 0) -> float:
    """The reference's sharpness metrics (lib.rs:1030-1166): 0 LAPM, 1 LAPV, 2 TENG(ksize), 3 GLVN."""
    g = np.ascontiguousarray(grey)
    out = C.c_double(0.0)
    rc = lib().orc_sharpness(_p(g), _depth(g), g.shape[1], g.shape[0], int(metric), int(ksize), C.byref(out))
    if rc:
        raise ValueError("orc_sharpness rc=%d" % rc)
    return out.value


def hybrid_match(frames, method: int = 8, ransac_reproj_threshold: float = 5.0, match_keep_ratio: float = 0.80,
                 match_ratio: float = 0.9, max_count=5000, epsilon=1e-5, gauss_filt_size=5, n_threads: int = 1):
    """BASELINE configs[4] (an extension beyond the reference), composed from the oracle's stages: ORB + RANSAC
    homography on the 8-bit grey ((grey16 + 128) / 257 for 16-bit frames) seeds findTransformECC (Homography) on float(grey);
    fold with alpha = 1/65535 (16-bit) or 1/255. Returns (image, warps [n,3,3] f32, iterations, seeds [n,3,3] f32)."""
    import math
    frames = [np.ascontiguousarray(f) for f in frames]
    n = len(frames)
    is16 = frames[0].dtype == np.uint16
    def greys_of(i):                                                    # (ORB input, ECC input) of frame i
        g = grey(frames[i])
        return (((g.astype(np.uint32) + 128) // 257).astype(np.uint8), g.astype(np.float32)) if is16 else (g, g)
    g8_0, gf_0 = greys_of(0)                                            # ECC input: float(grey16) / the 8-bit grey
    kp0, de0 = orb_detect_and_compute(g8_0)
    warps = np.zeros((n, 3, 3), np.float32); warps[0] = np.eye(3)
    seeds = np.zeros((n, 3, 3), np.float32); seeds[:] = np.eye(3)
    iters = np.zeros(n, np.int32)
    alpha = 1.0 / 65535.0 if is16 else 1.0 / 255.0
    def align(i):
        g8_i, gf_i = greys_of(i)
        kp, de = orb_detect_and_compute(g8_i)
        ms = []
        if len(kp0):
            knn = bf_knn2_hamming(de0, de)
            for q in range(len(kp0)):
                if knn[q, 0] < 0 or knn[q, 2] < 0:
                    continue
                if np.float32(knn[q, 1]) < np.float32(match_ratio) * np.float32(knn[q, 3]):    # lib.rs:224
                    ms.append((q, int(knn[q, 0]), float(knn[q, 1])))
            ms.sort(key=lambda m: m[2])                                                         # stable, lib.rs:233
            keep = int(math.floor(float(np.float32(len(ms)) * np.float32(match_keep_ratio)) + 0.5))   # round(), lib.rs:235
            ms = ms[:keep] if keep < len(ms) else ms
        if len(ms) >= 5:                                                                        # lib.rs:240
            src0 = np.array([[kp0[q][0], kp0[q][1]] for q, _, _ in ms], np.float32)
            dsti = np.array([[kp[t][0], kp[t][1]] for _, t, _ in ms], np.float32)
            H, _ = find_homography(dsti, src0, method, ransac_reproj_threshold)                 # frame i -> frame 0, lib.rs:267
            if H is not None and abs(np.linalg.det(H)) >= 1e-6 and abs(H[2, 2]) > 1e-12:
                seeds[i] = (H / H[2, 2]).astype(np.float32)
                seeds[i][2, 2] = 1.0
        rc, W, rho, its = find_transform_ecc(gf_i, gf_0, seeds[i], MOTION_HOMOGRAPHY, max_count, epsilon, gauss_filt_size)
        if rc:
            raise RuntimeError("hybrid_match: findTransformECC rc=%d on frame %d" % (rc, i))
        warps[i], iters[i] = W, its
        return W

    if n_threads <= 1:
        acc = warp_frame(frames[0], np.eye(3), alpha=alpha)
        for i in range(1, n):
            acc = warp_frame(frames[i], align(i).astype(np.float64), alpha=alpha, acc=acc)
        return scale(acc, n), warps, iters, seeds
    # frame-parallel like the reference's Rayon fold (lib.rs:188-335): every worker aligns and folds a contiguous run of
    # frames into its own accumulator (the C stages release the GIL), the partial sums are added in run order
    from concurrent.futures import ThreadPoolExecutor
    n_threads = max(1, min(n_threads, n))
    bounds = [1 + (n - 1) * k // n_threads for k in range(n_threads + 1)]
    runs = [range(bounds[k], bounds[k + 1]) for k in range(n_threads)]

    def fold_run(k):
        part = warp_frame(frames[0], np.eye(3), alpha=alpha) if k == 0 else None
        for i in runs[k]:
            W = align(i).astype(np.float64)
            part = warp_frame(frames[i], W, alpha=alpha, acc=part) if part is not None else warp_frame(frames[i], W, alpha=alpha)
        return part
    with ThreadPoolExecutor(n_threads) as ex:
        parts = [p for p in ex.map(fold_run, range(n_threads)) if p is not None]
    acc = parts[0]
    for p in parts[1:]:
        acc = acc + p
    return scale(acc, n), warps, iters, seeds
