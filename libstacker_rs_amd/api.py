"""Host-side mirror of libstacker's public API over the C ABI (include/stacker.h).

Same names, argument meaning and error behaviour as the reference (src/lib.rs):
``keypoint_match(files, KeyPointMatchParameters, scale_down_width) -> (dropped, image)``
(lib.rs:129-144) and ``ecc_match(files, EccMatchParameters, scale_down_width) -> image``
(lib.rs:702-717); `StackerError` variants follow lib.rs:27-45. "files" are decoded frames here
(numpy arrays on the host or torch tensors already resident in HBM, BGR interleaved like
imread(IMREAD_UNCHANGED) returns, utils.rs:132); the first one is the reference frame.

All arithmetic happens in libstacker_amd.so (hand-written HIP for gfx950). This module only
marshals pointers; it has no CPU fallback and raises if the library is missing.
"""
from __future__ import annotations

import ctypes as C
import os
import enum
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _ffi

# OpenCV constants the reference passes through
RANSAC, LMEDS, RHO, LEAST_SQUARES = 8, 4, 16, 0
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101, BORDER_TRANSPARENT = range(6)
HOST, DEVICE = 0, 1


class StackerError(Exception):
    """Base of the reference's StackerError enum (lib.rs:27-45)."""


class NotEnoughFiles(StackerError):
    pass


class InvalidParams(StackerError):
    pass


class ProcessingError(StackerError):
    pass


class OpenCvError(StackerError):
    """The reference's OpenCvError: failures the OpenCV backend would have raised (ECC no-convergence ...)."""


class IoError(StackerError):
    pass


class HipError(StackerError):
    pass


class NotImplementedYet(StackerError):
    pass


_STATUS_EXC = {1: NotEnoughFiles, 2: InvalidParams, 3: ProcessingError, 4: OpenCvError, 5: IoError, 6: HipError,
               7: NotImplementedYet}


class MotionType(enum.IntEnum):   # lib.rs:603-609 (= OpenCV MOTION_*)
    Translation = 0
    Euclidean = 1
    Affine = 2
    Homography = 3


@dataclass
class KeyPointMatchParameters:    # lib.rs:48-73, defaults utils.rs:250-261
    method: int = RANSAC
    ransac_reproj_threshold: float = 3.0
    match_keep_ratio: float = 0.75
    match_ratio: float = 0.8
    border_mode: int = BORDER_CONSTANT
    border_value: Sequence[float] = field(default_factory=lambda: (0.0, 0.0, 0.0, 0.0))

    def _c(self) -> _ffi.KeypointParams:
        bv = (list(self.border_value) + [0.0] * 4)[:4]
        return _ffi.KeypointParams(int(self.method), float(self.ransac_reproj_threshold), float(self.match_keep_ratio),
                                   float(self.match_ratio), int(self.border_mode), (C.c_double * 4)(*bv))


@dataclass
class EccMatchParameters:         # lib.rs:611-623 (no Default in the reference)
    motion_type: MotionType
    max_count: Optional[int]
    epsilon: Optional[float]
    gauss_filt_size: int

    def _c(self) -> _ffi.EccParams:
        return _ffi.EccParams(int(self.motion_type), int(self.max_count is not None), int(self.max_count or 0),
                              int(self.epsilon is not None), float(self.epsilon or 0.0), int(self.gauss_filt_size))

    def term_criteria(self):
        """TermCriteria{typ, max_count, epsilon} as utils.rs:159-170 builds it (COUNT=1, EPS=2)."""
        typ = (1 if self.max_count is not None else 0) | (2 if self.epsilon is not None else 0)
        return typ, (self.max_count or 0), (self.epsilon or 0.0)


# ---------------------------------------------------------------------------------------------
def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


_DEPTH = {"uint8": 8, "uint16": 16, "float32": 32}


class _Marshalled:
    """Pointers + geometry of a frame stack, keeping the owners alive."""

    def __init__(self, frames):
        if _is_torch(frames) and frames.dim() == 4 and frames.is_contiguous() and frames.shape[0] > 0 \
                and str(frames.dtype).replace("torch.", "") in _DEPTH:
            # one tensor holding the whole stack: the pointers are an arithmetic progression — no per-frame Python work
            # (unbinding 256 frames cost ~2 ms per call, 3 % of a 57 ms stack)
            self.keep = [frames]
            self.n = int(frames.shape[0])
            self.location = DEVICE if frames.is_cuda else HOST
            self.torch_device = frames.device if frames.is_cuda else None
            self.devices = {frames.device} if frames.is_cuda else set()
            self.h, self.w, self.c = (int(v) for v in frames.shape[1:])
            self.depth = _DEPTH[str(frames.dtype).replace("torch.", "")]
            step = self.h * self.w * self.c * (self.depth // 8)
            addr = (frames.data_ptr() + np.arange(self.n, dtype=np.uint64) * np.uint64(step)).astype(np.uint64)
            self.ptr_arr = (C.c_void_p * self.n).from_buffer_copy(addr.tobytes())
            self.c_frames = _ffi.Frames(C.cast(self.ptr_arr, C.POINTER(C.c_void_p)), self.n, self.w, self.h, self.c,
                                        self.depth, self.location, 0)
            return
        if _is_torch(frames) and frames.dim() == 4:
            frames = list(frames.unbind(0))
        elif isinstance(frames, np.ndarray) and frames.ndim == 4:
            frames = list(frames)
        frames = list(frames)
        self.keep = []
        self.n = len(frames)
        self.location = HOST
        self.torch_device = None
        self.devices = set()
        ptrs = []
        geo = None
        for f in frames:
            if _is_torch(f):
                if f.is_cuda:
                    self.location = DEVICE
                    if self.torch_device is None:
                        self.torch_device = f.device         # outputs go where the reference frame lives
                    self.devices.add(f.device)
                    t = f.contiguous()
                    self.keep.append(t)
                    ptrs.append(t.data_ptr())
                    g = (tuple(t.shape), str(t.dtype).replace("torch.", ""))
                else:
                    a = np.ascontiguousarray(f.numpy())
                    self.keep.append(a)
                    ptrs.append(a.ctypes.data)
                    g = (a.shape, str(a.dtype))
            else:
                a = np.ascontiguousarray(f)
                self.keep.append(a)
                ptrs.append(a.ctypes.data)
                g = (a.shape, str(a.dtype))
            if len(g[0]) == 2:
                g = ((g[0][0], g[0][1], 1), g[1])
            if geo is None:
                geo = g
            elif g != geo:
                raise InvalidParams("all frames must share size, channels and dtype")
        if self.n:
            (self.h, self.w, self.c), dt = geo
            if dt not in _DEPTH:
                raise InvalidParams(f"unsupported pixel type {dt}")
            self.depth = _DEPTH[dt]
        else:
            self.h = self.w = self.c = 0
            self.depth = 8
        self.ptr_arr = (C.c_void_p * max(self.n, 1))(*ptrs)
        self.c_frames = _ffi.Frames(C.cast(self.ptr_arr, C.POINTER(C.c_void_p)), self.n, self.w, self.h, self.c,
                                    self.depth, self.location, 0)


class Stacker:
    """One engine context (stk_ctx): one GPU, or — `devices=[...]` — several GPUs of the node behind ONE context, the
    moving frames sharded over them and the accumulators reduced with RCCL inside the library (stk_create_multi).
    Not thread-safe: one call at a time per instance."""

    def __init__(self, device: int = 0, devices: Optional[Sequence[int]] = None):
        self._lib = _ffi.load()
        h = C.c_void_p()
        if devices is not None:
            ids = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            st = self._lib.stk_create_multi(len(devices), ids, C.byref(h))
            device = int(devices[0]) if len(devices) else 0
        else:
            st = self._lib.stk_create(int(device), C.byref(h))
        if st != 0:
            raise HipError(f"stk_create(device={device}, devices={devices}) failed with status {st} (no usable GPU?)")
        self._h = h
        self.device = int(device)
        self.devices = [int(d) for d in devices] if devices is not None else [int(device)]
        self._bound_stream = None          # cuda_stream handle of the torch stream this context runs on, if any

    def _marshal(self, frames) -> "_Marshalled":
        """Pointers of a frame stack; device tensors must be complete before the engine's stream reads them. If this
        context runs ON torch's current stream of that device (use_torch_stream) stream order already guarantees it;
        otherwise — an unbound context, a `with torch.cuda.stream(...)` block, another device — the producer streams
        are drained first."""
        m = _Marshalled(frames)
        if m.location == DEVICE:
            import torch
            for d in m.devices:
                cur = torch.cuda.current_stream(d)
                if not (self._bound_stream is not None and d.index == self.device and cur.cuda_stream == self._bound_stream):
                    cur.synchronize()
        return m

    def _tensor_ready(self, t):
        import torch
        cur = torch.cuda.current_stream(t.device)
        if not (self._bound_stream is not None and t.device.index == self.device and cur.cuda_stream == self._bound_stream):
            cur.synchronize()

    def rccl_selftest(self, count: int = 1 << 20):
        """Load RCCL, build a communicator over this context's devices and check a sum-reduce (stk_rccl_selftest)."""
        self._check(self._lib.stk_rccl_selftest(self._h, int(count)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.stk_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers -------------------------------------------------------------------------------
    def _check(self, st: int):
        if st != 0:
            msg = self._lib.stk_last_error(self._h)
            raise _STATUS_EXC.get(st, StackerError)((msg or b"").decode("utf-8", "replace"))

    def set_option(self, name: str, value: int):
        self._check(self._lib.stk_set_option(self._h, name.encode(), int(value)))

    def set_stream(self, stream_ptr: int | None):
        self._check(self._lib.stk_set_stream(self._h, C.c_void_p(stream_ptr or 0)))
        self._bound_stream = None

    def use_torch_stream(self):
        """Run this context on torch's CURRENT stream of its device (captured now; call again after switching streams)."""
        import torch
        h = torch.cuda.current_stream(self.device).cuda_stream
        self.set_stream(h)
        self._bound_stream = h

    def timing(self) -> dict:
        t = _ffi.Timing()
        self._check(self._lib.stk_get_timing(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in _ffi.Timing._fields_}

    def _out_image(self, m: _Marshalled):
        """f32 HxWxC output placed where the inputs live."""
        if m.location == DEVICE:
            import torch
            out = torch.empty((m.h, m.w, m.c), dtype=torch.float32, device=m.torch_device)
            img = _ffi.ImageF32(out.data_ptr(), m.w, m.h, m.c, DEVICE, 0)
        else:
            out = np.empty((m.h, m.w, m.c), np.float32)
            img = _ffi.ImageF32(out.ctypes.data, m.w, m.h, m.c, HOST, 0)
        return out, img

    @staticmethod
    def _stats_list(stats, n):
        return [dict(status=s.status, iterations=s.iterations, rho=s.rho, n_keypoints=s.n_keypoints,
                     n_matches=s.n_matches, n_inliers=s.n_inliers, warp=np.array(list(s.warp)).reshape(3, 3))
                for s in stats[:n]]

    # -- whole-stack API (the reference's two entry points) ---------------------------------------
    def ecc_match(self, files, params: EccMatchParameters, scale_down_width: Optional[float] = None,
                  return_stats: bool = False):
        if isinstance(files, (list, tuple)) and len({tuple(f.shape[:2]) for f in files}) > 1:
            raise OpenCvError("the frames differ in size: the reference fails on such a stack in cv::add (lib.rs:809)")
        m = self._marshal(files)
        if m.n == 0:
            raise NotEnoughFiles("Not enough files")
        out, img = self._out_image(m)
        stats = (_ffi.FrameStats * m.n)()
        p = params._c()
        st = self._lib.stk_ecc_match(self._h, C.byref(m.c_frames), C.byref(p), float(scale_down_width or 0.0),
                                     C.byref(img), stats)
        self._check(st)
        return (out, self._stats_list(stats, m.n)) if return_stats else out

    def _keypoint_match_mixed(self, frames, params: KeyPointMatchParameters, return_stats: bool, scale_down_width=None):
        """Frames of differing size (host arrays, HxWx3 u8): ORB at each frame's own size, every frame warped into the FIRST
        frame's size, as the reference does (lib.rs:166, 200-204, 290-299) — stk_keypoint_match_mixed."""
        arrs = [np.ascontiguousarray(f.cpu().numpy() if _is_torch(f) else f) for f in frames]
        cn = arrs[0].shape[2] if arrs[0].ndim == 3 else 0
        for a in arrs:
            if a.ndim != 3 or a.shape[2] not in (3, 4) or a.shape[2] != cn or a.dtype != np.uint8:
                raise OpenCvError("ORB: 8-bit BGR / BGRA frames of one type expected")
        n = len(arrs)
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        geo = (_ffi.FrameGeometry * n)(*[_ffi.FrameGeometry(a.shape[1], a.shape[0], 0) for a in arrs])
        h0, w0 = arrs[0].shape[:2]
        fr = _ffi.Frames(C.cast(ptrs, C.POINTER(C.c_void_p)), n, w0, h0, cn, 8, HOST, 0)
        out = np.empty((h0, w0, cn), np.float32)
        img = _ffi.ImageF32(out.ctypes.data, w0, h0, cn, HOST, 0)
        stats = (_ffi.FrameStats * n)()
        dropped = C.c_int32(0)
        p = params._c()
        self._check(self._lib.stk_keypoint_match_mixed(self._h, C.byref(fr), geo, C.byref(p), float(scale_down_width or 0.0), C.byref(img), C.byref(dropped), stats))
        return (dropped.value, out, self._stats_list(stats, n)) if return_stats else (dropped.value, out)

    def keypoint_match(self, files, params: KeyPointMatchParameters, scale_down_width: Optional[float] = None,
                       return_stats: bool = False):
        if isinstance(files, (list, tuple)) and len({tuple(f.shape) for f in files}) > 1:
            return self._keypoint_match_mixed(files, params, return_stats, scale_down_width)
        m = self._marshal(files)
        if m.n == 0:
            raise NotEnoughFiles("Not enough files")
        out, img = self._out_image(m)
        stats = (_ffi.FrameStats * m.n)()
        dropped = C.c_int32(0)
        p = params._c()
        st = self._lib.stk_keypoint_match(self._h, C.byref(m.c_frames), C.byref(p), float(scale_down_width or 0.0),
                                          C.byref(img), C.byref(dropped), stats)
        self._check(st)
        return (dropped.value, out, self._stats_list(stats, m.n)) if return_stats else (dropped.value, out)

    # -- shard-level (one process per GPU; frames[0] = reference frame) ------------------------------
    def ecc_match_shard(self, files, params: EccMatchParameters, add_reference: bool, sum_out,
                        scale_down_width: Optional[float] = None, return_stats: bool = True):
        """Un-normalised f32 sum of this rank's aligned frames into `sum_out` (cuda tensor HxWx3)."""
        m = self._marshal(files)
        if m.n == 0:
            raise NotEnoughFiles("Not enough files")
        img = _ffi.ImageF32(sum_out.data_ptr(), m.w, m.h, 3, DEVICE, 0)
        added = C.c_int32(0)
        stats = (_ffi.FrameStats * m.n)()
        p = params._c()
        st = self._lib.stk_ecc_match_shard(self._h, C.byref(m.c_frames), C.byref(p), float(scale_down_width or 0.0),
                                           int(bool(add_reference)), C.byref(img), C.byref(added), stats)
        self._check(st)
        return added.value, (self._stats_list(stats, m.n) if return_stats else None)

    def keypoint_match_shard(self, files, params: KeyPointMatchParameters, add_reference: bool, sum_out,
                             scale_down_width: Optional[float] = None, return_stats: bool = True):
        m = self._marshal(files)
        if m.n == 0:
            raise NotEnoughFiles("Not enough files")
        img = _ffi.ImageF32(sum_out.data_ptr(), m.w, m.h, 3, DEVICE, 0)
        added, dropped = C.c_int32(0), C.c_int32(0)
        stats = (_ffi.FrameStats * m.n)()
        p = params._c()
        st = self._lib.stk_keypoint_match_shard(self._h, C.byref(m.c_frames), C.byref(p),
                                                float(scale_down_width or 0.0), int(bool(add_reference)),
                                                C.byref(img), C.byref(added), C.byref(dropped), stats)
        self._check(st)
        return added.value, dropped.value, (self._stats_list(stats, m.n) if return_stats else None)

    def finalize_mean(self, sum_img, n_frames: int, out=None):
        """img / n  (lib.rs:339-345, 836-839) on a cuda tensor; in place when out is None."""
        self._tensor_ready(sum_img)
        h, w, c = sum_img.shape
        out = sum_img if out is None else out
        a = _ffi.ImageF32(sum_img.data_ptr(), w, h, c, DEVICE, 0)
        b = _ffi.ImageF32(out.data_ptr(), w, h, c, DEVICE, 0)
        self._check(self._lib.stk_finalize_mean(self._h, C.byref(a), int(n_frames), C.byref(b)))
        return out

    # -- stage-level (parity tests) ------------------------------------------------------------------
    def grey(self, frame):
        m = self._marshal([frame])
        if m.location == DEVICE:
            import torch
            out = torch.empty((m.h, m.w), dtype=m.keep[0].dtype, device=m.torch_device)
            ptr = out.data_ptr()
        else:
            out = np.empty((m.h, m.w), m.keep[0].dtype)
            ptr = out.ctypes.data
        self._check(self._lib.stk_grey(self._h, C.byref(m.c_frames), C.c_void_p(ptr)))
        return out

    def convert_f32(self, frame, alpha: float = 1.0 / 255.0):
        m = self._marshal([frame])
        if m.location == DEVICE:
            import torch
            out = torch.empty(tuple(m.keep[0].shape), dtype=torch.float32, device=m.torch_device)
            ptr = out.data_ptr()
        else:
            out = np.empty(m.keep[0].shape, np.float32)
            ptr = out.ctypes.data
        self._check(self._lib.stk_convert_f32(self._h, C.byref(m.c_frames), float(alpha), C.c_void_p(ptr)))
        return out

    # -- BASELINE configs[4] (extension): ORB-seeded ECC on 8- or 16-bit stacks --------------------------------
    def hybrid_match(self, files, kp_params: KeyPointMatchParameters, ecc_params: EccMatchParameters, return_stats=False):
        """ORB + RANSAC homography as the initial warp of findTransformECC; 16-bit frames folded with alpha 1/65535."""
        m = self._marshal(files)
        if m.n == 0:
            raise NotEnoughFiles("Not enough files")
        out, img = self._out_image(m)
        stats = (_ffi.FrameStats * m.n)()
        kp, ep = kp_params._c(), ecc_params._c()
        self._check(self._lib.stk_hybrid_match(self._h, C.byref(m.c_frames), C.byref(kp), C.byref(ep), C.byref(img), stats))
        return (out, self._stats_list(stats, m.n)) if return_stats else out

    def hybrid_match_shard(self, files, kp_params: KeyPointMatchParameters, ecc_params: EccMatchParameters,
                           add_reference: bool, sum_out, return_stats: bool = True):
        m = self._marshal(files)
        if m.n == 0:
            raise NotEnoughFiles("Not enough files")
        img = _ffi.ImageF32(sum_out.data_ptr(), m.w, m.h, 3, DEVICE, 0)
        added = C.c_int32(0)
        stats = (_ffi.FrameStats * m.n)()
        kp, ep = kp_params._c(), ecc_params._c()
        self._check(self._lib.stk_hybrid_match_shard(self._h, C.byref(m.c_frames), C.byref(kp), C.byref(ep),
                                                     int(bool(add_reference)), C.byref(img), C.byref(added), stats))
        return added.value, (self._stats_list(stats, m.n) if return_stats else None)

    # -- file front-end: the reference's entry points take paths ------------------------------
    def imread(self, path):
        """imgcodecs::imread(path, IMREAD_UNCHANGED) for binary PNM, 8-bit PNG and 8/16-bit TIFF: HxW or HxWx3 (BGR) array."""
        w, h, c, d = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
        bp = os.fsencode(path)
        self._check(self._lib.stk_imread(self._h, bp, None, 0, C.byref(w), C.byref(h), C.byref(c), C.byref(d)))
        out = np.empty((h.value, w.value, c.value), np.uint8 if d.value == 8 else np.uint16)
        self._check(self._lib.stk_imread(self._h, bp, C.c_void_p(out.ctypes.data), out.nbytes, None, None, None, None))
        return out[..., 0] if c.value == 1 else out

    def _paths(self, files):
        enc = [os.fsencode(f) for f in files]
        arr = (C.c_char_p * max(len(enc), 1))(*enc)
        return enc, arr

    def _file_geometry(self, files):
        w, h, c, d = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._check(self._lib.stk_imread(self._h, os.fsencode(files[0]), None, 0, C.byref(w), C.byref(h), C.byref(c), C.byref(d)))
        return w.value, h.value, (4 if c.value == 4 else 3)       # (an RGBA file gives a CV_32FC4 stack, like the reference)

    def keypoint_match_files(self, files, params: KeyPointMatchParameters, scale_down_width: Optional[float] = None):
        """keypoint_match(files, params, scale_down_width) -> (dropped, HxWx3 f32)   lib.rs:129-137"""
        files = list(files)
        if not files:
            raise NotEnoughFiles("Not enough files")
        w, h, cn = self._file_geometry(files)
        out = np.empty((h, w, cn), np.float32)
        img = _ffi.ImageF32(out.ctypes.data, w, h, cn, HOST, 0)
        keep, arr = self._paths(files)
        dropped = C.c_int32(0)
        p = params._c()
        self._check(self._lib.stk_keypoint_match_files(self._h, arr, len(files), C.byref(p), float(scale_down_width or 0.0),
                                                       C.byref(img), C.byref(dropped), None))
        return dropped.value, out

    def ecc_match_files(self, files, params: EccMatchParameters, scale_down_width: Optional[float] = None):
        """ecc_match(files, params, scale_down_width) -> HxWx3 f32   lib.rs:702-710"""
        files = list(files)
        if not files:
            raise NotEnoughFiles("Not enough files")
        w, h, cn = self._file_geometry(files)
        out = np.empty((h, w, cn), np.float32)
        img = _ffi.ImageF32(out.ctypes.data, w, h, cn, HOST, 0)
        keep, arr = self._paths(files)
        p = params._c()
        self._check(self._lib.stk_ecc_match_files(self._h, arr, len(files), C.byref(p), float(scale_down_width or 0.0),
                                                  C.byref(img), None))
        return out

    def hybrid_match_files(self, files, kp_params: KeyPointMatchParameters, ecc_params: EccMatchParameters):
        """stk_hybrid_match on a list of paths (e.g. a 16-bit TIFF stack)."""
        files = list(files)
        if not files:
            raise NotEnoughFiles("Not enough files")
        w, h, cn = self._file_geometry(files)
        out = np.empty((h, w, cn), np.float32)
        img = _ffi.ImageF32(out.ctypes.data, w, h, cn, HOST, 0)
        keep, arr = self._paths(files)
        kp, ep = kp_params._c(), ecc_params._c()
        self._check(self._lib.stk_hybrid_match_files(self._h, arr, len(files), C.byref(kp), C.byref(ep), C.byref(img), None))
        return out

    def _sharpness(self, grey, metric: int, ksize: int = 0) -> float:
        g = np.ascontiguousarray(grey)
        if g.ndim != 2 or str(g.dtype) not in ("uint8", "float32"):
            raise InvalidParams("sharpness metrics take a single-channel uint8 or float32 image")
        h, w = g.shape
        out = C.c_double(0.0)
        self._check(self._lib.stk_sharpness(self._h, C.c_void_p(g.ctypes.data), _DEPTH[str(g.dtype)], w, h, HOST,
                                            int(metric), int(ksize), C.byref(out)))
        return out.value

    def sharpness_modified_laplacian(self, grey) -> float:
        """lib.rs:1032-1071 (LAPM, Nayar89)."""
        return self._sharpness(grey, 0)

    def sharpness_variance_of_laplacian(self, grey) -> float:
        """lib.rs:1075-1091 (LAPV, Pech2000)."""
        return self._sharpness(grey, 1)

    def sharpness_tenengrad(self, grey, k_size: int) -> float:
        """lib.rs:1103-1147 (TENG, Krotkov86); k_size must be 1, 3, 5 or 7."""
        return self._sharpness(grey, 2, k_size)

    def sharpness_normalized_gray_level_variance(self, grey) -> float:
        """lib.rs:1151-1166 (GLVN, Santos97)."""
        return self._sharpness(grey, 3)

    def grey_blur_f32(self, frame, ksize: int):
        """cvt_color(BGR2GRAY) + findTransformECC's GaussianBlur of one frame, fused (the per-frame ECC preparation)."""
        m = self._marshal([frame])
        if m.location == DEVICE:
            import torch
            out = torch.empty((m.h, m.w), dtype=torch.float32, device=m.torch_device)
            ptr = out.data_ptr()
        else:
            out = np.empty((m.h, m.w), np.float32)
            ptr = out.ctypes.data
        self._check(self._lib.stk_grey_blur_f32(self._h, C.byref(m.c_frames), int(ksize), C.c_void_p(ptr)))
        return out

    def gaussian_blur_f32(self, grey, ksize: int):
        g = np.ascontiguousarray(grey)
        h, w = g.shape
        out = np.empty((h, w), np.float32)
        self._check(self._lib.stk_gaussian_blur_f32(self._h, C.c_void_p(g.ctypes.data), _DEPTH[str(g.dtype)], w, h,
                                                    HOST, int(ksize), C.c_void_p(out.ctypes.data)))
        return out

    def find_transform_ecc(self, templ, inp, warp, params: EccMatchParameters):
        """video::find_transform_ecc(template, input, warp, ...) lib.rs:769-777 -> (warp3x3 f32, rho, iterations)."""
        t = np.ascontiguousarray(templ)
        i = np.ascontiguousarray(inp)
        if t.shape != i.shape or t.dtype != i.dtype:
            raise InvalidParams("template and input must share size and type")
        wm = np.eye(3, dtype=np.float32)
        wv = np.asarray(warp, np.float32)
        wm[: wv.shape[0], :] = wv
        rho, its = C.c_double(0), C.c_int32(0)
        p = params._c()
        st = self._lib.stk_find_transform_ecc(self._h, C.c_void_p(t.ctypes.data), C.c_void_p(i.ctypes.data),
                                              _DEPTH[str(t.dtype)], t.shape[1], t.shape[0], HOST, C.byref(p),
                                              C.c_void_p(wm.ctypes.data), C.byref(rho), C.byref(its))
        self._check(st)
        return wm, rho.value, its.value

    def warp_accumulate(self, frame, M, *, is_affine=False, border_mode=BORDER_CONSTANT, border_value=(0, 0, 0, 0),
                        alpha=1.0 / 255.0, acc=None):
        """warp_perspective/warp_affine(convert(frame, alpha), M) (+ acc). Returns the f32 image."""
        m = self._marshal([frame])
        Md = np.ascontiguousarray(np.asarray(M, np.float64).reshape(-1))
        if Md.size == 6:
            Md = np.concatenate([Md, [0.0, 0.0, 1.0]])
        bv = np.asarray((list(border_value) + [0.0] * 4)[:4], np.float64)
        accumulate = acc is not None
        if accumulate:
            out = acc
            if _is_torch(out):
                img = _ffi.ImageF32(out.data_ptr(), m.w, m.h, m.c, DEVICE if out.is_cuda else HOST, 0)
            else:
                img = _ffi.ImageF32(out.ctypes.data, m.w, m.h, m.c, HOST, 0)
        else:
            out, img = self._out_image(m)
        st = self._lib.stk_warp_accumulate(self._h, C.byref(m.c_frames), C.c_void_p(Md.ctypes.data), int(is_affine),
                                           int(border_mode), C.c_void_p(bv.ctypes.data), float(alpha),
                                           int(accumulate), C.byref(img))
        self._check(st)
        return out

    def scale_image_grey(self, grey, scale_down: float):
        """utils::scale_image (utils.rs:186-214) on a grey image, 8-bit or f32 (the depths cvtColor leaves where the reference
        shrinks a grey): INTER_AREA, smaller dimension -> scale_down."""
        f32 = np.asarray(grey).dtype == np.float32
        g = np.ascontiguousarray(grey, np.float32 if f32 else np.uint8)
        h, w = g.shape
        f = float(np.float32(scale_down)) / min(w, h)              # utils.rs:191-199: the smaller dimension becomes scale_down
        out = np.empty(max(h * w, (int(w * f) + 1) * (int(h * f) + 1)), g.dtype)
        nw, nh = C.c_int32(0), C.c_int32(0)
        fn = self._lib.stk_scale_image_grey_f32 if f32 else self._lib.stk_scale_image_grey
        self._check(fn(self._h, C.c_void_p(g.ctypes.data), w, h, HOST, float(scale_down), C.c_void_p(out.ctypes.data), C.byref(nw), C.byref(nh)))
        return out[: nw.value * nh.value].reshape(nh.value, nw.value).copy()

    def orb_detect_and_compute(self, grey, max_keypoints: int = 2000):
        g = np.ascontiguousarray(grey, np.uint8)
        h, w = g.shape
        kps = np.zeros((max_keypoints, 7), np.float32)
        des = np.zeros((max_keypoints, 32), np.uint8)
        n = C.c_int32(0)
        st = self._lib.stk_orb_detect_and_compute(self._h, C.c_void_p(g.ctypes.data), w, h, HOST, int(max_keypoints),
                                                  C.c_void_p(kps.ctypes.data), C.c_void_p(des.ctypes.data), C.byref(n))
        self._check(st)
        return kps[: n.value].copy(), des[: n.value].copy()

    def bf_knn2_hamming(self, query, train):
        q = np.ascontiguousarray(query, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(train, np.uint8).reshape(-1, 32)
        out = np.full((q.shape[0], 4), -1, np.int32)
        st = self._lib.stk_bf_knn2_hamming(self._h, C.c_void_p(q.ctypes.data), q.shape[0], C.c_void_p(t.ctypes.data),
                                           t.shape[0], C.c_void_p(out.ctypes.data))
        self._check(st)
        return out

    def find_homography(self, src_pts, dst_pts, method: int = RANSAC, ransac_reproj_threshold: float = 3.0):
        s = np.ascontiguousarray(src_pts, np.float32).reshape(-1, 2)
        d = np.ascontiguousarray(dst_pts, np.float32).reshape(-1, 2)
        H = np.zeros(9, np.float64)
        mask = np.zeros(s.shape[0], np.uint8)
        found = C.c_int32(0)
        st = self._lib.stk_find_homography(self._h, C.c_void_p(s.ctypes.data), C.c_void_p(d.ctypes.data), s.shape[0],
                                           int(method), float(ransac_reproj_threshold), C.c_void_p(H.ctypes.data),
                                           C.c_void_p(mask.ctypes.data), C.byref(found))
        self._check(st)
        return (H.reshape(3, 3) if found.value else None), mask


_default: dict[int, Stacker] = {}


def default_stacker(device: int = 0) -> Stacker:
    if device not in _default:
        _default[device] = Stacker(device)
    return _default[device]


def keypoint_match(files, params: KeyPointMatchParameters, scale_down_width: Optional[float] = None):
    """Drop-in for libstacker::keypoint_match (lib.rs:129-144): returns (dropped, image)."""
    return default_stacker().keypoint_match(files, params, scale_down_width)


def ecc_match(files, params: EccMatchParameters, scale_down_width: Optional[float] = None):
    """Drop-in for libstacker::ecc_match (lib.rs:702-717): returns the stacked f32 image."""
    return default_stacker().ecc_match(files, params, scale_down_width)
