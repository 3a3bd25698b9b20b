"""keypoint_match on frames of DIFFERING size: the reference reads every file on its own, runs ORB at the
frame's own size and warps it into the first frame's (lib.rs:166, 200-204, 290-299); its ECC path fails on such a stack in
cv::add (lib.rs:809)."""
import ctypes as C

import numpy as np
import pytest

import oracle
from conftest import assert_stack_close
from libstacker_rs_amd import (EccMatchParameters, KeyPointMatchParameters, MotionType, NotImplementedYet, OpenCvError, RANSAC,
                               _ffi, synth)
from test_gpu_files import write_pnm

pytestmark = pytest.mark.gpu
KP = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)


def _mixed_stack():
    """Four 640x480 frames, three of them cut to sizes of their own (top-left crops keep the scene's coordinates)."""
    frames, G = synth.make_stack(4, 640, 480)
    fr = frames.numpy()
    return [fr[0], np.ascontiguousarray(fr[1][:400, :600]), np.ascontiguousarray(fr[2][:470, :520]), np.ascontiguousarray(fr[3][:333, :639])], G


def test_mixed_sizes_match_the_oracle(stacker):
    fr, G = _mixed_stack()
    d_ref, ref, Hs, status = oracle.keypoint_match(fr, details=True)
    dropped, out, stats = stacker.keypoint_match(fr, KP, return_stats=True)
    assert out.shape == fr[0].shape and dropped == d_ref == 0
    for i in range(1, len(fr)):
        assert stats[i]["status"] == int(status[i]) == 0
        assert np.allclose(stats[i]["warp"], Hs[i], rtol=2e-7, atol=1e-9)          # the LM floor, test_gpu_homography.py
        assert synth.corner_error(stats[i]["warp"], G[i], 640, 480) <= 1.0          # generator ground truth
    assert_stack_close(out, ref)


def test_mixed_sizes_with_scale_down_match_the_oracle(stacker):
    """keypoint_match_scale_down on such a stack (lib.rs:355-601): the width check is against the FIRST frame only, every grey is
    scaled so that ITS OWN smaller dimension equals scale_down_width (frames of differing size get differing factors) and the
    small-image homography is rescaled by that frame's own ratios (utils.rs:229-239) — reproduced as written, whatever it means
    geometrically for frames whose factors differ."""
    fr, G = _mixed_stack()
    from libstacker_rs_amd import InvalidParams
    for sd in (300.0, 420.0):                              # 420 > frame 3's height 333: that grey is ENLARGED
        d_ref, ref, Hs, status = oracle.keypoint_match(fr, details=True, scale_down_width=sd)
        dropped, out, stats = stacker.keypoint_match(fr, KP, scale_down_width=sd, return_stats=True)
        assert out.shape == fr[0].shape and dropped == d_ref
        for i in range(1, len(fr)):
            assert (stats[i]["status"] == 0) == (int(status[i]) == 0)
            if int(status[i]) == 0:
                # (frames scaled by differing factors: a 0.69 x similarity on few hundred matches — the LM floor in the two small
                # perspective entries is a few 1e-6 relative, cf. test_cpu_oracle.py::test_homography_lm_floor; what counts is
                # where the corners land)
                assert np.allclose(stats[i]["warp"], Hs[i], rtol=5e-6, atol=1e-9)
                assert synth.corner_error(stats[i]["warp"], Hs[i], fr[i].shape[1], fr[i].shape[0]) <= 1e-3
        assert_stack_close(out, ref)
    with pytest.raises(InvalidParams):
        stacker.keypoint_match(fr, KP, scale_down_width=640.0)         # lib.rs:377: >= the first frame's width


def test_frame_by_frame_route_equals_the_batched_pipeline(stacker):
    """A stack of ONE size whose frames sit in buffers of differing row stride cannot take the batched pipeline (one
    geometry per launch) and goes frame by frame through the stage-level entry points: same homographies, same image, bit
    for bit — a frame's stages do not depend on what shares their launches."""
    frames, _ = synth.make_stack(5, 320, 240)
    fr = frames.numpy()
    d0, base, s0 = stacker.keypoint_match(list(fr), KP, return_stats=True)
    n = len(fr)
    padded = []
    for i, f in enumerate(fr):
        buf = np.zeros((240, 320 * 3 + 4 * (i % 3)), np.uint8)                    # row strides 960, 964, 968, ...
        buf[:, :960] = f.reshape(240, 960)
        padded.append(buf)
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in padded])
    geo = (_ffi.FrameGeometry * n)(*[_ffi.FrameGeometry(320, 240, b.shape[1]) for b in padded])
    frs = _ffi.Frames(C.cast(ptrs, C.POINTER(C.c_void_p)), n, 320, 240, 3, 8, 0, 0)
    out = np.empty((240, 320, 3), np.float32)
    img = _ffi.ImageF32(out.ctypes.data, 320, 240, 3, 0, 0)
    stats = (_ffi.FrameStats * n)()
    dropped = C.c_int32(-1)
    p = KP._c()
    stacker._check(stacker._lib.stk_keypoint_match_mixed(stacker._h, C.byref(frs), geo, C.byref(p), 0.0, C.byref(img), C.byref(dropped), stats))
    assert dropped.value == d0
    for i in range(1, n):
        assert np.array_equal(np.array(list(stats[i].warp)).reshape(3, 3), s0[i]["warp"])
        assert (stats[i].n_keypoints, stats[i].n_matches, stats[i].n_inliers) == (s0[i]["n_keypoints"], s0[i]["n_matches"], s0[i]["n_inliers"])
    assert np.array_equal(out, base)
    # geometry == NULL and geometry with equal entries are the plain call
    dropped.value = -1
    tight = (C.c_void_p * n)(*[f.ctypes.data for f in fr])
    frs2 = _ffi.Frames(C.cast(tight, C.POINTER(C.c_void_p)), n, 320, 240, 3, 8, 0, 0)
    stacker._check(stacker._lib.stk_keypoint_match_mixed(stacker._h, C.byref(frs2), None, C.byref(p), 0.0, C.byref(img), C.byref(dropped), None))
    assert dropped.value == d0 and np.array_equal(out, base)


def test_files_of_differing_size(stacker, tmp_path):
    fr, _ = _mixed_stack()
    paths = []
    for i, f in enumerate(fr):
        paths.append(tmp_path / f"m{i}.ppm")
        write_pnm(paths[-1], f)
    d_f, out_f = stacker.keypoint_match_files(paths, KP)
    d_a, out_a = stacker.keypoint_match(fr, KP)
    assert d_f == d_a and np.array_equal(out_f, out_a)
    # ecc_match: the reference's `&acc + &warped` fails on it (cv::add, lib.rs:809)
    with pytest.raises(OpenCvError) as ei:
        stacker.ecc_match_files(paths, EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5))
    assert "differs in size" in str(ei.value)
    # a file that cannot be read at all still outranks the size question
    with pytest.raises(OpenCvError) as ei:
        stacker.keypoint_match_files(paths[:2] + [tmp_path / "missing.ppm"] + paths[2:], KP)
    assert "missing" in str(ei.value)
