"""The reference's example program end to end (examples/main.rs:27-133) on the engine: four sharpness scores per file on
the grey image, sort by TENG (low quality first), drop the worst file, reverse so that the sharpest frame becomes the
reference, then the four stacking calls with the example's own parameters — keypoint_match(None), keypoint_match(Some(400)),
ecc_match(None), ecc_match(Some(400)) — each compared with the oracle run on the same ordered list. The image set the
example expects (image_stacking_py, README.md:18) is not in the container: 10 synthetic 800x600 frames with graded blur
stand in for it (SURVEY 8d, configs[0])."""
import numpy as np
import pytest

import oracle
from conftest import assert_ecc_stack_close, assert_stack_close
from libstacker_rs_amd import EccMatchParameters, KeyPointMatchParameters, MotionType, RANSAC, synth

pytestmark = pytest.mark.gpu


def test_example_main_flow(stacker):
    frames, _ = synth.make_stack(10, 800, 600)
    fr = [f.copy() for f in frames.numpy()]
    # make the quality differ: frame 4 is badly defocused (the one main.rs:64 throws away), two more slightly
    for idx, k in ((4, 7), (7, 3), (2, 3)):
        b = np.stack([np.clip(np.rint(oracle.gaussian_blur_f32(fr[idx][..., c].copy(), k)), 0, 255) for c in range(3)], -1).astype(np.uint8)
        if k == 7:
            b = np.stack([np.clip(np.rint(oracle.gaussian_blur_f32(b[..., c].copy(), 7)), 0, 255) for c in range(3)], -1).astype(np.uint8)
        fr[idx] = b
    scored = []
    for i, f in enumerate(fr):                                         # main.rs:37-49 (IMREAD_GRAYSCALE stand-in: BGR2GRAY)
        g = oracle.grey(f)
        m = (stacker.sharpness_modified_laplacian(g), stacker.sharpness_variance_of_laplacian(g),
             stacker.sharpness_tenengrad(g, 3), stacker.sharpness_normalized_gray_level_variance(g))
        assert m == (oracle.sharpness(g, 0), oracle.sharpness(g, 1), oracle.sharpness(g, 2, 3), oracle.sharpness(g, 3))
        scored.append((i, m))
    scored.sort(key=lambda t: t[1][2])                                 # main.rs:53: by TENG, low quality first (stable)
    assert scored[0][0] == 4                                           # the defocused frame ranks last in quality
    order = [i for i, _ in scored][1:][::-1]                           # main.rs:64: skip(1).rev()
    files = [fr[i] for i in order]
    kp = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)               # main.rs:69-76
    ecc = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)     # main.rs:107-112

    dropped, img, stats = stacker.keypoint_match(files, kp, return_stats=True)                      # main.rs:67-78
    d_o, ref, Hs, _ = oracle.keypoint_match(files, details=True)
    assert dropped == d_o == 0
    assert_stack_close(img, ref, bulk=9e-6)

    dropped4, img4 = stacker.keypoint_match(files, kp, scale_down_width=400.0)                      # main.rs:86-97
    d4_o, ref4 = oracle.keypoint_match(files, scale_down_width=400.0)
    assert dropped4 == d4_o == 0
    assert_stack_close(img4, ref4, bulk=9e-6)

    e_img, e_stats = stacker.ecc_match(files, ecc, return_stats=True)                               # main.rs:105-114
    e_ref, warps, iters = oracle.ecc_match(files, max_count=5000, epsilon=1e-5, gauss_filt_size=5)
    assert_ecc_stack_close(e_img, e_ref, files, warps, label="example ecc", iters=[s["iterations"] for s in e_stats[1:]], iters_ref=iters[1:])

    e4, e4_stats = stacker.ecc_match(files, ecc, scale_down_width=400.0, return_stats=True)         # main.rs:119-128
    e4_ref, w4, it4 = oracle.ecc_match(files, max_count=5000, epsilon=1e-5, gauss_filt_size=5, scale_down_width=400.0)
    assert_ecc_stack_close(e4, e4_ref, files, w4, label="example ecc 400", iters=[s["iterations"] for s in e4_stats[1:]], iters_ref=it4[1:])
    # the four results show the same scene (what the example's four windows are for)
    for a in (img4, e_img, e4):
        assert np.abs(a[40:-40, 40:-40] - img[40:-40, 40:-40]).mean() < 0.01
