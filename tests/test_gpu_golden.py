"""HIP path against the committed golden vectors (tests/golden/golden_v1.npz), through the C ABI."""
import os

import numpy as np
import pytest

from conftest import assert_stack_close

from libstacker_rs_amd import EccMatchParameters, KeyPointMatchParameters, MotionType, RANSAC, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def crop(a):
    return a[:48, :64]


def test_golden_pixel_stages(stacker, gold):
    f0 = gold["frames"][0]
    assert np.array_equal(stacker.grey(f0), gold["grey"])
    assert np.array_equal(stacker.grey(f0.astype(np.uint16) * 257), gold["grey16"])
    assert np.array_equal(stacker.convert_f32(f0[:16, :16].copy()), gold["convert"])
    for k in (3, 5, 7):
        assert np.array_equal(crop(stacker.gaussian_blur_f32(gold["grey"], k)), gold[f"blur{k}"])
    assert np.max(np.abs(crop(stacker.warp_accumulate(f0, gold["warp_M"])) - gold["warp_exact"])) <= 1e-6
    assert np.max(np.abs(crop(stacker.warp_accumulate(f0, gold["warp_M"], border_mode=4)) - gold["warp_reflect"])) <= 1e-6
    assert np.max(np.abs(crop(stacker.warp_accumulate(f0, gold["warp_A"], is_affine=True)) - gold["warp_affine"])) <= 1e-6
    stacker.set_option("warp_subpixel_bits", 5)
    try:
        assert np.max(np.abs(crop(stacker.warp_accumulate(f0, gold["warp_M"])) - gold["warp_classic"])) <= 1e-6
    finally:
        stacker.set_option("warp_subpixel_bits", 0)


def test_golden_ecc(stacker, gold):
    fr = gold["frames"]
    g0, g1 = stacker.grey(fr[0]), stacker.grey(fr[1])
    for name, mot in (("homography", MotionType.Homography), ("affine", MotionType.Affine),
                      ("euclidean", MotionType.Euclidean), ("translation", MotionType.Translation)):
        W, rho, its = stacker.find_transform_ecc(g1, g0, np.eye(3 if mot == MotionType.Homography else 2, 3),
                                                 EccMatchParameters(mot, 5, None, 5))
        assert its == 5
        # five fixed iterations: only f32 reduction order differs from the oracle
        assert synth.corner_error(W, gold[f"ecc_{name}_warp"], 160, 120) <= 5e-3
        assert abs(rho - float(gold[f"ecc_{name}_rho"])) <= 1e-6
    out, stats = stacker.ecc_match(list(fr), EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5), return_stats=True)
    for i in (1, 2):
        assert synth.corner_error(stats[i]["warp"], gold["ecc_match_warps"][i], 160, 120) <= 0.05
        assert stats[i]["iterations"] == int(gold["ecc_match_iters"][i])
    # the golden image is a 64x48 crop: evaluate on the interior pixels that fall inside it
    from conftest import interior_mask
    m = crop(interior_mask(out.shape[:2], [gold["ecc_match_warps"][i] for i in (1, 2)]))
    rel = (np.abs(crop(out) - gold["ecc_match_image"]) / np.maximum(np.abs(gold["ecc_match_image"]), 1e-3))[m]
    print("golden ecc stack: max rel %.3e over %d interior px" % (rel.max(), m.sum()))
    assert rel.max() <= 1e-4


def test_golden_keypoint_path(stacker, gold):
    fr = gold["frames"]
    kp0, de0 = stacker.orb_detect_and_compute(gold["grey"])
    assert np.array_equal(kp0, gold["orb_kp0"]) and np.array_equal(de0, gold["orb_de0"])
    assert np.array_equal(stacker.bf_knn2_hamming(gold["orb_de0"], gold["orb_de1"]), gold["knn01"])
    dropped, out, stats = stacker.keypoint_match(list(fr), KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9), return_stats=True)
    assert dropped == 0
    for i in (1, 2):
        assert np.allclose(stats[i]["warp"], gold["kp_match_H"][i], rtol=2e-7, atol=1e-9)   # the LM floor, see test_gpu_homography.py
    assert_stack_close(crop(out), gold["kp_match_image"])
