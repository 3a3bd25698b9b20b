"""GPU parity of the keypoint_match path (ORB, Hamming 2-NN, RANSAC homography, fold) against the oracle."""
import numpy as np
import pytest

import oracle
from conftest import assert_stack_close
from libstacker_rs_amd import (InvalidParams, KeyPointMatchParameters, NotEnoughFiles, NotImplementedYet, OpenCvError,
                               RANSAC, LMEDS, RHO, synth)

pytestmark = pytest.mark.gpu

# findHomography vs the oracle: inlier masks identical; H to 2e-7 relative. Not tighter because LMSolver accepts a step only
# if it lowers the f64 cost S, which resolves the minimiser to ~sqrt(eps * S / curvature) ~ 2e-8 px in translation: the
# oracle itself moves by 1.5e-8 when its points are merely re-ordered (tests/test_cpu_oracle.py::test_homography_lm_floor).
H_RTOL = 2e-7

PARAMS = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)          # examples/main.rs:69-76


@pytest.fixture(scope="module")
def kp_stack():
    frames, G = synth.make_stack(4, 640, 480)
    return frames.numpy(), G


def test_orb_bit_exact_vs_oracle(stacker, kp_stack):
    frames, _ = kp_stack
    for f in frames[:2]:
        g = oracle.grey(f)
        kp, de = stacker.orb_detect_and_compute(g, 4096)
        kpo, deo = oracle.orb_detect_and_compute(g)
        assert kp.shape == kpo.shape and len(kp) >= 400
        assert np.array_equal(kp, kpo)            # integer stages + identical f32 formulas: exact
        assert np.array_equal(de, deo)            # 256-bit descriptors: bit exact
        assert list(np.bincount(kp[:, 5].astype(int), minlength=8)[:3]) == [109, 90, 75]   # nfeatures per level


def test_orb_odd_sizes_and_flat_image(stacker):
    rng = np.random.default_rng(3)
    g = rng.integers(0, 256, (211, 333), dtype=np.uint8)
    kp, de = stacker.orb_detect_and_compute(g)
    kpo, deo = oracle.orb_detect_and_compute(g)
    assert np.array_equal(kp, kpo) and np.array_equal(de, deo)
    flat = np.full((120, 160), 77, np.uint8)      # no corners at all
    kp, de = stacker.orb_detect_and_compute(flat)
    assert len(kp) == 0 and len(de) == 0
    tiny = rng.integers(0, 256, (40, 50), dtype=np.uint8)   # smaller than 2*edgeThreshold: nothing survives
    kp, _ = stacker.orb_detect_and_compute(tiny)
    assert len(kp) == 0 and len(oracle.orb_detect_and_compute(tiny)[0]) == 0


def test_knn2_hamming_bit_exact(stacker):
    rng = np.random.default_rng(0)
    q = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (257, 32), dtype=np.uint8)
    t[10] = t[200] = q[5]                         # exact ties: the lower train index must win
    t[11] = q[6]
    got, ref = stacker.bf_knn2_hamming(q, t), oracle.bf_knn2_hamming(q, t)
    assert np.array_equal(got, ref)
    assert got[5, 0] == 10 and got[5, 2] == 200 and got[5, 1] == 0
    one = stacker.bf_knn2_hamming(q[:4], t[:1])   # fewer than 2 train rows: second neighbour missing
    assert np.array_equal(one, oracle.bf_knn2_hamming(q[:4], t[:1])) and (one[:, 2] == -1).all()
    none = stacker.bf_knn2_hamming(q[:4], t[:0])
    assert (none[:, 0] == -1).all()
    # the scan is split over eight lanes per query (rows t = lane mod 8) and merged: every train count around the split, with
    # many exact ties (few distinct rows), must give the sequential scan's (best, second) — ties to the lower train index
    few = rng.integers(0, 256, (3, 32), dtype=np.uint8)
    for nt in (1, 2, 3, 7, 8, 9, 15, 16, 17, 63, 64, 65, 500):
        tt = few[rng.integers(0, 3, nt)]
        qq = np.concatenate([few, q[:13]])
        assert np.array_equal(stacker.bf_knn2_hamming(qq, tt), oracle.bf_knn2_hamming(qq, tt)), nt
    with pytest.raises(Exception):
        stacker.bf_knn2_hamming(q[:2], np.zeros((65537, 32), np.uint8))


def test_find_homography_known_answers(stacker):
    Ht = np.array([[1.02, 0.03, 5.0], [-0.01, 0.98, -3.0], [1e-5, -2e-5, 1.0]])
    rng = np.random.default_rng(1)
    src = rng.uniform(0, 600, (120, 2)).astype(np.float32)
    p = np.c_[src, np.ones(len(src))] @ Ht.T
    dst = (p[:, :2] / p[:, 2:]).astype(np.float32)
    # 4 exact correspondences -> plain DLT
    H, _ = stacker.find_homography(src[:4], dst[:4], RANSAC, 3.0)
    assert synth.corner_error(H, Ht, 640, 480) < 1e-2
    # RANSAC with 30% gross outliers
    dst_o = dst.copy()
    dst_o[::3] += rng.uniform(30, 80, dst_o[::3].shape).astype(np.float32)
    H, mask = stacker.find_homography(src, dst_o, RANSAC, 3.0)
    Ho, masko = oracle.find_homography(src, dst_o, 8, 3.0)
    assert np.array_equal(mask, masko) and np.allclose(H, Ho, rtol=H_RTOL, atol=1e-9)
    assert mask[::3].sum() == 0 and mask.sum() == len(src) - len(src[::3])
    assert synth.corner_error(H, Ht, 640, 480) < 0.05
    # least squares (method 0)
    H0, _ = stacker.find_homography(src, dst, 0, 3.0)
    assert synth.corner_error(H0, Ht, 640, 480) < 0.05
    with pytest.raises(OpenCvError):
        stacker.find_homography(src[:3], dst[:3], RANSAC, 3.0)        # fewer than 4 pairs
    with pytest.raises(NotImplementedYet):
        stacker.find_homography(src, dst, RHO, 3.0)
    # LMEDS: least median of squares, < 45% outliers by construction of the method
    Hl, maskl = stacker.find_homography(src, dst_o, LMEDS, 3.0)
    Hlo, masklo = oracle.find_homography(src, dst_o, 4, 3.0)
    assert np.array_equal(maskl, masklo) and np.allclose(Hl, Hlo, rtol=H_RTOL, atol=1e-9)
    assert maskl[::3].sum() == 0 and synth.corner_error(Hl, Ht, 640, 480) < 0.05
    # collinear points: no model
    line = np.c_[np.arange(20), 2 * np.arange(20)].astype(np.float32)
    Hn, _ = stacker.find_homography(line, line, RANSAC, 3.0)
    assert Hn is None and oracle.find_homography(line, line, 8, 3.0)[0] is None


def test_rng_known_answer():
    # cv::RNG(-1): state = state_lo * 4164903690 + state_hi, first outputs
    st = 0xFFFFFFFFFFFFFFFF
    exp = []
    for _ in range(3):
        st = ((st & 0xFFFFFFFF) * 4164903690 + (st >> 32)) & 0xFFFFFFFFFFFFFFFF
        exp.append(st & 0xFFFFFFFF)
    assert oracle.rng_sequence(3) == exp


def test_keypoint_match_stack_matches_oracle(stacker, kp_stack):
    frames, G = kp_stack
    dropped, out, stats = stacker.keypoint_match(list(frames), PARAMS, return_stats=True)
    d_o, ref, Hs, status = oracle.keypoint_match(list(frames), details=True)
    assert dropped == 0 and d_o == 0
    for i in range(1, len(frames)):
        assert np.allclose(stats[i]["warp"], Hs[i], rtol=H_RTOL, atol=1e-9)   # same matches -> same RANSAC decisions
        assert synth.corner_error(stats[i]["warp"], G[i], 640, 480) <= 1.0   # vs generator ground truth
        assert stats[i]["n_matches"] >= 100
    assert_stack_close(out, ref)                                   # <= 1e-6 per folded frame


def test_keypoint_match_drops_unmatchable_frame(stacker, kp_stack):
    frames, _ = kp_stack
    bad = np.full_like(frames[0], 128)            # featureless frame: no keypoints -> < 5 matches -> dropped
    stack = [frames[0], frames[1], bad, frames[2]]
    dropped, out, stats = stacker.keypoint_match(stack, PARAMS, return_stats=True)
    d_o, ref = oracle.keypoint_match(stack)
    assert dropped == 1 and d_o == 1 and stats[2]["status"] == 1
    assert_stack_close(out, ref)       # divisor is n - dropped = 3 (documented semantics, lib.rs:98)
    # the same stack without the bad frame gives the same image
    d2, out2 = stacker.keypoint_match([frames[0], frames[1], frames[2]], PARAMS)
    assert d2 == 0 and np.array_equal(out, out2)


def test_keypoint_match_all_dropped_is_invalid_params(stacker, kp_stack):
    frames, _ = kp_stack
    flat = np.full_like(frames[0], 50)
    # frame 0 itself is always kept (lib.rs:194-196), so use a stack whose moving frames all fail: still OK
    dropped, out = stacker.keypoint_match([frames[0], flat, flat], PARAMS)
    assert dropped == 2 and np.max(np.abs(out - oracle.convert_f32(frames[0]))) <= 1e-6


def test_keypoint_match_errors(stacker, kp_stack):
    frames, _ = kp_stack
    with pytest.raises(NotEnoughFiles):
        stacker.keypoint_match([], PARAMS)
    with pytest.raises(OpenCvError):
        stacker.keypoint_match([f.astype(np.uint16) for f in frames[:2]], PARAMS)      # ORB needs 8-bit
    with pytest.raises(InvalidParams):
        stacker.keypoint_match(list(frames[:2]), PARAMS, scale_down_width=640.0)        # >= full width, lib.rs:377
    # frames of differing SIZE are a legal stack on this path since round 4 (lib.rs:200-204, 290-299: tests/test_gpu_mixed.py);
    # frames of differing TYPE are not
    d, out = stacker.keypoint_match([frames[0], np.ascontiguousarray(frames[1][:400])], PARAMS)
    assert out.shape == frames[0].shape
    with pytest.raises(OpenCvError):
        stacker.keypoint_match([frames[0], np.ascontiguousarray(np.concatenate([frames[1][:400], frames[1][:400, :, :1]], -1))], PARAMS)
    # a method findHomography does not know makes it throw, and keypoint_match turns the error into a skipped frame
    # (lib.rs:275: Err(_) => return Ok(None)): every moving frame is dropped and the result is frame 0 alone (round 4; the
    # engine used to fail the call). USAC's numbers (32 .. 38, OpenCV >= 4.5) and RHO are real methods: NotImplemented.
    from libstacker_rs_amd import NotImplementedYet
    d, out = stacker.keypoint_match(list(frames[:3]), KeyPointMatchParameters(7, 5.0, 0.80, 0.9))
    d_o, ref = oracle.keypoint_match(list(frames[:3]), method=7)
    assert d == d_o == 2 and np.array_equal(out, ref)
    assert np.array_equal(out, frames[0].astype(np.float32) * np.float32(1.0 / 255.0))
    for m in (16, 32, 38):
        with pytest.raises(NotImplementedYet):
            stacker.keypoint_match(list(frames[:2]), KeyPointMatchParameters(m, 5.0, 0.80, 0.9))


def test_keypoint_match_border_mode_and_value(stacker, kp_stack):
    frames, _ = kp_stack
    p = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9, oracle.BORDER_REPLICATE, (0, 0, 0, 0))
    _, out = stacker.keypoint_match(list(frames[:3]), p)
    _, ref = oracle.keypoint_match(list(frames[:3]), border_mode=oracle.BORDER_REPLICATE)
    assert_stack_close(out, ref)
    p = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9, oracle.BORDER_CONSTANT, (0.5, 0.25, 1.0, 0))
    _, out = stacker.keypoint_match(list(frames[:3]), p)
    _, ref = oracle.keypoint_match(list(frames[:3]), border_value=(0.5, 0.25, 1.0, 0))
    assert_stack_close(out, ref)


def test_scale_image_inter_area_bit_exact(stacker):
    rng = np.random.default_rng(5)
    g = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    for sd in (200.0, 240.0, 333.0, 97.5):                  # 240 -> exact 2x (resizeAreaFast), others fractional
        nw, nh = oracle.scaled_size(640, 480, sd)
        got = stacker.scale_image_grey(g, sd)
        assert got.shape == (nh, nw)
        assert np.array_equal(got, oracle.resize_area_u8(g, nw, nh))
    tall = rng.integers(0, 256, (300, 120), dtype=np.uint8)   # width < height: the WIDTH becomes scale_down
    got = stacker.scale_image_grey(tall, 60.0)
    assert got.shape == (150, 60) and np.array_equal(got, oracle.resize_area_u8(tall, 60, 150))
    # exact halving rounds a half UP ((a + b + c + d + 2) >> 2), the other ratios to even (cvRound): closed form
    two = np.zeros((16, 16), np.uint8); two[0::2, 0::2] = 2                 # every 2 x 2 cell sums to 2: 0.5
    assert (stacker.scale_image_grey(two, 8.0) == 1).all()
    quarter = np.zeros((32, 32), np.uint8); quarter[0::4, 0::4] = 8         # every 4 x 4 cell sums to 8: 0.5 -> 0
    assert (stacker.scale_image_grey(quarter, 8.0) == 0).all()
    # scale_down above the smaller dimension ENLARGES (the reference checks it against the width only): INTER_AREA's bilinear emulation
    for sd in (560.0, 600.5, 960.0):                          # 480 -> 560 (x 7/6), 600 (fractional), 960 (x 2: pixel replication)
        nw, nh = oracle.scaled_size(640, 480, sd)
        got = stacker.scale_image_grey(g, sd)
        assert got.shape == (nh, nw) and nw > 640 and np.array_equal(got, oracle.resize_area_u8(g, nw, nh)), sd
    assert np.array_equal(stacker.scale_image_grey(g, 960.0), np.repeat(np.repeat(g, 2, axis=0), 2, axis=1))
    # a 32FC1 grey (float stacks; round 4): the same tables without the rounding, bit for bit against the oracle
    f = rng.random((480, 640), dtype=np.float32) * 255
    for sd in (200.0, 240.0, 160.0, 333.0, 97.5):              # 240 -> 2 x 2 (vector form), 160 -> 3 x 3 (groups of four), others fractional
        nw, nh = oracle.scaled_size(640, 480, sd)
        got = stacker.scale_image_grey(f, sd)
        assert got.dtype == np.float32 and got.shape == (nh, nw)
        assert np.array_equal(got, oracle.resize_area_f32(f, nw, nh)), sd
    for sd in (560.0, 600.5):
        nw, nh = oracle.scaled_size(640, 480, sd)
        assert np.array_equal(stacker.scale_image_grey(f, sd), oracle.resize_area_f32(f, nw, nh)), sd


def test_keypoint_match_scale_down_between_height_and_width_enlarges_like_the_reference(stacker, kp_stack):
    """lib.rs:377 rejects scale_down_width >= the frame's WIDTH only; scale_image (utils.rs:186-214) then makes the SMALLER dimension
    equal to it — for landscape frames a value between height and width enlarges the greys ORB runs on (round 4; the engine
    used to answer InvalidParams)."""
    frames, G = kp_stack
    h, w = frames[0].shape[:2]
    assert h < w
    sd = float(h) * 1.125
    dropped, out, stats = stacker.keypoint_match(list(frames[:4]), PARAMS, scale_down_width=sd, return_stats=True)
    d_o, ref, Hs, status = oracle.keypoint_match(list(frames[:4]), details=True, scale_down_width=sd)
    assert dropped == d_o
    for i in range(1, 4):
        assert (stats[i]["status"] == 0) == (status[i] == 0)
        if status[i] == 0:
            assert np.allclose(stats[i]["warp"], Hs[i], rtol=H_RTOL, atol=1e-9)
    assert_stack_close(out, ref)


def test_keypoint_match_scale_down_matches_oracle(stacker, kp_stack):
    frames, G = kp_stack
    dropped, out, stats = stacker.keypoint_match(list(frames), PARAMS, scale_down_width=300.0, return_stats=True)
    d_o, ref, Hs, status = oracle.keypoint_match(list(frames), details=True, scale_down_width=300.0)
    assert dropped == d_o == 0
    for i in range(1, len(frames)):
        assert np.allclose(stats[i]["warp"], Hs[i], rtol=H_RTOL, atol=1e-9)         # incl. the 4-entry rescale (utils.rs:236-239)
        assert synth.corner_error(stats[i]["warp"], G[i], 640, 480) <= 2.5    # coarser: features found at 0.62x
    assert_stack_close(out, ref)


def test_keypoint_match_lmeds_matches_oracle(stacker, kp_stack):
    frames, G = kp_stack
    p = KeyPointMatchParameters(LMEDS, 5.0, 0.80, 0.9)
    dropped, out, stats = stacker.keypoint_match(list(frames[:3]), p, return_stats=True)
    d_o, ref, Hs, status = oracle.keypoint_match(list(frames[:3]), method=4, details=True)
    assert dropped == d_o == 0
    for i in (1, 2):
        assert np.allclose(stats[i]["warp"], Hs[i], rtol=H_RTOL, atol=1e-9)
        assert synth.corner_error(stats[i]["warp"], G[i], 640, 480) <= 1.0
    assert_stack_close(out, ref)
    with pytest.raises(NotImplementedYet):
        stacker.keypoint_match(list(frames[:2]), KeyPointMatchParameters(RHO, 5.0, 0.80, 0.9))


def test_frame_sharded_ranks_reproduce_the_single_gpu_stack(stacker, kp_stack):
    import torch
    from libstacker_rs_amd.shard import shard_moving_frames
    frames, _ = kp_stack
    bad = np.full_like(frames[0], 128)                       # one frame that gets dropped, on the second rank
    stack = np.stack([frames[0], frames[1], frames[2], bad, frames[3]])
    n = len(stack)
    d_full, full, full_stats = stacker.keypoint_match(list(stack), PARAMS, return_stats=True)
    total = torch.zeros((stack.shape[1], stack.shape[2], 3), dtype=torch.float32, device="cuda")
    added_total = dropped_total = 0
    for rank in range(2):
        mine = shard_moving_frames(n, 2, rank)
        acc = torch.empty_like(total)
        sub = torch.from_numpy(np.ascontiguousarray(stack[[0] + mine])).cuda()
        added, dropped, stats = stacker.keypoint_match_shard(sub, PARAMS, rank == 0, acc)
        for j, g in enumerate(mine):
            assert stats[1 + j]["status"] == full_stats[g]["status"]
            assert np.array_equal(stats[1 + j]["warp"], full_stats[g]["warp"])
        total += acc
        added_total += added
        dropped_total += dropped
    assert dropped_total == d_full == 1 and added_total == n - 1
    out = stacker.finalize_mean(total, added_total).cpu().numpy()
    assert np.max(np.abs(out - full)) <= 1e-6


@pytest.mark.parametrize("shape", [(128, 128), (129, 257), (127, 383), (96, 520), (161, 130), (256, 384), (70, 70), (63, 200)])
def test_orb_tile_boundaries_bit_exact(stacker, shape):
    # the FAST/NMS, blur and pyramid kernels work on 128 x 32 tiles with halos: sizes at, just below and just above the tile
    # multiples (and images barely larger than the 2 x 31 px border) must give the oracle's keypoints and descriptors exactly
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    base = rng.integers(0, 256, (shape[0] // 4 + 2, shape[1] // 4 + 2), dtype=np.uint8)
    g = np.kron(base, np.ones((4, 4), np.uint8))[: shape[0], : shape[1]]        # blocky texture: many real corners
    g = np.clip(g.astype(np.int16) + rng.integers(-6, 7, g.shape), 0, 255).astype(np.uint8)
    kp, de = stacker.orb_detect_and_compute(g, 4096)
    kpo, deo = oracle.orb_detect_and_compute(g)
    assert kp.shape == kpo.shape and np.array_equal(kp, kpo) and np.array_equal(de, deo)
    if min(shape) > 100:
        assert len(kp) > 50


def test_keypoint_lanes_do_not_change_results(stacker):
    """Device-resident stacks of >= 16 frames are cut into kp_lanes runs of >= 8 frames that go through the pipeline side by
    side (helper contexts of the same device), and the fold follows them run by run in stack order. Frames are independent
    of each other (lib.rs:185-290 is a Rayon map body): status, keypoint / match / inlier counts, H and the stacked image —
    bit for bit, the fold accumulates in stack order whatever the cut — must not depend on the lanes, including a frame
    that is dropped and repeated calls on the same context."""
    import torch
    from libstacker_rs_amd import KeyPointMatchParameters, RANSAC, synth
    frames, _ = synth.make_stack(35, 640, 480, device="cuda")
    frames[13] = 128                                        # featureless frames in two different lanes: dropped there
    frames[30] = 7
    kp = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)
    res = {}
    try:
        for lanes in (4, 1, 3, 2, 4):
            stacker.set_option("kp_lanes", lanes)
            d, out, stats = stacker.keypoint_match(frames, kp, return_stats=True)
            cur = (d, out.cpu().numpy(), [(s["status"], s["n_keypoints"], s["n_matches"], s["n_inliers"]) for s in stats],
                   np.stack([s["warp"] for s in stats]))
            if lanes in res:
                assert cur[0] == res[lanes][0] and np.array_equal(cur[1], res[lanes][1]) and cur[2] == res[lanes][2]
            res[lanes] = cur
    finally:
        stacker.set_option("kp_lanes", 3)
    assert res[1][0] == 2
    for lanes in (2, 3, 4):
        assert res[lanes][0] == res[1][0] and res[lanes][2] == res[1][2] and np.array_equal(res[lanes][3], res[1][3])
        assert np.array_equal(res[lanes][1], res[1][1])
    # the scale-down variant (ORB on INTER_AREA-shrunk greys, lib.rs:355-601) through the lanes as well
    outs = []
    try:
        for lanes in (4, 1):
            stacker.set_option("kp_lanes", lanes)
            d, out, stats = stacker.keypoint_match(frames, kp, scale_down_width=400.0, return_stats=True)
            outs.append((d, out.cpu().numpy(), np.stack([s["warp"] for s in stats])))
    finally:
        stacker.set_option("kp_lanes", 3)
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][1], outs[1][1])
    with pytest.raises(Exception):
        stacker.set_option("kp_lanes", 9)


def test_orb_patch_blur_equals_the_whole_level_blur(stacker):
    """The descriptor kernel blurs (GaussianBlur 7x7 sigma 2, orb.cpp's fixed per-pixel operation order) only the 45 x 40 window
    around each kept keypoint (orb_patch_blur = 1, the default); blurring every level whole first (0) must give the same
    keypoints and descriptor bytes — single images of awkward sizes, and a stack through keypoint_match."""
    from libstacker_rs_amd import KeyPointMatchParameters, RANSAC, synth
    rng = np.random.default_rng(77)
    imgs = []
    for (h, w) in [(480, 640), (301, 517), (200, 129), (1080, 1920)]:
        yy, xx = np.mgrid[0:h, 0:w]
        g = ((np.sin(xx * 0.13) + np.cos(yy * 0.11) + np.sin((xx + yy) * 0.07)) * 40 + 128)
        imgs.append(np.clip(g + rng.integers(-25, 26, g.shape), 0, 255).astype(np.uint8))
    frames, _ = synth.make_stack(17, 640, 480, device="cuda")
    kp = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)
    got = {}
    try:
        for mode in (1, 0):
            stacker.set_option("orb_patch_blur", mode)
            single = [stacker.orb_detect_and_compute(g, 4096) for g in imgs]
            d, out, stats = stacker.keypoint_match(frames, kp, return_stats=True)
            got[mode] = (single, d, out.cpu().numpy(), np.stack([s["warp"] for s in stats]))
    finally:
        stacker.set_option("orb_patch_blur", 1)
    for (k1, d1), (k0, d0) in zip(got[1][0], got[0][0]):
        assert len(k1) > 20 and np.array_equal(k1, k0) and np.array_equal(d1, d0)
    assert got[1][1] == got[0][1] and np.array_equal(got[1][3], got[0][3]) and np.array_equal(got[1][2], got[0][2])
    # and against the oracle (whole-level blur, oracle_orb.cpp:102), on the 1080p image
    ko, do = oracle.orb_detect_and_compute(imgs[3])
    assert np.array_equal(got[1][0][3][0], ko) and np.array_equal(got[1][0][3][1], do)


def test_orb_table_driven_pyramid_equals_the_per_tile_tables(stacker):
    """The pyramid steps read offset / weight tables computed once per geometry (orb_resize_tables = 1, the default) and stage
    the source footprint with one replicated column / row behind the image; computing the tables per tile (0, round 2's
    kernel) must give the same keypoints and descriptors — widths and heights around the tile and 16-byte boundaries."""
    rng = np.random.default_rng(5)
    got = {}
    imgs = [rng.integers(0, 256, (h, w), dtype=np.uint8) for (h, w) in [(480, 640), (301, 517), (200, 129), (95, 1000), (1080, 1920), (130, 131)]]
    imgs = [np.clip(np.kron(rng.integers(0, 256, ((h + 7) // 8, (w + 7) // 8)), np.ones((8, 8)))[:h, :w] * 0.7 + im * 0.3, 0, 255).astype(np.uint8)
            for im, (h, w) in zip(imgs, [i.shape for i in imgs])]
    try:
        for mode in (1, 0):
            stacker.set_option("orb_resize_tables", mode)
            got[mode] = [stacker.orb_detect_and_compute(g, 4096) for g in imgs]
    finally:
        stacker.set_option("orb_resize_tables", 1)
    n_total = 0
    for (k1, d1), (k0, d0) in zip(got[1], got[0]):
        assert np.array_equal(k1, k0) and np.array_equal(d1, d0)
        n_total += len(k1)
    assert n_total > 500
    ko, do = oracle.orb_detect_and_compute(imgs[1])
    assert np.array_equal(got[1][1][0], ko) and np.array_equal(got[1][1][1], do)


def test_orb_device_cull_equals_the_host_cull(stacker):
    """The Harris cull (KeyPointsFilter::retainBest: the n_l best per level and everything that ties with the n_l-th) and the
    ordering (response descending, y, x) run on the device by default (one bitonic sort per level and frame on an integer
    key); the host's nth_element + sort (orb_device_cull = 0) must give the same keypoints and descriptors — on textured
    images, on a periodic pattern (hundreds of exactly tied responses: the tie rule, and short lists longer than the device
    sorts, which fall back to the host per level), and through keypoint_match."""
    from libstacker_rs_amd import KeyPointMatchParameters, RANSAC, synth
    rng = np.random.default_rng(11)
    imgs = []
    for (h, w) in [(480, 640), (301, 517), (1080, 1920)]:
        base = np.kron(rng.integers(0, 256, ((h + 5) // 6, (w + 5) // 6)), np.ones((6, 6)))[:h, :w]
        imgs.append(np.clip(base * 0.8 + rng.integers(0, 50, (h, w)), 0, 255).astype(np.uint8))
    yy, xx = np.mgrid[0:600, 0:800]
    imgs.append((((xx // 8 + yy // 8) % 2) * 200 + 20).astype(np.uint8))          # checkerboard: every corner ties with hundreds of others
    tile = rng.integers(0, 256, (16, 16)).astype(np.uint8)
    imgs.append(np.tile(tile, (40, 50)))                                          # 640 x 800 of one repeated 16 x 16 tile
    frames, _ = synth.make_stack(17, 640, 480, device="cuda")
    kp = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)
    got = {}
    try:
        for mode in (1, 0):
            stacker.set_option("orb_device_cull", mode)
            single = [stacker.orb_detect_and_compute(g, 4096) for g in imgs]
            d, out, stats = stacker.keypoint_match(frames, kp, return_stats=True)
            got[mode] = (single, d, out.cpu().numpy(), np.stack([s["warp"] for s in stats]))
    finally:
        stacker.set_option("orb_device_cull", 1)
    for i, ((k1, d1), (k0, d0)) in enumerate(zip(got[1][0], got[0][0])):
        assert k1.shape == k0.shape and np.array_equal(k1, k0) and np.array_equal(d1, d0), i
    assert sum(len(k) for k, _ in got[1][0]) > 1500
    assert got[1][1] == got[0][1] and np.array_equal(got[1][3], got[0][3]) and np.array_equal(got[1][2], got[0][2])
    for i in (1, 3, 4):                                                           # and against the oracle
        ko, do = oracle.orb_detect_and_compute(imgs[i])
        assert np.array_equal(got[1][0][i][0], ko) and np.array_equal(got[1][0][i][1], do), i
