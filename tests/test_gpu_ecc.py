"""GPU parity of findTransformECC / ecc_match against the CPU oracle and the generator's ground truth."""
import numpy as np
import pytest
import torch

import oracle
from conftest import assert_ecc_stack_close
from libstacker_rs_amd import EccMatchParameters, MotionType, OpenCvError, NotEnoughFiles, synth

pytestmark = pytest.mark.gpu

PARAMS = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)     # examples/main.rs:107-112


def test_find_transform_ecc_homography_matches_oracle(stacker, small_stack):
    frames, G = small_stack
    g0 = oracle.grey(frames[0])
    for i in range(1, len(frames)):
        gi = oracle.grey(frames[i])
        W, rho, its = stacker.find_transform_ecc(gi, g0, np.eye(3), PARAMS)
        rc, Wo, rho_o, its_o = oracle.find_transform_ecc(gi, g0, np.eye(3), oracle.MOTION_HOMOGRAPHY, 5000, 1e-5, 5)
        assert rc == 0
        # <= 0.05 px corner displacement vs the oracle, the same iteration count
        assert synth.corner_error(W, Wo, 320, 240) <= 0.05
        assert its == its_o
        assert abs(rho - rho_o) <= 1e-5
        assert synth.corner_error(W, G[i], 320, 240) <= 0.25      # vs generator ground truth


@pytest.mark.parametrize("motion,omotion", [(MotionType.Translation, oracle.MOTION_TRANSLATION),
                                            (MotionType.Euclidean, oracle.MOTION_EUCLIDEAN),
                                            (MotionType.Affine, oracle.MOTION_AFFINE)])
def test_find_transform_ecc_other_motions(stacker, small_stack, motion, omotion):
    frames, _ = small_stack
    g0, g1 = oracle.grey(frames[0]), oracle.grey(frames[1])
    p = EccMatchParameters(motion, 200, 1e-6, 3)
    W, rho, its = stacker.find_transform_ecc(g1, g0, np.eye(2, 3), p)
    rc, Wo, rho_o, its_o = oracle.find_transform_ecc(g1, g0, np.eye(2, 3), omotion, 200, 1e-6, 3)
    assert rc == 0
    assert synth.corner_error(W, Wo, 320, 240) <= 0.05
    assert abs(rho - rho_o) <= 1e-5


def test_fixed_iteration_count_no_eps(stacker, small_stack):
    frames, _ = small_stack
    g0, g1 = oracle.grey(frames[0]), oracle.grey(frames[2])
    p = EccMatchParameters(MotionType.Homography, 3, None, 5)
    W, rho, its = stacker.find_transform_ecc(g1, g0, np.eye(3), p)
    rc, Wo, rho_o, its_o = oracle.find_transform_ecc(g1, g0, np.eye(3), oracle.MOTION_HOMOGRAPHY, 3, None, 5)
    assert its == 3 and its_o == 3
    assert synth.corner_error(W, Wo, 320, 240) <= 0.01
    # per-iteration agreement is much tighter than at an eps-terminated stop
    np.testing.assert_allclose(W, Wo, rtol=0, atol=2e-5)


def test_translated_pattern_recovers_shift(stacker):
    # closed-form known answer: smooth pattern shifted by (3, -2) px
    yy, xx = np.mgrid[0:200, 0:260].astype(np.float64)
    def pat(x, y):
        return 120 + 60 * np.sin(x / 9.0) * np.cos(y / 7.0) + 40 * np.sin((x + 2 * y) / 23.0)
    ref = np.clip(pat(xx, yy), 0, 255).astype(np.uint8)
    mov = np.clip(pat(xx + 3.0, yy - 2.0), 0, 255).astype(np.uint8)    # mov(x) = ref(x + (3,-2))
    p = EccMatchParameters(MotionType.Translation, 300, 1e-8, 5)
    W, rho, its = stacker.find_transform_ecc(mov, ref, np.eye(2, 3), p)
    assert abs(W[0, 2] - 3.0) < 0.03 and abs(W[1, 2] + 2.0) < 0.03
    assert rho > 0.999


def test_uncorrelated_images_raise_opencv_error(stacker):
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    b = np.full((64, 64), 7, np.uint8)                       # constant input: zero variance -> NaN rho
    with pytest.raises(OpenCvError):
        stacker.find_transform_ecc(a, b, np.eye(3), PARAMS)
    rc, *_ = oracle.find_transform_ecc(a, b, np.eye(3), oracle.MOTION_HOMOGRAPHY, 5000, 1e-5, 5)
    assert rc != 0


def test_criteria_without_count_or_eps_is_an_error(stacker, small_stack):
    frames, _ = small_stack
    with pytest.raises(OpenCvError):
        stacker.ecc_match(list(frames), EccMatchParameters(MotionType.Homography, None, None, 5))


def test_ecc_match_empty_list(stacker):
    with pytest.raises(NotEnoughFiles):
        stacker.ecc_match([], PARAMS)


def test_ecc_match_stack_matches_oracle(stacker, small_stack):
    frames, G = small_stack
    out, stats = stacker.ecc_match(list(frames), PARAMS, return_stats=True)
    ref, warps, iters = oracle.ecc_match(list(frames), max_count=5000, epsilon=1e-5, gauss_filt_size=5)
    for i in range(1, len(frames)):
        assert synth.corner_error(stats[i]["warp"], warps[i], 320, 240) <= 0.05
        assert synth.corner_error(stats[i]["warp"], G[i], 320, 240) <= 0.25
    # stacked output, end to end: max relative error over the pixels >= 2 px inside every warped border
    err, _ = assert_ecc_stack_close(out, ref, frames, warps, label="320x240", iters=[s["iterations"] for s in stats[1:]], iters_ref=iters[1:])
    # given the oracle's warps, the fold itself is exact to f32 round-off (<= 1e-6 per frame)
    acc = None
    for i, f in enumerate(frames):
        M = np.eye(3) if i == 0 else warps[i].astype(np.float64)
        acc = stacker.warp_accumulate(f, M, acc=acc)
    got = acc * np.float32(1.0 / len(frames))
    assert np.max(np.abs(got - ref)) <= 1e-4 * np.max(np.abs(ref))      # north-star tolerance
    assert np.max(np.abs(got - ref)) <= 4e-6


def test_ecc_match_single_frame_is_convert(stacker, small_stack):
    frames, _ = small_stack
    out = stacker.ecc_match([frames[0]], PARAMS)
    assert np.array_equal(out, oracle.convert_f32(frames[0]))


def test_ecc_match_device_resident_equals_host_fed(stacker, small_stack):
    import torch
    frames, _ = small_stack
    host = stacker.ecc_match(list(frames), PARAMS)
    dev = stacker.ecc_match(torch.from_numpy(frames).cuda(), PARAMS)
    assert np.array_equal(dev.cpu().numpy(), host)              # deterministic: fixed reduction order


def test_direct_variant_agrees_with_production(stacker, small_stack):
    # same per-pixel arithmetic, different accumulation structure (66 per-lane sums vs per-strip Y-moments): an
    # independent cross-check of the production kernel; only the f32 summation order differs
    frames, _ = small_stack
    base, s0 = stacker.ecc_match(list(frames), PARAMS, return_stats=True)
    stacker.set_option("ecc_variant", 0)
    try:
        out0, s3 = stacker.ecc_match(list(frames), PARAMS, return_stats=True)
    finally:
        stacker.set_option("ecc_variant", 3)
    for a, d in zip(s0[1:], s3[1:]):
        assert a["iterations"] == d["iterations"]
        assert synth.corner_error(a["warp"], d["warp"], 320, 240) <= 0.02
    # a caller-supplied start whose m22 is not 1 takes the direct kernel and must leave the option as it was
    g0 = oracle.grey(frames[0])
    big = np.array([[1.25, 0, -40.0], [0, 1.25, -30.0], [0, 0, 1.0]], np.float32)
    p1 = EccMatchParameters(MotionType.Homography, 1, None, 5)
    Wd, rho_d, _ = stacker.find_transform_ecc(g0, g0, big, p1)
    Ws, rho_s, _ = stacker.find_transform_ecc(g0, g0, big * np.float32(2.0), p1)     # the same map, scaled: m22 = 2
    assert abs(rho_s - rho_d) <= 1e-5
    out1, s1 = stacker.ecc_match(list(frames), PARAMS, return_stats=True)
    assert np.array_equal(out1, base)                                                  # variant restored: same bits as before
    with pytest.raises(Exception):
        stacker.set_option("ecc_variant", 1)                                           # deleted in round 2


@pytest.mark.parametrize("w,h,n,strength", [(1000, 700, 4, 1.0), (1920, 1080, 4, 4.0), (2000, 1200, 3, 12.0)])
def test_lds_ring_path_is_bit_identical_to_the_gather_path(stacker, w, h, n, strength):
    """The homography pass fetches the frame-0 rows of a column strip through a per-wave LDS ring (hand-issued LDS-DMA,
    counted waits) wherever the strip's source footprint allows it, and gathers every tap from global memory elsewhere
    (strip ends, frame borders, strong rotation: the third case has strips of both kinds). Both read the same taps and run
    the same arithmetic: warps, iteration counts and the stacked image must not differ in a single bit."""
    frames, _ = synth.make_stack(n, w, h, strength=strength)
    dev = frames.cuda()
    ring, s_ring = stacker.ecc_match(dev, PARAMS, return_stats=True)
    stacker.set_option("ecc_ring", 0)
    try:
        gather, s_gather = stacker.ecc_match(dev, PARAMS, return_stats=True)
    finally:
        stacker.set_option("ecc_ring", 1)
    assert [s["iterations"] for s in s_ring] == [s["iterations"] for s in s_gather]
    assert all(np.array_equal(a["warp"], b["warp"]) for a, b in zip(s_ring, s_gather))
    assert np.array_equal(ring.cpu().numpy(), gather.cpu().numpy())


@pytest.mark.parametrize("motion", [MotionType.Affine, MotionType.Euclidean, MotionType.Translation])
def test_lds_ring_path_other_motions(stacker, motion):
    # the same kernel template with one plain accumulator per sum: ring and gather routes must agree bit for bit here too
    frames, _ = synth.make_stack(4, 1920, 1080, strength=0.5)
    dev = frames.cuda()
    p = EccMatchParameters(motion, 30, 1e-4, 5)
    ring, s_ring = stacker.ecc_match(dev, p, return_stats=True)
    stacker.set_option("ecc_ring", 0)
    try:
        gather, s_gather = stacker.ecc_match(dev, p, return_stats=True)
    finally:
        stacker.set_option("ecc_ring", 1)
    assert [s["iterations"] for s in s_ring] == [s["iterations"] for s in s_gather]
    assert all(np.array_equal(a["warp"], b["warp"]) for a, b in zip(s_ring, s_gather))
    assert np.array_equal(ring.cpu().numpy(), gather.cpu().numpy())


def test_lds_ring_run_time_check_falls_back_to_the_gather_loop(stacker):
    """The ring's safety rests on bounds derived from a strip's corners plus a scalar check, row by row, that what a fetch
    reads has landed and has not been overwritten. A strip that fails the check is redone by the gather loop (its
    accumulators are private and cleared) and counted — it used to poison a sum with NaN and fail the whole stack with
    OpenCV's NaN error. The debug option ecc_ring_lookahead = 1 makes the check fire on ordinary strips: the results must
    still equal the gather route's bit for bit, and the production lookahead must count no fall-back at all."""
    frames, _ = synth.make_stack(5, 1920, 1080, strength=1.0)
    dev = frames.cuda()
    ref, s_ref = stacker.ecc_match(dev, PARAMS, return_stats=True)
    assert stacker.timing()["ecc_ring_fallbacks"] == 0
    stacker.set_option("ecc_ring_lookahead", 1)
    try:
        out, s_out = stacker.ecc_match(dev, PARAMS, return_stats=True)
        n_fallbacks = stacker.timing()["ecc_ring_fallbacks"]
    finally:
        stacker.set_option("ecc_ring_lookahead", 5)
    assert n_fallbacks > 0, "lookahead 1 did not provoke the run-time check: the test exercises nothing"
    assert [s["iterations"] for s in s_out] == [s["iterations"] for s in s_ref]
    assert all(np.array_equal(a["warp"], b["warp"]) for a, b in zip(s_out, s_ref))
    assert np.array_equal(out.cpu().numpy(), ref.cpu().numpy())
    stacker.set_option("ecc_ring", 0)
    try:
        gather = stacker.ecc_match(dev, PARAMS)
        assert stacker.timing()["ecc_ring_fallbacks"] == 0
    finally:
        stacker.set_option("ecc_ring", 1)
    assert np.array_equal(gather.cpu().numpy(), ref.cpu().numpy())
    with pytest.raises(Exception):
        stacker.set_option("ecc_ring_lookahead", 0)


def test_lds_ring_against_gather_on_random_start_warps(stacker):
    """The ring route decides per column strip whether its source footprint fits (window of 76 columns, lanes at most 2.5
    rows apart, source row rising by 0.6 .. 1.4 per template row, corners with w >= 1/4) and falls back to the gather loop
    otherwise. Random start homographies — rotations up to 8 degrees, scale 0.8 .. 1.25, perspective, shifts — put strips on
    every side of those limits; two fixed iterations from each start must give the same bits by both routes."""
    rng = np.random.default_rng(7)
    frames, _ = synth.make_stack(2, 1280, 960)
    g0, g1 = oracle.grey(frames.numpy()[0]), oracle.grey(frames.numpy()[1])
    p2 = EccMatchParameters(MotionType.Homography, 2, None, 5)
    w, h = 1280, 960
    n_checked = 0
    for k in range(24):
        th = np.radians(rng.uniform(-8, 8)) * (k % 3 != 0)
        sc = rng.uniform(0.8, 1.25) if k % 2 else rng.uniform(0.97, 1.03)
        C = np.array([[1, 0, w / 2], [0, 1, h / 2], [0, 0, 1.0]])
        R = np.array([[np.cos(th) * sc, -np.sin(th) * sc, rng.uniform(-30, 30)], [np.sin(th) * sc, np.cos(th) * sc, rng.uniform(-30, 30)],
                      [rng.uniform(-2e-5, 2e-5), rng.uniform(-2e-5, 2e-5), 1.0]])
        W0 = C @ R @ np.linalg.inv(C)
        W0 = (W0 / W0[2, 2]).astype(np.float32)
        W0[2, 2] = 1.0
        outs = []
        for ring in (1, 0):
            stacker.set_option("ecc_ring", ring)
            try:
                outs.append(stacker.find_transform_ecc(g0, g1, W0.copy(), p2))
            except OpenCvError as e:                       # a start this far off may fail like OpenCV's does: then by both routes
                outs.append(str(e))
            finally:
                stacker.set_option("ecc_ring", 1)
        if isinstance(outs[0], str) or isinstance(outs[1], str):
            assert isinstance(outs[0], str) and isinstance(outs[1], str), (k, outs)
            continue
        n_checked += 1
        assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1] and outs[0][2] == outs[1][2], k
    assert n_checked >= 12


def test_frames_entering_idle_slots_do_not_change_results(stacker):
    """With the templates prepared on a second stream while the first frames already iterate (device-resident stacks of
    more than 2 x ecc_slots frames) and with host-fed stacks, frames enter slots that have been idle for some launches.
    That path once let a slot's ticket count drift by a launch (the idle slot's frame was taken while other workgroups
    of the same solve launch had not read the slot yet), so a frame came out an ulp or an iteration off, run to run.
    Every run must give the bits of the all-templates-first run."""
    frames, _ = synth.make_stack(90, 640, 480)
    dev = frames.cuda()
    stacker.set_option("prep_overlap", 0)
    try:
        ref, s_ref = stacker.ecc_match(dev, PARAMS, return_stats=True)
    finally:
        stacker.set_option("prep_overlap", 1)
    ref = ref.cpu().numpy()
    pinned = frames.pin_memory()
    for rep in range(6):
        for src in (dev, pinned):
            out, st = stacker.ecc_match(src, PARAMS, return_stats=True)
            assert [s["iterations"] for s in st] == [s["iterations"] for s in s_ref], rep
            assert all(np.array_equal(a["warp"], b["warp"]) for a, b in zip(st, s_ref)), rep
            assert np.array_equal(out.cpu().numpy() if hasattr(out, "cpu") else out, ref), rep


def test_slot_count_does_not_change_results(stacker, small_stack):
    frames, _ = small_stack
    base, s0 = stacker.ecc_match(list(frames), PARAMS, return_stats=True)
    for slots in (1, 2, 3):
        stacker.set_option("ecc_slots", slots)
        try:
            out, s1 = stacker.ecc_match(list(frames), PARAMS, return_stats=True)
        finally:
            stacker.set_option("ecc_slots", 0)
        assert np.array_equal(out, base)
        assert [s["iterations"] for s in s1] == [s["iterations"] for s in s0]


def test_ecc_match_scaling_down_matches_oracle(stacker):
    frames, G = synth.make_stack(3, 640, 480)
    frames = frames.numpy()
    for motion, omotion in ((MotionType.Homography, oracle.MOTION_HOMOGRAPHY), (MotionType.Affine, oracle.MOTION_AFFINE)):
        p = EccMatchParameters(motion, 5000, 1e-5, 5)
        out, stats = stacker.ecc_match(list(frames), p, scale_down_width=240.0, return_stats=True)
        ref, warps, iters = oracle.ecc_match(list(frames), motion=omotion, scale_down_width=240.0)
        for i in (1, 2):
            assert synth.corner_error(stats[i]["warp"], warps[i], 640, 480) <= 0.1      # 0.05 px at half size
            assert stats[i]["iterations"] == int(iters[i])
        assert_ecc_stack_close(out, ref, frames, [w[:2] if motion != MotionType.Homography else w for w in warps], label="scaled %s" % motion.name,
                               iters=[s["iterations"] for s in stats[1:]], iters_ref=iters[1:])
    for i in (1, 2):                                             # full-size truth (homography run)
        assert synth.corner_error(stats[i]["warp"], G[i], 640, 480) <= 4.0 or motion != MotionType.Homography
    # scale_down between the height and the width: the greys are ENLARGED (INTER_AREA's bilinear emulation), as in the reference
    out, stats = stacker.ecc_match(list(frames), PARAMS, scale_down_width=540.0, return_stats=True)
    ref, warps, iters = oracle.ecc_match(list(frames), scale_down_width=540.0)
    for i in (1, 2):
        assert synth.corner_error(stats[i]["warp"], warps[i], 640, 480) <= 0.1
        assert stats[i]["iterations"] == int(iters[i])
    assert_ecc_stack_close(out, ref, frames, warps, label="scale_down 540 on 640x480 (enlarging)", iters=[s["iterations"] for s in stats[1:]], iters_ref=iters[1:])
    from libstacker_rs_amd import InvalidParams
    with pytest.raises(InvalidParams):
        stacker.ecc_match(list(frames), PARAMS, scale_down_width=640.0)     # lib.rs:876
    with pytest.raises(InvalidParams):
        stacker.ecc_match(list(frames), PARAMS, scale_down_width=10.0)      # lib.rs:883


def test_frame_sharded_ranks_reproduce_the_single_gpu_stack(stacker, small_stack):
    # what bench.py / a multi-GPU host does: every rank folds its contiguous range of moving frames
    # (rank 0 also frame 0), the f32 sums are added (RCCL reduce in production) and rank 0 divides by n.
    # Per-frame warps are identical to the unsharded run bit for bit; the image differs only by the order of the adds.
    import torch
    from libstacker_rs_amd.shard import shard_moving_frames
    frames, _ = small_stack
    n = len(frames)
    full, full_stats = stacker.ecc_match(list(frames), PARAMS, return_stats=True)
    for world in (2, 3):
        total = torch.zeros((frames.shape[1], frames.shape[2], 3), dtype=torch.float32, device="cuda")
        added_total = 0
        for rank in range(world):
            mine = shard_moving_frames(n, world, rank)
            acc = torch.empty_like(total)
            sub = torch.from_numpy(np.ascontiguousarray(frames[[0] + mine])).cuda()
            added, stats = stacker.ecc_match_shard(sub, PARAMS, rank == 0, acc)
            assert added == len(mine) + (1 if rank == 0 else 0)
            for j, g in enumerate(mine):
                assert np.array_equal(stats[1 + j]["warp"], full_stats[g]["warp"])
                assert stats[1 + j]["iterations"] == full_stats[g]["iterations"]
            total += acc
            added_total += added
        assert added_total == n
        out = stacker.finalize_mean(total, added_total).cpu().numpy()
        assert np.max(np.abs(out - full)) <= 1e-6


def test_ecc_match_f32_frames(stacker, small_stack):
    # CV_32FC3 inputs (float TIFF/EXR): cvtColor gives 32FC1, which findTransformECC accepts;
    # convert(CV_32F, 1/255) still scales by 1/255 (utils.rs:133)
    frames, G = small_stack
    f32 = [f.astype(np.float32) for f in frames[:3]]
    out, stats = stacker.ecc_match(f32, PARAMS, return_stats=True)
    ref, warps, iters = oracle.ecc_match(f32, max_count=5000, epsilon=1e-5, gauss_filt_size=5)
    for i in (1, 2):
        assert synth.corner_error(stats[i]["warp"], warps[i], 320, 240) <= 0.05
        assert stats[i]["iterations"] == int(iters[i])
    assert_ecc_stack_close(out, ref, f32, warps, label="f32 frames", iters=[s["iterations"] for s in stats[1:]], iters_ref=iters[1:])
    # ecc_match_scaling_down on the same float stack (round 4): the 32FC1 grey is shrunk as it is (lib.rs:896, 921); 120 halves the
    # 320 x 240 frames exactly (2 x 2 cells), 100 -> fractional coverage
    for sd in (120.0, 100.0):
        out, stats = stacker.ecc_match(f32, PARAMS, scale_down_width=sd, return_stats=True)
        ref, warps, iters = oracle.ecc_match(f32, max_count=5000, epsilon=1e-5, gauss_filt_size=5, scale_down_width=sd)
        for i in (1, 2):
            assert synth.corner_error(stats[i]["warp"], warps[i], 320, 240) <= 0.1
            assert stats[i]["iterations"] == int(iters[i])
        assert_ecc_stack_close(out, ref, f32, warps, label="f32 frames, scale_down %g" % sd, iters=[s["iterations"] for s in stats[1:]], iters_ref=iters[1:])
    # 16-bit frames: the reference's grey is 16UC1, which findTransformECC rejects -> OpenCvError
    with pytest.raises(OpenCvError):
        stacker.ecc_match([f.astype(np.uint16) for f in frames[:2]], PARAMS)
    with pytest.raises(OpenCvError):
        stacker.ecc_match([f.astype(np.uint16) for f in frames[:2]], PARAMS, scale_down_width=120.0)


@pytest.mark.parametrize("depth,gauss", [(8, 3), (8, 5), (8, 7), (16, 5)])
def test_streaming_grey_blur_is_bit_identical_to_the_tiled_kernel(stacker, depth, gauss):
    # the templates of a run of frames come from grey_blur_stream_kernel (no LDS, cross-lane halo, one launch per run); the
    # tiled kernel — itself bit-exact against the oracle (test_gpu_stages / test_gpu_fullsize) — must give the same planes,
    # hence the same ECC trajectory and the same stacked bits. Widths: several waves across, a ragged last wave, one quad row
    import torch
    from libstacker_rs_amd import KeyPointMatchParameters, RANSAC
    for w, h in ((1000, 70), (256, 33), (8, 40)):
        frames, _ = synth.make_stack(5, max(w, 64), max(h, 64), depth=depth)
        fr = frames[:, :h, :w].contiguous().cuda()
        p = EccMatchParameters(MotionType.Homography, 4, None, gauss)
        run = (lambda: stacker.ecc_match(fr, p, return_stats=True)) if depth == 8 else \
              (lambda: stacker.hybrid_match(fr, KeyPointMatchParameters(RANSAC, 5.0, 0.8, 0.9), p, return_stats=True))
        try:
            a, sa = run()
        except Exception as e:                      # a sliver this small may not correlate: both paths must then fail alike
            a, sa = None, repr(type(e))
        stacker.set_option("prep_stream", 0)
        try:
            try:
                b, sb = run()
            except Exception as e:
                b, sb = None, repr(type(e))
        finally:
            stacker.set_option("prep_stream", 1)
        if a is None or b is None:
            assert a is None and b is None and sa == sb
            continue
        assert torch.equal(a, b)
        assert all(np.array_equal(x["warp"], y["warp"]) and x["rho"] == y["rho"] for x, y in zip(sa, sb))


def test_slot_groups_do_not_change_results(stacker):
    """ecc_groups = 2 cuts the slots in two halves that run their (iterate, solve) launch sequences on two streams, sharing the
    device-side queue (one half's solve runs under the other's iteration pass). Which slot, group or launch a frame lands
    in must not show: iterations, warps and the stacked image bit for bit — device-resident, host-fed, and with fewer slots
    than frames (the queue refills both groups)."""
    frames, _ = synth.make_stack(70, 640, 480)
    dev = frames.cuda()
    pinned = frames.pin_memory()
    res = {}
    try:
        for groups in (1, 2, 2):
            stacker.set_option("ecc_groups", groups)
            for name, src, slots in (("dev", dev, 0), ("host", pinned, 0), ("dev-12-slots", dev, 12)):
                stacker.set_option("ecc_slots", slots)
                out, st = stacker.ecc_match(src, PARAMS, return_stats=True)
                cur = (out.cpu().numpy() if hasattr(out, "cpu") else out, [s["iterations"] for s in st], np.stack([s["warp"] for s in st]))
                if name in res:
                    assert cur[1] == res[name][1] and np.array_equal(cur[2], res[name][2]) and np.array_equal(cur[0], res[name][0]), (groups, name)
                else:
                    res[name] = cur
    finally:
        stacker.set_option("ecc_groups", 0)
        stacker.set_option("ecc_slots", 0)
    assert np.array_equal(res["dev"][0], res["host"][0]) and np.array_equal(res["dev"][0], res["dev-12-slots"][0])
    with pytest.raises(Exception):
        stacker.set_option("ecc_groups", 3)
