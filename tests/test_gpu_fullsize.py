"""Parity at BASELINE.json's full sizes (4K ECC, 1080p keypoints): the oracle where it finishes in seconds, otherwise
size-independent properties — generator ground truth, exact identities of the warp, determinism, shard invariance."""
import numpy as np
import pytest
import torch

import oracle
from conftest import assert_ecc_stack_close, assert_stack_close
from libstacker_rs_amd import EccMatchParameters, KeyPointMatchParameters, MotionType, RANSAC, synth
from libstacker_rs_amd.shard import shard_moving_frames

pytestmark = pytest.mark.gpu

# findHomography vs the oracle: inlier masks identical; H to 2e-7 relative. Not tighter because LMSolver accepts a step only
# if it lowers the f64 cost S, which resolves the minimiser to ~sqrt(eps * S / curvature) ~ 2e-8 px in translation: the
# oracle itself moves by 1.5e-8 when its points are merely re-ordered (tests/test_cpu_oracle.py::test_homography_lm_floor).
H_RTOL = 2e-7

W4K, H4K = 3840, 2160
ECC = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)           # examples/main.rs:107-112
KP = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)


@pytest.fixture(scope="module")
def stack4k():
    frames, G = synth.make_stack(9, W4K, H4K, device="cuda")
    return frames, G


def test_grey_blur_4k_bit_exact(stacker, stack4k):
    frames, _ = stack4k
    f = frames[1].cpu().numpy()
    assert np.array_equal(stacker.grey_blur_f32(frames[1], 5).cpu().numpy(), oracle.gaussian_blur_f32(oracle.grey(f), 5))


def test_warp_identities_4k(stacker, stack4k):
    frames, _ = stack4k
    f = frames[2]
    conv = oracle.convert_f32(f.cpu().numpy())
    ident = stacker.warp_accumulate(f, np.eye(3))                        # identity warp == convertTo(1/255), exactly
    assert np.array_equal(ident.cpu().numpy(), conv)
    twice = stacker.warp_accumulate(f, np.eye(3), acc=ident.clone())     # linearity of the fold: v + v is exact
    assert np.array_equal(twice.cpu().numpy(), conv + conv)
    # integer translation == shifted copy, zeros where the source falls outside (BORDER_CONSTANT 0)
    M = np.array([[1, 0, 17.0], [0, 1, -9.0], [0, 0, 1.0]])             # moves content by (+17, -9): dst(x, y) = src(x - 17, y + 9)
    got = stacker.warp_accumulate(f, M).cpu().numpy()
    ref = np.zeros_like(conv)
    ref[: H4K - 9, 17:] = conv[9:, : W4K - 17]
    assert np.array_equal(got, ref)


def test_ecc_4k_ground_truth_oracle_and_determinism(stacker, stack4k):
    frames, G = stack4k
    out, stats = stacker.ecc_match(frames, ECC, return_stats=True)
    for i in range(1, len(G)):
        assert stats[i]["status"] == 0 and 3 <= stats[i]["iterations"] <= 40
        assert synth.corner_error(stats[i]["warp"], G[i], W4K, H4K) <= 0.5       # generator ground truth
    # one frame against the CPU oracle at full size (findTransformECC on the grey images)
    g0, g1 = oracle.grey(frames[0].cpu().numpy()), oracle.grey(frames[1].cpu().numpy())
    rc, Wo, rho_o, its_o = oracle.find_transform_ecc(g1, g0, np.eye(3), oracle.MOTION_HOMOGRAPHY, 5000, 1e-5, 5)
    assert rc == 0 and stats[1]["iterations"] == its_o
    assert synth.corner_error(stats[1]["warp"], Wo, W4K, H4K) <= 0.05
    # same input, same bits; device-resident and host-fed agree
    out2 = stacker.ecc_match(frames, ECC)
    assert torch.equal(out, out2)
    host = stacker.ecc_match(list(frames[:3].cpu().numpy()), ECC)
    dev3 = stacker.ecc_match(frames[:3], ECC)
    assert np.array_equal(dev3.cpu().numpy(), host)


def test_ecc_4k_stacked_image_matches_the_oracle(stacker, stack4k):
    """North star: "stacked output within 1e-4 relative" at the metric's frame size. ecc_match on 4 x 4K frames end to end
    (prep, ECC, warpPerspective, fold, 1/n) against the oracle's ecc_match: iteration counts, warps (<= 0.05 px) and the
    stacked PIXELS — max |a - b| / max(|b|, 1e-3) over the pixels >= 2 px inside every warped border; the bar is 1e-4 or,
    where one f32 ulp of the warp matrix is worth more than that (x ~ 3800: 2.3e-4 px), 3 x the oracle's own 1-ulp floor
    (conftest.assert_ecc_stack_close prints both numbers; DESIGN.md section 2 carries them)."""
    frames, _ = stack4k
    sub = frames[:4]
    host = [f for f in sub.cpu().numpy()]
    out, stats = stacker.ecc_match(sub, ECC, return_stats=True)
    assert stacker.timing()["ecc_ring_fallbacks"] == 0
    ref, warps, iters = oracle.ecc_match(host, max_count=5000, epsilon=1e-5, gauss_filt_size=5, n_threads=4)
    for i in range(1, 4):
        assert stats[i]["iterations"] == int(iters[i])
        assert synth.corner_error(stats[i]["warp"], warps[i], W4K, H4K) <= 0.05
    assert_ecc_stack_close(out.cpu().numpy(), ref, host, warps, label="4 x 3840x2160",
                           iters=[s["iterations"] for s in stats[1:]], iters_ref=[int(k) for k in iters[1:]])


def test_warp_accumulate_4k_u8_vs_oracle(stacker, stack4k):
    # the fold's kernel at the metric's size on a non-identity homography (rotation, scale, perspective, sub-pixel shift):
    # u8 fast path vs the oracle's warp_frame with the same matrix, <= 1e-6 abs (identical f32 op sequence)
    frames, G = stack4k
    f = frames[3]
    th = np.radians(0.4)
    C = np.array([[1, 0, W4K / 2], [0, 1, H4K / 2], [0, 0, 1.0]])
    M = C @ np.array([[1.003 * np.cos(th), -1.003 * np.sin(th), 6.37], [1.003 * np.sin(th), 1.003 * np.cos(th), -4.81],
                      [1.1e-6, -0.7e-6, 1.0]]) @ np.linalg.inv(C)
    fh = f.cpu().numpy()
    for mat in (M, np.asarray(G[3], np.float64)):
        got = stacker.warp_accumulate(f, mat).cpu().numpy()
        ref = oracle.warp_frame(fh, mat)
        assert np.max(np.abs(got - ref)) <= 1e-6
    base = oracle.convert_f32(frames[0].cpu().numpy())               # and folded onto an existing accumulator
    got = stacker.warp_accumulate(f, M, acc=torch.from_numpy(base.copy()).cuda()).cpu().numpy()
    assert np.max(np.abs(got - oracle.warp_frame(fh, M, acc=base.copy()))) <= 1e-6


def test_ecc_4k_shard_invariance(stacker, stack4k):
    # 8 moving frames over 2 ranks: every frame is summed over its fixed workgroup partition (288 per 4K frame), so its
    # warp is bit-identical to the single-GPU run for any split.
    frames, _ = stack4k
    n = frames.shape[0]
    full, full_stats = stacker.ecc_match(frames, ECC, return_stats=True)
    total = torch.zeros((H4K, W4K, 3), dtype=torch.float32, device="cuda")
    for rank in range(2):
        mine = shard_moving_frames(n, 2, rank)
        acc = torch.empty_like(total)
        added, stats = stacker.ecc_match_shard(frames[[0] + mine], ECC, rank == 0, acc)
        for j, g in enumerate(mine):
            assert np.array_equal(stats[1 + j]["warp"], full_stats[g]["warp"])       # bit-identical per-frame result
        total += acc
    out = stacker.finalize_mean(total, n)
    assert float((out - full).abs().max()) <= 1e-6


def test_keypoint_1080p_oracle_and_ground_truth(stacker):
    frames, G = synth.make_stack(4, 1920, 1080, device="cuda")
    g1 = oracle.grey(frames[1].cpu().numpy())
    kp, de = stacker.orb_detect_and_compute(g1, 4096)
    kpo, deo = oracle.orb_detect_and_compute(g1)
    assert len(kp) >= 450 and np.array_equal(kp, kpo) and np.array_equal(de, deo)    # bit-exact at full size
    dropped, out, stats = stacker.keypoint_match(frames, KP, return_stats=True)
    assert dropped == 0
    for i in range(1, 4):
        assert synth.corner_error(stats[i]["warp"], G[i], 1920, 1080) <= 1.0
    # worker count does not change anything: frames are independent and folded in frame order
    stacker.set_option("kp_workers", 1)
    try:
        d1, out1 = stacker.keypoint_match(frames, KP)
    finally:
        stacker.set_option("kp_workers", 12)
    assert d1 == 0 and torch.equal(out, out1)


def _jpeg_roundtrip(frames_u8):
    """JPEG-encode/decode (quality 90) with the Pillow of the image's conda interpreter, if there is one: the reference's
    inputs are camera JPEGs (BASELINE configs[0]); returns None where that interpreter is absent."""
    import os
    import subprocess
    import tempfile
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py):
        return None
    with tempfile.TemporaryDirectory() as d:
        src, dst = os.path.join(d, "in.npy"), os.path.join(d, "out.npy")
        np.save(src, frames_u8)
        code = ("import io, sys, numpy as np\nfrom PIL import Image\na = np.load(sys.argv[1])\nout = []\n"
                "for f in a:\n    b = io.BytesIO(); Image.fromarray(f[..., ::-1]).save(b, format='JPEG', quality=90)\n"
                "    out.append(np.asarray(Image.open(io.BytesIO(b.getvalue())))[..., ::-1])\n"
                "np.save(sys.argv[2], np.stack(out))\n")
        r = subprocess.run([py, "-c", code, src, dst], capture_output=True, timeout=120)
        if r.returncode != 0 or not os.path.exists(dst):
            return None
        return np.ascontiguousarray(np.load(dst))


def test_config0_nine_800x600_frames_keypoint_match(stacker):
    # BASELINE configs[0]: the image_stacking_py set (9 JPEGs of about 800x600) is not in the container; SURVEY 8d
    # substitutes 9 synthetic frames of that size, JPEG round-tripped where a Pillow is available
    frames, G = synth.make_stack(9, 800, 600)
    fr = frames.numpy()
    jp = _jpeg_roundtrip(fr)
    if jp is not None:
        assert jp.shape == fr.shape and np.abs(jp.astype(int) - fr.astype(int)).mean() < 6    # JPEG noise, same content
        fr = jp
    dropped, out, stats = stacker.keypoint_match(list(fr), KP, return_stats=True)
    d_o, ref, Hs, status = oracle.keypoint_match(list(fr), details=True)
    assert dropped == d_o == 0
    for i in range(1, 9):
        assert np.allclose(stats[i]["warp"], Hs[i], rtol=H_RTOL, atol=1e-9)
        assert synth.corner_error(stats[i]["warp"], G[i], 800, 600) <= (1.5 if jp is not None else 1.0)
    assert_stack_close(out, ref, bulk=9e-6)                        # <= 1e-6 per folded frame
    e_out, e_stats = stacker.ecc_match(list(fr), ECC, return_stats=True)
    e_ref, warps, iters = oracle.ecc_match(list(fr), max_count=5000, epsilon=1e-5, gauss_filt_size=5)
    assert_ecc_stack_close(e_out, e_ref, fr, warps, label="config0 800x600", iters=[s["iterations"] for s in e_stats[1:]], iters_ref=iters[1:])


def test_ecc_1080p_sixteen_slot_plan(stacker):
    # BASELINE configs[2]'s shape: 1920x1080, ecc_match Homography / 5000 / 1e-5 / gauss 5. 33 frames = 32 moving frames:
    # all in flight at once by default (32 slots); with ecc_slots = 5 the device queue refills slots six times over. Either
    # way, and split over 2 ranks, every frame keeps its own fixed workgroup partition: identical bits.
    frames, G = synth.make_stack(33, 1920, 1080, device="cuda")
    n = frames.shape[0]
    out, stats = stacker.ecc_match(frames, ECC, return_stats=True)
    for i in range(1, n):
        assert stats[i]["status"] == 0 and 3 <= stats[i]["iterations"] <= 60
        assert synth.corner_error(stats[i]["warp"], G[i], 1920, 1080) <= 0.25            # generator ground truth
    # the oracle on a frame of the first slot generation and on one that took a refilled slot
    g0 = oracle.grey(frames[0].cpu().numpy())
    for i in (1, 29):
        gi = oracle.grey(frames[i].cpu().numpy())
        rc, Wo, rho_o, its_o = oracle.find_transform_ecc(gi, g0, np.eye(3), oracle.MOTION_HOMOGRAPHY, 5000, 1e-5, 5)
        assert rc == 0 and stats[i]["iterations"] == its_o
        assert synth.corner_error(stats[i]["warp"], Wo, 1920, 1080) <= 0.05
        assert abs(stats[i]["rho"] - rho_o) <= 1e-5
    # end to end against the oracle's stack (4 frames: the oracle finishes in seconds), max relative error
    sub = frames[:4]
    o4, s4 = stacker.ecc_match(sub, ECC, return_stats=True)
    fr4 = list(sub.cpu().numpy())
    ref, warps, iters = oracle.ecc_match(fr4, max_count=5000, epsilon=1e-5, gauss_filt_size=5)
    assert_ecc_stack_close(o4.cpu().numpy(), ref, fr4, warps, label="1080p", iters=[s["iterations"] for s in s4[1:]], iters_ref=iters[1:])
    # shard invariance with a real accumulator reduce
    total = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda")
    for rank in range(2):
        mine = shard_moving_frames(n, 2, rank)
        assert len(mine) == 16
        acc = torch.empty_like(total)
        added, st = stacker.ecc_match_shard(frames[[0] + mine], ECC, rank == 0, acc)
        for j, g in enumerate(mine):
            assert np.array_equal(st[1 + j]["warp"], stats[g]["warp"]) and st[1 + j]["iterations"] == stats[g]["iterations"]
        total += acc
    assert float((stacker.finalize_mean(total, n) - out).abs().max()) <= 1e-6
    assert torch.equal(out, stacker.ecc_match(frames, ECC))                                 # same input, same bits
    stacker.set_option("ecc_slots", 5)                                                      # queue refill path: 32 frames through 5 slots
    try:
        out5, st5 = stacker.ecc_match(frames, ECC, return_stats=True)
    finally:
        stacker.set_option("ecc_slots", 0)
    assert torch.equal(out5, out) and all(np.array_equal(a["warp"], b["warp"]) for a, b in zip(st5, stats))


def test_hybrid_4k_16bit(stacker):
    # BASELINE configs[4]'s shape: 3840x2160 16-bit BGR, ORB-seeded ECC (an extension beyond the reference).
    from libstacker_rs_amd import EccMatchParameters as EP
    frames, G = synth.make_stack(6, W4K, H4K, device="cuda", depth=16)
    n = frames.shape[0]
    ecc = EP(MotionType.Homography, 200, 1e-5, 5)
    out, stats = stacker.hybrid_match(frames, KP, ecc, return_stats=True)
    for i in range(1, n):
        assert stats[i]["status"] == 0 and stats[i]["n_matches"] >= 100 and stats[i]["n_inliers"] >= 50
        assert synth.corner_error(stats[i]["warp"], G[i], W4K, H4K) <= 0.5                # generator ground truth
    assert 0.0 <= float(out.min()) and float(out.max()) <= 1.0 + 1e-6                     # alpha = 1/65535
    assert torch.equal(out, stacker.hybrid_match(frames, KP, ecc))                        # deterministic
    # stage parity at full size. bgr16 -> grey8 feeds ORB: one differing grey level would move corners, so the
    # keypoint count of the fused bgr16_to_grey8 + ORB path must equal the oracle's on (grey16 + 128) / 257
    f1 = frames[1].cpu().numpy()
    g16 = oracle.grey(f1)
    g8 = ((g16.astype(np.uint32) + 128) // 257).astype(np.uint8)
    kpo, _ = oracle.orb_detect_and_compute(g8)
    assert stats[1]["n_keypoints"] == len(kpo)
    # ECC's template of a 16-bit frame: GaussianBlur(float(grey16)) in one fused pass, bit-exact
    assert np.array_equal(stacker.grey_blur_f32(frames[1], 5).cpu().numpy(), oracle.gaussian_blur_f32(g16.astype(np.float32), 5))
    # warp_accumulate_u16c3 (interior fast path + border waves) against the oracle, both convert scales
    M = np.linalg.inv(G[1])
    got = stacker.warp_accumulate(frames[1], M, alpha=1.0 / 65535.0).cpu().numpy()
    assert np.max(np.abs(got - oracle.warp_frame(f1, M, alpha=1.0 / 65535.0))) <= 1e-6
    got = stacker.warp_accumulate(frames[1], M).cpu().numpy()                             # the reference's literal 1/255: range 0..257
    assert np.max(np.abs(got - oracle.warp_frame(f1, M))) <= 257e-6
    # ORB seed shortens ECC at 4K as well
    cold = EP(MotionType.Homography, 200, 1e-5, 5)
    assert sum(s["iterations"] for s in stats[1:]) <= 8 * (n - 1)


@pytest.mark.timeout(900)
def test_config3_256_frames_4k_ecc_match_and_eight_way_shards(stacker):
    # BASELINE configs[3] at its full size on one GPU: 256 x 3840x2160 BGR u8 (6.4 GB of frames + 8.5 GB of templates),
    # ecc_match Homography / 5000 / 1e-5 / gauss 5. Every frame against the generator's ground truth; then the stack cut
    # into the 8 contiguous ranges the 8 GPUs of a node would get (32 frames each, run one after the other on this GPU)
    # and reduced: per-frame warps bit-identical, image equal up to the order of the f32 adds.
    n = 256
    frames, G = synth.make_stack(n, W4K, H4K, device="cuda")
    out, stats = stacker.ecc_match(frames, ECC, return_stats=True)
    worst = 0.0
    for i in range(1, n):
        assert stats[i]["status"] == 0 and 3 <= stats[i]["iterations"] <= 40
        worst = max(worst, synth.corner_error(stats[i]["warp"], G[i], W4K, H4K))
    assert worst <= 0.5, worst
    total = torch.zeros((H4K, W4K, 3), dtype=torch.float32, device="cuda")
    added_total = 0
    for rank in range(8):
        mine = shard_moving_frames(n, 8, rank)
        acc = torch.empty_like(total)
        added, st = stacker.ecc_match_shard(frames[[0] + mine], ECC, rank == 0, acc)
        for j, g in enumerate(mine):
            assert np.array_equal(st[1 + j]["warp"], stats[g]["warp"]) and st[1 + j]["iterations"] == stats[g]["iterations"]
        total += acc
        added_total += added
    assert added_total == n
    assert float((stacker.finalize_mean(total, n) - out).abs().max()) <= 2e-6
    # run-to-run determinism at the size where it once failed: the LDS ring of the ECC pass took a row's extreme source rows
    # from its end lanes, an interior lane rounded one row further, and in 3 % of the runs read a row still in flight
    # (frame 254 of this very stack came out an ulp off). 40 more runs, every warp bit for bit.
    ref_w = np.stack([s["warp"] for s in stats])
    for rep in range(40):
        out2, st2 = stacker.ecc_match(frames, ECC, return_stats=True)
        assert np.array_equal(np.stack([s["warp"] for s in st2]), ref_w), rep
        assert torch.equal(out2, out), rep
    del frames, total
    torch.cuda.empty_cache()


@pytest.mark.timeout(1500)
def test_config4_1024_frames_4k_16bit_hybrid(stacker):
    # BASELINE configs[4] at its full size: 1024 x 3840x2160 16-bit BGR (51 GB of frames, 34 GB of templates, ORB in
    # batches inside a 32 GiB workspace), ORB-seeded ECC refine (an extension beyond the reference).
    # Size-independent checks: every frame against the generator's ground truth, unit range of the mean, determinism of
    # a sub-range against the full run (per-frame results do not depend on what else is in the stack).
    from libstacker_rs_amd import EccMatchParameters as EP
    n = 1024
    chunks = []
    G = []
    for c0 in range(0, n, 128):                            # generate in chunks: torch.stack of everything would double the footprint
        f, g = synth.make_stack(0, W4K, H4K, device="cuda", depth=16, indices=list(range(c0, c0 + 128)))
        chunks.append(f); G.append(g)
    G = np.concatenate(G)
    frames = [fr for c in chunks for fr in c.unbind(0)]    # a list of device tensors: not evenly spaced in memory
    ecc = EP(MotionType.Homography, 200, 1e-5, 5)
    out, stats = stacker.hybrid_match(frames, KP, ecc, return_stats=True)
    worst = 0.0
    for i in range(1, n):
        assert stats[i]["status"] == 0 and stats[i]["n_inliers"] >= 50
        worst = max(worst, synth.corner_error(stats[i]["warp"], G[i], W4K, H4K))
    assert worst <= 0.5, worst
    assert 0.0 <= float(out.min()) and float(out.max()) <= 1.0 + 1e-6
    sub = [frames[0]] + frames[700:716]
    o2, s2 = stacker.hybrid_match(sub, KP, ecc, return_stats=True)
    for j in range(1, len(sub)):
        assert np.array_equal(s2[j]["warp"], stats[699 + j]["warp"])
    del frames, chunks
    torch.cuda.empty_cache()
