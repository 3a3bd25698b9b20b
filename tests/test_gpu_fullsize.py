"""Parity at BASELINE.json's full sizes (4K ECC, 1080p keypoints): the oracle where it finishes in seconds, otherwise
size-independent properties — generator ground truth, exact identities of the warp, determinism, shard invariance."""
import numpy as np
import pytest
import torch

import oracle
from libstacker_rs_amd import EccMatchParameters, KeyPointMatchParameters, MotionType, RANSAC, synth
from libstacker_rs_amd.shard import shard_moving_frames

pytestmark = pytest.mark.gpu

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
    assert rc == 0 and abs(stats[1]["iterations"] - its_o) <= 1
    assert synth.corner_error(stats[1]["warp"], Wo, W4K, H4K) <= 0.05
    # same input, same bits; device-resident and host-fed agree
    out2 = stacker.ecc_match(frames, ECC)
    assert torch.equal(out, out2)
    host = stacker.ecc_match(list(frames[:3].cpu().numpy()), ECC)
    dev3 = stacker.ecc_match(frames[:3], ECC)
    assert np.array_equal(dev3.cpu().numpy(), host)


def test_ecc_4k_shard_invariance(stacker, stack4k):
    # 8 moving frames over 2 ranks = 4 each = the slot count at 4K: every frame is summed over the same workgroup
    # partition as in the single-GPU run, so its warp is bit-identical. (A shard with FEWER moving frames than slots
    # spreads each frame over more workgroups and differs at f32 round-off, ~1e-7: DESIGN.md section 4.)
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
        assert np.allclose(stats[i]["warp"], Hs[i], rtol=0, atol=1e-10)
        assert synth.corner_error(stats[i]["warp"], G[i], 800, 600) <= (1.5 if jp is not None else 1.0)
    assert np.max(np.abs(out - ref)) <= 9e-6                       # <= 1e-6 per folded frame
    assert np.max(np.abs(out - ref)) <= 1e-4 * np.max(np.abs(ref))  # north-star tolerance
    e_out = stacker.ecc_match(list(fr), ECC)
    e_ref, warps, iters = oracle.ecc_match(list(fr), max_count=5000, epsilon=1e-5, gauss_filt_size=5)
    rel = np.abs(e_out - e_ref) / np.maximum(np.abs(e_ref), 1e-3)
    assert np.percentile(rel[4:-4, 4:-4], 99.5) < 4e-3
