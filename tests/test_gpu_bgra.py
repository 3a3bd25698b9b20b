"""Four-channel (BGRA) stacks. imread(IMREAD_UNCHANGED) keeps a PNG's alpha plane, cvtColor(BGR2GRAY)
takes four channels and ignores the fourth (utils.rs:132-142), and everything after that — convertTo, warpPerspective,
`&acc + &warped`, the final division — treats the frame as whatever it is: the reference returns a CV_32FC4 image whose
fourth channel is the aligned, averaged alpha / 255. The oracle has no four-channel entry point; the checks compose it:
channels 0-2 of the BGRA result must be the BGR result BIT FOR BIT (alpha influences nothing), and channel 3 must be the
oracle's one-channel fold of the alpha planes under the oracle's own warps."""
import numpy as np
import pytest

import oracle
from conftest import assert_stack_close
from libstacker_rs_amd import EccMatchParameters, KeyPointMatchParameters, MotionType, RANSAC, synth

pytestmark = pytest.mark.gpu
ECC = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
KP = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)


def _bgra_stack(n=4, w=320, h=240):
    frames, G = synth.make_stack(n, w, h)
    bgr = frames.numpy()
    rng = np.random.default_rng(3)
    # a smooth alpha plane that moves with the scene (the warped alpha then lines up, as a real matte would): the red channel, inverted
    alpha = (255 - bgr[..., 2:3]).astype(np.uint8)
    alpha[..., 0] ^= rng.integers(0, 2, (n, h, w), dtype=np.uint8)             # and not a function of B, G, R alone
    return bgr, np.ascontiguousarray(np.concatenate([bgr, alpha], -1)), G


def _alpha_reference(bgra, warps, ok=None):
    """The oracle's fold of the alpha planes under its own warps: (A_0 / 255 + sum warp(A_i / 255, W_i)) / n_used."""
    n = len(bgra)
    acc = oracle.warp_frame(np.ascontiguousarray(bgra[0][..., 3]), np.eye(3))
    used = 1
    for i in range(1, n):
        if ok is not None and not ok[i]:
            continue
        acc = oracle.warp_frame(np.ascontiguousarray(bgra[i][..., 3]), np.asarray(warps[i], np.float64), acc=acc)
        used += 1
    return oracle.scale(acc, used)[..., 0]


def test_grey_of_bgra_ignores_alpha(stacker):
    bgr, bgra, _ = _bgra_stack(2)
    assert np.array_equal(stacker.grey(bgra[1]), oracle.grey(bgr[1]))
    assert np.array_equal(stacker.grey(bgra[1].astype(np.uint16) * 257), oracle.grey(bgr[1].astype(np.uint16) * 257))
    f32 = bgra[1].astype(np.float32)
    assert np.array_equal(stacker.grey(f32), oracle.grey(np.ascontiguousarray(f32[..., :3])))


def test_ecc_match_bgra(stacker):
    bgr, bgra, _ = _bgra_stack()
    out4, stats4 = stacker.ecc_match(list(bgra), ECC, return_stats=True)
    out3, stats3 = stacker.ecc_match(list(bgr), ECC, return_stats=True)
    assert out4.shape == (240, 320, 4)
    assert np.array_equal(out4[..., :3], out3)
    for a, b in zip(stats4[1:], stats3[1:]):
        assert a["iterations"] == b["iterations"] and np.array_equal(a["warp"], b["warp"])
    ref, warps, iters = oracle.ecc_match(list(bgr), max_count=5000, epsilon=1e-5, gauss_filt_size=5)
    assert [s["iterations"] for s in stats4[1:]] == [int(i) for i in iters[1:]]
    assert_stack_close(out4[..., 3], _alpha_reference(bgra, warps))
    # device-resident frames take the same route
    import torch
    dev = torch.from_numpy(bgra).to("cuda:0")
    assert np.array_equal(stacker.ecc_match(dev, ECC).cpu().numpy(), out4)


def test_keypoint_match_bgra(stacker):
    bgr, bgra, _ = _bgra_stack()
    d4, out4, stats4 = stacker.keypoint_match(list(bgra), KP, return_stats=True)
    d3, out3 = stacker.keypoint_match(list(bgr), KP)
    assert d4 == d3 == 0 and out4.shape == (240, 320, 4)
    assert np.array_equal(out4[..., :3], out3)
    d_ref, ref, Hs, status = oracle.keypoint_match(list(bgr), details=True)
    assert_stack_close(out4[..., 3], _alpha_reference(bgra, Hs, ok=[s == 0 for s in status]))
    # frames of differing size and four channels: the frame-by-frame route
    cut = [bgra[0], np.ascontiguousarray(bgra[1][:200, :300]), np.ascontiguousarray(bgra[2][:230, :310]), bgra[3]]
    d5, out5 = stacker.keypoint_match(cut, KP)
    d6, out6 = stacker.keypoint_match([np.ascontiguousarray(f[..., :3]) for f in cut], KP)
    assert d5 == d6 and out5.shape == (240, 320, 4) and np.array_equal(out5[..., :3], out6)


def test_rgba_png_files(stacker, tmp_path, write_png):
    """An RGBA PNG stack through the path-based entry points: decoded to B G R A, stacked to four channels."""
    import ctypes as C
    try:
        C.CDLL("libpng16.so.16")
    except OSError:
        pytest.skip("libpng16.so.16 is not installed here")
    bgr, bgra, _ = _bgra_stack(3)
    paths = []
    for i, f in enumerate(bgra):
        paths.append(tmp_path / f"a{i}.png")
        write_png(paths[-1], np.ascontiguousarray(f[..., [2, 1, 0, 3]]))       # the writer stores four-channel arrays as given: R G B A
    assert np.array_equal(stacker.imread(paths[1]), bgra[1])
    assert np.array_equal(stacker.ecc_match_files(paths, ECC), stacker.ecc_match(list(bgra), ECC))
    d_f, out_f = stacker.keypoint_match_files(paths, KP)
    d_a, out_a = stacker.keypoint_match(list(bgra), KP)
    assert d_f == d_a and np.array_equal(out_f, out_a)
