"""CPU tests: the oracle against the committed golden vectors and against closed-form known answers.

The oracle is the repo's CPU restatement of the OpenCV algorithms libstacker calls ("parity
unpinned": no OpenCV and no reference fixtures exist here). These tests pin it to
(a) tests/golden/golden_v1.npz (drift guard), (b) analytic identities, (c) the synthetic generator's
ground-truth homographies.
"""
import os

import numpy as np
import pytest

import oracle
from libstacker_rs_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def crop(a):
    return a[:48, :64]


# ---- golden vectors ---------------------------------------------------------------------------------
def test_golden_integer_stages_bit_exact(gold):
    f0 = gold["frames"][0]
    assert np.array_equal(oracle.grey(f0), gold["grey"])
    assert np.array_equal(oracle.grey(f0.astype(np.uint16) * 257), gold["grey16"])
    assert np.array_equal(oracle.convert_f32(f0[:16, :16]), gold["convert"])
    for k in (3, 5, 7):
        assert np.array_equal(crop(oracle.gaussian_blur_f32(gold["grey"], k)), gold[f"blur{k}"])
    np.testing.assert_allclose(crop(oracle.gaussian_blur_f32(gold["grey"], 9)), gold["blur9"], rtol=1e-6)
    gx, gy = oracle.gradients(oracle.gaussian_blur_f32(gold["grey"], 5))
    assert np.array_equal(crop(gx), gold["grad_x"]) and np.array_equal(crop(gy), gold["grad_y"])


def test_golden_warps(gold):
    f0 = gold["frames"][0]
    M, A = gold["warp_M"], gold["warp_A"]
    assert np.array_equal(crop(oracle.warp_frame(f0, M)), gold["warp_exact"])
    assert np.array_equal(crop(oracle.warp_frame(f0, M, subpixel_bits=5)), gold["warp_classic"])
    assert np.array_equal(crop(oracle.warp_frame(f0, M, border_mode=oracle.BORDER_REFLECT_101)), gold["warp_reflect"])
    assert np.array_equal(crop(oracle.warp_frame(f0, A, is_affine=True)), gold["warp_affine"])
    # exact f32 path and the classic 1/32-px path agree to the quantisation error
    assert np.max(np.abs(gold["warp_exact"] - gold["warp_classic"])) < 0.02


def test_golden_ecc(gold):
    fr = gold["frames"]
    g0, g1 = oracle.grey(fr[0]), oracle.grey(fr[1])
    for name, mot in (("homography", 3), ("affine", 2), ("euclidean", 1), ("translation", 0)):
        rc, W, rho, its = oracle.find_transform_ecc(g1, g0, np.eye(3 if mot == 3 else 2, 3), mot, 5, None, 5)
        assert rc == 0 and its == 5
        np.testing.assert_allclose(W, gold[f"ecc_{name}_warp"], rtol=0, atol=1e-6)
        assert abs(rho - float(gold[f"ecc_{name}_rho"])) < 1e-9
    img, warps, iters = oracle.ecc_match(list(fr))
    assert list(iters) == list(gold["ecc_match_iters"])
    np.testing.assert_allclose(warps, gold["ecc_match_warps"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(crop(img), gold["ecc_match_image"], rtol=0, atol=2e-6)
    for i in (1, 2):    # and the recovered warps agree with the generator's truth
        assert synth.corner_error(warps[i], gold["truth_G"][i], 160, 120) < 0.25


def test_golden_keypoint_path(gold):
    fr = gold["frames"]
    g0, g1 = oracle.grey(fr[0]), oracle.grey(fr[1])
    kp0, de0 = oracle.orb_detect_and_compute(g0)
    kp1, de1 = oracle.orb_detect_and_compute(g1)
    assert np.array_equal(kp0, gold["orb_kp0"]) and np.array_equal(de0, gold["orb_de0"])
    assert np.array_equal(kp1, gold["orb_kp1"]) and np.array_equal(de1, gold["orb_de1"])
    assert np.array_equal(oracle.bf_knn2_hamming(de0, de1), gold["knn01"])
    assert np.array_equal(np.array(oracle.rng_sequence(8), np.uint32), gold["rng_first8"])
    d, img, Hs, status = oracle.keypoint_match(list(fr), details=True)
    assert d == int(gold["kp_match_dropped"]) == 0
    np.testing.assert_allclose(Hs, gold["kp_match_H"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(crop(img), gold["kp_match_image"], rtol=0, atol=2e-6)
    for i in (1, 2):
        assert synth.corner_error(Hs[i], gold["truth_G"][i], 160, 120) < 1.0


# ---- closed-form known answers -------------------------------------------------------------------------
def test_grey_formula_known_pixels():
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 77]]], np.uint8)
    exp = [(b * 3735 + g * 19235 + r * 9798 + 16384) >> 15 for b, g, r in px[0].astype(int)]
    assert list(oracle.grey(px)[0]) == exp and exp[0] == 255 and exp[2] == 29 and exp[3] == 150 and exp[4] == 76
    px16 = px.astype(np.uint16) * 257
    exp16 = [(b * 1868 + g * 9617 + r * 4899 + 8192) >> 14 for b, g, r in px16[0].astype(int)]
    assert list(oracle.grey(px16)[0]) == exp16
    f = np.array([[[0.5, 0.25, 1.0]]], np.float32)
    assert oracle.grey(f)[0, 0] == np.float32(np.float32(np.float32(0.5) * np.float32(0.114) + np.float32(0.25) * np.float32(0.587)) + np.float32(1.0) * np.float32(0.299))


def test_convert_is_multiply_by_f32_reciprocal():
    a = np.arange(256, dtype=np.uint8).reshape(16, 16)
    assert np.array_equal(oracle.convert_f32(a), a.astype(np.float32) * np.float32(1.0 / 255.0))
    assert oracle.convert_f32(np.array([[65535]], np.uint16))[0, 0] == np.float32(65535) * np.float32(1 / 255.0)   # range 0..257 quirk


def test_gaussian_fixed_taps_and_constant_image():
    lib = oracle.lib()
    import ctypes as C
    for n, taps in ((1, [1]), (3, [.25, .5, .25]), (5, [.0625, .25, .375, .25, .0625]),
                    (7, [.03125, .109375, .21875, .28125, .21875, .109375, .03125])):
        k = (C.c_float * n)()
        assert lib.orc_gaussian_kernel(n, k) == 0 and list(k) == taps
    k = (C.c_float * 9)()
    lib.orc_gaussian_kernel(9, k)
    assert abs(sum(k) - 1) < 1e-6 and list(k) == list(k)[::-1]            # sigma = 0.3*((9-1)*0.5-1)+0.8 = 1.7
    assert lib.orc_gaussian_kernel(4, k) != 0
    const = np.full((9, 11), 200, np.uint8)
    assert np.array_equal(oracle.gaussian_blur_f32(const, 5), np.full((9, 11), 200, np.float32))
    gx, gy = oracle.gradients(np.tile(np.arange(8, dtype=np.float32) * 3, (6, 1)))
    assert np.all(gx[:, 1:-1] == 3) and np.all(gx[:, [0, -1]] == 0) and np.all(gy == 0)   # REFLECT_101: zero at the rim


def test_warp_identity_shift_and_singular():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    assert np.array_equal(oracle.warp_frame(img, np.eye(3)), oracle.convert_f32(img))
    M = np.array([[1, 0, 4], [0, 1, 2], [0, 0, 1.0]])
    ref = np.zeros((20, 30, 3), np.float32)
    ref[2:, 4:] = oracle.convert_f32(img)[:-2, :-4]
    assert np.array_equal(oracle.warp_frame(img, M), ref)
    assert np.array_equal(oracle.warp_frame(img, np.zeros((3, 3)), border_value=(0.5, 0.5, 0.5, 0)),
                          np.full((20, 30, 3), 0.5, np.float32))
    half = oracle.warp_frame(img, np.array([[1, 0, 0.5], [0, 1, 0], [0, 0, 1.0]]))      # half-pixel shift = mean of neighbours
    c = oracle.convert_f32(img)
    np.testing.assert_allclose(half[:, 1:], 0.5 * (c[:, :-1] + c[:, 1:]), atol=1e-7)
    acc = np.ones((20, 30, 3), np.float32)
    assert np.array_equal(oracle.warp_frame(img, np.eye(3), acc=acc), np.float32(1) + c)


def test_scale_is_multiply_by_reciprocal():
    a = np.array([1.0, 2.0, 3.0, 1e-3], np.float32)
    assert np.array_equal(oracle.scale(a, 3), a * np.float32(1.0 / 3.0))


def test_ecc_recovers_known_translation_and_reports_failures():
    yy, xx = np.mgrid[0:120, 0:160].astype(np.float64)

    def pat(x, y):
        return 120 + 60 * np.sin(x / 9.0) * np.cos(y / 7.0) + 40 * np.sin((x + 2 * y) / 23.0)
    ref = np.clip(pat(xx, yy), 0, 255).astype(np.uint8)
    mov = np.clip(pat(xx + 3.0, yy - 2.0), 0, 255).astype(np.uint8)
    rc, W, rho, its = oracle.find_transform_ecc(mov, ref, np.eye(2, 3), oracle.MOTION_TRANSLATION, 300, 1e-8, 5)
    assert rc == 0 and abs(W[0, 2] - 3) < 0.03 and abs(W[1, 2] + 2) < 0.03 and rho > 0.999
    rc, *_ = oracle.find_transform_ecc(mov, np.full_like(ref, 9), np.eye(3), oracle.MOTION_HOMOGRAPHY, 50, 1e-5, 5)
    assert rc == 1                                    # zero variance -> NaN rho -> StsNoConv
    rc, *_ = oracle.find_transform_ecc(mov, ref, np.eye(3), oracle.MOTION_HOMOGRAPHY, None, None, 5)
    assert rc == 3                                    # criteria without COUNT or EPS
    rc, W, rho, its = oracle.find_transform_ecc(mov, ref, np.eye(3), oracle.MOTION_HOMOGRAPHY, 7, None, 5)
    assert rc == 0 and its == 7                        # COUNT only: exactly max_count iterations


def test_orb_structure():
    fr, _ = synth.make_stack(1, 640, 480)
    g = oracle.grey(fr[0].numpy())
    kp, de = oracle.orb_detect_and_compute(g)
    assert list(np.bincount(kp[:, 5].astype(int), minlength=8)) == [109, 90, 75, 63, 52, 44, 36, 31]
    ws, hs, sc = oracle.orb_level_sizes(640, 480)
    assert (ws[0], hs[0], ws[1], hs[1]) == (640, 480, 533, 400)
    assert np.allclose(kp[:, 2], 31 * np.array(sc, np.float32)[kp[:, 5].astype(int)])
    for l in range(8):                                  # every keypoint is >= 31 px from its level's border
        m = kp[:, 5] == l
        x, y = kp[m, 0] / sc[l], kp[m, 1] / sc[l]
        assert x.min() >= 31 - 1e-3 and x.max() < ws[l] - 31 and y.min() >= 31 - 1e-3 and y.max() < hs[l] - 31
    assert (kp[:, 3] >= 0).all() and (kp[:, 3] <= 360).all()
    # FAST score map: a bright isolated dot is not a corner (its ring is uniformly darker -> it IS one), check symmetry
    img = np.full((32, 32), 50, np.uint8)
    img[16, 16] = 200
    s = oracle.fast_score_map(img)
    assert s[16, 16] == 149 and s.sum() == 149            # centre brighter than all 16 ring pixels by 150 -> score 149


def test_resize_linear_exact_properties():
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (48, 60), dtype=np.uint8)
    assert np.array_equal(oracle.resize_linear_exact(a, 60, 48), a)                  # same size: identity
    c = np.full((30, 36), 77, np.uint8)
    assert np.all(oracle.resize_linear_exact(c, 30, 25) == 77)
    r = oracle.resize_linear_exact(a, 50, 40)
    assert r.shape == (40, 50) and int(r.min()) >= int(a.min()) and int(r.max()) <= int(a.max())


def test_knn_ties_and_short_train():
    q = np.zeros((2, 32), np.uint8)
    t = np.zeros((3, 32), np.uint8)
    t[0, 0] = 0b111
    t[2, 0] = 0b1
    out = oracle.bf_knn2_hamming(q, t)
    assert list(out[0]) == [1, 0, 2, 1]
    t[2, 0] = 0
    assert list(oracle.bf_knn2_hamming(q, t)[0]) == [1, 0, 2, 0]                   # tie: lower index first
    assert list(oracle.bf_knn2_hamming(q, t[:1])[0]) == [0, 3, -1, -1]


def test_find_homography_exact_and_ransac():
    Ht = np.array([[0.97, -0.05, 12.0], [0.04, 1.03, -7.0], [3e-5, 1e-5, 1.0]])
    rng = np.random.default_rng(2)
    src = rng.uniform(0, 500, (80, 2)).astype(np.float32)
    p = np.c_[src, np.ones(80)] @ Ht.T
    dst = (p[:, :2] / p[:, 2:]).astype(np.float32)
    H, mask = oracle.find_homography(src[:4], dst[:4], 8, 3.0)
    assert synth.corner_error(H, Ht, 500, 500) < 2e-2 and mask.all()
    bad = dst.copy()
    bad[:20] += 50
    H, mask = oracle.find_homography(src, bad, 8, 3.0)
    assert mask[:20].sum() == 0 and mask[20:].all() and synth.corner_error(H, Ht, 500, 500) < 0.05
    assert abs(H[2, 2] - 1) < 1e-12
    with pytest.raises(ValueError):
        oracle.find_homography(src[:3], dst[:3], 8, 3.0)


def test_sharpness_known_answers():
    # constant image: every derivative filter gives 0, variance 0
    flat = np.full((20, 30), 77, np.uint8)
    assert [oracle.sharpness(flat, m, 3) for m in range(4)] == [0.0, 0.0, 0.0, 0.0]
    # GLVN of a half 0 / half 100 image: mean 50, variance 2500 -> 50
    half = np.zeros((10, 10), np.uint8); half[:, 5:] = 100
    assert oracle.sharpness(half, 3) == 50.0
    # horizontal ramp p = 3x: Sobel(3) gx = 8 * 3 = 24 in the interior and 0 on the two reflected border columns,
    # gy = 0 -> TENG = 24^2 * (w - 2) / w ; the second derivative is 0 inside, |(-1)(3) + 2(0) - 3| = 6 at x = 0
    # (and the same at x = w-1 by symmetry) -> LAPM = 12 / w
    w, h = 16, 9
    ramp = np.tile((3 * np.arange(w)).astype(np.uint8), (h, 1))
    assert oracle.sharpness(ramp, 2, 3) == 24.0 ** 2 * (w - 2) / w
    assert oracle.sharpness(ramp, 0) == pytest.approx(12.0 / w, rel=0, abs=1e-15)
    # single bright pixel (value 8) in a 5x5 zero image, Laplacian(ksize 3) kernel [2 0 2; 0 -8 0; 2 0 2]:
    # values -64 at the pixel, 16 at the four diagonal neighbours -> mean = 0, E[v^2] = (4096 + 4 * 256) / 25
    dot = np.zeros((5, 5), np.uint8); dot[2, 2] = 8
    assert oracle.sharpness(dot, 1) == pytest.approx((4096 + 1024) / 25.0, rel=1e-15)
    with pytest.raises(ValueError):
        oracle.sharpness(flat, 2, 2)


def test_homography_lm_floor():
    """How well is findHomography's result defined at all? Re-ordering the correspondences changes nothing but the
    order of the oracle's f64 sums, yet H moves by up to ~1.5e-8 relative: LMSolver accepts a step only when it lowers
    the f64 cost, so the minimiser is resolved to ~sqrt(eps * S / curvature). This is the floor behind the 2e-7
    tolerance of the GPU-vs-oracle comparisons (tests/test_gpu_homography.py, H_RTOL)."""
    from libstacker_rs_amd import synth
    rng = np.random.default_rng(2)
    worst = 0.0
    for _ in range(20):
        n = int(rng.integers(50, 400))
        src = rng.uniform(0, 1920, (n, 2)).astype(np.float32)
        Hs = synth.random_homography(rng, 1920, 1080, strength=3.0)
        p = np.c_[src.astype(np.float64), np.ones(n)] @ Hs.T
        dst = (p[:, :2] / p[:, 2:] + rng.normal(0, 0.5, (n, 2))).astype(np.float32)
        H1, _ = oracle.find_homography(src, dst, 0, 3.0)
        perm = rng.permutation(n)
        H2, _ = oracle.find_homography(src[perm], dst[perm], 0, 3.0)
        worst = max(worst, float(np.max(np.abs(H1 - H2) / np.maximum(np.abs(H1), 1e-3))))
    assert 1e-10 < worst < 2e-7, worst


# ---------------------------------------------------------------------------------------------------------------------
# Author-independent evidence for the oracle's pure-math stages: scipy.ndimage re-computes them in float64 from the
# textbook definitions (bilinear resampling with a constant border, correlation with mirrored borders). This pins no
# OpenCV quirk — the parity status stays "unpinned" — but it is written by nobody involved in this repository.
# ---------------------------------------------------------------------------------------------------------------------
def _smooth_image(rng, h, w, cn=None):
    """A band-limited test image in [0, 255] (u8): sums of a few low-frequency sinusoids, so a sub-pixel difference in the
    sample position is worth little and the comparison isolates the arithmetic."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    def plane():
        v = np.zeros((h, w))
        for _ in range(4):
            fx, fy = rng.uniform(0.01, 0.06, 2)
            v += rng.uniform(0.5, 1.0) * np.sin(fx * x + fy * y + rng.uniform(0, 6.28))
        return v
    img = plane() if cn is None else np.stack([plane() for _ in range(cn)], -1)
    img = (img - img.min()) / (img.max() - img.min()) * 255.0
    return np.round(img).astype(np.uint8)


@pytest.mark.parametrize("cn", [1, 3])
def test_warp_frame_matches_scipy_map_coordinates(cn):
    """warp_perspective (exact f32 mode, BORDER_CONSTANT 0) against scipy's order-1 spline resampling with a constant
    grid extension: dst(x, y) = bilinear(src / 255, M^-1 (x, y, 1)), taps outside the image replaced by 0."""
    from scipy.ndimage import map_coordinates
    rng = np.random.default_rng(11)
    h, w = 96, 128
    src = _smooth_image(rng, h, w, None if cn == 1 else 3)
    for M in (np.array([[1.01, 0.02, -3.3], [-0.015, 0.99, 2.7], [2e-5, -1e-5, 1.0]]),
              np.array([[0.9, 0.1, 7.25], [-0.12, 1.05, -4.5], [0.0, 0.0, 1.0]]),
              np.eye(3)):
        got = oracle.warp_frame(src, M)
        Minv = np.linalg.inv(M)
        y, x = np.mgrid[0:h, 0:w].astype(np.float64)
        W = Minv[2, 0] * x + Minv[2, 1] * y + Minv[2, 2]
        X = (Minv[0, 0] * x + Minv[0, 1] * y + Minv[0, 2]) / W
        Y = (Minv[1, 0] * x + Minv[1, 1] * y + Minv[1, 2]) / W
        planes = src[..., None] if src.ndim == 2 else src
        ref = np.stack([map_coordinates(planes[..., c].astype(np.float64) / 255.0, [Y, X], order=1, mode="grid-constant", cval=0.0)
                        for c in range(planes.shape[-1])], -1)
        # the oracle evaluates the map in f32 (OpenCV >= 4.11 semantics): positions agree to ~2e-5 px here, and the image's
        # slope is below 0.03 per pixel, except across the border, where one position's rounding decides a whole tap
        d = np.abs(got.astype(np.float64) - ref)
        inside = (X > 0.01) & (X < w - 1.01) & (Y > 0.01) & (Y < h - 1.01)
        assert d[inside].max() <= 2e-6, d[inside].max()
        assert np.percentile(d, 99.5) <= 2e-6 and d.max() <= 0.02      # the rim: a few pixels within 1e-5 px of a tap boundary


@pytest.mark.parametrize("ksize", [3, 5, 7, 9, 13])
def test_gaussian_blur_matches_scipy_correlate(ksize):
    """GaussianBlur(ksize, sigma 0) with BORDER_REFLECT_101 against scipy's separable correlation with mirrored borders.
    Taps: OpenCV's fixed tables up to 7, exp(-x^2 / 2 sigma^2) normalised with sigma = 0.3 ((k - 1) / 2 - 1) + 0.8 beyond."""
    from scipy.ndimage import correlate1d
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (61, 83)).astype(np.uint8)
    fixed = {3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if ksize in fixed:
        taps = np.array(fixed[ksize])
    else:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
        xs = np.arange(ksize) - (ksize - 1) / 2
        taps = np.exp(-xs * xs / (2 * sigma * sigma))
        taps /= taps.sum()
    ref = correlate1d(correlate1d(img.astype(np.float64), taps, axis=1, mode="mirror"), taps, axis=0, mode="mirror")
    got = oracle.gaussian_blur_f32(img, ksize)
    assert np.max(np.abs(got - ref)) <= 255 * 4e-7 * 4, np.max(np.abs(got - ref))     # f32 taps and sums against f64


def test_gradients_match_scipy_correlate():
    """The ECC image gradients: correlation with [-0.5, 0, 0.5] along x and along y, BORDER_REFLECT_101."""
    from scipy.ndimage import correlate1d
    rng = np.random.default_rng(9)
    img = rng.uniform(0, 255, (40, 57)).astype(np.float32)
    gx, gy = oracle.gradients(img)
    k = np.array([-0.5, 0.0, 0.5])
    assert np.max(np.abs(gx - correlate1d(img.astype(np.float64), k, axis=1, mode="mirror"))) <= 1e-4
    assert np.max(np.abs(gy - correlate1d(img.astype(np.float64), k, axis=0, mode="mirror"))) <= 1e-4


def test_orb_harris_response_and_angle_match_scipy():
    """ORB's per-keypoint Harris response and intensity-centroid angle, recomputed for the oracle's level-0 keypoints with
    scipy / numpy from the textbook definitions (Sobel 3x3 gradients, 7x7 box sums, det - 0.04 trace^2 scaled by
    (1 / (4 * 7 * 255))^4; first-order moments over the radius-15 disc, atan2 in degrees). Independent of the oracle's code
    path for these two quantities; says nothing about FAST, the pyramid or BRIEF."""
    from scipy.ndimage import correlate
    from libstacker_rs_amd import synth
    frames, _ = synth.make_stack(1, 640, 480)
    g = oracle.grey(frames.numpy()[0])
    kp, de = oracle.orb_detect_and_compute(g)
    lvl0 = kp[kp[:, 5] == 0]
    assert len(lvl0) >= 50
    I = g.astype(np.int64)
    sx = correlate(I, np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]]), mode="constant")
    sy = correlate(I, np.array([[-1, -2, -1], [0, 0, 0], [1, 2, 1]]), mode="constant")
    box = np.ones((7, 7), np.int64)
    a, b, c = correlate(sx * sx, box, mode="constant"), correlate(sy * sy, box, mode="constant"), correlate(sx * sy, box, mode="constant")
    scale4 = np.float32(1.0 / (4 * 7 * 255.0)) ** 4
    # the radius-15 disc: u_max per row as ORB builds it (rounded quarter circle, made symmetric about the diagonal)
    hp = 15
    vmax, vmin = int(np.floor(hp * np.sqrt(2) / 2 + 1)), int(np.ceil(hp * np.sqrt(2) / 2))
    umax = np.zeros(hp + 2, int)
    for v in range(vmax + 1):
        umax[v] = int(np.rint(np.sqrt(hp * hp - v * v)))
    v0 = 0
    for v in range(hp, vmin - 1, -1):
        while umax[v0] == umax[v0 + 1]:
            v0 += 1
        umax[v] = v0
        v0 += 1
    worst_r = worst_a = 0.0
    for x, y, size, angle, resp in lvl0[:, :5]:
        xi, yi = int(x), int(y)
        assert (xi, yi) == (x, y) and size == 31.0
        fa, fb, fc = np.float32(a[yi, xi]), np.float32(b[yi, xi]), np.float32(c[yi, xi])
        r = (fa * fb - fc * fc - np.float32(0.04) * (fa + fb) * (fa + fb)) * scale4
        worst_r = max(worst_r, abs(float(r) - resp) / max(abs(resp), 1e-12))
        m10 = m01 = 0
        for v in range(-hp, hp + 1):
            u = umax[abs(v)]
            row = I[yi + v, xi - u:xi + u + 1]
            m10 += int(np.dot(np.arange(-u, u + 1), row))
            m01 += v * int(row.sum())
        ref = np.degrees(np.arctan2(m01, m10)) % 360.0
        worst_a = max(worst_a, min(abs(ref - angle), 360.0 - abs(ref - angle)))
    assert worst_r <= 2e-6, worst_r            # f32 evaluation order of the determinant
    assert worst_a <= 0.35, worst_a            # fastAtan2 is a 0.3-degree polynomial


def test_fast_score_map_matches_the_definition():
    """FAST-9/16 from its definition, by brute force over thresholds: a pixel is a corner at threshold t iff nine contiguous
    pixels of the 16-pixel Bresenham ring are all brighter than centre + t or all darker than centre - t; its score is the
    largest t >= threshold for which it still is one (OpenCV's cornerScore), 0 for non-corners."""
    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, (40, 56)).astype(np.uint8)
    img[10:30, 20:40] = np.clip(img[10:30, 20:40].astype(int) + 90, 0, 255).astype(np.uint8)     # a bright block: real corners
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    h, w = img.shape
    I = img.astype(int)
    ref = np.zeros((h, w), int)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            d = np.array([I[y + dy, x + dx] - I[y, x] for dx, dy in ring])
            best = 0
            for t in range(20, 256):
                br, dk = d > t, d < -t
                ok = any(all(br[(k + j) % 16] for j in range(9)) or all(dk[(k + j) % 16] for j in range(9)) for k in range(16))
                if not ok:
                    break
                best = t
            ref[y, x] = best
    got = oracle.fast_score_map(img, 20).astype(int)
    assert np.array_equal(got[3:h - 3, 3:w - 3], ref[3:h - 3, 3:w - 3])
    assert (ref > 0).sum() > 20


def test_resize_area_known_answers_and_f32():
    """scale_image's resize(INTER_AREA) (utils.rs:186-214). Exact halving is OpenCV's 2 x 2 special case: (a + b + c + d + 2) >> 2,
    which rounds a half UP where cvRound (every other ratio) rounds it to even — pinned here by closed form (round 4: the oracle
    and the kernel both used cvRound for it until now). The f32 form (a 32FC1 grey) against the plain area average."""
    src = np.array([[1, 1, 2, 0, 255, 255, 3, 3],
                    [0, 0, 0, 0, 255, 254, 2, 2]], np.uint8)
    got = oracle.resize_area_u8(src, 4, 1)
    assert got.tolist() == [[1, 1, 255, 3]]                  # sums 2, 2, 1019, 10: 0.5 -> 1 (cvRound would give 0), 254.75 -> 255, 2.5 -> 3 (cvRound: 2)
    third = np.array([[1, 1, 0], [0, 0, 0], [0, 0, 0], [9, 9, 9], [9, 9, 9], [9, 9, 5]], np.uint8)
    assert oracle.resize_area_u8(third, 1, 2).tolist() == [[0], [9]]        # 2 / 9 -> 0; 77 / 9 = 8.56 -> 9: cvRound(sum * (1.f / 9))
    half_even = np.array([[1, 1, 1, 1], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]], np.uint8)       # 4 / 16 ... use 4 x 4 -> 1 x 1: 0.25 -> 0
    assert oracle.resize_area_u8(half_even, 1, 1).tolist() == [[0]]
    eight = np.zeros((4, 4), np.uint8); eight[0] = 2                                                # 8 / 16 = 0.5 -> cvRound -> 0 (half to even)
    assert oracle.resize_area_u8(eight, 1, 1).tolist() == [[0]]
    rng = np.random.default_rng(11)
    f = rng.random((48, 64), dtype=np.float32)
    for dw, dh in ((32, 24), (16, 12), (21, 16), (64 // 3, 16), (50, 37)):
        got = oracle.resize_area_f32(f, dw, dh)
        sx, sy = 64 / dw, 48 / dh
        ref = np.empty((dh, dw))
        for y in range(dh):                                   # exact area integral of the piecewise-constant image
            for x in range(dw):
                x0, x1, y0, y1 = x * sx, min((x + 1) * sx, 64), y * sy, min((y + 1) * sy, 48)
                xs = np.clip(np.minimum(np.arange(64) + 1, x1) - np.maximum(np.arange(64), x0), 0, None)
                ys = np.clip(np.minimum(np.arange(48) + 1, y1) - np.maximum(np.arange(48), y0), 0, None)
                ref[y, x] = ys @ f.astype(np.float64) @ xs / ((x1 - x0) * (y1 - y0))
        assert np.abs(got - ref).max() < 2e-6, (dw, dh)
    u = rng.integers(0, 256, (48, 64), dtype=np.uint8)        # the 8-bit form is the rounded f32 form except at exact halves
    for dw, dh in ((16, 12), (21, 16), (50, 37)):
        a, b = oracle.resize_area_u8(u, dw, dh).astype(int), oracle.resize_area_f32(u.astype(np.float32), dw, dh)
        assert np.abs(a - b).max() <= 0.5 + 1e-4


def test_resize_area_enlarging_is_the_bilinear_emulation():
    """scale_image enlarges when scale_down exceeds the frame's smaller dimension (the reference only checks it against the WIDTH,
    lib.rs:377, 876): resize(INTER_AREA) then runs "some variant of bilinear interpolation" [OCV-RECALL] — weights from
    fx = (dx + 1) - (sx + 1) * inv_scale. Closed forms: an integer enlargement replicates pixels, 1.5 x alternates copies and
    half-way blends (8 bit: rounded half up by the fixed-point column pass)."""
    rng = np.random.default_rng(13)
    u = rng.integers(0, 256, (6, 8), dtype=np.uint8)
    assert np.array_equal(oracle.resize_area_u8(u, 16, 12), np.repeat(np.repeat(u, 2, axis=0), 2, axis=1))
    assert np.array_equal(oracle.resize_area_u8(u, 24, 18), np.repeat(np.repeat(u, 3, axis=0), 3, axis=1))
    f = rng.random((6, 8), dtype=np.float32)
    assert np.array_equal(oracle.resize_area_f32(f, 16, 12), np.repeat(np.repeat(f, 2, axis=0), 2, axis=1))
    row = np.array([[10, 11, 20, 40]], np.uint8)
    assert oracle.resize_area_u8(row, 6, 1).tolist() == [[10, 11, 11, 20, 30, 40]]      # S0, (S0 + S1) / 2 rounded up, S1, S2, (S2 + S3) / 2, S3
    got = oracle.resize_area_f32(row.astype(np.float32), 6, 1)
    assert got.tolist() == [[10.0, 10.5, 11.0, 20.0, 30.0, 40.0]]
    # any enlargement: the weights of a pixel are an area-overlap split, so a constant image stays constant and the mean is kept
    big = oracle.resize_area_f32(f, 13, 11)
    assert abs(float(big.mean()) - float(f.mean())) < 0.02 and big.min() >= f.min() - 1e-6 and big.max() <= f.max() + 1e-6
    assert (oracle.resize_area_u8(np.full((5, 7), 93, np.uint8), 12, 9) == 93).all()
