"""GPU parity of the per-pixel stages against the CPU oracle (through the C ABI)."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _rng_img(h, w, c, dtype, seed=0):
    rng = np.random.default_rng(seed)
    if dtype == np.uint8:
        return rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    if dtype == np.uint16:
        return rng.integers(0, 65536, (h, w, c), dtype=np.uint16)
    return rng.random((h, w, c), dtype=np.float32)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_grey_bit_exact(stacker, dtype):
    img = _rng_img(97, 131, 3, dtype)
    got = stacker.grey(img)
    ref = oracle.grey(img)
    assert got.dtype == ref.dtype
    assert np.array_equal(got, ref)          # integer formula: bit exact; f32: same three roundings


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_convert_bit_exact(stacker, dtype):
    img = _rng_img(64, 80, 3, dtype, 1)
    assert np.array_equal(stacker.convert_f32(img), oracle.convert_f32(img))


@pytest.mark.parametrize("ksize", [1, 3, 5, 7])
def test_gaussian_blur_small_kernels_bit_exact(stacker, ksize):
    # dyadic taps on u8 input: every partial sum is exactly representable, so any order agrees
    g = _rng_img(150, 201, 1, np.uint8, 2)[..., 0]
    assert np.array_equal(stacker.gaussian_blur_f32(g, ksize), oracle.gaussian_blur_f32(g, ksize))


@pytest.mark.parametrize("ksize", [9, 15, 31, 33, 63])
def test_gaussian_blur_large_kernels(stacker, ksize):
    g = _rng_img(90, 140, 1, np.uint8, 3)[..., 0]
    got, ref = stacker.gaussian_blur_f32(g, ksize), oracle.gaussian_blur_f32(g, ksize)
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-4)


def test_gaussian_blur_tiny_image_reflects(stacker):
    g = _rng_img(3, 5, 1, np.uint8, 4)[..., 0]
    assert np.array_equal(stacker.gaussian_blur_f32(g, 7), oracle.gaussian_blur_f32(g, 7))


@pytest.mark.parametrize("shape", [(200, 300), (211, 333), (75, 520), (40, 8), (33, 4), (2160 // 8, 3840 // 8)])
@pytest.mark.parametrize("ksize", [3, 5, 7])
def test_fused_grey_blur_bit_exact(stacker, shape, ksize):
    # the per-frame ECC preparation; (200,300)/(75,520)/... take the dword-load fast path (rows 4-byte aligned),
    # (211,333) the generic one; tiles of 128x32 outputs are crossed in both directions
    import torch
    h, w = shape
    img = _rng_img(h, w, 3, np.uint8, 7)
    ref = oracle.gaussian_blur_f32(oracle.grey(img), ksize)
    assert np.array_equal(stacker.grey_blur_f32(img, ksize), ref)
    dev = stacker.grey_blur_f32(torch.from_numpy(img).cuda(), ksize)
    assert np.array_equal(dev.cpu().numpy(), ref)


@pytest.mark.parametrize("shape", [(120, 200), (97, 131)])
def test_fused_grey_blur_u16_bit_exact(stacker, shape):
    # 16-bit BGR (hybrid path): grey by the 16U formula, blur of float(grey16); (120,200) takes the dword-load fast path
    img = _rng_img(shape[0], shape[1], 3, np.uint16, 12)
    ref = oracle.gaussian_blur_f32(oracle.grey(img).astype(np.float32), 5)
    assert np.array_equal(stacker.grey_blur_f32(img, 5), ref)


def test_fused_grey_blur_f32_input_and_large_kernel(stacker):
    img = _rng_img(60, 90, 3, np.float32, 8)
    got = stacker.grey_blur_f32(img, 5)
    np.testing.assert_allclose(got, oracle.gaussian_blur_f32(oracle.grey(img), 5), rtol=2e-6, atol=1e-6)
    img8 = _rng_img(60, 90, 3, np.uint8, 9)
    np.testing.assert_allclose(stacker.grey_blur_f32(img8, 9), oracle.gaussian_blur_f32(oracle.grey(img8), 9), rtol=2e-6, atol=1e-4)


H_CASES = {
    "identity": np.eye(3),
    "shift_int": np.array([[1, 0, 5], [0, 1, -3], [0, 0, 1.0]]),
    "shift_frac": np.array([[1, 0, 2.25], [0, 1, 1.5], [0, 0, 1.0]]),
    "projective": np.array([[1.01, 0.02, -3.3], [-0.015, 0.99, 4.1], [2e-5, -1e-5, 1.0]]),
    "big_rotation": np.array([[0.8, -0.6, 40.0], [0.6, 0.8, -30.0], [0, 0, 1.0]]),
}


@pytest.mark.parametrize("name", list(H_CASES))
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_warp_perspective_matches_oracle(stacker, name, dtype):
    img = _rng_img(120, 160, 3, dtype, 5)
    M = H_CASES[name]
    got = stacker.warp_accumulate(img, M)
    ref = oracle.warp_frame(img, M)
    # identical f32 operation sequence (explicit fma, true division): <= 1e-6 abs
    scale = 1.0 if dtype != np.uint16 else 257.0
    assert np.max(np.abs(got - ref)) <= 1e-6 * scale


def test_warp_identity_is_convert(stacker):
    img = _rng_img(50, 70, 3, np.uint8, 6)
    assert np.array_equal(stacker.warp_accumulate(img, np.eye(3)), oracle.convert_f32(img))


def test_warp_integer_shift_is_shifted_copy(stacker):
    img = _rng_img(40, 60, 3, np.uint8, 7)
    M = np.array([[1, 0, 4], [0, 1, 2], [0, 0, 1.0]])       # frame_i -> frame_0: moves content by (+4,+2)
    got = stacker.warp_accumulate(img, M)
    ref = np.zeros((40, 60, 3), np.float32)
    ref[2:, 4:] = oracle.convert_f32(img)[:-2, :-4]
    assert np.array_equal(got, ref) and not np.isnan(got).any()


@pytest.mark.parametrize("mode", [oracle.BORDER_CONSTANT, oracle.BORDER_REPLICATE, oracle.BORDER_REFLECT,
                                  oracle.BORDER_WRAP, oracle.BORDER_REFLECT_101])
def test_warp_border_modes(stacker, mode):
    img = _rng_img(64, 48, 3, np.uint8, 8)
    M = np.array([[1.05, 0.1, -9.0], [-0.08, 0.97, 7.5], [1e-4, 5e-5, 1.0]])
    bv = (0.25, 0.5, 0.75, 0)
    got = stacker.warp_accumulate(img, M, border_mode=mode, border_value=bv)
    ref = oracle.warp_frame(img, M, border_mode=mode, border_value=bv)
    assert np.max(np.abs(got - ref)) <= 1e-6


def test_warp_affine_and_accumulate(stacker):
    img = _rng_img(72, 96, 3, np.uint8, 9)
    M = np.array([[0.99, 0.03, 1.7], [-0.03, 1.01, -2.2]])
    acc0 = np.random.default_rng(1).random((72, 96, 3), dtype=np.float32)
    got = stacker.warp_accumulate(img, M, is_affine=True, acc=acc0.copy())
    ref = oracle.warp_frame(img, M, is_affine=True, acc=acc0.copy())
    assert np.max(np.abs(got - ref)) <= 1e-6


def test_warp_single_channel(stacker):
    img = _rng_img(33, 45, 1, np.float32, 10)
    M = H_CASES["projective"]
    got = stacker.warp_accumulate(img, M, alpha=1.0)
    ref = oracle.warp_frame(img, M, alpha=1.0)
    assert np.max(np.abs(got - ref)) <= 1e-6


def test_warp_classic_quantised_mode(stacker):
    img = _rng_img(80, 100, 3, np.uint8, 11)
    M = H_CASES["projective"]
    stacker.set_option("warp_subpixel_bits", 5)
    try:
        got = stacker.warp_accumulate(img, M)
    finally:
        stacker.set_option("warp_subpixel_bits", 0)
    ref = oracle.warp_frame(img, M, subpixel_bits=5)
    assert np.max(np.abs(got - ref)) <= 1e-6


def test_warp_degenerate_matrix_gives_border(stacker):
    img = _rng_img(16, 16, 3, np.uint8, 12)
    got = stacker.warp_accumulate(img, np.zeros((3, 3)))       # singular: inverse is the zero matrix
    ref = oracle.warp_frame(img, np.zeros((3, 3)))
    assert np.array_equal(got, ref) and not np.isnan(got).any()


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_warp_u8_fast_path_is_bit_identical_to_the_generic_kernel(stacker, dtype):
    """warp_accumulate_u8c3_kernel (and its 16-bit sibling) shares one reciprocal chain between X / W and Y / W (the compiler's own IEEE expansion
    without the range scaling) and skips clamps and border selects on interior waves: wherever all four taps are inside
    the frame it must return the very bits of the generic kernel (true `/`, per-tap selects), which BORDER_REPLICATE selects."""
    _fast_path_vs_generic(stacker, dtype, 333, 517)             # not a multiple of the 64 x 4 tile; odd row size: 8-byte gathers
    _fast_path_vs_generic(stacker, dtype, 333, 516)             # rows a multiple of 4 bytes: dword-aligned 12-byte windows (round 3)


def _fast_path_vs_generic(stacker, dtype, h, w):
    rng = np.random.default_rng(3)
    top = 255 if dtype == np.uint8 else 65535
    frame = rng.integers(0, top + 1, (h, w, 3)).astype(dtype)
    ones = np.full((h, w, 3), 255, dtype)
    from libstacker_rs_amd import synth
    mats = [np.eye(3), synth.random_homography(rng, w, h, 8.0), synth.random_homography(rng, w, h, 30.0),
            np.array([[0.7, 0.2, 15.3], [-0.25, 0.9, 40.1], [4e-4, -3e-4, 1.0]]),          # strong perspective: W far from 1
            np.array([[1e-3, 0, 0], [0, 1e-3, 0], [0, 0, 1e-3]]),                            # tiny W (still in the safe range)
            np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1e-20]])]                                 # |1 / W| = 1e20: leaves the shared chain for `/`
    for M in mats:
        fast = stacker.warp_accumulate(frame, M)
        slow = stacker.warp_accumulate(frame, M, border_mode=1)
        cov = stacker.warp_accumulate(ones, M)
        inside = (cov == np.float32(255) * np.float32(1.0 / 255.0)).all(axis=2)
        assert np.array_equal(fast[inside], slow[inside])
        assert inside.mean() > 0.5 or abs(np.linalg.det(M)) < 1e-6 or M[2, 2] < 1e-6
        # and an accumulating second pass over several frames (the unrolled frame loop, odd tail)
    acc = stacker.warp_accumulate(frame, mats[1])
    accg = stacker.warp_accumulate(frame, mats[1], border_mode=1)
    cov = stacker.warp_accumulate(ones, mats[1])
    inside = (cov == 1.0).all(axis=2)
    acc2 = stacker.warp_accumulate(frame, mats[1], acc=acc.copy())
    accg2 = stacker.warp_accumulate(frame, mats[1], border_mode=1, acc=accg.copy())
    assert np.array_equal(acc2[inside], accg2[inside])


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_warp_fast_paths_equal_the_generic_kernel_everywhere_also_for_non_finite_maps(stacker, dtype):
    """BORDER_CONSTANT on every pixel, rim and border included: the u8 / u16 fast kernels against the generic kernel, which
    the same pixel values reach as a float32 frame. Also for maps with inf / NaN coordinates (a NaN X or Y with
    a finite W once passed the fast path's range test, v_max drops NaN operands, and accumulated NaN; OpenCV's
    saturate_cast sends such a coordinate to the border): inverse entries that overflow f32 (inf * 0 = NaN in column 0),
    an affine map with a NaN translation, and matrices whose W range is tested per frame on the host (round 3)."""
    for w in (203, 204):                                        # odd rows (8-byte gathers) and dword-aligned rows (12-byte windows)
        _fast_paths_everywhere(stacker, dtype, 131, w)
    # frames of one and two rows (the u8 launcher took the fast kernel for sh == 1, whose interior bound
    # (unsigned)(sh - 2) then wraps and lets every pixel — the non-finite sentinel too — gather unclamped)
    for h in (1, 2):
        _fast_paths_everywhere(stacker, dtype, h, 204)


def _fast_paths_everywhere(stacker, dtype, h, w):
    rng = np.random.default_rng(11)
    top = 255 if dtype == np.uint8 else 65535
    frame = rng.integers(0, top + 1, (h, w, 3)).astype(dtype)
    as_f32 = frame.astype(np.float32)
    from libstacker_rs_amd import synth
    bv = (0.25, 0.5, 0.75, 0.0)
    cases = [(np.eye(3), False), (synth.random_homography(rng, w, h, 12.0), False),
             (np.array([[0.7, 0.2, 15.3], [-0.25, 0.9, 40.1], [4e-4, -3e-4, 1.0]]), False),
             (np.array([[1, 0, 0], [0, 1, 0], [-0.02, 0, 1.0]]), False),                     # W changes sign inside the image
             (np.diag([1e-39, 1.0, 1.0]), False), (np.diag([1.0, 1e-39, 1.0]), False),       # inverse entries overflow f32 -> inf, inf * 0 = NaN
             (np.array([[1, 0, 1e30], [0, 1, 0], [0, 0, 1.0]]), False),                      # huge finite coordinates
             (np.array([[1.0, 0.01, np.nan], [0.0, 1.0, 3.0]]), True),                       # affine, NaN translation
             (np.array([[1.0, 0.01, 2.5], [0.02, 1.0, -3.25]]), True)]
    for M, aff in cases:
        for value in ((0, 0, 0, 0), bv):
            fast = stacker.warp_accumulate(frame, M, is_affine=aff, border_value=value)
            slow = stacker.warp_accumulate(as_f32, M, is_affine=aff, border_value=value)
            assert np.array_equal(fast, slow, equal_nan=False), (M, value)
            assert np.isfinite(fast).all()
