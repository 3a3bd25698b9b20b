"""GPU parity of the reference's sharpness metrics (lib.rs:1030-1166) against the CPU oracle."""
import numpy as np
import pytest

import oracle
from libstacker_rs_amd import InvalidParams, synth

pytestmark = pytest.mark.gpu


def _metrics(s, g, k=3):
    return (s.sharpness_modified_laplacian(g), s.sharpness_variance_of_laplacian(g), s.sharpness_tenengrad(g, k),
            s.sharpness_normalized_gray_level_variance(g))


def _oracle_metrics(g, k=3):
    return (oracle.sharpness(g, 0), oracle.sharpness(g, 1), oracle.sharpness(g, 2, k), oracle.sharpness(g, 3))


@pytest.mark.parametrize("shape", [(97, 131), (240, 320), (5, 3), (1, 17), (33, 1)])
def test_u8_metrics_are_bit_exact(stacker, shape):
    # 8-bit input: integer-valued filters, int64 sums -> the same double as the oracle, whatever the summation order
    g = np.random.default_rng(shape[0]).integers(0, 256, shape, dtype=np.uint8)
    assert _metrics(stacker, g) == _oracle_metrics(g)


@pytest.mark.parametrize("k", [1, 3, 5, 7])
def test_tenengrad_kernel_sizes(stacker, k):
    g = np.random.default_rng(k).integers(0, 256, (120, 90), dtype=np.uint8)
    assert stacker.sharpness_tenengrad(g, k) == oracle.sharpness(g, 2, k)
    with pytest.raises(InvalidParams):
        stacker.sharpness_tenengrad(g, 4)                                   # lib.rs:1105-1109


def test_f32_metrics_match_to_f64_roundoff(stacker):
    g = np.random.default_rng(5).random((150, 170), dtype=np.float32) * 255
    for a, b in zip(_metrics(stacker, g, 5), _oracle_metrics(g, 5)):
        assert abs(a - b) <= 1e-11 * abs(b)


def test_metrics_order_blurred_below_sharp(stacker):
    # what examples/main.rs:40-60 uses them for: rank frames, the blurred one comes out last
    frames, _ = synth.make_stack(1, 640, 480)
    sharp = oracle.grey(frames[0].numpy())
    blurred = np.clip(np.rint(oracle.gaussian_blur_f32(sharp, 7)), 0, 255).astype(np.uint8)
    ms, mb = _metrics(stacker, sharp), _metrics(stacker, blurred)
    assert ms[0] > mb[0] and ms[1] > mb[1] and ms[2] > mb[2]
    flat = np.full((64, 64), 9, np.uint8)
    assert _metrics(stacker, flat) == (0.0, 0.0, 0.0, 0.0)
