#!/usr/bin/env python3
"""Generates tests/golden/golden_v1.npz — small input/output vectors for the hot path.

Where they come from: the reference (Rust + OpenCV 4.12) cannot be built or imported in this
container and holds no fixtures of its own, so these vectors are produced by
the repo's CPU oracle (oracle/, a restatement of the OpenCV algorithms; "parity unpinned") from
seeded synthetic inputs. They pin the oracle against drift and give the HIP path a second,
file-based reference that travels to the GPU box. Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from libstacker_rs_amd import synth  # noqa: E402


def crop(a):
    """Top-left 48 x 64 block (contains the warped border) — keeps the fixture file small."""
    return np.ascontiguousarray(a[:48, :64])


def main():
    out = {}
    frames, G = synth.make_stack(3, 160, 120, seed=7)
    frames = frames.numpy()
    out["frames"] = frames                       # 3 x 120 x 160 x 3 u8
    out["truth_G"] = G
    f0 = frames[0]
    out["grey"] = oracle.grey(f0)
    out["grey16"] = oracle.grey(f0.astype(np.uint16) * 257)
    out["convert"] = oracle.convert_f32(f0[:16, :16])
    for k in (3, 5, 7, 9):
        out[f"blur{k}"] = crop(oracle.gaussian_blur_f32(out["grey"], k))
    gx, gy = oracle.gradients(oracle.gaussian_blur_f32(out["grey"], 5))
    out["grad_x"], out["grad_y"] = crop(gx), crop(gy)
    M = np.array([[1.01, 0.02, -3.3], [-0.015, 0.99, 4.1], [2e-5, -1e-5, 1.0]])
    out["warp_M"] = M
    out["warp_exact"] = crop(oracle.warp_frame(f0, M))
    out["warp_classic"] = crop(oracle.warp_frame(f0, M, subpixel_bits=5))
    out["warp_reflect"] = crop(oracle.warp_frame(f0, M, border_mode=oracle.BORDER_REFLECT_101))
    A = np.array([[0.99, 0.03, 1.7], [-0.03, 1.01, -2.2]])
    out["warp_A"] = A
    out["warp_affine"] = crop(oracle.warp_frame(f0, A, is_affine=True))
    # ECC: 5 fixed iterations (no eps) per motion type, template = frame 1, input = frame 0
    g0, g1 = oracle.grey(frames[0]), oracle.grey(frames[1])
    for name, mot in (("homography", 3), ("affine", 2), ("euclidean", 1), ("translation", 0)):
        rc, W, rho, its = oracle.find_transform_ecc(g1, g0, np.eye(3 if mot == 3 else 2, 3), mot, 5, None, 5)
        assert rc == 0 and its == 5
        out[f"ecc_{name}_warp"] = W
        out[f"ecc_{name}_rho"] = np.float64(rho)
    img, warps, iters = oracle.ecc_match(list(frames), max_count=5000, epsilon=1e-5, gauss_filt_size=5)
    out["ecc_match_image"], out["ecc_match_warps"], out["ecc_match_iters"] = crop(img), warps, iters
    # ORB / matching / homography
    kp0, de0 = oracle.orb_detect_and_compute(g0)
    kp1, de1 = oracle.orb_detect_and_compute(g1)
    out["orb_kp0"], out["orb_de0"], out["orb_kp1"], out["orb_de1"] = kp0, de0, kp1, de1
    out["knn01"] = oracle.bf_knn2_hamming(de0, de1)
    out["rng_first8"] = np.array(oracle.rng_sequence(8), np.uint32)
    d, kimg, Hs, status = oracle.keypoint_match(list(frames), details=True)
    out["kp_match_image"], out["kp_match_H"], out["kp_match_dropped"] = crop(kimg), Hs, np.int32(d)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(kp0), "keypoints; ecc iters", list(iters), "dropped", d)


if __name__ == "__main__":
    main()
