"""Where does the end-to-end ecc_match stack differ from the oracle's, and by how much?

Prints, per moving frame, iterations / rho / corner displacement of the GPU warp against the oracle's, then the MAX
relative error of the stacked image over the pixels SURVEY 8d names (>= 2 px from every frame's warped border) and
where the largest one sits. Run on the GPU box: python tools/ecc_parity_diag.py [W H N]."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from libstacker_rs_amd import EccMatchParameters, MotionType, Stacker, synth  # noqa: E402


def interior_mask(shape_hw, warps, margin=2):
    from scipy.ndimage import binary_erosion
    h, w = shape_hw
    ones = np.full((h, w, 1), 255, np.uint8)
    m = np.ones((h, w), bool)
    for W in warps:
        cov = oracle.warp_frame(ones, np.asarray(W, np.float64))[..., 0]
        m &= cov >= 1.0 - 1e-6
    return binary_erosion(m, structure=np.ones((2 * margin + 1, 2 * margin + 1), bool), border_value=0)


def main():
    w, h, n = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (640, 480, 6)
    frames, G = synth.make_stack(n, w, h)
    frames = frames.numpy()
    st = Stacker(0)
    p = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
    out, stats = st.ecc_match(list(frames), p, return_stats=True)
    ref, warps, iters = oracle.ecc_match(list(frames), max_count=5000, epsilon=1e-5, gauss_filt_size=5)
    for i in range(1, n):
        ce = synth.corner_error(stats[i]["warp"], warps[i], w, h)
        print(f"frame {i}: iters gpu {stats[i]['iterations']} oracle {iters[i]}  rho gpu {stats[i]['rho']:.9f}  "
              f"corner delta {ce:.5f} px  vs truth gpu {synth.corner_error(stats[i]['warp'], G[i], w, h):.4f} "
              f"oracle {synth.corner_error(warps[i], G[i], w, h):.4f}")
    m = interior_mask((h, w), [warps[i] for i in range(1, n)])
    rel = np.abs(out - ref) / np.maximum(np.abs(ref), 1e-3)
    relm = rel[m]
    k = np.unravel_index(np.argmax(np.where(m[..., None], rel, 0)), rel.shape)
    print(f"interior px {m.sum()} of {h * w}; max rel {relm.max():.3e} at {k}; p99.9 {np.percentile(relm, 99.9):.3e} "
          f"p99 {np.percentile(relm, 99):.3e} mean {relm.mean():.3e}; abs max {np.abs(out - ref)[m].max():.3e}")
    # the same with a fixed iteration count: separates "stop one iteration apart" from round-off in the sums
    for cnt in (3, 8):
        pf = EccMatchParameters(MotionType.Homography, cnt, None, 5)
        o2, s2 = st.ecc_match(list(frames), pf, return_stats=True)
        r2, w2, _ = oracle.ecc_match(list(frames), max_count=cnt, epsilon=None, gauss_filt_size=5)
        m2 = interior_mask((h, w), [w2[i] for i in range(1, n)])
        rel2 = (np.abs(o2 - r2) / np.maximum(np.abs(r2), 1e-3))[m2]
        ce = max(synth.corner_error(s2[i]["warp"], w2[i], w, h) for i in range(1, n))
        print(f"fixed {cnt} iterations: max corner delta {ce:.6f} px, stack max rel {rel2.max():.3e}")


if __name__ == "__main__":
    main()
