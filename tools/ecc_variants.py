"""The ECC iteration kernels against each other: per stack size, iteration counts, final warps and the stacked image of
each `ecc_variant`, and the time of the call. Run on the GPU box: python tools/ecc_variants.py [variants ...]."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libstacker_rs_amd import EccMatchParameters, MotionType, Stacker, synth  # noqa: E402


def main():
    variants = [int(a) for a in sys.argv[1:]] or [3, 0]
    st = Stacker(0)
    p = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
    for (w, h, n, reps) in ((320, 240, 6, 1), (1000, 700, 5, 1), (200, 150, 3, 1), (1920, 1080, 17, 2), (3840, 2160, 64, 3)):
        frames, _ = synth.make_stack(n, w, h)
        dev = frames.cuda()
        base = None
        for v in variants:
            st.set_option("ecc_variant", v)
            out, stats = st.ecc_match(dev, p, return_stats=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                out = st.ecc_match(dev, p)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            its = [s["iterations"] for s in stats]
            warps = np.stack([s["warp"] for s in stats])
            o = out.cpu().numpy()
            line = f"{w}x{h}x{n} variant {v}: {dt * 1e3:8.2f} ms  iters sum {sum(its)}"
            if base is None:
                base = (its, warps, o)
            else:
                ce = max(synth.corner_error(warps[i], base[1][i], w, h) for i in range(1, n))
                line += (f"  same iters: {its == base[0]}  max corner delta {ce:.2e} px  stack max abs diff "
                         f"{np.abs(o - base[2]).max():.2e}")
                if its != base[0]:
                    line += f"  {its} vs {base[0]}"
            print(line, flush=True)
    st.set_option("ecc_variant", 3)


if __name__ == "__main__":
    main()
