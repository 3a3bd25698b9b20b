"""The LDS-ring path of the column-walking ECC kernel against the same kernel gathering from global memory (option
ecc_ring 0): the two read the same taps and run the same arithmetic, so iteration counts, warps and the stacked image must
be bit-identical. Prints that and the time of each. Run on the GPU box."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libstacker_rs_amd import EccMatchParameters, MotionType, Stacker, synth  # noqa: E402

st = Stacker(0)
p = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
for (w, h, n, reps, strength) in ((320, 240, 6, 1, 1.0), (1000, 700, 5, 1, 1.0), (1920, 1080, 9, 2, 1.0), (1920, 1080, 5, 1, 4.0),
                                  (3840, 2160, 64, 3, 1.0)):
    frames, _ = synth.make_stack(n, w, h, strength=strength)
    dev = frames.cuda()
    res = {}
    for ring in (0, 1):
        st.set_option("ecc_ring", ring)
        out, stats = st.ecc_match(dev, p, return_stats=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            st.ecc_match(dev, p)
        torch.cuda.synchronize()
        res[ring] = (out.cpu().numpy(), [s["iterations"] for s in stats], np.stack([s["warp"] for s in stats]),
                     (time.perf_counter() - t0) / reps)
    same = np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1] and np.array_equal(res[0][2], res[1][2])
    print(f"{w}x{h}x{n} strength {strength}: gather {res[0][3] * 1e3:.2f} ms, ring {res[1][3] * 1e3:.2f} ms, iterations "
          f"{sum(res[1][1])}, bit-identical: {same}" + ("" if same else f"  max |diff| {np.abs(res[0][0] - res[1][0]).max():.3e} iters {res[0][1]} vs {res[1][1]}"),
          flush=True)
st.set_option("ecc_ring", 1)
