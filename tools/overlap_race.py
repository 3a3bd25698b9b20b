"""Is ecc_match deterministic run after run, with and without the overlapped template preparation and the LDS ring?
256 x 4K frames, repeated calls: every call must succeed and return the same bits. Run on the GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libstacker_rs_amd import EccMatchParameters, MotionType, Stacker, synth  # noqa: E402

st = Stacker(0)
p = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
frames, _ = synth.make_stack(int(sys.argv[1]) if len(sys.argv) > 1 else 256, 3840, 2160, device="cuda")
ref = None
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for overlap, ring in ((0, 0), (1, 0), (0, 1), (1, 1)):
    st.set_option("prep_overlap", overlap)
    st.set_option("ecc_ring", ring)
    bad = 0
    diff = 0
    for rep in range(REPS):
        # dirty the template buffer's old contents between calls: a stale read shows up as a wrong result
        try:
            out, stats = st.ecc_match(frames, p, return_stats=True)
        except Exception as e:
            bad += 1
            print("   ", str(e)[-60:], flush=True)
            continue
        its = [s["iterations"] for s in stats]
        warps = np.stack([s["warp"] for s in stats])
        if ref is None:
            ref = (out.clone(), its, warps)
        elif its != ref[1] or not torch.equal(out, ref[0]):
            diff += 1
            wd = [i for i in range(len(its)) if not np.array_equal(warps[i], ref[2][i])]
            print(f"    rep {rep}: frames with another warp: {wd[:20]}{'...' if len(wd) > 20 else ''} ({len(wd)}); iterations differ in "
                  f"{sum(a != b for a, b in zip(its, ref[1]))}; image max |diff| {float((out - ref[0]).abs().max()):.3e}", flush=True)
    print(f"prep_overlap {overlap} ecc_ring {ring}: {bad} failed, {diff} differing of {REPS}", flush=True)
