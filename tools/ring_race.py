"""Run-to-run determinism of ecc_match with the LDS ring, for several stack sizes / slot counts. GPU box."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libstacker_rs_amd import EccMatchParameters, MotionType, Stacker, synth
st = Stacker(0)
p = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
st.set_option("prep_overlap", 0)
for n, slots, reps in ((256, 0, 150), (255, 0, 60), (131, 0, 60)):
    frames, _ = synth.make_stack(n, 3840, 2160, device="cuda")
    st.set_option("ecc_slots", slots)
    ref = None; bad = collections = 0
    which = {}
    for rep in range(reps):
        out, stats = st.ecc_match(frames, p, return_stats=True)
        warps = np.stack([s["warp"] for s in stats]); its = [s["iterations"] for s in stats]
        if ref is None: ref = (warps, its)
        else:
            wd = [i for i in range(n) if not np.array_equal(warps[i], ref[0][i])]
            if wd:
                bad += 1
                for i in wd: which[i] = which.get(i, 0) + 1
    print(f"n {n} slots {slots}: {bad} of {reps} runs differ; frames {which}; iterations of those {[ref[1][i] for i in which]} max its {max(ref[1])}", flush=True)
    del frames; torch.cuda.empty_cache()
