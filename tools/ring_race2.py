"""More run-to-run determinism of ecc_match with the LDS ring: other sizes and stronger motion. GPU box."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libstacker_rs_amd import EccMatchParameters, MotionType, Stacker, synth
st = Stacker(0)
p = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
for (w, h, n, strength, reps) in ((1000, 700, 80, 1.0, 80), (1000, 700, 80, 3.0, 80), (1920, 1080, 120, 2.0, 80), (2000, 1200, 60, 1.5, 60),
                                  (3840, 2160, 100, 2.0, 60), (3840, 2160, 130, 0.5, 60)):
    frames, _ = synth.make_stack(n, w, h, device="cuda", strength=strength)
    ref = None; bad = 0; which = {}
    for rep in range(reps):
        try:
            out, stats = st.ecc_match(frames, p, return_stats=True)
        except Exception as e:
            print("   raised", str(e)[-70:]); break
        warps = np.stack([s["warp"] for s in stats])
        if ref is None: ref = (warps, out.clone())
        else:
            wd = [i for i in range(n) if not np.array_equal(warps[i], ref[0][i])]
            if wd or not torch.equal(out, ref[1]):
                bad += 1
                for i in wd: which[i] = which.get(i, 0) + 1
    print(f"{w}x{h} x{n} strength {strength}: {bad} of {reps} runs differ {which}", flush=True)
    del frames; torch.cuda.empty_cache()
