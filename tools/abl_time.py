"""Scratch timing of the ECC iteration launch for A/B builds (STACKER_AMD_LIB=...): tolerant of failing solves."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, torch
from libstacker_rs_amd import Stacker, EccMatchParameters, MotionType, StackerError, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 17
frames, _ = synth.make_stack(n, 3840, 2160, device="cuda")
st = Stacker(0)
st.set_option("profile", 2)
p = EccMatchParameters(MotionType.Homography, 6, None, 5)
for rep in range(2):
    try:
        st.ecc_match(frames, p)
    except StackerError as e:
        print("solve failed as expected for an ablated build:", str(e)[:60])
    t = st.timing()
    print(f"rep {rep}: launches timed {t['ecc_iter_timed']}, avg {1e3 * t['ecc_iter_ms'] / max(t['ecc_iter_timed'], 1):.1f} us, slot-iterations {t['ecc_slot_iterations']}")
