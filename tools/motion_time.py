"""Timing of ecc_match for the non-homography motion types (variant 3 = column-walking kernel, 0 = direct)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libstacker_rs_amd import Stacker, EccMatchParameters, MotionType, synth
frames, _ = synth.make_stack(17, 3840, 2160, device="cuda", strength=0.3)
st = Stacker(0)
for motion in (MotionType.Affine, MotionType.Euclidean, MotionType.Translation):
    p = EccMatchParameters(motion, 30, None, 5)               # fixed 30 iterations: comparable work
    for variant in (3, 0):
        st.set_option("ecc_variant", variant)
        st.ecc_match(frames, p)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st.ecc_match(frames, p)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        t = st.timing()
        print(f"{motion.name:12s} variant {variant}: {1e3 * dt:7.2f} ms, align {t['align_ms']:.2f} ms, {t['ecc_slot_iterations']} slot-iterations")
st.set_option("ecc_variant", 3)
