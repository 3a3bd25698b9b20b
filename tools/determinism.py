"""Run-to-run determinism of every entry point the bench measures: repeated calls on the same input must return the same
bits (warps, iteration counts, dropped counts, image). GPU box: python tools/determinism.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libstacker_rs_amd import EccMatchParameters, KeyPointMatchParameters, MotionType, RANSAC, Stacker, synth

st = Stacker(0)
ecc = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
kp = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)


def sig(res):
    parts = []
    for r in (res if isinstance(res, tuple) else (res,)):
        if isinstance(r, list):
            parts.append(np.stack([np.asarray(s["warp"], np.float64) for s in r]).tobytes())
            parts.append(np.asarray([s.get("iterations", 0) for s in r]).tobytes())
        elif hasattr(r, "cpu"):
            parts.append(r.cpu().numpy().tobytes())
        elif isinstance(r, np.ndarray):
            parts.append(r.tobytes())
        else:
            parts.append(repr(r).encode())
    return parts


def check(name, fn, reps):
    ref = sig(fn())
    bad = 0
    for _ in range(reps):
        if sig(fn()) != ref:
            bad += 1
    print(f"{name}: {bad} of {reps} runs differ", flush=True)


f1080, _ = synth.make_stack(64, 1920, 1080, device="cuda")
check("ecc_match 64 x 1080p", lambda: st.ecc_match(f1080, ecc, return_stats=True), 80)
check("keypoint_match 64 x 1080p", lambda: st.keypoint_match(f1080, kp, return_stats=True), 40)
for m in (MotionType.Affine, MotionType.Euclidean, MotionType.Translation):
    check(f"ecc_match {m.name} 24 x 1080p", lambda: st.ecc_match(f1080[:24], EccMatchParameters(m, 50, 1e-4, 5), return_stats=True), 30)
host = f1080.cpu().pin_memory()
check("ecc_match 64 x 1080p host-fed", lambda: st.ecc_match(host, ecc, return_stats=True), 40)
check("keypoint_match 64 x 1080p host-fed", lambda: st.keypoint_match(host, kp, return_stats=True), 20)
del f1080, host
f16, _ = synth.make_stack(48, 3840, 2160, device="cuda", depth=16)
check("hybrid_match 48 x 4K 16-bit", lambda: st.hybrid_match(f16, kp, EccMatchParameters(MotionType.Homography, 200, 1e-5, 5), return_stats=True), 20)
del f16
torch.cuda.empty_cache()
f4k, _ = synth.make_stack(40, 3840, 2160, device="cuda")
multi = Stacker(0, devices=[0, 0])
check("ecc_match 40 x 4K, two-member context on one GPU", lambda: multi.ecc_match(f4k, ecc, return_stats=True), 20)
check("ecc_match 40 x 4K scale_down 1000", lambda: st.ecc_match(f4k, ecc, scale_down_width=1000.0, return_stats=True), 20)
