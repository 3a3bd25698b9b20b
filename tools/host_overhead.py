"""Fixed host cost of one call through the Python mirror and the C ABI: a stack so small that the kernels take microseconds.
usage: python tools/host_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libstacker_rs_amd import EccMatchParameters, KeyPointMatchParameters, MotionType, RANSAC, Stacker, synth

st = Stacker(0)
st.set_option("profile", 1)
frames, _ = synth.make_stack(9, 256, 192, device="cuda")
ecc = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
acc = torch.empty((192, 256, 3), dtype=torch.float32, device="cuda")
out = torch.empty_like(acc)


def timed(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print("ecc_match_shard  9 x 256x192: %.1f us per call" % timed(lambda: st.ecc_match_shard(frames, ecc, True, acc)))
t = st.timing()
print("   of which stage timers say: prep %.1f + align %.1f + warp %.1f us" % (t["prep_ms"] * 1e3, t["align_ms"] * 1e3, t["warp_ms"] * 1e3))
print("timing():           %.1f us" % timed(st.timing))
print("finalize_mean:      %.1f us" % timed(lambda: st.finalize_mean(acc, 9, out)))
kp = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)
frames2, _ = synth.make_stack(9, 640, 480, device="cuda")
acc2 = torch.empty((480, 640, 3), dtype=torch.float32, device="cuda")
print("keypoint_match_shard 9 x 640x480: %.1f us per call" % timed(lambda: st.keypoint_match_shard(frames2, kp, True, acc2), 100))
t = st.timing()
print("   of which stage timers say: align %.1f + warp %.1f us" % (t["align_ms"] * 1e3, t["warp_ms"] * 1e3))
