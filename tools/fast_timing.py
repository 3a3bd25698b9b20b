"""Per-pass wall-clock of one workgroup of fast_nms_tiled_kernel (build with -DSTK_FAST_TIMING). GPU box only."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libstacker_rs_amd import KeyPointMatchParameters, RANSAC, Stacker, synth, _ffi
frames, _ = synth.make_stack(8, 1920, 1080, device="cuda")
st = Stacker(0)
acc = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda")
for _ in range(2):
    st.keypoint_match_shard(frames, KeyPointMatchParameters(RANSAC, 5.0, 0.8, 0.9), True, acc)
lib = _ffi.load()
out = (C.c_ulonglong * 16)()
lib.stk_debug_fast_timing(out)
t = [out[i] for i in range(6)]
names = ["tile load + zero", "pass 1 compass", "pass 2 ring masks", "pass 3 strength", "pass 4 nms"]
print("last level launched; candidates A", out[8], "corners B", out[9])
for i, n in enumerate(names):
    print(f"{n:20s} {(t[i + 1] - t[i]) * 10:8d} ns")
print("total", (t[5] - t[0]) * 10, "ns")
