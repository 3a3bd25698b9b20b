"""Time of the 16-bit fused warp + accumulate launch (hybrid path's fold): n 4K u16 frames. GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libstacker_rs_amd import EccMatchParameters, KeyPointMatchParameters, MotionType, RANSAC, Stacker, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
frames, _ = synth.make_stack(n, 3840, 2160, device="cuda", depth=16)
st = Stacker(0)
acc = torch.empty((2160, 3840, 3), dtype=torch.float32, device="cuda")
best = 1e9
for _ in range(3):
    st.hybrid_match_shard(frames, KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9), EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5), True, acc, return_stats=False)
    best = min(best, st.timing()["warp_ms"])
print(f"{os.environ.get('STACKER_AMD_LIB', 'default')}: u16 warp {best:.3f} ms for {n} frames = {(n * 6 + 12) * 3840 * 2160 / 1e9 / best * 1e3:.0f} GB/s", flush=True)
