"""Time the fused warp + accumulate launch for the tile / unroll variants (option warp_tune). GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libstacker_rs_amd import EccMatchParameters, MotionType, Stacker, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames, _ = synth.make_stack(n, 3840, 2160, device="cuda")
st = Stacker(0)
p = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
acc = torch.empty((2160, 3840, 3), dtype=torch.float32, device="cuda")
for v in [int(a, 16) for a in sys.argv[2:]] or (0x00, 0x01, 0x02, 0x10, 0x11, 0x12, 0x20, 0x22):
    st.set_option("warp_tune", v)
    best = 1e9
    for _ in range(3):
        st.ecc_match_shard(frames, p, True, acc)
        best = min(best, st.timing()["warp_ms"])
    gb = (n * 3 + 12) * 3840 * 2160 / 1e9
    print(f"tune {v:#04x}: warp {best:.3f} ms for {n} frames = {gb / best * 1e3:.0f} GB/s", flush=True)
