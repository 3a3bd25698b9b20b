"""Host-fed rate from PAGEABLE memory (what a caller hands over when it has not decoded into stk_host_alloc buffers)
next to pinned memory. GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libstacker_rs_amd import EccMatchParameters, MotionType, Stacker, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames, _ = synth.make_stack(n, 3840, 2160, device="cuda")
st = Stacker(0)
p = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
acc = torch.empty((2160, 3840, 3), dtype=torch.float32, device="cuda")
host_pageable = [np.array(f.cpu().numpy()) for f in frames]          # ordinary malloc'ed arrays
host_pinned = frames.cpu().pin_memory()
for name, src in (("pinned", host_pinned), ("pageable", host_pageable)):
    st.ecc_match_shard(src, p, True, acc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        st.ecc_match_shard(src, p, True, acc)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    t = st.timing()
    print(f"{name:9s}: {n / dt:8.1f} frames/s, {1e3 * dt:7.1f} ms per stack, copy stream busy {t['h2d_ms']:.1f} ms = {t['h2d_bytes'] / t['h2d_ms'] / 1e6:.1f} GB/s", flush=True)
