"""Time of the ECC iteration pass alone: N 4K frames, a fixed number of iterations (no eps), every launch bracketed by an
event pair (profile = 2). For A/B builds: STACKER_AMD_LIB=libstacker_rs_amd/ab/libX.so python tools/ecc_iter_time.py [n] [iters]."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libstacker_rs_amd import EccMatchParameters, MotionType, Stacker, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 33
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    opts = [a.split("=") for a in sys.argv[3:]]
    frames, _ = synth.make_stack(n, 3840, 2160, device="cuda")
    st = Stacker(0)
    if os.environ.get("TORCH_STREAM"):
        st.use_torch_stream()
    for k, v in opts:
        st.set_option(k, int(v))
    p = EccMatchParameters(MotionType.Homography, iters, None, 5) if iters > 0 else EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
    acc = torch.empty((2160, 3840, 3), dtype=torch.float32, device="cuda")
    st.set_option("profile", 2)
    st.set_option("profile_stride", 1)
    for _ in range(2):
        st.ecc_match_shard(frames, p, True, acc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    ms = cnt = 0
    for _ in range(reps):
        st.ecc_match_shard(frames, p, True, acc)
        t = st.timing()
        ms += t["ecc_iter_ms"]; cnt += t["ecc_iter_timed"]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{os.environ.get('STACKER_AMD_LIB', 'default')}: {n} frames x {iters} iterations: {ms / max(cnt, 1):.4f} ms per launch "
          f"({cnt // reps} launches timed per call), call {dt * 1e3:.2f} ms, fallbacks {t['ecc_ring_fallbacks']}", flush=True)


if __name__ == "__main__":
    main()
