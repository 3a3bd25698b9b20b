"""Persistent ECC scheduler vs the launch-per-iteration form: same bits (iterations, warps, stacked image), and the time of
each. GPU box only: python tools/persist_check.py [WxHxN ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from libstacker_rs_amd import EccMatchParameters, MotionType, Stacker, synth

sizes = [(320, 240, 5), (640, 480, 9), (1000, 700, 3), (1920, 1080, 17), (3840, 2160, 33)]
if len(sys.argv) > 1:
    sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
st = Stacker(0)
if os.environ.get("TORCH_STREAM"):
    st.use_torch_stream()
REPS = int(os.environ.get("REPS", "3"))
if os.environ.get("WGS"):
    st.set_option("ecc_persist_wgs", int(os.environ["WGS"]))
for motion in (MotionType.Homography, MotionType.Affine):
    p = EccMatchParameters(motion, 5000, 1e-5, 5)
    for (w, h, n) in sizes:
        if motion != MotionType.Homography and w > 1920:
            continue
        frames, _ = synth.make_stack(n, w, h, device='cuda')
        dev = frames
        res = {}
        for persist in (2, 0):
            st.set_option("ecc_persist", persist)
            out, stats = st.ecc_match(dev, p, return_stats=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = REPS
            for _ in range(reps):
                out2 = st.ecc_match(dev, p)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            tm = st.timing()
            assert torch.equal(out, out2), "run-to-run difference"
            res[1 if persist else 0] = (out.cpu().numpy(), [s["iterations"] for s in stats], np.stack([s["warp"] for s in stats]), dt, tm["align_ms"], tm["prep_ms"])
        same = res[1][1] == res[0][1] and np.array_equal(res[1][2], res[0][2]) and np.array_equal(res[1][0], res[0][0])
        print(f"{motion.name} {w}x{h}x{n}: persistent {res[1][3] * 1e3:8.2f} ms, per-iteration launches {res[0][3] * 1e3:8.2f} ms, "
              f"align {res[1][4]:.3f} vs {res[0][4]:.3f} ms (prep {res[1][5]:.3f} vs {res[0][5]:.3f}), iterations {sum(res[1][1])}, same bits: {same}", flush=True)
        if not same:
            print("   iters", res[1][1], res[0][1])
st.set_option("ecc_persist", 1)
