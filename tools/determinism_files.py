"""Run-to-run determinism of the file front end (decoder pool + per-frame gate + uploader + engine) and of host-fed stacks
with several upload batch sizes. GPU box."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libstacker_rs_amd import EccMatchParameters, KeyPointMatchParameters, MotionType, RANSAC, Stacker, synth
st = Stacker(0)
ecc = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
kp = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)
frames, _ = synth.make_stack(48, 1280, 960)
fr = frames.numpy()
d = tempfile.mkdtemp()
paths = []
for i, f in enumerate(fr):
    p = os.path.join(d, f"f{i:03d}.ppm")
    with open(p, "wb") as fh:
        fh.write(b"P6\n1280 960\n255\n" + np.ascontiguousarray(f[..., ::-1]).tobytes())
    paths.append(p)
ref_e = st.ecc_match(list(fr), ecc)
ref_k = st.keypoint_match(list(fr), kp)
bad = 0
for rep in range(25):
    if not np.array_equal(st.ecc_match_files(paths, ecc), ref_e): bad += 1
    dk, ok = st.keypoint_match_files(paths, kp)
    if dk != ref_k[0] or not np.array_equal(ok, ref_k[1]): bad += 1
print(f"*_match_files, 48 x 1280x960 PPM: {bad} of 50 calls differ from the frame-based result", flush=True)
for batch in (1, 2, 5, 8, 16, 64):
    st.set_option("upload_batch", batch)
    bad = 0
    for rep in range(10):
        if not np.array_equal(st.ecc_match(list(fr), ecc), ref_e): bad += 1
        dk, ok = st.keypoint_match(list(fr), kp)
        if dk != ref_k[0] or not np.array_equal(ok, ref_k[1]): bad += 1
    print(f"host-fed, upload_batch {batch}: {bad} of 20 calls differ", flush=True)
st.set_option("upload_batch", 8)
