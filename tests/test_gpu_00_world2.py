"""N > 1 through the PRODUCT's shard entry points with a real cross-process reduce: two ranks on the one GPU of the box
(gloo for the exchange; RCCL wants a GPU per rank), compared with the single-context run. Named *_00_* so that it runs
before any test of this session has initialised the GPU in the parent process (the children are separate programs
started with subprocess before the parent touches HIP). The multi-GPU RCCL run itself is the driver's (bench.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_two_ranks_through_the_shard_entry_points(tmp_path):
    import socket
    with socket.socket() as sk:                      # a free rendezvous port (the box is shared with nothing, but be polite)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    res = tmp_path / "world2.json"
    env = dict(os.environ, WORLD2_RESULT=str(res), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "world2_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    v = json.loads(res.read_text())
    assert v["ok"], v
    assert v["dropped"] == 1 and v["single_dropped"] == 1 and v["added"] == 8          # n - dropped, summed over ranks
    assert v["kp_per_frame_equal"]                         # every frame's status and H, bit for bit
    assert v["kp_max_abs"] <= 1e-6                         # image: only the order of the f32 adds differs
    assert v["ecc_added"] == 8 and v["ecc_max_abs"] <= 1e-6   # per-frame warps identical for any split (fixed workgroup partition)
