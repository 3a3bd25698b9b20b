"""bench.py's N > 1 code — the code the driver's 1/2/4/8-GPU SCALE run executes — rehearsed on the box's one GPU: two
torchrun ranks on cuda:0 with the accumulator reduce through gloo (--rehearse-on-one-gpu), with and without the
double-buffered reduce overlap, and the one-process form (--single-process, stk_create_multi). Checks the control flow
(shards, double buffering, fence, MAX of the times, the JSON contract), not a performance number. Named *_00_* so the
children start before this process has touched the GPU; replaces nothing in the reference — it guards the measurement of
its fold / reduce (lib.rs:188-335, 746-833)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
COMMON = ["--gpus", "2", "--rehearse-on-one-gpu", "--workload", "ecc_small", "--steps", "2", "--warmup", "1",
          "--no-cpu-baseline", "--host-fed-steps", "0"]


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]                # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def _check_contract(v, n_gpus):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in v, key
    assert v["n_gpus"] == n_gpus and v["steps"] == 2 and v["warmup"] == 1
    assert v["value"] > 0 and v["ms_per_step"] > 0 and v["higher_is_better"] is True
    assert v["scaling"] == "strong" and v["unit"] == "frames/s" and v["data"] == "synthetic" and v["vs_baseline"] is None
    assert v["config"]["frames_total"] == 8 and v["config"]["frames_per_gpu"] == 4
    # value = frames of the WHOLE stack per second
    assert abs(v["value"] - 8 * 2 / (v["ms_per_step"] * 2e-3)) <= 1e-2 * v["value"]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("extra", [[], ["--no-reduce-overlap"]], ids=["reduce-overlapped", "reduce-waited"])
def test_bench_two_ranks_under_torchrun(extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH] + COMMON + extra
    r = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=840, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    v = _json_line(r.stdout)
    _check_contract(v, 2)
    assert v["stages"]["frames_folded_last_step"] == v["config"]["frames_total"] == 8     # both shards arrived in the sum
    assert v["stages"]["frames_dropped_last_step"] == 0
    assert ("overlapped" in v["config"]["accumulator_reduce"]) == (not extra)
    assert "roofline" in v and v["roofline"]["frac"] > 0


@pytest.mark.timeout(900)
def test_bench_single_process_two_members():
    cmd = [sys.executable, BENCH, "--single-process"] + COMMON
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=840, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    v = _json_line(r.stdout)
    _check_contract(v, 2)
    assert "ONE process" in v["config"]["parallelism"]


@pytest.mark.timeout(900)
def test_bench_four_ranks_under_torchrun():
    """Four ranks on the box's one GPU (the process guard allows six): two frames per rank, three of them without the reference
    frame in their fold — the shard arithmetic, the double-buffered reduce and the counters behind the accumulator at N = 4."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH, "--gpus", "4", "--rehearse-on-one-gpu", "--workload", "ecc_small", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--host-fed-steps", "0"]
    r = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=840, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    v = _json_line(r.stdout)
    assert v["n_gpus"] == 4 and v["config"]["frames_total"] == 8 and v["config"]["frames_per_gpu"] == 2
    assert v["stages"]["frames_folded_last_step"] == 8 and v["stages"]["frames_dropped_last_step"] == 0
    assert v["scaling"] == "strong" and v["value"] > 0


@pytest.mark.timeout(900)
def test_bench_two_ranks_keypoint_workload():
    """configs[1] cut over two ranks (32 frames each, every rank in three lanes; rank 1 folds without the reference frame):
    all 64 frames arrive in the reduced sum."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH, "--gpus", "2", "--rehearse-on-one-gpu", "--workload", "keypoint_1080p", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--host-fed-steps", "0"]
    r = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=840, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    v = _json_line(r.stdout)
    assert v["n_gpus"] == 2 and v["config"]["frames_total"] == 64 and v["config"]["frames_per_gpu"] == 32
    assert v["stages"]["frames_folded_last_step"] == 64 and v["stages"]["frames_dropped_last_step"] == 0
