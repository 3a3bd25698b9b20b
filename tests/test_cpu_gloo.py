"""world_size-2 test of the N>1 path on CPU (gloo): shard -> per-rank fold -> sum-reduce -> finalize.

On the GPU each rank's fold is stk_ecc_match_shard / stk_keypoint_match_shard; here the CPU oracle
stands in for the per-rank worker so the sharding arithmetic, the single collective and the final
1/(n - dropped) scale are exercised exactly as bench.py drives them.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from libstacker_rs_amd import shard, synth


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, frames, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        mine = shard.shard_frame_list(list(frames), world, rank)
        h, w, _ = frames[0].shape
        # per-rank un-normalised fold (what stk_ecc_match_shard returns): frame 0 only on rank 0
        g0 = oracle.grey(mine[0])
        acc = np.zeros((h, w, 3), np.float32)
        added = 0
        if rank == 0:
            acc = oracle.warp_frame(mine[0], np.eye(3), acc=acc)
            added += 1
        for f in mine[1:]:
            rc, W, _, _ = oracle.find_transform_ecc(oracle.grey(f), g0, np.eye(3), oracle.MOTION_HOMOGRAPHY, 5000, 1e-5, 5)
            assert rc == 0
            acc = oracle.warp_frame(f, W.astype(np.float64), acc=acc)
            added += 1
        t_acc = torch.from_numpy(acc)
        counts = torch.tensor([added, 0], dtype=torch.int64)
        shard.reduce_to_root(t_acc, counts)
        if rank == 0:
            # divisor = frames actually added over all ranks (= n - dropped: `added` never counts a dropped frame)
            out = oracle.scale(t_acc.numpy(), int(counts[0]))
            np.save(os.path.join(tmpdir, "out.npy"), out)
            np.save(os.path.join(tmpdir, "counts.npy"), counts.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_shard_reduce_matches_single_process(tmp_path):
    frames, _ = synth.make_stack(5, 160, 120, seed=7)
    frames = frames.numpy()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, frames, str(tmp_path)), nprocs=2, join=True)
    out = np.load(tmp_path / "out.npy")
    counts = np.load(tmp_path / "counts.npy")
    assert list(counts) == [5, 0]
    ref, _, _ = oracle.ecc_match(list(frames), n_threads=1)
    # same frames, same warps; only the f32 summation order differs (two partial sums instead of one)
    assert np.max(np.abs(out - ref)) <= 1e-6
