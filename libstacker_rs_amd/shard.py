"""Frame sharding across ranks (one process per GPU) and the single exchange step of the path.

The reference folds frames with a Rayon try_fold/try_reduce (lib.rs:188-335, 746-833): every frame
i > 0 depends only on frame 0 and itself, and the accumulators are summed pairwise. Here the moving
frames 1..n-1 are split into contiguous ranges, one per rank; every rank also holds frame 0 (its
derived planes / descriptors are recomputed locally, cheaper than a broadcast), exactly one rank adds
frame 0 itself, and the only collective is a sum-reduce of the f32 accumulator plus two counters to
rank 0 (RCCL over xGMI with backend "nccl"; gloo on CPU in the tests). Rank 0 then divides by
n - dropped (lib.rs:339-345, 836-839).
"""
from __future__ import annotations

from typing import List, Sequence


def shard_moving_frames(n_frames: int, world_size: int, rank: int) -> List[int]:
    """Global indices (1-based, frame 0 excluded) of the moving frames rank `rank` aligns: contiguous,
    sizes differing by at most one, earlier ranks take the remainder."""
    if n_frames <= 0 or world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad shard arguments")
    moving = n_frames - 1
    base, rem = divmod(moving, world_size)
    lo = 1 + rank * base + min(rank, rem)
    cnt = base + (1 if rank < rem else 0)
    return list(range(lo, lo + cnt))


def shard_frame_list(frames: Sequence, world_size: int, rank: int) -> list:
    """frames[0] followed by this rank's slice of frames[1:] — the `files` argument of the *_shard calls."""
    return [frames[0]] + [frames[i] for i in shard_moving_frames(len(frames), world_size, rank)]


def reduce_to_root(acc, counts, group=None):
    """Sum-reduce the accumulator tensor and the [added, dropped] counters to rank 0 (no-op for one rank)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(acc, dst=0, op=dist.ReduceOp.SUM, group=group)
        dist.reduce(counts, dst=0, op=dist.ReduceOp.SUM, group=group)
    return acc, counts
