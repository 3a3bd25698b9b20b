"""Rank body of tests/test_gpu_00_world2.py: two processes, both on cuda:0, each folds its contiguous range of the
moving frames through the product's *_shard entry points; the sums and counters are reduced with gloo (RCCL needs one
GPU per rank); rank 0 finalises and compares with the single-context result. Writes a JSON verdict."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libstacker_rs_amd import (EccMatchParameters, KeyPointMatchParameters, MotionType, RANSAC, Stacker, synth)  # noqa: E402
from libstacker_rs_amd.shard import shard_moving_frames  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    verdict = {"ok": False}
    try:
        frames, _ = synth.make_stack(9, 320, 240)
        frames = frames.numpy().copy()
        frames[5] = 128                                    # featureless frame: keypoint_match drops it, ecc cannot use it
        n = len(frames)
        st = Stacker(0)
        mine = shard_moving_frames(n, world, rank)
        sub = torch.from_numpy(np.ascontiguousarray(frames[[0] + mine])).cuda()
        kp = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)
        acc = torch.empty((240, 320, 3), dtype=torch.float32, device="cuda")
        added, dropped, stats = st.keypoint_match_shard(sub, kp, rank == 0, acc)
        t_acc, counts = acc.cpu(), torch.tensor([added, dropped], dtype=torch.int64)
        dist.reduce(t_acc, dst=0, op=dist.ReduceOp.SUM)
        dist.reduce(counts, dst=0, op=dist.ReduceOp.SUM)
        gathered = [None] * world
        dist.gather_object([(g, s["status"], s["warp"].tolist()) for g, s in zip(mine, stats[1:])], gathered if rank == 0 else None, dst=0)
        # ecc on the stack without the flat frame
        keep = [i for i in range(n) if i != 5]
        fr_e = frames[keep]
        mine_e = shard_moving_frames(len(fr_e), world, rank)
        sub_e = torch.from_numpy(np.ascontiguousarray(fr_e[[0] + mine_e])).cuda()
        ep = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
        acc_e = torch.empty_like(acc)
        added_e, stats_e = st.ecc_match_shard(sub_e, ep, rank == 0, acc_e)
        t_e, c_e = acc_e.cpu(), torch.tensor([added_e], dtype=torch.int64)
        dist.reduce(t_e, dst=0, op=dist.ReduceOp.SUM)
        dist.reduce(c_e, dst=0, op=dist.ReduceOp.SUM)
        if rank == 0:
            tot_added, tot_dropped = int(counts[0]), int(counts[1])
            out = st.finalize_mean(t_acc.cuda(), tot_added).cpu().numpy()           # divisor = added = n - dropped
            d1, full, full_stats = st.keypoint_match(list(frames), kp, return_stats=True)
            per_frame_equal = all(status == full_stats[g]["status"] and np.array_equal(np.array(w), full_stats[g]["warp"])
                                  for part in gathered for g, status, w in part)
            out_e = st.finalize_mean(t_e.cuda(), int(c_e[0])).cpu().numpy()
            full_e = st.ecc_match(list(fr_e), ep)
            verdict = {"ok": True, "added": tot_added, "dropped": tot_dropped, "single_dropped": d1,
                       "kp_max_abs": float(np.max(np.abs(out - full))), "kp_per_frame_equal": bool(per_frame_equal),
                       "ecc_added": int(c_e[0]), "ecc_max_abs": float(np.max(np.abs(out_e - full_e)))}
    except Exception as e:  # noqa: BLE001
        verdict = {"ok": False, "error": repr(e)}
        raise
    finally:
        if rank == 0:
            with open(os.environ["WORLD2_RESULT"], "w") as f:
                json.dump(verdict, f)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
