#!/usr/bin/env python3
"""bench.py — frames/sec aligned+stacked on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over this rank's shard of a synthetic stack that is already
resident in HBM: ecc_match (grey -> blur -> ECC homography, 5000 iters / eps 1e-5 / gauss 5, the
reference example's parameters, examples/main.rs:107-112) -> warpPerspective -> f32 accumulate,
then the cross-rank reduce of the accumulator (RCCL) and the final 1/n scale on rank 0.

Default workload = BASELINE.json configs[3]: ONE 256-frame 3840x2160 BGR u8 stack, the moving frames
sharded contiguously over the N ranks (256/N frames per GPU; N = 8 gives the 32 per GPU configs[3] names;
N = 1 holds all 256: 6.4 GB of frames + 8.5 GB of ECC templates). Total work is fixed: "scaling": "strong".
One process per GPU, no data-path collective except the single accumulator reduce.
--frames-per-gpu F runs F frames on every rank instead (weak scaling, for experiments).

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable

WORKLOADS = {
    # name: (width, height, frames in the WHOLE stack, api)
    "ecc_4k": (3840, 2160, 256, "ecc"),                 # BASELINE configs[3] — the metric's configuration
    "ecc_1080p": (1920, 1080, 64, "ecc"),               # configs[2]
    "keypoint_1080p": (1920, 1080, 64, "keypoint"),     # configs[1]
    "ecc_small": (640, 480, 8, "ecc"),
    # BASELINE configs[4] (an extension beyond the reference): 16-bit 4K stack, ORB-seeded ECC refine; 1024 frames in
    # BASELINE, 256 here so that the u16 stack (12.7 GB) + templates + ORB workspace fit next to each other comfortably
    "hybrid_4k16": (3840, 2160, 256, "hybrid"),
}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="ecc_4k", choices=sorted(WORKLOADS))
    ap.add_argument("--frames-per-gpu", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-frames", type=int, default=0, help="frames in the CPU baseline sample (0 = cores+1)")
    ap.add_argument("--ecc-slots", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="engine tuning knob name=value (stk_set_option)")
    ap.add_argument("--no-reduce-overlap", action="store_true",
                    help="N > 1: wait for the accumulator reduce before starting the next step (default: the reduce of step k "
                         "runs under the alignment of step k+1, on a second accumulator)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 rehearsal on a one-GPU box: every rank uses cuda:0 and the reduce goes through gloo/CPU "
                         "(checks the sharded code path, not a performance number)")
    ap.add_argument("--profile-launches", type=int, default=1,
                    help="0: no per-launch timing; 1: HIP event pairs around every 3rd ECC iteration launch (roofline.achieved); "
                         "n > 1: around every n-th")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from libstacker_rs_amd import (EccMatchParameters, KeyPointMatchParameters, MotionType, RANSAC, Stacker, synth)
    from libstacker_rs_amd.shard import shard_moving_frames

    W, H, n_global, api = WORKLOADS[args.workload]      # frames in the whole stack (frame 0 = reference)
    scaling = "strong"
    if args.frames_per_gpu > 0:
        n_global, scaling = args.frames_per_gpu * world, "weak"
    fpg = n_global / world
    # contiguous shards of the moving frames 1..n-1; rank 0 also folds frame 0 itself in
    mine = shard_moving_frames(n_global, world, rank)

    t0 = time.time()
    scene = synth.render_scene(W, H)
    frames, G = synth.make_stack(0, W, H, scene=scene, device=dev, indices=[0] + mine, depth=16 if api == "hybrid" else 8)
    torch.cuda.synchronize()
    gen_s = time.time() - t0

    st = Stacker(local_rank)
    st.use_torch_stream()
    if args.ecc_slots:
        st.set_option("ecc_slots", args.ecc_slots)
    for kv in args.opt:
        k, v = kv.split("=")
        st.set_option(k, int(v))
    st.set_option("profile", 2 if args.profile_launches else 1)
    # an event pair around a launch keeps it from being dispatched back to back with its neighbours: bracketing every
    # launch costs ~5 % of `value`, so every 3rd launch is sampled (3 is coprime with the 4-launch polling chunk)
    st.set_option("profile_stride", max(1, args.profile_launches if args.profile_launches > 1 else 3))
    ecc_params = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
    kp_params = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)          # examples/main.rs:69-76
    # two accumulators: while step k's sum is being reduced over xGMI, step k+1 aligns into the other one
    accs = [torch.empty((H, W, 3), dtype=torch.float32, device=dev) for _ in range(2 if world > 1 else 1)]
    cnts = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(len(accs))]
    out = torch.empty_like(accs[0])
    overlap = world > 1 and not args.no_reduce_overlap
    pending = None                                   # the previous step's reduce in flight: (works, acc, counts, staging)
    step_no = 0
    totals = [0, 0]                                  # frames folded / dropped in the last finished stack (rank 0)

    agg = {"ecc_iter_ms": 0.0, "ecc_iter_timed": 0, "ecc_iter_launches": 0, "ecc_slot_iterations": 0, "prep_ms": 0.0, "align_ms": 0.0,
           "warp_ms": 0.0, "warp_frames": 0, "warp_launches": 0}
    last_stats = None

    def start_reduce(acc, counts):
        """The path's one exchange: sum of the per-rank accumulators (+ two counters) to rank 0. RCCL over xGMI."""
        if args.rehearse_on_one_gpu:                 # gloo on CPU copies: control-flow rehearsal only
            stage = (acc.cpu(), counts.cpu())
            works = [dist.reduce(stage[0], dst=0, op=dist.ReduceOp.SUM, async_op=True),
                     dist.reduce(stage[1], dst=0, op=dist.ReduceOp.SUM, async_op=True)]
            return (works, acc, counts, stage)
        works = [dist.reduce(acc, dst=0, op=dist.ReduceOp.SUM, async_op=True),
                 dist.reduce(counts, dst=0, op=dist.ReduceOp.SUM, async_op=True)]
        return (works, acc, counts, None)

    def finish_reduce(p):
        works, acc, counts, stage = p
        for w in works:
            w.wait()                                 # RCCL: the current stream waits for the collective
        if stage is not None:
            acc.copy_(stage[0]); counts.copy_(stage[1])
        if rank == 0:
            totals[0], totals[1] = int(counts[0].item()), int(counts[1].item())
            st.finalize_mean(acc, totals[0], out)

    def step(record: bool):
        nonlocal last_stats, pending, step_no
        acc, counts = accs[step_no % len(accs)], cnts[step_no % len(accs)]
        step_no += 1
        if api == "ecc":
            added, stats = st.ecc_match_shard(frames, ecc_params, rank == 0, acc)
            dropped = 0
        elif api == "hybrid":
            added, stats = st.hybrid_match_shard(frames, kp_params, ecc_params, rank == 0, acc)
            dropped = 0
        else:
            added, dropped, stats = st.keypoint_match_shard(frames, kp_params, rank == 0, acc)
        if record:
            t = st.timing()
            for k in agg:
                agg[k] += t[k]
            last_stats = stats
        counts[0] = added
        counts[1] = dropped
        if world == 1:
            totals[0], totals[1] = added, dropped
            st.finalize_mean(acc, added, out)
            return
        if pending is not None:                      # its reduce ran under the alignment that just finished
            finish_reduce(pending)
            pending = None
        p = start_reduce(acc, counts)
        if overlap:
            pending = p
        else:
            finish_reduce(p)
            torch.cuda.current_stream().synchronize()

    def drain():
        nonlocal pending
        if pending is not None:
            finish_reduce(pending)
            pending = None

    def fence():
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    frames_per_step = n_global
    value = frames_per_step * args.steps / elapsed

    if rank == 0:
        px = W * H
        res = {
            "metric": "frames/sec aligned+stacked (4K RGB, ECC homography)" if args.workload == "ecc_4k"
                      else f"frames/sec aligned+stacked ({args.workload})",
            "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {n_global}-frame {W}x{H} BGR {'u16' if api == 'hybrid' else 'u8'} stack, "
                                   + ("ORB-seeded ecc_match (extension, 16-bit) Homography max_count 5000 eps 1e-5 gauss 5" if api == "hybrid" else
                                      "ecc_match Homography max_count 5000 eps 1e-5 gauss 5" if api == "ecc"
                                      else "keypoint_match RANSAC thr 5.0 ratio 0.9 keep 0.80")
                                   + f", {fpg:g} frames/GPU resident in HBM, frame-sharded, one accumulator reduce",
                       "frames_per_gpu": fpg, "frames_total": n_global, "width": W, "height": H, "parallelism": f"frame-shard x{world}",
                       "accumulator_reduce": ("none (1 GPU)" if world == 1 else
                                              "RCCL reduce to rank 0, overlapped with the next step's alignment (double-buffered)"
                                              if overlap else "RCCL reduce to rank 0, waited for before the next step")},
        }
        # ---- roofline of the dominant kernel --------------------------------------------------
        if api in ("ecc", "hybrid") and agg["ecc_iter_timed"] > 0:
            # ECC iteration kernel: ALGORITHMIC bytes = 16 B/px per frame-iteration (template 4 B +
            # frame-0 image/gx/gy 12 B); one launch advances `slots` frames by one iteration.
            alg_bytes_total = 16.0 * px * agg["ecc_slot_iterations"]
            launches = agg["ecc_iter_launches"]                     # every launch of the timed region (incl. drained no-ops)
            avg_ms = agg["ecc_iter_ms"] / agg["ecc_iter_timed"]     # event-timed sample: every 3rd of them
            achieved = (alg_bytes_total / launches) / (avg_ms * 1e-3) / 1e9
            # HBM-side traffic per launch from the PMC passes of the SAME command (tools/profile_round.sh,
            # separate FETCH_SIZE / WRITE_SIZE passes; summary committed under profiles/): FETCH_SIZE is
            # doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950 (the float4 scale_kernel in the same
            # run calibrates exactly 1/2), WRITE_SIZE is taken as reported. null if no summary is present.
            traffic = None
            if args.workload == "ecc_4k" and world == 1 and not args.opt and not args.ecc_slots:
                try:
                    allk = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc_summary.json")))
                    pm = next(v for k, v in allk.items() if k.startswith("stk::ecc_iter_h8_kernel"))
                    traffic = round(2 * pm["FETCH_SIZE_bytes_per_dispatch"] + pm["WRITE_SIZE_bytes_per_dispatch"], 1)
                except Exception:
                    traffic = None
            res["roofline"] = {"kernel": "ecc_iter_h8_kernel (ECC iteration pass, homography)", "bound": "hbm", "achieved": round(achieved, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                               "traffic": traffic, "traffic_source": "rocprofv3 --pmc (profiles/r01/pmc_summary.json)" if traffic else None,
                               "avg_launch_ms": round(avg_ms, 5), "launches": launches, "launches_timed": agg["ecc_iter_timed"],
                               "alg_bytes_per_launch": round(alg_bytes_total / launches, 1)}
        its = [s["iterations"] for s in (last_stats or [])[1:]]
        src_b = 3 * px
        warp_bytes = (agg["warp_frames"] * src_b + agg["warp_launches"] * 2 * 12 * px)
        res["stages"] = {
            "prep_ms_per_step": round(agg["prep_ms"] / args.steps, 3),
            "align_ms_per_step": round(agg["align_ms"] / args.steps, 3),
            "warp_ms_per_step": round(agg["warp_ms"] / args.steps, 3),
            "ecc_iterations_mean": round(float(np.mean(its)), 2) if its else None,
            "ecc_iterations_max": int(max(its)) if its else None,
            "warp_accumulate_GBps_fused": round(warp_bytes / max(agg["warp_ms"], 1e-9) / 1e6, 1),
            "warp_accumulate_GBps_survey_bytes": round(agg["warp_frames"] * (src_b + 24 * px) / max(agg["warp_ms"], 1e-9) / 1e6, 1),
            "synthetic_generation_s": round(gen_s, 1),
            "frames_folded_last_step": totals[0], "frames_dropped_last_step": totals[1],
        }
        # ---- CPU baseline: the oracle (a port of the reference's OpenCV/Rayon path) on host cores ----
        if not args.no_cpu_baseline and world == 1:
            import oracle
            cores = os.cpu_count() or 1
            use = min(cores, 32)
            n_s = args.cpu_sample_frames or (use + 1)
            n_s = max(2, min(n_s, frames.shape[0]))
            sample = [f.cpu().numpy() for f in frames[:n_s]]
            tc = time.perf_counter()
            if api == "ecc":
                oracle.ecc_match(sample, max_count=5000, epsilon=1e-5, gauss_filt_size=5, n_threads=use)
            elif api == "hybrid":
                use, sample = 1, sample[:3]                     # the oracle's hybrid path is a serial Python composition
                n_s = len(sample)
                oracle.hybrid_match(sample)
            else:
                oracle.keypoint_match(sample, n_threads=use)
            cpu_s = time.perf_counter() - tc
            res["cpu_baseline"] = {"value": round(n_s / cpu_s, 4), "unit": "frames/s", "cores": use, "kind": "port",
                                   "sample": f"first {n_s} frames of the same {W}x{H} stack, one oracle pass ({cpu_s:.1f} s), "
                                             + ("serial Python composition of the oracle's stages" if api == "hybrid"
                                                else "frame-parallel OpenMP like the reference's Rayon fold")}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
