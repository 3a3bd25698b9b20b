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

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); what a float4 stream achieves is measured in the run

WORKLOADS = {
    # name: (width, height, frames in the WHOLE stack, api)
    "ecc_4k": (3840, 2160, 256, "ecc"),                 # BASELINE configs[3] — the metric's configuration
    "ecc_1080p": (1920, 1080, 64, "ecc"),               # configs[2]
    "keypoint_1080p": (1920, 1080, 64, "keypoint"),     # configs[1]
    "ecc_small": (640, 480, 8, "ecc"),
    # BASELINE configs[4] (an extension beyond the reference): the 1024-frame 16-bit 4K stack, ORB-seeded ECC refine
    # (51 GB of u16 frames + 34 GB of templates + the ORB workspace on one GPU; 128 frames per GPU at N = 8)
    "hybrid_4k16": (3840, 2160, 1024, "hybrid"),
}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # default: ~6 s of timed region on the default workload (57 ms per step), long enough for a 5 s utilisation sampler
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="ecc_4k", choices=sorted(WORKLOADS))
    ap.add_argument("--frames-per-gpu", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-frames", type=int, default=0, help="frames in the CPU baseline sample (0 = cores+1)")
    ap.add_argument("--ecc-slots", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="engine tuning knob name=value (stk_set_option)")
    ap.add_argument("--no-reduce-overlap", action="store_true",
                    help="N > 1: wait for the accumulator reduce before starting the next step (default: the reduce of step k "
                         "runs under the alignment of step k+1, on a second accumulator)")
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1 from ONE process: a multi-device context (stk_create_multi) shards, runs one host thread per GPU "
                         "and reduces with RCCL inside the library — the call shape of the Rust drop-in. Launch WITHOUT torchrun.")
    ap.add_argument("--host-fed-steps", type=int, default=2,
                    help="N = 1: extra steps after the timed region with the stack in PINNED HOST memory (H2D included); 0 = skip")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 rehearsal on a one-GPU box: every rank uses cuda:0 and the reduce goes through gloo/CPU "
                         "(checks the sharded code path, not a performance number)")
    ap.add_argument("--files", default="", choices=["", "tiff", "pnm"],
                    help="N = 1, ecc / keypoint workloads: ALSO time the reference's own call shape (a list of file paths, "
                         "lib.rs:129-137, 702-710): the stack is written once to --files-dir as uncompressed TIFF or binary PNM, "
                         "then stk_*_match_files decodes on host threads into pinned memory while the engine runs; never `value`")
    ap.add_argument("--files-dir", default="/dev/shm", help="where --files writes the stack (tmpfs: the figure is decode + engine, not disk)")
    ap.add_argument("--files-steps", type=int, default=2)
    ap.add_argument("--profile-launches", type=int, default=1,
                    help="0: no roofline sample; n >= 1: ONE extra step after the timed region (never inside it) with a HIP event "
                         "pair around every n-th ECC iteration launch (roofline.achieved)")
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
        n_global, scaling = args.frames_per_gpu * (args.gpus if args.single_process else world), "weak"
    ecc_params = EccMatchParameters(MotionType.Homography, 5000, 1e-5, 5)
    kp_params = KeyPointMatchParameters(RANSAC, 5.0, 0.80, 0.9)          # examples/main.rs:69-76
    depth = 16 if api == "hybrid" else 8
    px = W * H

    if args.single_process:
        return single_process(args, torch, np, synth, Stacker, shard_moving_frames, W, H, n_global, api, depth, ecc_params, kp_params, scaling)

    fpg = n_global / world
    # contiguous shards of the moving frames 1..n-1; rank 0 also folds frame 0 itself in
    mine = shard_moving_frames(n_global, world, rank)

    t0 = time.time()
    scene = synth.render_scene(W, H)
    frames, G = synth.make_stack(0, W, H, scene=scene, device=dev, indices=[0] + mine, depth=depth)
    torch.cuda.synchronize()
    gen_s = time.time() - t0

    st = Stacker(local_rank)
    st.use_torch_stream()
    if args.ecc_slots:
        st.set_option("ecc_slots", args.ecc_slots)
    for kv in args.opt:
        k, v = kv.split("=")
        st.set_option(k, int(v))
    # The timed region runs with stage timers only (profile = 1). An event pair around a launch keeps it from being
    # dispatched back to back with its neighbours (~5 % of the step if every launch is bracketed), so the per-launch
    # sample the roofline needs is taken in ONE extra step after the timed region, like the prep kernel's figure.
    st.set_option("profile", 1)
    stride = max(1, args.profile_launches)
    # two accumulators: while step k's sum is being reduced over xGMI, step k+1 aligns into the other one
    # (the two counters of a stack — frames folded, frames dropped — ride behind the accumulator in the same buffer, as floats:
    # ONE collective per stack instead of two, one small host -> device copy instead of two scalar writes, one read-back)
    n_acc = H * W * 3
    flats = [torch.empty(n_acc + 4, dtype=torch.float32, device=dev) for _ in range(2 if world > 1 else 1)]
    accs = [f[:n_acc].view(H, W, 3) for f in flats]
    cnts = [f[n_acc:] for f in flats]
    cnt_host = [torch.zeros(4, dtype=torch.float32).pin_memory() for _ in flats]
    out = torch.empty_like(accs[0])
    overlap = world > 1 and not args.no_reduce_overlap
    pending = None                                   # the previous step's reduce in flight: (work, flat buffer, staging)
    step_no = 0
    totals = [0, 0]                                  # frames folded / dropped in the last finished stack (rank 0)

    agg = {k: 0 for k in ("ecc_iter_launches", "ecc_slot_iterations", "ecc_ring_fallbacks", "prep_ms", "align_ms", "warp_ms",
                          "warp_frames", "warp_launches", "fast_ms", "fast_launches", "fast_pixels")}
    last_stats = None

    def start_reduce(flat):
        """The path's one exchange: sum of the per-rank accumulators (+ the two counters behind them) to rank 0. RCCL over xGMI."""
        if args.rehearse_on_one_gpu:                 # gloo on CPU copies: control-flow rehearsal only
            stage = flat.cpu()
            return (dist.reduce(stage, dst=0, op=dist.ReduceOp.SUM, async_op=True), flat, stage)
        return (dist.reduce(flat, dst=0, op=dist.ReduceOp.SUM, async_op=True), flat, None)

    def finish_reduce(p):
        work, flat, stage = p
        work.wait()                                  # RCCL: the current stream waits for the collective
        if stage is not None:
            flat.copy_(stage)
        if rank == 0:
            c = flat[n_acc:n_acc + 2].tolist()       # counters are small integers: exact in f32
            totals[0], totals[1] = int(round(c[0])), int(round(c[1]))
            st.finalize_mean(flat[:n_acc].view(H, W, 3), totals[0], out)

    def run_shard(src, acc, want_stats=False):
        # per-frame statistics are marshalled into Python objects only when asked for (the last step): that is host time of
        # the harness, not of the path
        if api == "ecc":
            added, stats = st.ecc_match_shard(src, ecc_params, rank == 0, acc, return_stats=want_stats)
            return added, 0, stats
        if api == "hybrid":
            added, stats = st.hybrid_match_shard(src, kp_params, ecc_params, rank == 0, acc, return_stats=want_stats)
            return added, 0, stats
        return st.keypoint_match_shard(src, kp_params, rank == 0, acc, return_stats=want_stats)

    def step(record: bool, last: bool = False):
        nonlocal last_stats, pending, step_no
        b = step_no % len(accs)
        acc, counts, flat = accs[b], cnts[b], flats[b]
        step_no += 1
        added, dropped, stats = run_shard(frames, acc, want_stats=last)
        if record:
            t = st.timing()
            for k in agg:
                agg[k] += t[k]
            if stats is not None:
                last_stats = stats
        if world == 1:
            totals[0], totals[1] = added, dropped
            st.finalize_mean(acc, added, out)
            return
        cnt_host[b][0] = float(added); cnt_host[b][1] = float(dropped)
        counts.copy_(cnt_host[b], non_blocking=True)
        if pending is not None:                      # its reduce ran under the alignment that just finished
            finish_reduce(pending)
            pending = None
        p = start_reduce(flat)
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
    for k in range(args.steps):
        step(True, last=k == args.steps - 1)
    fence()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    value = n_global * args.steps / elapsed

    # ---- after the timed region: ONE step with an event pair around every `stride`-th ECC iteration launch (rank 0) ----
    sample = None
    if rank == 0 and api in ("ecc", "hybrid") and args.profile_launches > 0:
        st.set_option("profile", 2)
        st.set_option("profile_stride", stride)
        try:
            run_shard(frames, accs[0])
            t = st.timing()
            sample = {k: t[k] for k in ("ecc_iter_ms", "ecc_iter_timed", "ecc_iter_launches", "ecc_slot_iterations")}
        finally:
            st.set_option("profile", 1)
    # ---- keypoint / hybrid: ONE step as a single pipeline (kp_lanes = 1), so that the stage timers of FAST and of the fold
    # bracket those kernels alone: in the timed steps the lanes' launches overlap one another, and a per-lane event pair
    # then measures a kernel that shares the device (its bytes / its time would overstate the rate) ----
    kp_sample = None
    if rank == 0 and api in ("keypoint", "hybrid") and args.profile_launches > 0:
        lanes_cfg = 3
        for kv in args.opt:
            if kv.split("=", 1)[0] == "kp_lanes":
                lanes_cfg = int(kv.split("=", 1)[1])
        st.set_option("kp_lanes", 1)
        try:
            run_shard(frames, accs[0])
            t = st.timing()
            kp_sample = {k: t[k] for k in ("fast_ms", "fast_pixels", "fast_launches", "warp_ms", "warp_frames", "warp_launches")}
        finally:
            st.set_option("kp_lanes", lanes_cfg)
    # ---- and what a plain float4 stream achieves on this card in this run: the 1/n scale kernel (read 4 B + write 4 B per
    # float of a W x H x 3 image), best of 5 by its own HIP-event timer ----
    stream_gbs = None
    if rank == 0:
        best = None
        for _ in range(5):
            st.finalize_mean(accs[0], max(totals[0], 1), out)
            ms = st.timing()["finalize_ms"]
            best = ms if best is None or ms < best else best
        if best and best > 0:
            stream_gbs = 8.0 * 3 * px / best / 1e6

    if rank == 0:
        res = result_header(args, value, elapsed, world, scaling, api, W, H, n_global, fpg,
                            "none (1 GPU)" if world == 1 else
                            "RCCL reduce to rank 0, overlapped with the next step's alignment (double-buffered)" if overlap
                            else "RCCL reduce to rank 0, waited for before the next step", f"frame-shard x{world}, one process per GPU")
        src_b = 3 * px * (2 if depth == 16 else 1)
        kernels = []
        # ---- roofline of the dominant kernel and of the others on the path (live HIP-event timings of THIS run) ----
        if sample is not None and sample["ecc_iter_timed"] > 0:
            # ECC iteration kernel: ALGORITHMIC bytes = 16 B/px per frame-iteration (template 4 B + frame-0 image/gx/gy
            # 12 B); one launch advances up to `slots` frames by one iteration. All figures of this block are
            # from the sample step (same stack, same schedule as every timed step).
            alg_bytes_total = 16.0 * px * sample["ecc_slot_iterations"]
            launches = sample["ecc_iter_launches"]                  # every launch of the sample step (incl. drained no-ops)
            avg_ms = sample["ecc_iter_ms"] / sample["ecc_iter_timed"]     # mean duration of the event-bracketed ones
            achieved = (alg_bytes_total / launches) / (avg_ms * 1e-3) / 1e9
            traffic, tsrc = pmc_traffic(args, world, "stk::ecc_iter_col_kernel<3>")
            res["roofline"] = {
                "kernel": "ecc_iter_col_kernel<homography> (ECC iteration pass)",
                # `bound` names the roofline the figure is priced against (the contract's vocabulary: hbm | mfma). `achieved`
                # is ALGORITHMIC bytes / time: most of those bytes (the frame-0 planes shared by the frames in flight) are
                # served from L2 / Infinity Cache, the HBM itself sees `traffic`; what limits the kernel is VALU issue.
                "bound": "hbm", "limiter": "valu-issue",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "peak_achievable": round(stream_gbs, 1) if stream_gbs else None,
                "peak_achievable_source": "scale_kernel (float4 stream, 4 B read + 4 B write per float of the W x H x 3 image), "
                                          "best of 5 launches after the timed region, its own HIP-event timer; a yardstick for the "
                                          "streaming kernels (fold, grey_blur), NOT for this one: its algorithmic bytes are served "
                                          "from L2 / Infinity Cache for the most part (see traffic)",
                "traffic": traffic, "traffic_source": tsrc,
                "hbm_frac_of_traffic": round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                "avg_launch_ms": round(avg_ms, 5), "launches": launches, "launches_timed": sample["ecc_iter_timed"],
                "timing": f"one extra step AFTER the timed region with a HIP event pair around every {stride}. launch, engine stream "
                          "(rocprofv3 --kernel-trace mean over ALL launches of the same command: profiles/); `value` is timed without event pairs",
                "alg_bytes_per_launch": round(alg_bytes_total / launches, 1)}
            valu = valu_roofline(args, world, "stk::ecc_iter_col_kernel<3>", px, sample, avg_ms)
            if valu:
                res["roofline"]["valu"] = valu
            kernels.append({"kernel": "ecc_iter_col_kernel<homography>", "bytes": "16 B/px/frame-iteration (algorithmic; cache-served in part)",
                            "GBps": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4), "ms_per_step": round(avg_ms * launches, 3)})
        # fold and FAST: from the single-pipeline sample step where there is one (keypoint, hybrid), else from the timed steps
        wsrc = kp_sample if kp_sample is not None else dict(agg, **{"_steps": args.steps})
        wsteps = 1 if kp_sample is not None else args.steps
        alone = "one extra single-pipeline step (kp_lanes = 1) after the timed region" if kp_sample is not None else "stage timers of the timed steps"
        if wsrc["warp_ms"] > 0:
            # fused fold: every frame's source read once + the accumulator written once per launch (no read: the launch
            # overwrites); SURVEY 8d's per-frame figure (source + accumulator read + write for EVERY frame) is the unfused cost
            wb = wsrc["warp_frames"] * src_b + wsrc["warp_launches"] * 12 * px
            g = wb / wsrc["warp_ms"] / 1e6
            kernels.append({"kernel": "warp_accumulate_%s" % ("u16c3" if depth == 16 else "u8c3"),
                            "bytes": "frames x %d B/px source + 12 B/px accumulator write per launch" % (src_b // px),
                            "GBps": round(g, 1), "frac": round(g / HBM_PEAK_GBS, 4), "ms_per_step": round(wsrc["warp_ms"] / wsteps, 3),
                            "timing": alone})
        if api == "ecc" and world == 1:
            # template preparation: in the timed steps it runs on the prep stream under the iteration of the first frames
            # (its time is inside align_ms); its own stage timer needs a step with the overlap switched off, outside the
            # timed region
            st.set_option("prep_overlap", 0)
            try:
                run_shard(frames, accs[0])
                prep_ms = st.timing()["prep_ms"]
            finally:
                st.set_option("prep_overlap", 1)
            pb = (src_b + 4 * px) * (frames.shape[0] - 1) + (src_b + 24 * px)
            g = pb / prep_ms / 1e6
            kernels.append({"kernel": "grey_blur_stream (+ ref_planes once)", "bytes": "3 B/px read + 4 B/px template write per frame",
                            "GBps": round(g, 1), "frac": round(g / HBM_PEAK_GBS, 4), "ms_per_step": round(prep_ms, 3),
                            "note": "one extra step with prep_overlap = 0 (not in the timed region); in the timed steps this runs "
                                    "under the ECC iteration of the first frames"})
        if wsrc["fast_ms"] > 0:
            # FAST-9/16 + NMS + Harris short list over the 8 pyramid levels: reads each level's u8 pixels once
            g = wsrc["fast_pixels"] / wsrc["fast_ms"] / 1e6
            kernels.append({"kernel": "fast_nms_tiled_all_kernel + threshold + pick + describe (all 8 levels, one launch each per batch)",
                            "bytes": "1 B/px of every pyramid level",
                            "GBps": round(g, 1), "frac": round(g / HBM_PEAK_GBS, 4), "ms_per_step": round(wsrc["fast_ms"] / wsteps, 3),
                            "limiter": "integer VALU (ring tests)", "timing": alone})
        if api == "keypoint" and kernels:
            k = kernels[-1]
            res["roofline"] = {"kernel": k["kernel"], "bound": "hbm", "limiter": "valu-issue", "achieved": k["GBps"], "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": k["frac"], "traffic": None,
                               "peak_achievable": round(stream_gbs, 1) if stream_gbs else None,
                               "timing": "HIP events around the FAST launches of every ORB batch, engine stream; " + alone}
        res["kernels"] = kernels
        its = [s["iterations"] for s in (last_stats or [])[1:]]
        res["stages"] = {
            "prep_ms_per_step": round(agg["prep_ms"] / args.steps, 3),
            "align_ms_per_step": round(agg["align_ms"] / args.steps, 3),
            "warp_ms_per_step": round(agg["warp_ms"] / args.steps, 3),
            "ecc_iterations_mean": round(float(np.mean(its)), 2) if its else None,
            "ecc_iterations_max": int(max(its)) if its else None,
            "synthetic_generation_s": round(gen_s, 1),
            "frames_folded_last_step": totals[0], "frames_dropped_last_step": totals[1],
            "ecc_ring_fallbacks_per_step": round(agg["ecc_ring_fallbacks"] / args.steps, 3),
        }
        # ---- host-fed: the same stack in PINNED HOST memory, H2D inside the timed region (SURVEY 8d metric ii) ----
        if world == 1 and args.host_fed_steps > 0:
            host = frames.cpu().pin_memory()
            run_shard(host, accs[0])                                        # warm-up: staging buffers, first-touch
            torch.cuda.synchronize()
            th = time.perf_counter()
            h2d_ms = h2d_b = 0.0
            for _ in range(args.host_fed_steps):
                added, dropped, _ = run_shard(host, accs[0])
                st.finalize_mean(accs[0], added, out)
                t = st.timing()
                h2d_ms += t["h2d_ms"]; h2d_b += t["h2d_bytes"]
            torch.cuda.synchronize()
            eh = time.perf_counter() - th
            link = h2d_b / max(h2d_ms, 1e-9) / 1e6
            res["host_fed"] = {"value": round(n_global * args.host_fed_steps / eh, 3), "unit": "frames/s", "steps": args.host_fed_steps,
                               "ms_per_step": round(1e3 * eh / args.host_fed_steps, 3),
                               "h2d_GBps": round(link, 2), "h2d_ms_per_step": round(h2d_ms / args.host_fed_steps, 3),
                               # a step cannot finish before its bytes have crossed the link: frames / h2d time = the PCIe-bound ceiling
                               "pcie_bound_ceiling": round(n_global * args.host_fed_steps / (h2d_ms * 1e-3), 1) if h2d_ms > 0 else None,
                               "frac_of_pcie_ceiling": round((n_global * args.host_fed_steps / eh) / (n_global * args.host_fed_steps / (h2d_ms * 1e-3)), 3) if h2d_ms > 0 else None,
                               "note": "frames in pinned host memory; copy stream -> prep stream -> gated ECC queue; never `value`"}
            del host
        # ---- the reference's own call shape: a list of file paths (decode on host threads inside the call) ----
        if world == 1 and args.files and api in ("ecc", "keypoint"):
            res["files"] = files_leg(args, st, frames, api, ecc_params, kp_params, n_global, W, H)
        # ---- CPU baseline: the oracle (a port of the reference's OpenCV/Rayon path) on host cores ----
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(args, api, frames, W, H)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def result_header(args, value, elapsed, world, scaling, api, W, H, n_global, fpg, reduce_desc, parallelism):
    return {
        "metric": "frames/sec aligned+stacked (4K RGB, ECC homography)" if args.workload == "ecc_4k"
                  else f"frames/sec aligned+stacked ({args.workload})",
        "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {n_global}-frame {W}x{H} BGR {'u16' if api == 'hybrid' else 'u8'} stack"
                               + (" = BASELINE configs[3]" if args.workload == "ecc_4k" and n_global == 256 else "") + ", "
                               + ("ORB-seeded ecc_match (extension, 16-bit) Homography max_count 5000 eps 1e-5 gauss 5" if api == "hybrid" else
                                  "ecc_match Homography max_count 5000 eps 1e-5 gauss 5" if api == "ecc"
                                  else "keypoint_match RANSAC thr 5.0 ratio 0.9 keep 0.80")
                               + f", {fpg:g} frames/GPU resident in HBM, frame-sharded, one accumulator reduce",
                   "frames_per_gpu": fpg, "frames_total": n_global, "width": W, "height": H, "parallelism": parallelism,
                   "accumulator_reduce": reduce_desc},
    }


# What a vector instruction costs a gfx950 SIMD (tools/valu_rates.hip, profiles/r04/valu_rates.txt; two or more waves per
# SIMD): plain f32 add / mul / fma and simple 32-bit integer ops ~2.15 cycles per wave64 instruction, every packed-f32,
# conversion, floor / fract, min / max, left shift, multiply, three-operand integer, DPP, compare ... ~4.3, v_rcp_f32 ~8.2.
SLOT_CYCLES = 2.15
PEAK_CLOCK_GHZ = 2.4
N_SIMD = 256 * 4


def valu_roofline(args, world, kernel_prefix, px, sample, avg_ms):
    """The roof that bounds the ECC pass: vector-instruction ISSUE. From the committed counter passes of the same command
    (profiles/r04/pmc_summary.json, pinned to the kernel sources like `traffic`): SQ_INSTS_VALU per launch -> instructions per
    pixel and iteration; the static cost of the ring loop in issue slots (tools/isa_loops.py on the same sources; a packed or
    half-rate instruction is two slots, v_rcp_f32 four) -> the fraction of the chip's issue slots the launch fills, at the
    2.4 GHz peak clock and at the clock the chip held (GRBM_GUI_ACTIVE / 8 / duration)."""
    pm, src = pmc_kernel(args, world, kernel_prefix)
    if not pm or "SQ_INSTS_VALU_per_dispatch" not in pm:
        return None
    px_iter = px * sample["ecc_slot_iterations"] / max(sample["ecc_iter_launches"], 1)       # pixel-iterations per launch, this run
    px_iter_pmc = pm.get("_px_iterations_per_dispatch") or px_iter
    instr_per_px = pm["SQ_INSTS_VALU_per_dispatch"] * 64.0 / px_iter_pmc
    slots_per_px = pm.get("_issue_slots_per_px")
    out = {"instr_per_px": round(instr_per_px, 2), "source": src,
           "slot": "one plain f32 / simple integer wave64 instruction = %.2f SIMD cycles (tools/valu_rates.hip)" % SLOT_CYCLES}
    if slots_per_px:
        slots_per_s = slots_per_px * px_iter / 64.0 / (avg_ms * 1e-3)             # wave-instruction slots the launch consumes per second
        peak = N_SIMD * PEAK_CLOCK_GHZ * 1e9 / SLOT_CYCLES
        out.update({"issue_slots_per_px": round(slots_per_px, 2), "slots_per_s": round(slots_per_s, 1), "slots_per_s_peak": round(peak, 1),
                    "frac": round(slots_per_s / peak, 4)})
        clk = pm.get("_clock_ghz_held")
        if clk:
            out.update({"clock_ghz_held": round(clk, 3), "frac_at_held_clock": round(slots_per_s / (N_SIMD * clk * 1e9 / SLOT_CYCLES), 4)})
    return out


def pmc_kernel(args, world, kernel_prefix):
    """The committed PMC summary entry of a kernel, or (None, reason) when it was taken with other kernel sources."""
    import hashlib
    if args.workload != "ecc_4k" or world != 1 or args.opt or args.ecc_slots or args.frames_per_gpu:
        return None, None
    for rnd in ("r04", "r03"):
        path = os.path.join(ROOT, "profiles", rnd, "pmc_summary.json")
        try:
            allk = json.load(open(path))
            pm = next(v for k, v in allk.items() if k.startswith(kernel_prefix))
            want = allk.get("_kernel_source_sha256")
            h = hashlib.sha256()
            for name in allk.get("_kernel_source_files", ["kernels_ecc_col.hip"]):
                h.update(open(os.path.join(ROOT, "libstacker_rs_amd", "csrc", name), "rb").read())
            if want is None or want != h.hexdigest():
                return None, f"profiles/{rnd}/pmc_summary.json is from older ECC sources: ignored"
            return pm, f"rocprofv3 --pmc, same command (profiles/{rnd}/pmc_summary.json)"
        except Exception:
            continue
    return None, None


def pmc_traffic(args, world, kernel_prefix):
    """HBM-side traffic per launch of the dominant kernel from the PMC passes of the SAME command (tools/profile_round.sh:
    separate FETCH_SIZE / WRITE_SIZE passes, summary committed under profiles/): FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950 (calibrated in the same run on the float4 scale kernel AND on the byte-stream grey kernel),
    WRITE_SIZE as reported. The summary records the SHA-256 of the kernel source it was taken with: a profile of an
    older kernel is ignored (null) rather than reported."""
    pm, src = pmc_kernel(args, world, kernel_prefix)
    if not pm:
        return None, src
    return round(2 * pm["FETCH_SIZE_bytes_per_dispatch"] + pm["WRITE_SIZE_bytes_per_dispatch"], 1), src


def files_leg(args, st, frames, api, ecc_params, kp_params, n_global, W, H):
    """ecc_match(files, ...) / keypoint_match(files, ...) as the reference is called (lib.rs:129-137, 702-710): the synthetic
    stack is written ONCE to tmpfs, then every step decodes it again (a pool of host threads, straight into pinned memory)
    while the engine uploads, prepares and aligns the frames that have arrived."""
    import shutil
    import struct
    import tempfile
    import numpy as np
    d = tempfile.mkdtemp(prefix="stk_bench_", dir=args.files_dir)
    try:
        paths, nbytes = [], 0
        tw = time.perf_counter()
        for i in range(frames.shape[0]):
            a = np.ascontiguousarray(frames[i].cpu().numpy()[..., ::-1])          # BGR -> RGB on disk
            path = os.path.join(d, "f%04d.%s" % (i, "tif" if args.files == "tiff" else "ppm"))
            with open(path, "wb") as f:
                if args.files == "pnm":
                    f.write(("P6\n%d %d\n255\n" % (W, H)).encode())
                else:                                                             # baseline TIFF: little-endian, uncompressed, one strip
                    n_tags, ifd_ofs = 10, 8
                    bps_ofs = ifd_ofs + 2 + 12 * n_tags + 4
                    data_ofs = bps_ofs + 8
                    tag = lambda t, typ, cnt, val: struct.pack("<HHII", t, typ, cnt, val)
                    ifd = struct.pack("<H", n_tags) + tag(256, 4, 1, W) + tag(257, 4, 1, H) + tag(258, 3, 3, bps_ofs) + tag(259, 3, 1, 1) \
                        + tag(262, 3, 1, 2) + tag(273, 4, 1, data_ofs) + tag(277, 3, 1, 3) + tag(278, 4, 1, H) + tag(279, 4, 1, a.nbytes) \
                        + tag(284, 3, 1, 1) + struct.pack("<I", 0)
                    f.write(b"II" + struct.pack("<HI", 42, ifd_ofs) + ifd + struct.pack("<HHHH", 8, 8, 8, 0))
                f.write(a.tobytes())
            nbytes += os.path.getsize(path)
            paths.append(path)
        write_s = time.perf_counter() - tw

        def run():
            if api == "ecc":
                return st.ecc_match_files(paths, ecc_params)
            return st.keypoint_match_files(paths, kp_params)
        run()                                                                     # warm-up: pinned block, page cache
        import torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.files_steps):
            run()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / args.files_steps
        workers = min(len(paths) - 1, max(1, min(os.cpu_count() or 1, 16)))
        return {"value": round(n_global / el, 3), "unit": "frames/s", "steps": args.files_steps, "ms_per_step": round(el * 1e3, 3),
                "format": "uncompressed RGB TIFF, one strip" if args.files == "tiff" else "binary PPM (P6)",
                "file_MB": round(nbytes / len(paths) / 1e6, 2), "decode_MBps": round(nbytes / el / 1e6, 1),
                "decode_threads": workers, "host_cores": os.cpu_count(), "dir": args.files_dir, "written_in_s": round(write_s, 1),
                "note": "stk_%s_match_files: decode (host threads, into one page-locked block) + H2D + the whole path per step, "
                        "files in tmpfs; never `value`" % ("ecc" if api == "ecc" else "keypoint")}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cpu_baseline(args, api, frames, W, H):
    import oracle
    cores = os.cpu_count() or 1
    use = min(cores, 32)
    n_s = args.cpu_sample_frames or (use + 1)
    n_s = max(2, min(n_s, frames.shape[0]))
    sample = [f.cpu().numpy() for f in frames[:n_s]]
    tc = time.perf_counter()
    if api == "ecc":
        oracle.ecc_match(sample, max_count=5000, epsilon=1e-5, gauss_filt_size=5, n_threads=use)
        how = f"frame-parallel OpenMP on {use} threads like the reference's Rayon fold"
    elif api == "hybrid":
        oracle.hybrid_match(sample, n_threads=use)      # the oracle's stages composed per frame, frames over a thread pool
        how = f"frame-parallel on {use} threads (one frame per thread: ORB + match + RANSAC + ECC + fold, partial sums combined)"
    else:
        oracle.keypoint_match(sample, n_threads=use)
        how = f"frame-parallel OpenMP on {use} threads like the reference's Rayon fold"
    cpu_s = time.perf_counter() - tc
    return {"value": round(n_s / cpu_s, 4), "unit": "frames/s", "cores": use, "kind": "port",
            "sample": f"first {n_s} frames of the same {W}x{H} stack, one oracle pass ({cpu_s:.1f} s), {how}"}


def single_process(args, torch, np, synth, Stacker, shard_moving_frames, W, H, n_global, api, depth, ecc_params, kp_params, scaling):
    """N GPUs from ONE process through a multi-device context: the library shards, threads and reduces (RCCL) itself."""
    n_dev = args.gpus
    ids = [0] * n_dev if args.rehearse_on_one_gpu else list(range(n_dev))
    scene = synth.render_scene(W, H)
    parts = [None] * n_global
    for r in range(n_dev):                                   # every device's range resident on that device
        mine = shard_moving_frames(n_global, n_dev, r)
        idx = ([0] if r == 0 else []) + mine
        if not idx:
            continue
        fr, _ = synth.make_stack(0, W, H, scene=scene, device=torch.device("cuda", ids[r]), indices=idx, depth=depth)
        for k, g in enumerate(idx):
            parts[g] = fr[k]
    st = Stacker(devices=ids)

    def step():
        if api == "ecc":
            return st.ecc_match(parts, ecc_params)
        if api == "hybrid":
            return st.hybrid_match(parts, kp_params, ecc_params)
        return st.keypoint_match(parts, kp_params)[1]

    def sync():
        for d in set(ids):
            torch.cuda.synchronize(d)
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    res = result_header(args, n_global * args.steps / elapsed, elapsed, n_dev, scaling, api, W, H, n_global, n_global / n_dev,
                        "RCCL ncclReduce inside the library (stk_create_multi)" if not args.rehearse_on_one_gpu
                        else "members share one GPU: local adds (rehearsal)", f"frame-shard x{n_dev}, ONE process, one host thread per GPU")
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
