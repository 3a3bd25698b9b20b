// kernels_ecc_solve.hip — the per-iteration "solve" step of findTransformECC (SURVEY.md §8a-E*
// steps b..k; reference call site lib.rs:769-777), one workgroup per slot:
//   1. fixed-order f64 reduction of the iteration kernel's block partials (layout [slot][sum][block],
//      one wavefront per sum, deterministic shuffle tree);
//   2. the normal equations exactly as ecc.cpp forms them: Hessian cast to f32, inverse by
//      hal::LU32f's elimination order (run element-parallel in LDS: every element sees the same
//      operation sequence as the serial loop), projections in f32, lambda in f64;
//   3. warp update, convergence test of the reference's for-loop, and the device-side frame queue.
#include "common.h"

namespace stk {

}  // namespace stk
#include "ecc_solve_body.h"
namespace stk {

// Stand-alone solve, one workgroup per slot: stage 1 reduces the slot's block partials (66 x nb doubles, 150 KB at 4K; each
// sum by one wavefront in a fixed order), stage 2 runs the normal equations and the loop control on the 66 sums.
// STK_SOLVE_G > 1 spreads stage 1 over G workgroups per slot that publish their sums and draw tickets, the last drawer
// running stage 2 (round 1's form, 19 -> 14 us at 4 slots). With ~43 slots in flight the two agent-scope fences of that
// hand-off cost more than the spread saves — 344 workgroups fencing, 15-23 us of a 32 us launch (-DSTK_SOLVE_TIMING) —
// and G = 1 needs no hand-off at all: 256-frame 4K step 58.2 ms at G = 8, 57.4 at G = 4, 57.3 at G = 1. Which workgroup
// reduces a sum does not affect any value.
#ifndef STK_SOLVE_G
#define STK_SOLVE_G 1
#endif
constexpr int SOLVE_G = STK_SOLVE_G;

__global__ __launch_bounds__(256) void ecc_solve_kernel(EccIterArgs a, int motion, EccCriteria crit, EccQueue* queue,
                                                       EccFrameResult* results, const float* init_warps) {
    const int slot = a.slot0 + (int)blockIdx.x / SOLVE_G, g = (int)blockIdx.x % SOLVE_G;
#ifdef STK_SOLVE_TIMING
    const long long t_start = wall_clock64();
#endif
    const int frame = a.slots[slot].frame;                  // tested only after the partial loads are in flight
    const int P = motion == STK_MOTION_HOMOGRAPHY ? 8 : motion == STK_MOTION_AFFINE ? 6 : motion == STK_MOTION_EUCLIDEAN ? 3 : 2;
    const int NS = P * (P + 1) / 2 + 3 * P + 6;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const double* base = a.partials + (size_t)slot * NS * a.nb;
    {
        constexpr int KB = (ECC_MAX_SUMS + SOLVE_G * 4 - 1) / (SOLVE_G * 4), JB = 5;   // sums per wave (66 / G / 4 rounded up), partials per lane in flight
        const int nbi = (a.nb + 63) >> 6;
        double acc[KB];
#pragma unroll
        for (int r = 0; r < KB; r++) acc[r] = 0;
        for (int j0 = 0; j0 < nbi; j0 += JB) {
            double v[KB][JB];
#pragma unroll
            for (int j = 0; j < JB; j++) {
                const int b = lane + 64 * (j0 + j);
#pragma unroll
                for (int r = 0; r < KB; r++) {
                    const int k = g + SOLVE_G * (wave + 4 * r);
                    v[r][j] = (b < a.nb && k < NS) ? base[(size_t)k * a.nb + b] : 0.0;
                }
            }
#pragma unroll
            for (int j = 0; j < JB; j++)
#pragma unroll
                for (int r = 0; r < KB; r++) acc[r] += v[r][j];
        }
#pragma unroll
        for (int r = 0; r < KB; r++) {
            double v = acc[r];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            const int k = g + SOLVE_G * (wave + 4 * r);
            if (lane == 0 && k < NS && frame >= 0) a.sums[(size_t)slot * ECC_MAX_SUMS + k] = v;
        }
    }
    if (frame < 0) {
        // Idle slot: look for a newly prepared frame — but only once all SOLVE_G workgroups of this launch have read
        // `frame`, i.e. from the one that draws the last ticket. (Taking it from workgroup 0 right away let later-starting
        // workgroups of the same launch see the slot as active: they published sums of stale partials and drew tickets,
        // the count carried into the next launch and stage 2 ran before all of that launch's sums were in — a frame
        // that entered an idle slot could come out one ulp or one iteration off, run to run.)
        __shared__ int last_idle;
        if (tid == 0) {
            if constexpr (SOLVE_G > 1) {
                const int t = atomicAdd(&a.tickets[slot], 1);
                last_idle = (t == SOLVE_G - 1);
                if (last_idle) a.tickets[slot] = 0;
            } else last_idle = 1;
        }
        __syncthreads();
        if (last_idle && tid == 0) {
            EccSlot* sl = a.slots + slot;
            slot_take_next(sl, queue, init_warps);
            if (sl->frame >= 0) sl->last_rho = 0;
        }
        return;
    }
#ifdef STK_SOLVE_TIMING
    const long long t_reduced = wall_clock64();
#endif
    if constexpr (SOLVE_G > 1) {
        __threadfence();                                     // publish this workgroup's sums before its ticket
        __shared__ int last;
        __syncthreads();
        if (tid == 0) {
            const int t = atomicAdd(&a.tickets[slot], 1);
            last = (t == SOLVE_G - 1);
            if (last) a.tickets[slot] = 0;                   // everyone has drawn: reset for the next launch
        }
        __syncthreads();
        if (!last) return;
        __threadfence();                                     // see the other workgroups' sums
    } else {
        __syncthreads();                                     // one workgroup per slot: its own sums, through its own L1 / L2
    }
#ifdef STK_SOLVE_TIMING
    if (tid == 0 && slot == a.slot0) { queue->dbg[0] = t_start; queue->dbg[1] = t_reduced; queue->dbg[2] = wall_clock64(); }
#endif
    ecc_solve_body<4, true>(a, slot, motion, crit, queue, results, init_warps);
}

hipError_t launch_ecc_solve(const EccIterArgs& a, int motion, EccCriteria crit, EccQueue* queue,
                            EccFrameResult* results, hipStream_t s, const float* init_warps) {
    ecc_solve_kernel<<<a.n_slots * SOLVE_G, 256, 0, s>>>(a, motion, crit, queue, results, init_warps);
    return hipGetLastError();
}

__global__ void ecc_init_kernel(EccSlot* slots, int n_slots, int* tickets, EccQueue* queue, int n_frames, EccFrameResult* results,
                                const float* init_warps, int ready0) {
    // one workgroup, everything in parallel: slot s starts with frame s (what handing the frames out one by one from an
    // empty queue gives), as far as frames are ready; the queue continues behind them
    for (int f = threadIdx.x; f < n_frames; f += blockDim.x) { results[f].status = 3; results[f].iters = 0; results[f].rho = -1; }
    const int avail = min(ready0 < 0 ? n_frames : ready0, n_frames);
    for (int s = threadIdx.x; s < n_slots; s += blockDim.x) {
        EccSlot* sl = slots + s;
        tickets[s] = 0;
        sl->last_rho = 0;
        if (s < avail) {
            sl->frame = s;
            sl->iter = 0;
            for (int k = 0; k < 9; k++) sl->warp[k] = init_warps ? init_warps[(size_t)s * 9 + k] : ((k % 4 == 0) ? 1.f : 0.f);
            sl->cI = 0; sl->cT = 0;
            sl->rho = -1;
        } else {
            sl->frame = -1;
        }
    }
    if (threadIdx.x == 0) {
        queue->next_frame = min(avail, n_slots); queue->n_frames = n_frames; queue->frames_done = 0; queue->ring_fallbacks = 0;
        queue->ready = ready0 < 0 ? n_frames : ready0;
    }
}

hipError_t launch_ecc_init(EccSlot* slots, int n_slots, int* tickets, EccQueue* queue, int n_frames, EccFrameResult* results,
                           const float* init_warps, hipStream_t s, int ready0) {
    ecc_init_kernel<<<1, 256, 0, s>>>(slots, n_slots, tickets, queue, n_frames, results, init_warps, ready0);
    return hipGetLastError();
}

__global__ void ecc_set_ready_kernel(EccQueue* queue, int ready) {
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(&queue->ready, ready, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
hipError_t launch_ecc_set_ready(EccQueue* queue, int ready, hipStream_t s) {
    ecc_set_ready_kernel<<<1, 64, 0, s>>>(queue, ready);
    return hipGetLastError();
}

}  // namespace stk
