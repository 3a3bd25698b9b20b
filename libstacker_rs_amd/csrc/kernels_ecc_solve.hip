// kernels_ecc_solve.hip — the per-iteration "solve" step of findTransformECC (SURVEY.md §8a-E*
// steps b..k; reference call site lib.rs:769-777) as its own launch, one workgroup per slot, for the
// launch-per-iteration form of the alignment (host-fed stacks); the routine itself is ecc_solve_body.h:
//   1. fixed-order f64 reduction of the pixel pass's unit partials (layout [slot][sum][unit]);
//   2. the normal equations exactly as ecc.cpp forms them: Hessian cast to f32, inverse by
//      hal::LU32f's elimination order (run element-parallel in LDS: every element sees the same
//      operation sequence as the serial loop), projections in f32, lambda in f64;
//   3. warp update, convergence test of the reference's for-loop, and the device-side frame queue.
// Also here: the queue / slot initialisation for both forms of the alignment.
#include "ecc_solve_body.h"

namespace stk {

// (Round 1 spread the reduction over 8 workgroups per slot with a ticket hand-off; with ~43 slots in flight the two
// agent-scope fences of that hand-off cost more than the spread saved. One workgroup per slot needs no hand-off.)
__global__ __launch_bounds__(256) void ecc_solve_kernel(EccIterArgs a, int motion, EccCriteria crit, EccQueue* queue,
                                                       EccFrameResult* results, const float* init_warps) {
    const int slot = a.slot0 + (int)blockIdx.x;
    EccSlot* sl = a.slots + slot;
    if (sl->frame < 0) {
        // idle slot: look for a newly prepared frame
        if (threadIdx.x == 0 && slot_take_next(sl, queue, init_warps) >= 0) sl->last_rho = 0;
        return;
    }
    __shared__ EccSolveLds L;
    ecc_solve_body<4>(ecc_solve_args(a), slot, motion, crit, queue, results, init_warps, L);
}

hipError_t launch_ecc_solve(const EccIterArgs& a, int motion, EccCriteria crit, EccQueue* queue,
                            EccFrameResult* results, hipStream_t s, const float* init_warps) {
    ecc_solve_kernel<<<a.n_slots, 256, 0, s>>>(a, motion, crit, queue, results, init_warps);
    return hipGetLastError();
}

__global__ void ecc_init_kernel(EccSlot* slots, int n_slots, EccQueue* queue, int n_frames, EccFrameResult* results,
                                const float* init_warps, int ready0, EccSched* sched, int nb) {
    // one workgroup, everything in parallel: slot s starts with frame s (what handing the frames out one by one from an
    // empty queue gives), as far as frames are ready; the queue continues behind them
    for (int f = threadIdx.x; f < n_frames; f += blockDim.x) { results[f].status = 3; results[f].iters = 0; results[f].rho = -1; }
    const int avail = min(ready0 < 0 ? n_frames : ready0, n_frames);
    for (int s = threadIdx.x; s < 64; s += blockDim.x) {
        const bool active = s < n_slots && s < avail;
        if (s < n_slots) {
            EccSlot* sl = slots + s;
            sl->last_rho = 0;
            if (active) {
                sl->frame = s;
                sl->iter = 0;
                for (int k = 0; k < 9; k++) sl->warp[k] = init_warps ? init_warps[(size_t)s * 9 + k] : ((k % 4 == 0) ? 1.f : 0.f);
                sl->cI = 0; sl->cT = 0;
                sl->rho = -1;
            } else {
                sl->frame = -1;
            }
        }
        if (sched) {                                        // persistent scheduler: the first iteration's units
            sched->done[s] = 0;
            sched->frame_of[s] = active ? s : 0x7fffffff;
            for (int c = 0; c < 8; c++) sched->W[c][s] = active ? ecc_ticket_word(1, nb / 8) : 0;
        }
    }
    if (threadIdx.x == 0) {
        queue->next_frame = min(avail, n_slots); queue->n_frames = n_frames; queue->frames_done = 0; queue->ring_fallbacks = 0;
        queue->ready = ready0 < 0 ? n_frames : ready0;
        if (sched) { sched->live = min(avail, n_slots); for (int k = 0; k < 63 + 128; k++) sched->pad[k] = 0; }
    }
}

hipError_t launch_ecc_init(EccSlot* slots, int n_slots, EccQueue* queue, int n_frames, EccFrameResult* results,
                           const float* init_warps, hipStream_t s, int ready0, EccSched* sched, int nb) {
    ecc_init_kernel<<<1, 256, 0, s>>>(slots, n_slots, queue, n_frames, results, init_warps, ready0, sched, nb);
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
