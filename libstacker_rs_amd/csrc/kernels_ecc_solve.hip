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

constexpr int SOLVE_WAVES_STANDALONE = 16;

__global__ __launch_bounds__(1024) void ecc_solve_kernel(EccIterArgs a, int motion, EccCriteria crit, EccQueue* queue,
                                                        EccFrameResult* results, const float* init_warps) {
    ecc_solve_body<SOLVE_WAVES_STANDALONE>(a, a.slot0 + (int)blockIdx.x, motion, crit, queue, results, init_warps);
}

hipError_t launch_ecc_solve(const EccIterArgs& a, int motion, EccCriteria crit, EccQueue* queue,
                            EccFrameResult* results, hipStream_t s) {
    ecc_solve_kernel<<<a.n_slots, 64 * SOLVE_WAVES_STANDALONE, 0, s>>>(a, motion, crit, queue, results, nullptr);
    return hipGetLastError();
}

__global__ void ecc_init_kernel(EccSlot* slots, int n_slots, EccQueue* queue, int n_frames, EccFrameResult* results,
                                const float* init_warps) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    queue->next_frame = 0; queue->n_frames = n_frames; queue->frames_done = 0; queue->pad = 0;
    for (int f = 0; f < n_frames; f++) { results[f].status = 3; results[f].iters = 0; results[f].rho = -1; }
    for (int s = 0; s < n_slots; s++) {
        slots[s].last_rho = 0;
        slot_take_next(slots + s, queue, init_warps);
    }
}

hipError_t launch_ecc_init(EccSlot* slots, int n_slots, EccQueue* queue, int n_frames, EccFrameResult* results,
                           const float* init_warps, hipStream_t s) {
    ecc_init_kernel<<<1, 64, 0, s>>>(slots, n_slots, queue, n_frames, results, init_warps);
    return hipGetLastError();
}

}  // namespace stk
