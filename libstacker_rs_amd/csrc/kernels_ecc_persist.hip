// kernels_ecc_persist.hip — findTransformECC for every frame of a shard in ONE launch (reference call site lib.rs:769-777
// inside the Rayon fold lib.rs:746-833; algorithm SURVEY.md 8a-E*).
//
// The launch-per-iteration form (kernels_ecc_col.hip + kernels_ecc_solve.hip) pays, per iteration of the frames in
// flight: two kernel boundaries, the fill and drain of a 12 000-workgroup grid, a solve launch that holds the whole GPU
// for the latency of one 8 x 8 elimination, and — at the end of a shard — launches that carry one or two frames on a
// quarter of the machine, plus a host round trip to learn that the queue has drained. None of that is work.
//
// Here 4 workgroups per CU stay resident and schedule themselves (EccSched, common.h):
//   * the work of a slot's current iteration is a set of UNITS (ecc_col_unit.h: a fixed image region reduced to one set
//     of partial sums — the same units, the same summation partition and therefore the same bits as the launch-per-
//     iteration form); a workgroup draws a ticket for a unit with one atomic add on the slot's ticket word, preferring
//     the region class of its XCD and the slot that is furthest behind (ecc_acquire);
//   * the workgroup whose unit completes an iteration (an arrival counter) reduces the partials, solves the normal
//     equations, applies the reference's loop test, hands the slot its next iteration — or the next frame of the queue —
//     and opens the new units; everybody else keeps working on other slots meanwhile: no barrier between iterations or
//     frames, the 20 us of a solve cost one workgroup instead of the machine;
//   * when no slot holds a frame any more every workgroup leaves, and stream order lets the fold follow without a host
//     round trip.
// Cross-workgroup data (slot state, unit partials, scheduler words) moves by agent-scope write-through stores and
// cache-bypassing loads, ordered by the ticket / arrival atomics: stores -> s_waitcnt vmcnt(0) in every storing wave ->
// workgroup barrier -> ONE lane's atomic; the consumer loads after its own atomic has returned (MI355X_MICROARCH.md,
// inter-workgroup visibility, sc1 hand-off). A workgroup never waits for a workgroup that is not running: the only
// waits are polls for new tickets while other RUNNING workgroups hold the units that will produce them.
// Only frames that were ready when the kernel was launched are taken (queue->ready is fixed for its lifetime): what a
// concurrent preparation kernel writes is never read here.
#include "ecc_col_unit.h"
#include "ecc_solve_body.h"

namespace stk {

template <int MOTION>
union EccPersistLds {                       // a workgroup either runs a unit or solves: the solve borrows the rings' memory
    EccUnitLds<MOTION> unit;
    EccSolveLds solve;
};

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

struct EccTicket { int slot, unit; };       // slot < 0: nothing left anywhere, leave

// The solve, OUT OF LINE: inlined into the persistent kernel it shares the unit's register allocation (128 VGPRs, all
// taken by the pixel loops) and its serial tail runs on spilled values — 90 us per solve instead of 20, which a lone
// frame's iteration pays in full (152 us against 63 + 20 for a launch per iteration). As a call it gets registers of its own.
template <int MOTION>
__device__ __attribute__((noinline)) void ecc_solve_call(EccSolveArgs a, int slot, EccCriteria crit, EccQueue* queue,
                                                         EccFrameResult* results, const float* init_warps, EccSolveLds* L) {
    ecc_solve_body<4>(a, slot, MOTION, crit, queue, results, init_warps, *L);
}

// Wave 0 draws the next ticket. `look`: the ticket words of the workgroup's own class, lane = slot, already loaded by the
// caller (so that the load shares a round trip with the arrival of the previous ticket), or -1 to load them here.
// Policy (variants measured on 32- and 128-frame 4K stacks, round 3: all within 1 % of each other except the phase, which
// is worth 3 % on a 32-frame shard): take the slot that is furthest BEHIND in virtual time = iterations done + sixteenths
// of the current one, odd slots counted half an iteration late. All slots then sweep the image regions together — the
// frame-0 rows of a region are fetched into the XCD's L2 once for all of them, as the launch-per-iteration grid order
// arranges — but in two groups half an iteration apart, so the solves of one group run under the units of the other.
// Among equals every workgroup starts at a different slot (rot): a class's 128 workgroups do not all draw on one word.
__device__ __forceinline__ EccTicket ecc_acquire(const EccIterArgs& a, int cls, int rot, int look) {
    EccSched* sc = a.sched;
    const int lane = threadIdx.x & 63;
    for (bool first = true;; first = false) {
        for (int dc = 0; dc < 8; dc++) {                     // own class first, then the neighbours' (stealing keeps the tail busy)
            const int c = (cls + dc) & 7;
            const int w = (first && dc == 0 && look != -1) ? look : (lane < a.n_slots ? ld_agent(&sc->W[c][lane]) : 0);
            const int nxt = ecc_ticket_next(w), upc = ecc_ticket_units(w);
            bool avail = nxt < upc;
            const int prog = (ecc_ticket_gen(w) * 16 + (upc > 0 ? (nxt * 16) / upc : 0)) * 2 + (lane & 1) * 16;
            const int key = prog * 64 + ((lane - rot) & 63);
            for (;;) {
                const int best = wave_min_i32(avail ? key : 0x7fffffff);
                if (best == 0x7fffffff) break;
                const int slot = (best + rot) & 63;
                int t = 0;                                   // the draw: one atomic add by lane 0; valid iff below the word's own bound
                if (lane == 0) t = atomicAdd(&sc->W[c][slot], 1);
                t = __builtin_amdgcn_readfirstlane(t);
                if (ecc_ticket_next(t) < ecc_ticket_units(t)) return EccTicket{slot, ecc_ticket_next(t) * 8 + c};   // class c: regions c, c + 8, ...
                if (lane == slot) avail = false;             // raced: somebody took the last ticket between the look and the draw
            }
        }
        if (ld_agent(&sc->live) <= 0) return EccTicket{-1, 0};
        __builtin_amdgcn_s_sleep(32);                        // ~1 us: units are tens of us long
    }
}

template <int MOTION>
__global__ __launch_bounds__(256, STK_COL_WG) void ecc_persist_kernel(EccIterArgs a, EccCriteria crit, EccQueue* queue,
                                                                      EccFrameResult* results, const float* init_warps) {
    __shared__ EccPersistLds<MOTION> lds;
    __shared__ EccTicket sh_tk;
    __shared__ int sh_last;
#ifdef STK_UNIT_CUT
    __shared__ int sh_cut;
    if (threadIdx.x == 0) sh_cut = a.n_slots > 100 ? 4 : 1;
#endif
    const int cls = (int)blockIdx.x & 7, rot = ((int)blockIdx.x >> 3) % a.n_slots;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 0) {
        const EccTicket t = ecc_acquire(a, cls, rot, -1);
        if (threadIdx.x == 0) sh_tk = t;
    }
    __syncthreads();
    for (;;) {
        const EccTicket tk = sh_tk;
        if (tk.slot < 0) return;                             // uniform: every wave of the workgroup leaves here
#ifdef STK_UNIT_CUT
        ecc_col_unit<MOTION>(a, tk.slot, tk.unit, lds.unit, max(1, min(4, sh_cut)));
#else
#ifdef STK_PERSIST_TIMING
        const long long t_unit0 = wall_clock64();
#endif
        ecc_col_unit<MOTION>(a, tk.slot, tk.unit, lds.unit); // ends with every storing wave's s_waitcnt vmcnt(0)
#ifdef STK_PERSIST_TIMING
        const long long t_unit1 = wall_clock64();
        if (blockIdx.x == 264 && threadIdx.x == 0) {        // one ordinary workgroup's first units: start and end
            long long* dbg2 = reinterpret_cast<long long*>(a.sched->pad) + 32;
            const int k = a.sched->pad[61];
            if (k < 24) { dbg2[2 * k] = t_unit0; dbg2[2 * k + 1] = t_unit1; a.sched->pad[190 - k] = tk.unit; a.sched->pad[61] = k + 1; }
        }
#endif
#endif
        __syncthreads();                                     // all partials of this unit have left the workgroup
        if (wave == 0) {
            // arrive for this unit and look for the next one at the same time: the two round trips overlap
            int t = 0;
            if (lane == 0) t = atomicAdd(&a.sched->done[tk.slot], 1);
            const int look = lane < a.n_slots ? ld_agent(&a.sched->W[cls][lane]) : 0;
            const bool last = __builtin_amdgcn_readfirstlane(t) == a.nb - 1;
            if (threadIdx.x == 0) sh_last = last;
            if (!last) {
                const EccTicket nt = ecc_acquire(a, cls, rot, look);
                if (threadIdx.x == 0) sh_tk = nt;
            }
        }
        __syncthreads();
        if (sh_last) {
            // this unit completed the slot's iteration: every unit's partials were stored and signalled before our
            // arrival returned; they are read with agent-scope loads only
#ifdef STK_PERSIST_TIMING
            long long* dbg = reinterpret_cast<long long*>(a.sched->pad);
            const int it = a.sched->pad[62];
            if (threadIdx.x == 0 && it < 6) { dbg[it * 5 + 0] = t_unit0; dbg[it * 5 + 1] = t_unit1; dbg[it * 5 + 2] = wall_clock64(); }
#endif
            ecc_solve_call<MOTION>(ecc_solve_args(a), tk.slot, crit, queue, results, init_warps, &lds.solve);
#ifdef STK_PERSIST_TIMING
            if (threadIdx.x == 0 && it < 6) dbg[it * 5 + 3] = wall_clock64();
#endif
            if (wave == 0) {                                 // (the lane that armed the next iteration is in this wave)
                const EccTicket nt = ecc_acquire(a, cls, rot, -1);
                if (threadIdx.x == 0) sh_tk = nt;
            }
#ifdef STK_PERSIST_TIMING
            if (threadIdx.x == 0 && it < 6) { dbg[it * 5 + 4] = wall_clock64(); a.sched->pad[62] = it + 1; }
#endif
            __syncthreads();
        }
    }
}

hipError_t launch_ecc_persist(const EccIterArgs& a, int motion, EccCriteria crit, EccQueue* queue, EccFrameResult* results,
                              const float* init_warps, int n_workgroups, hipStream_t s) {
    // the ticket word's fields hold these by construction (common.h): units per class, and every workgroup's one failed draw
    if (!a.sched || a.n_slots < 1 || a.n_slots > 64 || a.nb % 8 || a.nb / 8 >= (1 << ECC_TICKET_UNITS_BITS) ||
        n_workgroups < 8 || n_workgroups + a.nb / 8 >= (1 << ECC_TICKET_NEXT_BITS))
        return hipErrorInvalidValue;
    const int grid = n_workgroups;
    switch (motion) {
        case STK_MOTION_HOMOGRAPHY: ecc_persist_kernel<STK_MOTION_HOMOGRAPHY><<<grid, 256, 0, s>>>(a, crit, queue, results, init_warps); break;
        case STK_MOTION_AFFINE: ecc_persist_kernel<STK_MOTION_AFFINE><<<grid, 256, 0, s>>>(a, crit, queue, results, init_warps); break;
        case STK_MOTION_EUCLIDEAN: ecc_persist_kernel<STK_MOTION_EUCLIDEAN><<<grid, 256, 0, s>>>(a, crit, queue, results, init_warps); break;
        case STK_MOTION_TRANSLATION: ecc_persist_kernel<STK_MOTION_TRANSLATION><<<grid, 256, 0, s>>>(a, crit, queue, results, init_warps); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace stk
