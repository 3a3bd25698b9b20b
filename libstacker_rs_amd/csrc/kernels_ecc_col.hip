// kernels_ecc_col.hip — the column-walking iteration pass of findTransformECC as ONE LAUNCH PER ITERATION (lib.rs:769-777;
// SURVEY.md 8a-E*): every workgroup runs the unit(s) of one (slot, region) pair (ecc_col_unit.h) and the solve kernel
// follows. Used for host-fed stacks, whose frames arrive while the queue already runs; device-resident stacks go through
// the persistent scheduler (kernels_ecc_persist.hip), which runs the very same units.
#include "ecc_col_unit.h"

namespace stk {

template <int MOTION>
__global__ __launch_bounds__(256, STK_COL_WG) void ecc_iter_col_kernel(EccIterArgs a) {
    const int bid = (int)blockIdx.x;
    const int xcd = bid & 7, q = bid >> 3;
    const int slot = a.slot0 + q % a.n_slots;
    const int region = (q / a.n_slots) * 8 + xcd;
    const EccSlot* sl = a.slots + slot;
    if (sl->frame < 0) return;
    __shared__ EccUnitLds<MOTION> lds;
#ifdef STK_UNIT_CUT
    ecc_col_unit<MOTION>(a, slot, region, lds, 1);
#else
    ecc_col_unit<MOTION>(a, slot, region, lds);
#endif
}

void ecc_set_col_units(EccIterArgs& a) {
    const long long units = (long long)((a.tw + 63) >> 6) * a.th;       // (column strip, row) pairs of a frame
    a.units_q = (int)(units / (a.nb * 4)); a.units_r = (int)(units % (a.nb * 4));
}

hipError_t launch_ecc_iter_col(const EccIterArgs& a, int motion, hipStream_t s) {
    const int grid = a.nb * a.n_slots;
    switch (motion) {
        case STK_MOTION_HOMOGRAPHY: ecc_iter_col_kernel<STK_MOTION_HOMOGRAPHY><<<grid, 256, 0, s>>>(a); break;
        case STK_MOTION_AFFINE: ecc_iter_col_kernel<STK_MOTION_AFFINE><<<grid, 256, 0, s>>>(a); break;
        case STK_MOTION_EUCLIDEAN: ecc_iter_col_kernel<STK_MOTION_EUCLIDEAN><<<grid, 256, 0, s>>>(a); break;
        case STK_MOTION_TRANSLATION: ecc_iter_col_kernel<STK_MOTION_TRANSLATION><<<grid, 256, 0, s>>>(a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace stk
