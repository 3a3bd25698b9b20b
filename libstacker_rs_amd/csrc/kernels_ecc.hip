// kernels_ecc.hip — the fused iteration pass of findTransformECC (lib.rs:769-777; algorithm
// SURVEY.md §8a-E*). The solve step and the device-side frame queue are in kernels_ecc_solve.hip.
//
//  ONE pass per iteration and slot: per template pixel, projective coordinates -> bilinear taps of
//  (I, gx, gy) from the zero-padded frame-0 planes -> nearest mask -> Jacobian in registers -> 66
//  moment sums (36 Hessian, 8 J.Iw, 8 J.T.m, 8 J.m, 6 scalars) accumulated in f32 per lane, reduced
//  with wavefront shuffles, then across the four waves through LDS in f64, written as f64 block
//  partials. OpenCV materialises ~150 f32 planes per iteration for the same result; here the
//  algorithmic traffic is 16 B/px.
//
//  Variants of the pass (option `ecc_variant`; same sums, they differ in the f32 summation order only):
//   * 3 (default): ecc_iter_col_kernel<MOTION> in kernels_ecc_col.hip — a wave walks DOWN a 64-pixel column strip, so X
//     is a lane constant and Y a scalar; frame-0 rows through a per-wave LDS ring; one cross-lane fold per strip.
//   * 0: direct (this file) — one wave per template row, lanes stream aligned 16-byte template quads, one accumulator per
//     sum and lane, every tap a global gather (the first version; kept as a cross-check).
//  History of the homography pass (4-slot 4K launch, then 32-slot): direct 182 us -> row-walking factorised pass 98 us
//  (round 1; VALU-issue-bound at ~97 instructions per pixel) -> 0.70 ms per 32-slot launch (round 2, fixed 288 blocks) ->
//  column-walking pass 0.61 ms -> with the LDS ring 0.58 ms.
//
//  Several frames ("slots") iterate concurrently in one launch. blockIdx is decoded so that the
//  blocks working on the SAME image region for different slots share blockIdx % 8, i.e. one XCD and
//  its L2: the frame-0 planes they all read are fetched from HBM/MALL once per XCD.
#include "ecc_pixel.h"

namespace stk {

// warped source coordinate of template pixel (x, y): sx, sy and the Jacobian helpers
template <int MOTION>
__device__ __forceinline__ void warp_coord(const SlotConst& c, float fx, float fy, float& sx, float& sy, float& rden,
                                           float& hx, float& hy) {
    sx = __builtin_fmaf(c.m0, fx, __builtin_fmaf(c.m1, fy, c.m2));
    sy = __builtin_fmaf(c.m3, fx, __builtin_fmaf(c.m4, fy, c.m5));
    rden = 1.0f; hx = 0.0f; hy = 0.0f;
    if constexpr (MOTION == STK_MOTION_HOMOGRAPHY) {
        // v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division: the result is still deterministic,
        // and a last-bit difference in 1/w moves the sample point by < 3e-4 px at 4K
        const float rw = __builtin_amdgcn_rcpf(__builtin_fmaf(c.m6, fx, __builtin_fmaf(c.m7, fy, c.m8)));
        rden = rw;
        if (!c.den_is_w) rden = __builtin_amdgcn_rcpf(__builtin_fmaf(c.m6, fx, __builtin_fmaf(c.m7, fy, 1.0f)));   // den = X*h2 + Y*h5 + 1
        hx = -sx * rden; hy = -sy * rden;                                                          // hatX, hatY
        sx *= rw; sy *= rw;
    }
}

// Everything after the taps: mask, Jacobian, the 66 fused multiply-adds.
template <int MOTION, int NS>
__device__ __forceinline__ void accumulate_pixel(const SlotConst& c, const float* warp, int x, int y, float fx, float fy,
                                                 float sx, float sy, float rden, float hx, float hy, float Iw, float gxw,
                                                 float gyw, float tval, float (&acc)[NS]) {
    constexpr int P = MotionTraits<MOTION>::P;
    constexpr int NH = P * (P + 1) / 2;
    // INTER_NEAREST mask: rounded source coordinate inside the input image. OpenCV rounds a double
    // (homography) or 10-bit fixed-point (affine) coordinate; the f32 value decides except within
    // 0.01 px of a boundary of the valid range, where the exact form is redone.
    // Pixels whose source coordinate is strictly inside [0, W-1] x [0, H-1] (all but a thin rim) skip all of that.
    bool inside = (sx > 0.0f) & (sx < c.mxw) & (sy > 0.0f) & (sy < c.mxh);
    if (!inside) {
        const float rx = __builtin_rintf(sx), ry = __builtin_rintf(sy);
        inside = (rx >= 0.0f) & (rx <= c.mxw) & (ry >= 0.0f) & (ry <= c.mxh);
        const bool edge = (__builtin_fabsf(sx + 0.5f) < 0.01f) | (__builtin_fabsf(sx - (c.mxw + 0.5f)) < 0.01f) |
                          (__builtin_fabsf(sy + 0.5f) < 0.01f) | (__builtin_fabsf(sy - (c.mxh + 0.5f)) < 0.01f);
        if (edge) inside = nearest_inside_exact<MOTION>(x, y, warp, c.iw, c.ih);
    }
    const float mf = inside ? 1.0f : 0.0f;

    float J[P];
    if constexpr (MOTION == STK_MOTION_HOMOGRAPHY) {
        const float ja = gxw * rden, jb = gyw * rden;
        const float jt = hx * ja + hy * jb;
        J[0] = ja * fx; J[1] = jb * fx; J[2] = jt * fx;
        J[3] = ja * fy; J[4] = jb * fy; J[5] = jt * fy;
        J[6] = ja; J[7] = jb;
    } else if constexpr (MOTION == STK_MOTION_AFFINE) {
        J[0] = gxw * fx; J[1] = gyw * fx; J[2] = gxw * fy; J[3] = gyw * fy; J[4] = gxw; J[5] = gyw;
    } else if constexpr (MOTION == STK_MOTION_EUCLIDEAN) {
        const float ex = -(fx * c.m3) - (fy * c.m0);     // h0 = m00 (cos), h1 = m10 (sin)
        const float ey = (fx * c.m0) - (fy * c.m3);
        J[0] = gxw * ex + gyw * ey; J[1] = gxw; J[2] = gyw;
    } else {
        J[0] = gxw; J[1] = gyw;
    }
    // centred samples: u = Iw - cI inside the mask (Iw outside), v = (T - cT) inside, 0 outside
    const float u = inside ? Iw - c.cI : Iw;
    const float v = inside ? tval - c.cT : 0.0f;
    int idx = 0;
#pragma unroll
    for (int k = 0; k < P; k++)
#pragma unroll
        for (int l = k; l < P; l++) { acc[idx] = __builtin_fmaf(J[k], J[l], acc[idx]); idx++; }
#pragma unroll
    for (int k = 0; k < P; k++) {
        acc[NH + k] = __builtin_fmaf(J[k], u, acc[NH + k]);
        acc[NH + P + k] = __builtin_fmaf(J[k], v, acc[NH + P + k]);
        acc[NH + 2 * P + k] = __builtin_fmaf(J[k], mf, acc[NH + 2 * P + k]);
    }
    const float um = u * mf;
    acc[NH + 3 * P + 0] += mf;
    acc[NH + 3 * P + 1] += um;
    acc[NH + 3 * P + 2] = __builtin_fmaf(um, u, acc[NH + 3 * P + 2]);
    acc[NH + 3 * P + 3] += v;
    acc[NH + 3 * P + 4] = __builtin_fmaf(v, v, acc[NH + 3 * P + 4]);
    acc[NH + 3 * P + 5] = __builtin_fmaf(um, v, acc[NH + 3 * P + 5]);
}

// direct path: taps gathered from global memory (zero border makes them unconditional)
template <int MOTION, int NS>
__device__ __forceinline__ void pixel_direct(const SlotConst& c, const EccIterArgs& a, const float* warp, int x, int y,
                                             float tval, float (&acc)[NS]) {
    const float fx = (float)x, fy = (float)y;
    float sx, sy, rden, hx, hy;
    warp_coord<MOTION>(c, fx, fy, sx, sy, rden, hx, hy);
    const float flx = __builtin_floorf(sx), fly = __builtin_floorf(sy);
    const float ax = sx - flx, ay = sy - fly;
    // clamp the integer coordinate into the zero border; NaN -> -2 (all taps zero)
    const int ix = (int)__builtin_fminf(__builtin_fmaxf(flx, -2.0f), c.fiw);
    const int iy = (int)__builtin_fminf(__builtin_fmaxf(fly, -2.0f), c.fih);
    const int rs = a.ref.stride;
    const int off = iy * rs + ix;
    const f32x2_a4 i0 = *(const f32x2_a4*)(a.ref.I + off), i1 = *(const f32x2_a4*)(a.ref.I + off + rs);
    const f32x2_a4 x0 = *(const f32x2_a4*)(a.ref.gx + off), x1 = *(const f32x2_a4*)(a.ref.gx + off + rs);
    const f32x2_a4 y0 = *(const f32x2_a4*)(a.ref.gy + off), y1 = *(const f32x2_a4*)(a.ref.gy + off + rs);
    const float Iw = bilerp4(i0.x, i0.y, i1.x, i1.y, ax, ay);
    const float gxw = bilerp4(x0.x, x0.y, x1.x, x1.y, ax, ay);
    const float gyw = bilerp4(y0.x, y0.y, y1.x, y1.y, ax, ay);
    accumulate_pixel<MOTION, NS>(c, warp, x, y, fx, fy, sx, sy, rden, hx, hy, Iw, gxw, gyw, tval, acc);
}

template <int NS>
__device__ __forceinline__ void block_reduce_store(float (&acc)[NS], const EccIterArgs& a, int slot, int region) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NS; k++) {
        float v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        acc[k] = v;
    }
    __shared__ double red[4][NS];
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NS; k++) red[wave][k] = (double)acc[k];
    }
    __syncthreads();
    if (threadIdx.x < NS) {
        const int k = threadIdx.x;
        const double s = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
        a.partials[((size_t)slot * NS + k) * a.nb + region] = s;   // [slot][sum][block]
    }
}

// ---------------------------------------------------------------------------------------------------
// direct variant: one wave per template row, lanes stream aligned 16-byte template quads
// ---------------------------------------------------------------------------------------------------
template <int MOTION>
__global__ __launch_bounds__(256) void ecc_iter_kernel(EccIterArgs a) {
    constexpr int P = MotionTraits<MOTION>::P;
    constexpr int NS = P * (P + 1) / 2 + 3 * P + 6;
    // XCD-aware decode: bid = xcd + 8 * (slot + n_slots * (region / 8))
    const int bid = (int)blockIdx.x;
    const int xcd = bid & 7, q = bid >> 3;
    const int slot = a.slot0 + q % a.n_slots;
    const int region = (q / a.n_slots) * 8 + xcd;
    const EccSlot* sl = a.slots + slot;
    const int frame = sl->frame;
    if (frame < 0) return;                                    // idle slot: whole block leaves
    SlotConst c;
    load_slot_const(sl, a, c);
    const float* __restrict__ T = a.templates + (size_t)frame * a.templ_plane_stride;

    float acc[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) acc[k] = 0.f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qw = (a.tw + 3) >> 2;                           // quads per row
    for (int y = region * 4 + wave; y < a.th; y += a.nb * 4) {
        const float* trow = T + (size_t)y * a.templ_row_stride;
        // (prefetching the template quads two iterations ahead in registers was measured: no gain, r01)
        for (int qx = lane; qx < qw; qx += 64) {
            const float4 t4 = *reinterpret_cast<const float4*>(trow + qx * 4);
#pragma unroll 2
            for (int j = 0; j < 4; j++) {
                const int x = qx * 4 + j;
                const float tvj = j == 0 ? t4.x : j == 1 ? t4.y : j == 2 ? t4.z : t4.w;
                if (x < a.tw) pixel_direct<MOTION, NS>(c, a, sl->warp, x, y, tvj, acc);
            }
        }
    }
    block_reduce_store<NS>(acc, a, slot, region);
}

// variant 3: the production kernel (column-walking pass, kernels_ecc_col.hip, every motion model); variant 0: the first,
// direct version (one accumulator per sum and lane, every tap a global gather) — kept as an independent cross-check in the
// tests and for a caller-supplied initial homography whose m22 is not 1. (Measured and deleted: LDS-tiled with LDS-DMA and
// row-sharing slots in round 1; the row-walking factorised homography pass and the row-walking affine-family pass in round
// 2, when the column-walking one overtook them.)
hipError_t launch_ecc_iter(const EccIterArgs& a, int motion, int variant, hipStream_t s) {
    const int grid = a.nb * a.n_slots;
    if (grid <= 0) return hipSuccess;
    if (variant == 3) {
        EccIterArgs b = a;
        const long long units = (long long)((a.tw + 63) >> 6) * a.th;
        if (units > 0x3fffffffLL) return hipErrorInvalidValue;
        b.units_q = (int)(units / (a.nb * 4)); b.units_r = (int)(units % (a.nb * 4));
        return launch_ecc_iter_col(b, motion, s);
    }
    switch (motion) {
        case STK_MOTION_HOMOGRAPHY: ecc_iter_kernel<STK_MOTION_HOMOGRAPHY><<<grid, 256, 0, s>>>(a); break;
        case STK_MOTION_AFFINE: ecc_iter_kernel<STK_MOTION_AFFINE><<<grid, 256, 0, s>>>(a); break;
        case STK_MOTION_EUCLIDEAN: ecc_iter_kernel<STK_MOTION_EUCLIDEAN><<<grid, 256, 0, s>>>(a); break;
        case STK_MOTION_TRANSLATION: ecc_iter_kernel<STK_MOTION_TRANSLATION><<<grid, 256, 0, s>>>(a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace stk
