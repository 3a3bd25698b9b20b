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
//   * 3 (default, homography): ecc_iter_h8_kernel — row-factorised Hessian, lane-adjacent pixels, taps through
//     one 32-bit offset on scalar bases, two-stage software pipeline (see the comment above that kernel).
//     Translation / euclidean / affine: ecc_iter_affine_kernel — the same data movement with plain accumulators.
//   * 0: direct — one wave per template row, lanes stream aligned 16-byte template quads, 66 per-lane
//     accumulators, every tap a global gather (the first version; kept as a cross-check).
//   * 1: tiled — a workgroup walks 64x16-pixel tiles; the source footprint of a tile is copied for all three
//     planes into LDS with LDS-DMA (global_load_lds_dwordx4), double-buffered, and every tap is an LDS read.
//   * 2: row-sharing — the waves of a workgroup are the slots, all on the same template row.
//  Measured: fabric traffic per launch equals the algorithmic bytes (no re-reads); ablations put the
//  arithmetic alone at 107 us and the coordinate + gather side alone at 86 us of a 116 us full 4-slot 4K launch.
//  Variants 1 and 2 and template prefetch left the time unchanged; removing VALU instructions, contiguous gathers and
//  pipelining the loads bought the rest. 182 -> 98 us per 4-slot 4K launch over round 1.
//
//  Several frames ("slots") iterate concurrently in one launch. blockIdx is decoded so that the
//  blocks working on the SAME image region for different slots share blockIdx % 8, i.e. one XCD and
//  its L2: the frame-0 planes they all read are fetched from HBM/MALL once per XCD.
#include "common.h"

namespace stk {

typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));

template <int MOTION> struct MotionTraits;
template <> struct MotionTraits<STK_MOTION_TRANSLATION> { static constexpr int P = 2; };
template <> struct MotionTraits<STK_MOTION_EUCLIDEAN> { static constexpr int P = 3; };
template <> struct MotionTraits<STK_MOTION_AFFINE> { static constexpr int P = 6; };
template <> struct MotionTraits<STK_MOTION_HOMOGRAPHY> { static constexpr int P = 8; };

__device__ __forceinline__ float bilerp4(float p00, float p01, float p10, float p11, float ax, float ay) {
    const float v0 = __builtin_fmaf(ax, p01 - p00, p00);
    const float v1 = __builtin_fmaf(ax, p11 - p10, p10);
    return __builtin_fmaf(ay, v1 - v0, v0);
}

__device__ __forceinline__ int sat_round_d(double v) {
    if (!(v > -2147483648.0)) return (int)0x80000000;
    if (!(v < 2147483647.0)) return 0x7fffffff;
    return (int)__builtin_rint(v);
}

// The mask pixel exactly as the classic INTER_NEAREST remap path computes it (imgwarp.cpp):
// homography: double coordinates, cvRound; affine family: AB_BITS = 10 fixed point.
template <int MOTION>
__device__ __forceinline__ bool nearest_inside_exact(int x, int y, const float* m, int iw, int ih) {
    int mx, my;
    if constexpr (MOTION == STK_MOTION_HOMOGRAPHY) {
        double W = (double)m[6] * x + (double)m[7] * y + (double)m[8];
        W = W != 0 ? 1.0 / W : 0;
        const double fX = fmax(-2147483648.0, fmin(2147483647.0, ((double)m[0] * x + (double)m[1] * y + (double)m[2]) * W));
        const double fY = fmax(-2147483648.0, fmin(2147483647.0, ((double)m[3] * x + (double)m[4] * y + (double)m[5]) * W));
        mx = sat_round_d(fX); my = sat_round_d(fY);
    } else {
        const int adx = sat_round_d((double)m[0] * x * 1024), bdx = sat_round_d((double)m[3] * x * 1024);
        const int X0 = sat_round_d(((double)m[1] * y + (double)m[2]) * 1024) + 512;
        const int Y0 = sat_round_d(((double)m[4] * y + (double)m[5]) * 1024) + 512;
        mx = (X0 + adx) >> 10; my = (Y0 + bdx) >> 10;
    }
    return ((unsigned)mx < (unsigned)iw) & ((unsigned)my < (unsigned)ih);
}

// Per-launch constants of one slot, kept in scalar registers.
struct SlotConst {
    float m0, m1, m2, m3, m4, m5, m6, m7, m8;
    float cI, cT;
    bool den_is_w;
    float fiw, fih, mxw, mxh;
    int iw, ih;
};

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

__device__ __forceinline__ void load_slot_const(const EccSlot* sl, const EccIterArgs& a, SlotConst& c) {
    c.m0 = sl->warp[0]; c.m1 = sl->warp[1]; c.m2 = sl->warp[2];
    c.m3 = sl->warp[3]; c.m4 = sl->warp[4]; c.m5 = sl->warp[5];
    c.m6 = sl->warp[6]; c.m7 = sl->warp[7]; c.m8 = sl->warp[8];
    c.cI = sl->cI; c.cT = sl->cT;
    c.den_is_w = (c.m8 == 1.0f);
    c.iw = a.ref.w; c.ih = a.ref.h;
    c.fiw = (float)a.ref.w; c.fih = (float)a.ref.h;
    c.mxw = (float)(a.ref.w - 1); c.mxh = (float)(a.ref.h - 1);
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

// ---------------------------------------------------------------------------------------------------
// homography, row-factorised Hessian (variant 3). The pass is bound by VALU issue with few resident
// waves (66 accumulators per lane), so this variant removes both work and registers:
//   J = (a, b, t) (x) (X, Y, 1) minus t.1, hence every Hessian entry is  sum q * X^i * Y^j  with
//   q in {aa, ab, at, bb, bt, tt}. Y is constant along a template row, so a lane only accumulates the
//   18 X-moments  sum q*X^2, sum q*X, sum q  of its row (18 FMAs per pixel instead of 36). At the end of
//   the row the wave reduces them with shuffles and lane L < 36 folds "its" entry, scaled by Y^j, into
//   ONE f64 accumulator — the 36 long-lived per-lane f32 accumulators disappear.
// Same sums as the other variants up to f32 rounding (products are associated differently).
// ---------------------------------------------------------------------------------------------------
// packed upper-triangular entry (row-major, J order aX bX tX aY bY tY a b) -> X-moment index, power of Y
__constant__ unsigned char c_hess_src[36] = {0, 1, 2, 6, 7, 8, 6, 7,  3, 4, 7, 9, 10, 7, 9,  5, 8, 10, 11, 8, 10,
                                              12, 13, 14, 12, 13,  15, 16, 13, 15,  17, 14, 16,  12, 13,  15};
__constant__ unsigned char c_hess_ypow[36] = {0, 0, 0, 1, 1, 1, 0, 0,  0, 0, 1, 1, 1, 0, 0,  0, 1, 1, 1, 0, 0,
                                               2, 2, 2, 1, 1,  2, 2, 1, 1,  2, 1, 1,  0, 0,  0};

// Sum over the 64 lanes with six DPP-modified adds (row_shr 1/2/4/8 inside each row of 16, then row_bcast 15 and 31
// across rows; shifted-in lanes contribute 0), the total of lane 63 returned in a scalar register.
// Fixed order, so deterministic; ds_bpermute-based shuffles cost an LDS round trip per step instead.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
    return v + __builtin_bit_cast(float, moved);
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = dpp_add<0x111, 0xf>(v);      // row_shr:1
    v = dpp_add<0x112, 0xf>(v);      // row_shr:2
    v = dpp_add<0x114, 0xf>(v);      // row_shr:4
    v = dpp_add<0x118, 0xf>(v);      // row_shr:8  -> lane 15 of each row = row total
    v = dpp_add<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 = wave total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// one pixel in flight between the two pipeline stages of the row-factorised pass
#ifndef STK_H8_WG
#define STK_H8_WG 4
#endif
struct H8Px {
    float sx, sy, rw, ax, ay, tval;
    f32x2_a4 i0, i1;
    f32x4_a8 g0, g1;
    int x;
};

__global__ __launch_bounds__(256, STK_H8_WG) void ecc_iter_h8_kernel(EccIterArgs a) {   // STK_H8_WG workgroups per CU
    constexpr int MOTION = STK_MOTION_HOMOGRAPHY;
    constexpr int P = 8, NH = 36, NR = 3 * P + 6, NS = NH + NR;   // NR = 30 per-lane sums besides the Hessian
    const int bid = (int)blockIdx.x;
    const int xcd = bid & 7, q = bid >> 3;
    const int slot = a.slot0 + q % a.n_slots;
    const int region = (q / a.n_slots) * 8 + xcd;
    const EccSlot* sl = a.slots + slot;
    const int frame = sl->frame;
    if (frame < 0) return;
    SlotConst c;
    load_slot_const(sl, a, c);
    const float* __restrict__ T = a.templates + (size_t)frame * a.templ_plane_stride;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hsrc = c_hess_src[min(lane, 35)], hyp = c_hess_ypow[min(lane, 35)];

    float acc[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) acc[k] = 0.f;
    double hacc = 0.0;                                        // lane L < 36: Hessian entry L
    // Tap addressing: one unsigned 32-bit byte offset per pixel, shared by the planes, on top of scalar
    // base pointers moved to the corner of the zero border (so the offset is never negative) — the
    // loads take the `saddr + voffset` form and the per-pixel 64-bit pointer arithmetic disappears.
    // gx and gy are read from the interleaved (gx, gy) plane: one 16-byte load per tap row.
    const int rs = a.ref.stride;
    const int corner = REF_PAD * rs + REF_PAD;
    const char* __restrict__ Ib = reinterpret_cast<const char*>(a.ref.I - corner);
    const char* __restrict__ Gb = reinterpret_cast<const char*>(a.ref.gxy - 2 * (size_t)corner);
    const char* __restrict__ Ib1 = Ib + (size_t)rs * 4;       // the tap row below: same vector offset, scalar base + 1 row
    const char* __restrict__ Gb1 = Gb + (size_t)rs * 8;
    const int nchunk = (a.tw + 63) >> 6;
    for (int y = region * 4 + wave; y < a.th; y += a.nb * 4) {
        const float fy = (float)y;
        const float rowX = __builtin_fmaf(c.m1, fy, c.m2), rowY = __builtin_fmaf(c.m4, fy, c.m5);
        const float rowW = __builtin_fmaf(c.m7, fy, c.m8);   // m22 == 1 is guaranteed by the launcher (den == w)
        float h2[6], h1[6], h0[6];                            // sum q*X^2, sum q*X, sum q over this lane's pixels of the row
#pragma unroll
        for (int k = 0; k < 6; k++) { h2[k] = 0.f; h1[k] = 0.f; h0[k] = 0.f; }
        const float* trow = T + (size_t)y * a.templ_row_stride;
        // lanes take ADJACENT pixels (x = 64 k + lane): the 64 tap addresses of a load are then nearly
        // contiguous (3-5 cache lines per wave-load instead of 8-16 with one quad per lane).
        // Two-stage software pipeline: the coordinates and the four loads of pixel k+1 are issued (stage A)
        // before the ~70 arithmetic instructions of pixel k (stage B), so a wave always has one pixel's
        // gathers in flight behind its own arithmetic instead of relying on the other 2 waves of the SIMD.
        // (A third pixel in flight was measured: 168 VGPRs, 1% faster — not worth sitting on the register limit.)
        auto stage_a = [&](int x, H8Px& p) {
            const int xc = min(x, a.tw - 1);                  // past the row end: a harmless repeat, skipped in stage B
            p.x = x;
            p.tval = trow[xc];
            const float fx = (float)xc;
            float sx = __builtin_fmaf(c.m0, fx, rowX), sy = __builtin_fmaf(c.m3, fx, rowY);
            const float rw = __builtin_amdgcn_rcpf(__builtin_fmaf(c.m6, fx, rowW));
            p.rw = rw;
            sx *= rw; sy *= rw;                               // hatX = -X'/den and hatY = -Y'/den are exactly -sx, -sy (den == w)
            p.sx = sx; p.sy = sy;
            const float flx = __builtin_floorf(sx), fly = __builtin_floorf(sy);
            p.ax = sx - flx; p.ay = sy - fly;
            // clamp into the zero border with one v_med3_f32 each; NaN -> -2 (all taps zero)
            const int ix = (int)__builtin_amdgcn_fmed3f(flx, -2.0f, c.fiw);
            const int iy = (int)__builtin_amdgcn_fmed3f(fly, -2.0f, c.fih);
            const unsigned bo = (unsigned)(__mul24(iy, rs) + ix + corner) << 2;
            p.i0 = *(const f32x2_a4*)(Ib + bo); p.i1 = *(const f32x2_a4*)(Ib1 + bo);
            p.g0 = *(const f32x4_a8*)(Gb + 2u * bo); p.g1 = *(const f32x4_a8*)(Gb1 + 2u * bo);
        };
        auto stage_b = [&](const H8Px& p) {
            if (p.x >= a.tw) return;
            const float fx = (float)p.x, sx = p.sx, sy = p.sy, rden = p.rw, tval = p.tval;
            const float Iw = bilerp4(p.i0.x, p.i0.y, p.i1.x, p.i1.y, p.ax, p.ay);
            const float gxw = bilerp4(p.g0.x, p.g0.z, p.g1.x, p.g1.z, p.ax, p.ay);
            const float gyw = bilerp4(p.g0.y, p.g0.w, p.g1.y, p.g1.w, p.ax, p.ay);
            bool inside = (sx > 0.0f) & (sx < c.mxw) & (sy > 0.0f) & (sy < c.mxh);
            if (!inside) {
                const float rx = __builtin_rintf(sx), ry = __builtin_rintf(sy);
                inside = (rx >= 0.0f) & (rx <= c.mxw) & (ry >= 0.0f) & (ry <= c.mxh);
                const bool edge = (__builtin_fabsf(sx + 0.5f) < 0.01f) | (__builtin_fabsf(sx - (c.mxw + 0.5f)) < 0.01f) |
                                  (__builtin_fabsf(sy + 0.5f) < 0.01f) | (__builtin_fabsf(sy - (c.mxh + 0.5f)) < 0.01f);
                if (edge) inside = nearest_inside_exact<MOTION>(p.x, y, sl->warp, c.iw, c.ih);
            }
            const float mf = inside ? 1.0f : 0.0f;
            const float ja = gxw * rden, jb = gyw * rden;
            const float jt = (-sx) * ja + (-sy) * jb;
            // Hessian: X-moments of the six products
            const float qv[6] = {ja * ja, ja * jb, ja * jt, jb * jb, jb * jt, jt * jt};
            const float xx = fx * fx;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                h2[k] = __builtin_fmaf(qv[k], xx, h2[k]);
                h1[k] = __builtin_fmaf(qv[k], fx, h1[k]);
                h0[k] += qv[k];
            }
            const float J[8] = {ja * fx, jb * fx, jt * fx, ja * fy, jb * fy, jt * fy, ja, jb};
            const float u = inside ? Iw - c.cI : Iw;
            const float v = inside ? tval - c.cT : 0.0f;
#pragma unroll
            for (int k = 0; k < P; k++) {
                acc[k] = __builtin_fmaf(J[k], u, acc[k]);
                acc[P + k] = __builtin_fmaf(J[k], v, acc[P + k]);
                acc[2 * P + k] = __builtin_fmaf(J[k], mf, acc[2 * P + k]);
            }
            const float um = u * mf;
            acc[3 * P + 0] += mf;
            acc[3 * P + 1] += um;
            acc[3 * P + 2] = __builtin_fmaf(um, u, acc[3 * P + 2]);
            acc[3 * P + 3] += v;
            acc[3 * P + 4] = __builtin_fmaf(v, v, acc[3 * P + 4]);
            acc[3 * P + 5] = __builtin_fmaf(um, v, acc[3 * P + 5]);
        };
        H8Px p0, p1;
        stage_a(lane, p0);
        for (int k = 0; k < nchunk; k += 2) {
            stage_a((k + 1) * 64 + lane, p1);
            stage_b(p0);
            stage_a((k + 2) * 64 + lane, p0);
            stage_b(p1);
        }
        // end of row: wave-reduce the 18 X-moments with DPP adds (no LDS traffic), total read from lane 63 into a
        // scalar register; then lane L < 36 takes entry L times Y^j in f64
        float sel = 0.f;
#pragma unroll
        for (int k = 0; k < 18; k++) {
            const float r = wave_sum_dpp(k < 6 ? h2[k] : k < 12 ? h1[k - 6] : h0[k - 12]);
            sel = (hsrc == k) ? r : sel;
        }
        const double dy = (double)fy;
        hacc += (double)sel * (hyp == 0 ? 1.0 : hyp == 1 ? dy : dy * dy);
    }

    // block reduction: the 30 per-lane sums by shuffles, the 36 Hessian entries are already one per lane
#pragma unroll
    for (int k = 0; k < NR; k++) {
        float r = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
        acc[k] = r;
    }
    __shared__ double red[4][NS];
    if (lane < NH) red[wave][lane] = hacc;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NR; k++) red[wave][NH + k] = (double)acc[k];
    }
    __syncthreads();
    if (threadIdx.x < NS) {
        const int k = threadIdx.x;
        const double s = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
        a.partials[((size_t)slot * NS + k) * a.nb + region] = s;   // [slot][sum][block]
    }
}

// ---------------------------------------------------------------------------------------------------
// translation / euclidean / affine (variant 3 for these motions): the data movement of the row-factorised kernel
// (lane-adjacent pixels, one 32-bit tap offset on scalar bases, (gx, gy) interleaved, two-stage software pipeline)
// with the plain per-lane moment accumulators — these motions have 15 / 21 / 45 sums, so there is nothing to factorise.
// Per-pixel arithmetic is accumulate_pixel's, i.e. identical to the direct variant up to the f32 summation order.
// ---------------------------------------------------------------------------------------------------
template <int MOTION>
__global__ __launch_bounds__(256, 3) void ecc_iter_affine_kernel(EccIterArgs a) {
    constexpr int P = MotionTraits<MOTION>::P;
    constexpr int NS = P * (P + 1) / 2 + 3 * P + 6;
    const int bid = (int)blockIdx.x;
    const int xcd = bid & 7, q = bid >> 3;
    const int slot = a.slot0 + q % a.n_slots;
    const int region = (q / a.n_slots) * 8 + xcd;
    const EccSlot* sl = a.slots + slot;
    const int frame = sl->frame;
    if (frame < 0) return;
    SlotConst c;
    load_slot_const(sl, a, c);
    const float* __restrict__ T = a.templates + (size_t)frame * a.templ_plane_stride;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) acc[k] = 0.f;
    const int rs = a.ref.stride;
    const int corner = REF_PAD * rs + REF_PAD;
    const char* __restrict__ Ib = reinterpret_cast<const char*>(a.ref.I - corner);
    const char* __restrict__ Gb = reinterpret_cast<const char*>(a.ref.gxy - 2 * (size_t)corner);
    const char* __restrict__ Ib1 = Ib + (size_t)rs * 4;
    const char* __restrict__ Gb1 = Gb + (size_t)rs * 8;
    const int nchunk = (a.tw + 63) >> 6;
    for (int y = region * 4 + wave; y < a.th; y += a.nb * 4) {
        const float fy = (float)y;
        const float rowX = __builtin_fmaf(c.m1, fy, c.m2), rowY = __builtin_fmaf(c.m4, fy, c.m5);
        const float* trow = T + (size_t)y * a.templ_row_stride;
        auto stage_a = [&](int x, H8Px& p) {
            const int xc = min(x, a.tw - 1);                  // past the row end: a harmless repeat, skipped in stage B
            p.x = x;
            p.tval = trow[xc];
            const float fx = (float)xc;
            const float sx = __builtin_fmaf(c.m0, fx, rowX), sy = __builtin_fmaf(c.m3, fx, rowY);
            p.sx = sx; p.sy = sy; p.rw = 1.0f;
            const float flx = __builtin_floorf(sx), fly = __builtin_floorf(sy);
            p.ax = sx - flx; p.ay = sy - fly;
            const int ix = (int)__builtin_amdgcn_fmed3f(flx, -2.0f, c.fiw);
            const int iy = (int)__builtin_amdgcn_fmed3f(fly, -2.0f, c.fih);
            const unsigned bo = (unsigned)(__mul24(iy, rs) + ix + corner) << 2;
            p.i0 = *(const f32x2_a4*)(Ib + bo); p.i1 = *(const f32x2_a4*)(Ib1 + bo);
            p.g0 = *(const f32x4_a8*)(Gb + 2u * bo); p.g1 = *(const f32x4_a8*)(Gb1 + 2u * bo);
        };
        auto stage_b = [&](const H8Px& p) {
            if (p.x >= a.tw) return;
            const float Iw = bilerp4(p.i0.x, p.i0.y, p.i1.x, p.i1.y, p.ax, p.ay);
            const float gxw = bilerp4(p.g0.x, p.g0.z, p.g1.x, p.g1.z, p.ax, p.ay);
            const float gyw = bilerp4(p.g0.y, p.g0.w, p.g1.y, p.g1.w, p.ax, p.ay);
            accumulate_pixel<MOTION, NS>(c, sl->warp, p.x, y, (float)p.x, fy, p.sx, p.sy, 1.0f, 0.0f, 0.0f, Iw, gxw, gyw, p.tval, acc);
        };
        H8Px p0, p1;
        stage_a(lane, p0);
        for (int k = 0; k < nchunk; k += 2) {
            stage_a((k + 1) * 64 + lane, p1);
            stage_b(p0);
            stage_a((k + 2) * 64 + lane, p0);
            stage_b(p1);
        }
    }
    block_reduce_store<NS>(acc, a, slot, region);
}

// variant 3: the production kernels (row-factorised homography pass / pipelined affine family); variant 0: the first,
// direct version (66 per-lane accumulators) — kept as an independent cross-check in the tests and for a caller-supplied
// initial homography whose m22 is not 1. (Two more variants — LDS-tiled with LDS-DMA, and row-sharing slots — were
// measured in round 1, never won, and were deleted in round 2.)
hipError_t launch_ecc_iter(const EccIterArgs& a, int motion, int variant, hipStream_t s) {
    const int grid = a.nb * a.n_slots;
    if (grid <= 0) return hipSuccess;
    if (variant == 3) {
        switch (motion) {
            case STK_MOTION_HOMOGRAPHY: ecc_iter_h8_kernel<<<grid, 256, 0, s>>>(a); break;
            case STK_MOTION_AFFINE: ecc_iter_affine_kernel<STK_MOTION_AFFINE><<<grid, 256, 0, s>>>(a); break;
            case STK_MOTION_EUCLIDEAN: ecc_iter_affine_kernel<STK_MOTION_EUCLIDEAN><<<grid, 256, 0, s>>>(a); break;
            case STK_MOTION_TRANSLATION: ecc_iter_affine_kernel<STK_MOTION_TRANSLATION><<<grid, 256, 0, s>>>(a); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
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
