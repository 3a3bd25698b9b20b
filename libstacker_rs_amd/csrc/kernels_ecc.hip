// kernels_ecc.hip — findTransformECC (lib.rs:769-777; algorithm SURVEY.md §8a-E*) as two kernels
// per iteration, with the whole iteration loop resident on the device:
//
//  ecc_iter   ONE fused pass per iteration and slot: per template pixel, projective coordinates ->
//             bilinear gathers of (I, gx, gy) from the zero-padded frame-0 planes -> nearest mask ->
//             Jacobian in registers -> 66 moment sums (36 Hessian, 8 J.Iw, 8 J.T.m, 8 J.m, 6 scalars)
//             accumulated in f32 per lane, reduced with wavefront shuffles, then LDS across the
//             four waves, written as f64 block partials. OpenCV materialises ~150 f32 planes per
//             iteration for the same result; here the algorithmic traffic is 16 B/px.
//  ecc_solve  one block per slot: fixed-order f64 sum of the block partials (deterministic), then
//             the 8x8 normal equations exactly as OpenCV forms them (f32 Hessian, LU inverse in
//             f32, lambda in f64), warp update, convergence test, and — when a frame finishes —
//             hand the slot the next frame from the device-side queue. The host never sees an
//             iteration; it only polls EccQueue::frames_done.
//
// Several frames ("slots") iterate concurrently in one launch. blockIdx is decoded so that the
// blocks working on the SAME image region for different slots share blockIdx % 8, i.e. one XCD
// and its L2: the frame-0 planes they all gather from are fetched from HBM/MALL once per XCD.
#include "common.h"

namespace stk {

typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

template <int MOTION> struct MotionTraits;
template <> struct MotionTraits<STK_MOTION_TRANSLATION> { static constexpr int P = 2; };
template <> struct MotionTraits<STK_MOTION_EUCLIDEAN> { static constexpr int P = 3; };
template <> struct MotionTraits<STK_MOTION_AFFINE> { static constexpr int P = 6; };
template <> struct MotionTraits<STK_MOTION_HOMOGRAPHY> { static constexpr int P = 8; };

__device__ __forceinline__ float bilerp(f32x2_a4 top, f32x2_a4 bot, float ax, float ay) {
    const float v0 = __builtin_fmaf(ax, top.y - top.x, top.x);
    const float v1 = __builtin_fmaf(ax, bot.y - bot.x, bot.x);
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
__device__ __noinline__ bool nearest_inside_exact(int x, int y, const float* m, int iw, int ih) {
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

template <int MOTION>
__global__ __launch_bounds__(256) void ecc_iter_kernel(EccIterArgs a) {
    constexpr int P = MotionTraits<MOTION>::P;
    constexpr int NH = P * (P + 1) / 2;
    constexpr int NS = NH + 3 * P + 6;

    // XCD-aware decode: bid = xcd + 8 * (slot + n_slots * (region / 8))
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = bid >> 3;
    const int slot = q % a.n_slots;
    const int region = (q / a.n_slots) * 8 + xcd;

    const EccSlot* sl = a.slots + slot;
    const int frame = sl->frame;
    if (frame < 0) return;                                    // idle slot: whole block leaves

    const float m0 = sl->warp[0], m1 = sl->warp[1], m2 = sl->warp[2];
    const float m3 = sl->warp[3], m4 = sl->warp[4], m5 = sl->warp[5];
    const float m6 = sl->warp[6], m7 = sl->warp[7], m8 = sl->warp[8];
    const float cI = sl->cI, cT = sl->cT;
    const bool den_is_w = (m8 == 1.0f);

    const float* __restrict__ T = a.templates + (size_t)frame * a.templ_plane_stride;
    const float* __restrict__ RI = a.ref.I;
    const float* __restrict__ RX = a.ref.gx;
    const float* __restrict__ RY = a.ref.gy;
    const int rs = a.ref.stride;
    const float fiw = (float)a.ref.w, fih = (float)a.ref.h;
    const float mxw = (float)(a.ref.w - 1), mxh = (float)(a.ref.h - 1);

    float acc[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) acc[k] = 0.f;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qw = (a.tw + 3) >> 2;                           // quads per row

    for (int y = region * 4 + wave; y < a.th; y += a.nb * 4) {
        const float fy = (float)y;
        const float rowX = __builtin_fmaf(m1, fy, m2);
        const float rowY = __builtin_fmaf(m4, fy, m5);
        const float rowW = __builtin_fmaf(m7, fy, m8);
        const float rowD = __builtin_fmaf(m7, fy, 1.0f);
        const float* trow = T + (size_t)y * a.templ_row_stride;
        for (int qx = lane; qx < qw; qx += 64) {
            const float4 t4 = *reinterpret_cast<const float4*>(trow + qx * 4);
            const float tv[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int x = qx * 4 + j;
                if (x < a.tw) {
                    const float fx = (float)x;
                    float sx = __builtin_fmaf(m0, fx, rowX);
                    float sy = __builtin_fmaf(m3, fx, rowY);
                    float rden = 1.0f, hx = 0.0f, hy = 0.0f;      // 1/den, hatX, hatY of the homography Jacobian
                    if constexpr (MOTION == STK_MOTION_HOMOGRAPHY) {
                        const float rw = 1.0f / __builtin_fmaf(m6, fx, rowW);
                        rden = rw;
                        if (!den_is_w) rden = 1.0f / __builtin_fmaf(m6, fx, rowD);   // den = X*h2 + Y*h5 + 1
                        hx = -sx * rden; hy = -sy * rden;
                        sx *= rw; sy *= rw;
                    }
                    const float flx = __builtin_floorf(sx), fly = __builtin_floorf(sy);
                    const float ax = sx - flx, ay = sy - fly;
                    // clamp the integer coordinate into the zero border; NaN -> -2 (all taps zero)
                    const int ix = (int)__builtin_fminf(__builtin_fmaxf(flx, -2.0f), fiw);
                    const int iy = (int)__builtin_fminf(__builtin_fmaxf(fly, -2.0f), fih);
                    const int off = iy * rs + ix;
                    const float Iw = bilerp(*(const f32x2_a4*)(RI + off), *(const f32x2_a4*)(RI + off + rs), ax, ay);
                    const float gxw = bilerp(*(const f32x2_a4*)(RX + off), *(const f32x2_a4*)(RX + off + rs), ax, ay);
                    const float gyw = bilerp(*(const f32x2_a4*)(RY + off), *(const f32x2_a4*)(RY + off + rs), ax, ay);
                    // INTER_NEAREST mask: rounded source coordinate inside the input image. OpenCV rounds
                    // a double (homography) or 10-bit fixed-point (affine) coordinate; the f32 value decides
                    // except within 0.01 px of a boundary of the valid range, where the exact form is redone.
                    const float rx = __builtin_rintf(sx), ry = __builtin_rintf(sy);
                    bool inside = (rx >= 0.0f) & (rx <= mxw) & (ry >= 0.0f) & (ry <= mxh);
                    const bool edge = (__builtin_fabsf(sx + 0.5f) < 0.01f) | (__builtin_fabsf(sx - (mxw + 0.5f)) < 0.01f) |
                                      (__builtin_fabsf(sy + 0.5f) < 0.01f) | (__builtin_fabsf(sy - (mxh + 0.5f)) < 0.01f);
                    if (edge) inside = nearest_inside_exact<MOTION>(x, y, sl->warp, a.ref.w, a.ref.h);
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
                        const float ex = -(fx * m3) - (fy * m0);     // h0 = m00 (cos), h1 = m10 (sin)
                        const float ey = (fx * m0) - (fy * m3);
                        J[0] = gxw * ex + gyw * ey; J[1] = gxw; J[2] = gyw;
                    } else {
                        J[0] = gxw; J[1] = gyw;
                    }
                    // centred samples: u = Iw - cI inside the mask (Iw outside), v = (T - cT) inside, 0 outside
                    const float u = inside ? Iw - cI : Iw;
                    const float v = inside ? tv[j] - cT : 0.0f;
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
            }
        }
    }

    // wavefront reduction (64 lanes), then the four waves through LDS in f64
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

hipError_t launch_ecc_iter(const EccIterArgs& a, int motion, hipStream_t s) {
    const int grid = a.nb * a.n_slots;
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
