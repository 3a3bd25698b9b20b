// kernels_warp.hip — warpPerspective / warpAffine (INTER_LINEAR) fused with the u8/u16 -> f32
// convert (x 1/255) and with the running f32 accumulator:
//     acc(p) (+)= sum over frames f of  bilinear( convert(frame_f), Minv_f . p )
// Reference: utils.rs:133 (convert), lib.rs:290-299 / 780-803 (warp), lib.rs:306-316 / 807-814 (add).
// The reference materialises a converted f32 frame (A2), a warped f32 frame (F1) and a fresh sum
// (G1) per frame: 12+12+12+12+12 B/px/channel-triple of traffic. Here one thread owns one
// destination pixel, keeps its three accumulator channels in registers across ALL frames of the
// launch and touches HBM for: the source taps (u8: ~3 B/px/frame, gathered, L2-friendly because the
// maps are near-identity) + one 12-byte accumulator read + one 12-byte accumulator write.
// HBM-bound; no LDS needed for the gather (footprints of neighbouring lanes overlap in L1/L2).
#include "common.h"

namespace stk {

__device__ __forceinline__ int border_interp(int p, int len, int mode) {
    if ((unsigned)p < (unsigned)len) return p;
    if (mode == STK_BORDER_REPLICATE) return p < 0 ? 0 : len - 1;
    if (mode == STK_BORDER_REFLECT || mode == STK_BORDER_REFLECT_101) {
        const int delta = mode == STK_BORDER_REFLECT_101;
        if (len == 1) return 0;
        do {
            if (p < 0) p = -p - 1 + delta;
            else p = len - 1 - (p - len) - delta;
        } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    if (mode == STK_BORDER_WRAP) {
        if (p < 0) p -= ((p - len + 1) / len) * len;
        if (p >= len) p %= len;
        return p;
    }
    return -1;   // BORDER_CONSTANT
}

__device__ __forceinline__ int sat_int_d(double v) {
    if (!(v > -2147483648.0)) return (int)0x80000000;
    if (!(v < 2147483647.0)) return 0x7fffffff;
    return (int)__builtin_rint(v);
}

template <typename T, int CN>
__global__ __launch_bounds__(256) void warp_accumulate_kernel(WarpArgs a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.dw || y >= a.dh) return;
    float* accp = a.acc + (size_t)y * a.acc_stride + (size_t)x * CN;
    float sum[CN];
#pragma unroll
    for (int c = 0; c < CN; c++) sum[c] = a.accumulate ? accp[c] : 0.0f;

    const float fx = (float)x, fy = (float)y;
    const int mode = a.border_mode;
    for (int f = 0; f < a.n_frames; f++) {
        const WarpFrame* fr = a.frames + f;
        const T* __restrict__ src = (const T*)fr->src;
        int ix, iy;
        float ax = 0, ay = 0;
        float w00 = 0, w01 = 0, w10 = 0, w11 = 0;
        bool finite = true;
        if (a.subpixel_bits == 0) {
            // OpenCV >= 4.11 kernels: f32 matrix, fma chains, true division, floor, lerp by fma
            float X = __builtin_fmaf(fr->M[0], fx, __builtin_fmaf(fr->M[1], fy, fr->M[2]));
            float Y = __builtin_fmaf(fr->M[3], fx, __builtin_fmaf(fr->M[4], fy, fr->M[5]));
            if (!a.is_affine) {
                const float W = __builtin_fmaf(fr->M[6], fx, __builtin_fmaf(fr->M[7], fy, fr->M[8]));
                X = X / W; Y = Y / W;
            }
            finite = (__builtin_fabsf(X) < 1e9f) & (__builtin_fabsf(Y) < 1e9f);   // false for NaN / inf
            const float flx = __builtin_floorf(X), fly = __builtin_floorf(Y);
            ix = finite ? (int)flx : -100000; iy = finite ? (int)fly : -100000;
            ax = X - flx; ay = Y - fly;
        } else {
            // classic remap path: 1/32-pixel quantised coordinates, 4-weight table
            int Xi, Yi;
            const double* M = fr->Md;
            if (a.is_affine) {
                const int adx = sat_int_d(M[0] * x * 1024), bdx = sat_int_d(M[3] * x * 1024);
                const int X0 = sat_int_d((M[1] * y + M[2]) * 1024) + 16;
                const int Y0 = sat_int_d((M[4] * y + M[5]) * 1024) + 16;
                Xi = (X0 + adx) >> 5; Yi = (Y0 + bdx) >> 5;
            } else {
                double W = M[6] * x + M[7] * y + M[8];
                W = W != 0 ? 32.0 / W : 0;
                const double Xd = fmax(-2147483648.0, fmin(2147483647.0, (M[0] * x + M[1] * y + M[2]) * W));
                const double Yd = fmax(-2147483648.0, fmin(2147483647.0, (M[3] * x + M[4] * y + M[5]) * W));
                Xi = sat_int_d(Xd); Yi = sat_int_d(Yd);
            }
            ix = Xi >> 5; iy = Yi >> 5;
            const float qx = (float)(Xi & 31) * (1.f / 32), qy = (float)(Yi & 31) * (1.f / 32);
            const float ux = 1.f - qx, uy = 1.f - qy;
            w00 = uy * ux; w01 = uy * qx; w10 = qy * ux; w11 = qy * qx;
        }
        int x0 = border_interp(ix, a.sw, mode), x1 = border_interp(ix + 1, a.sw, mode);
        int y0 = border_interp(iy, a.sh, mode), y1 = border_interp(iy + 1, a.sh, mode);
        if (!finite) { x0 = x1 = y0 = y1 = (mode == STK_BORDER_CONSTANT) ? -1 : 0; }
        const bool v00 = (x0 >= 0) & (y0 >= 0), v01 = (x1 >= 0) & (y0 >= 0);
        const bool v10 = (x0 >= 0) & (y1 >= 0), v11 = (x1 >= 0) & (y1 >= 0);
        // clamped addresses keep every load in bounds; out-of-image taps are replaced afterwards
        const int cx0 = max(x0, 0), cx1 = max(x1, 0), cy0 = max(y0, 0), cy1 = max(y1, 0);
        const T* r0 = src + (size_t)cy0 * a.src_stride;
        const T* r1 = src + (size_t)cy1 * a.src_stride;
#pragma unroll
        for (int c = 0; c < CN; c++) {
            const float p00 = v00 ? (float)r0[cx0 * CN + c] * a.alpha : a.bv[c];
            const float p01 = v01 ? (float)r0[cx1 * CN + c] * a.alpha : a.bv[c];
            const float p10 = v10 ? (float)r1[cx0 * CN + c] * a.alpha : a.bv[c];
            const float p11 = v11 ? (float)r1[cx1 * CN + c] * a.alpha : a.bv[c];
            float v;
            if (a.subpixel_bits == 0) {
                const float t0 = __builtin_fmaf(ax, p01 - p00, p00);
                const float t1 = __builtin_fmaf(ax, p11 - p10, p10);
                v = __builtin_fmaf(ay, t1 - t0, t0);
            } else {
                v = p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11;
            }
            sum[c] = sum[c] + v;
        }
    }
#pragma unroll
    for (int c = 0; c < CN; c++) accp[c] = sum[c];
}

hipError_t launch_warp_accumulate(const WarpArgs& a, int depth, hipStream_t s) {
    dim3 grid((a.dw + 63) / 64, (a.dh + 3) / 4);
#define STK_WARP_CASE(T, CN) warp_accumulate_kernel<T, CN><<<grid, 256, 0, s>>>(a)
    if (depth == 8 && a.cn == 3) STK_WARP_CASE(uint8_t, 3);
    else if (depth == 8 && a.cn == 1) STK_WARP_CASE(uint8_t, 1);
    else if (depth == 16 && a.cn == 3) STK_WARP_CASE(uint16_t, 3);
    else if (depth == 16 && a.cn == 1) STK_WARP_CASE(uint16_t, 1);
    else if (depth == 32 && a.cn == 3) STK_WARP_CASE(float, 3);
    else if (depth == 32 && a.cn == 1) STK_WARP_CASE(float, 1);
    else return hipErrorInvalidValue;
#undef STK_WARP_CASE
    return hipGetLastError();
}

}  // namespace stk
