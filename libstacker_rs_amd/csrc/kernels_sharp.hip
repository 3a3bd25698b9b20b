// kernels_sharp.hip — the reference's four sharpness metrics (lib.rs:1030-1166: LAPM, LAPV, TENG, GLVN), the
// example's pre-filter (examples/main.rs:40-47). Each is a small stencil into CV_64F followed by cv::mean or
// cv::meanStdDev; here the stencil and the reduction are one pass: a lane evaluates the filter at its pixels from
// the (L1/L2-cached) source, accumulates the sum (and the sum of squares), and a workgroup writes one partial pair.
// The host adds the partials in index order (stacker.cpp) and applies the reference's closing formula.
//   8-bit input: every filtered value is an integer (LAPM: a multiple of 1/4, computed x4), so the accumulation is
//   done in int64 and is exact — the result does not depend on the summation order (bit-exact vs the oracle).
//   f32 input: f64 arithmetic in the order of the separable filter (row pass, then column pass), f64 sums in a fixed
//   order: deterministic, equal to the oracle's sequential sum to ~1e-13 relative.
#include "common.h"

namespace stk {

__device__ __forceinline__ int sharp_reflect101(int p, int len) {
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}
__device__ __forceinline__ int sharp_replicate(int p, int len) { return p < 0 ? 0 : (p >= len ? len - 1 : p); }

struct SharpKernels { double smooth[7], deriv[7]; int n_smooth, n_deriv; };   // TENG: getDerivKernels' integer taps

// S = long long (8-bit input) or double (f32 input)
template <typename T, typename S, int METRIC>
__global__ __launch_bounds__(256) void sharpness_kernel(const T* __restrict__ src, int w, int h, SharpKernels k,
                                                        S* __restrict__ partials) {
    S a = 0, b = 0;
    const size_t n = (size_t)w * h;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / w), x = (int)(i - (size_t)y * w);
        auto px = [&](int xx, int yy) -> S { return (S)src[(size_t)yy * w + xx]; };
        if constexpr (METRIC == 0) {
            // LAPM: m = [-1 2 -1]; g = [1 2 1] / 4 (the /4 is applied by the host for integer input)
            const S g[3] = {(S)(sizeof(T) == 1 ? 1 : 0.25), (S)(sizeof(T) == 1 ? 2 : 0.5), (S)(sizeof(T) == 1 ? 1 : 0.25)};
            int xs[3], ys[3];
            for (int t = 0; t < 3; t++) { xs[t] = sharp_reflect101(x + t - 1, w); ys[t] = sharp_reflect101(y + t - 1, h); }
            S lx = 0, ly = 0;
            for (int j = 0; j < 3; j++) {                       // column pass over row-filtered values
                const S rm = ((S)(-1) * px(xs[0], ys[j]) + (S)2 * px(xs[1], ys[j])) + (S)(-1) * px(xs[2], ys[j]);
                lx += g[j] * rm;
                const S rg = (g[0] * px(xs[0], ys[j]) + g[1] * px(xs[1], ys[j])) + g[2] * px(xs[2], ys[j]);
                ly += (S)(j == 1 ? 2 : -1) * rg;
            }
            a += (lx < 0 ? -lx : lx) + (ly < 0 ? -ly : ly);
        } else if constexpr (METRIC == 1) {
            const int xm = sharp_replicate(x - 1, w), xp = sharp_replicate(x + 1, w);
            const int ym = sharp_replicate(y - 1, h), yp = sharp_replicate(y + 1, h);
            const S v = ((((S)2 * px(xm, ym) + (S)2 * px(xp, ym)) - (S)8 * px(x, y)) + (S)2 * px(xm, yp)) + (S)2 * px(xp, yp);
            a += v; b += v * v;
        } else if constexpr (METRIC == 2) {
            const int rs = k.n_smooth / 2, rd = k.n_deriv / 2;
            S gx = 0, gy = 0;
            for (int j = 0; j < k.n_smooth; j++) {              // gx: derivative along x (row pass), smoothing along y
                const int yy = sharp_reflect101(y + j - rs, h);
                S r = 0;
                for (int t = 0; t < k.n_deriv; t++) r += (S)k.deriv[t] * px(sharp_reflect101(x + t - rd, w), yy);
                gx += (S)k.smooth[j] * r;
            }
            for (int j = 0; j < k.n_deriv; j++) {               // gy: smoothing along x, derivative along y
                const int yy = sharp_reflect101(y + j - rd, h);
                S r = 0;
                for (int t = 0; t < k.n_smooth; t++) r += (S)k.smooth[t] * px(sharp_reflect101(x + t - rs, w), yy);
                gy += (S)k.deriv[j] * r;
            }
            a += gx * gx + gy * gy;
        } else {
            const S v = px(x, y);
            a += v; b += v * v;
        }
    }
    // fixed-order workgroup reduction through LDS
    __shared__ S ra[256], rb[256];
    ra[threadIdx.x] = a; rb[threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { ra[threadIdx.x] += ra[threadIdx.x + o]; rb[threadIdx.x] += rb[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = ra[0]; partials[2 * blockIdx.x + 1] = rb[0]; }
}

hipError_t launch_sharpness(const void* grey, int depth, int w, int h, int metric, int ksize, void* partials, int n_blocks,
                            hipStream_t s) {
    SharpKernels k{};
    k.n_smooth = 1; k.n_deriv = 3;
    const double sm3[] = {1, 2, 1}, sm5[] = {1, 4, 6, 4, 1}, sm7[] = {1, 6, 15, 20, 15, 6, 1};
    const double d3[] = {-1, 0, 1}, d5[] = {-1, -2, 0, 2, 1}, d7[] = {-1, -4, -5, 0, 5, 4, 1};
    k.smooth[0] = 1;
    for (int i = 0; i < 3; i++) k.deriv[i] = d3[i];
    if (ksize == 3) { k.n_smooth = 3; for (int i = 0; i < 3; i++) k.smooth[i] = sm3[i]; }
    if (ksize == 5) { k.n_smooth = k.n_deriv = 5; for (int i = 0; i < 5; i++) { k.smooth[i] = sm5[i]; k.deriv[i] = d5[i]; } }
    if (ksize == 7) { k.n_smooth = k.n_deriv = 7; for (int i = 0; i < 7; i++) { k.smooth[i] = sm7[i]; k.deriv[i] = d7[i]; } }
#define STK_SHARP(T, S, M) sharpness_kernel<T, S, M><<<n_blocks, 256, 0, s>>>((const T*)grey, w, h, k, (S*)partials)
    if (depth == 8) {
        switch (metric) { case 0: STK_SHARP(uint8_t, long long, 0); break; case 1: STK_SHARP(uint8_t, long long, 1); break;
                          case 2: STK_SHARP(uint8_t, long long, 2); break; default: STK_SHARP(uint8_t, long long, 3); break; }
    } else {
        switch (metric) { case 0: STK_SHARP(float, double, 0); break; case 1: STK_SHARP(float, double, 1); break;
                          case 2: STK_SHARP(float, double, 2); break; default: STK_SHARP(float, double, 3); break; }
    }
#undef STK_SHARP
    return hipGetLastError();
}

}  // namespace stk
