// kernels_prep.hip — per-frame preparation kernels (gfx950):
//   grey           cvt_color(BGR2GRAY) on the integer image            (utils.rs:136-142)
//   convert_f32    Mat::convert_to(CV_32F, alpha)                      (utils.rs:133,     SURVEY A2)
//   grey_blur      BGR -> grey -> GaussianBlur(float(grey), g x g, sigma 0, REFLECT_101) in ONE pass:
//                  the grey tile with its halo is staged in LDS, row-filtered into a second LDS
//                  tile and column-filtered to HBM (findTransformECC setup step 3, SURVEY §8a-E*)
//   ref_planes     blurred frame 0 -> zero-padded I / gx / gy planes    (ECC setup step 5)
// HBM-bound: grey_blur reads 3 B/px (u8 BGR) and writes 4 B/px.
#include "common.h"

namespace stk {

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// ---- grey -------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t grey_u8(unsigned b, unsigned g, unsigned r) {
    return (uint8_t)((b * 3735u + g * 19235u + r * 9798u + (1u << 14)) >> 15);
}
__device__ __forceinline__ uint16_t grey_u16(unsigned b, unsigned g, unsigned r) {
    return (uint16_t)((b * 1868u + g * 9617u + r * 4899u + (1u << 13)) >> 14);
}
__device__ __forceinline__ float grey_f32(float b, float g, float r) {
    return b * 0.114f + g * 0.587f + r * 0.299f;   // -ffp-contract=off: three roundings, as the oracle
}

// blockIdx.z = frame of a batch: source frames `src_frame_stride` elements apart, outputs `out_frame_stride` apart
// CN = 3 (BGR) or 4 (BGRA: cvtColor(BGR2GRAY) takes four channels and ignores the fourth, utils.rs:136-142)
template <typename T, int CN>
__global__ __launch_bounds__(256) void grey_kernel(const T* __restrict__ src, size_t stride, int w, int h,
                                                   T* __restrict__ out, size_t src_frame_stride, size_t out_frame_stride) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    src += blockIdx.z * src_frame_stride; out += blockIdx.z * out_frame_stride;
    const T* p = src + (size_t)y * stride + (size_t)x * CN;
    T v;
    if constexpr (sizeof(T) == 1) v = grey_u8(p[0], p[1], p[2]);
    else if constexpr (sizeof(T) == 2) v = grey_u16(p[0], p[1], p[2]);
    else v = grey_f32(p[0], p[1], p[2]);
    out[(size_t)y * w + x] = v;
}

// 8-bit fast path: four pixels per lane from three aligned dwords, one dword store (rows and outputs 4-byte aligned)
__global__ __launch_bounds__(256) void grey_u8x4_kernel(const uint8_t* __restrict__ src, size_t stride, int w, int h,
                                                        uint8_t* __restrict__ out, size_t src_frame_stride, size_t out_frame_stride) {
    const int q = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (q * 4 >= w) return;
    src += blockIdx.z * src_frame_stride; out += blockIdx.z * out_frame_stride;
    const uint32_t* p = reinterpret_cast<const uint32_t*>(src + (size_t)y * stride + (size_t)q * 12);
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];   // b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
    const uint32_t g0 = grey_u8(d0 & 255u, (d0 >> 8) & 255u, (d0 >> 16) & 255u);
    const uint32_t g1 = grey_u8(d0 >> 24, d1 & 255u, (d1 >> 8) & 255u);
    const uint32_t g2 = grey_u8((d1 >> 16) & 255u, d1 >> 24, d2 & 255u);
    const uint32_t g3 = grey_u8((d2 >> 8) & 255u, (d2 >> 16) & 255u, d2 >> 24);
    *reinterpret_cast<uint32_t*>(out + (size_t)y * w + (size_t)q * 4) = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
}

// n_frames > 1: frames `src_frame_bytes` apart in memory, grey images `out_frame_elems` elements apart
hipError_t launch_grey(const void* bgr, int depth, int w, int h, size_t stride_bytes, void* out, hipStream_t s,
                       int n_frames, size_t src_frame_bytes, size_t out_frame_elems, int cn) {
    if (cn != 3 && cn != 4) return hipErrorInvalidValue;
    if (cn == 3 && depth == 8 && w % 4 == 0 && stride_bytes % 4 == 0 && src_frame_bytes % 4 == 0 && out_frame_elems % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(bgr) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 3) == 0) {
        dim3 g4((w / 4 + 255) / 256, h, n_frames);
        grey_u8x4_kernel<<<g4, 256, 0, s>>>((const uint8_t*)bgr, stride_bytes, w, h, (uint8_t*)out, src_frame_bytes, out_frame_elems);
        return hipGetLastError();
    }
    dim3 grid((w + 255) / 256, h, n_frames);
#define STK_GREY(T, CN, div) grey_kernel<T, CN><<<grid, 256, 0, s>>>((const T*)bgr, stride_bytes / div, w, h, (T*)out, src_frame_bytes / div, out_frame_elems)
    if (cn == 3) { if (depth == 8) STK_GREY(uint8_t, 3, 1); else if (depth == 16) STK_GREY(uint16_t, 3, 2); else STK_GREY(float, 3, 4); }
    else { if (depth == 8) STK_GREY(uint8_t, 4, 1); else if (depth == 16) STK_GREY(uint16_t, 4, 2); else STK_GREY(float, 4, 4); }
#undef STK_GREY
    return hipGetLastError();
}

// 16-bit grey -> 8-bit grey for ORB on 16-bit stacks (BASELINE configs[4], an extension: SURVEY 8d): (g + 128) / 257
__global__ __launch_bounds__(256) void grey16_to_8_kernel(const uint16_t* __restrict__ src, size_t n, uint8_t* __restrict__ dst) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    for (; i < n; i += step) dst[i] = (uint8_t)(((unsigned)src[i] + 128u) / 257u);
}

// BGR 16-bit frame -> 8-bit grey in one pass: grey16 by the 16U formula, then (g + 128) / 257. Batched over frames.
__global__ __launch_bounds__(256) void bgr16_to_grey8_kernel(const uint16_t* __restrict__ src, size_t stride, int w, int h,
                                                             uint8_t* __restrict__ out, size_t src_frame_stride, size_t out_frame_stride) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    src += blockIdx.z * src_frame_stride; out += blockIdx.z * out_frame_stride;
    const uint16_t* p = src + (size_t)y * stride + (size_t)x * 3;
    out[(size_t)y * w + x] = (uint8_t)(((unsigned)grey_u16(p[0], p[1], p[2]) + 128u) / 257u);
}

// fast path: four pixels per lane from six aligned dwords (24 bytes), one dword store (rows, frames and outputs 4-byte aligned)
__global__ __launch_bounds__(256) void bgr16_to_grey8_x4_kernel(const uint16_t* __restrict__ src, size_t stride, int w, int h,
                                                                uint8_t* __restrict__ out, size_t src_frame_stride, size_t out_frame_stride) {
    const int q = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (q * 4 >= w) return;
    src += blockIdx.z * src_frame_stride; out += blockIdx.z * out_frame_stride;
    const uint32_t* p = reinterpret_cast<const uint32_t*>(src + (size_t)y * stride + (size_t)q * 12);
    uint32_t d[6];                                                   // b0 g0 | r0 b1 | g1 r1 | b2 g2 | r2 b3 | g3 r3 (8-byte aligned)
    __builtin_memcpy(d, p, 24);
    const unsigned g0 = ((unsigned)grey_u16(d[0] & 0xffffu, d[0] >> 16, d[1] & 0xffffu) + 128u) / 257u;
    const unsigned g1 = ((unsigned)grey_u16(d[1] >> 16, d[2] & 0xffffu, d[2] >> 16) + 128u) / 257u;
    const unsigned g2 = ((unsigned)grey_u16(d[3] & 0xffffu, d[3] >> 16, d[4] & 0xffffu) + 128u) / 257u;
    const unsigned g3 = ((unsigned)grey_u16(d[4] >> 16, d[5] & 0xffffu, d[5] >> 16) + 128u) / 257u;
    *reinterpret_cast<uint32_t*>(out + (size_t)y * w + (size_t)q * 4) = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
}

// eight pixels per lane: three aligned 16-byte loads (48 bytes), one 8-byte store; the lane also walks down R rows so that
// every lane has 3 R loads in flight (round 3: the four-pixel kernel moved 58 MB per 4K frame at 2.5 TB/s, a third of the
// peak and 6.7 % of the hybrid step)
template <int R>
__global__ __launch_bounds__(256) void bgr16_to_grey8_x8_kernel(const uint16_t* __restrict__ src, size_t stride, int w, int h,
                                                                uint8_t* __restrict__ out, size_t src_frame_stride, size_t out_frame_stride) {
    const int q = blockIdx.x * 256 + threadIdx.x, y0 = blockIdx.y * R;
    if (q * 8 >= w) return;
    src += blockIdx.z * src_frame_stride; out += blockIdx.z * out_frame_stride;
    uint4 d[R][3];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int y = min(y0 + r, h - 1);
        const uint4* p = reinterpret_cast<const uint4*>(src + (size_t)y * stride + (size_t)q * 24);
        d[r][0] = p[0]; d[r][1] = p[1]; d[r][2] = p[2];
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (y0 + r >= h) break;
        // 24 halfwords b0 g0 r0 b1 ... r7 in 12 dwords
        const uint32_t v[12] = {d[r][0].x, d[r][0].y, d[r][0].z, d[r][0].w, d[r][1].x, d[r][1].y, d[r][1].z, d[r][1].w,
                                d[r][2].x, d[r][2].y, d[r][2].z, d[r][2].w};
        auto hw = [&](int k) -> unsigned { return (k & 1) ? (v[k >> 1] >> 16) : (v[k >> 1] & 0xffffu); };
        unsigned g[8];
#pragma unroll
        for (int k = 0; k < 8; k++) g[k] = ((unsigned)grey_u16(hw(3 * k), hw(3 * k + 1), hw(3 * k + 2)) + 128u) / 257u;
        uint2 o;
        o.x = g[0] | (g[1] << 8) | (g[2] << 16) | (g[3] << 24);
        o.y = g[4] | (g[5] << 8) | (g[6] << 16) | (g[7] << 24);
        *reinterpret_cast<uint2*>(out + (size_t)(y0 + r) * w + (size_t)q * 8) = o;
    }
}

hipError_t launch_bgr16_to_grey8(const void* bgr16, int w, int h, size_t stride_bytes, uint8_t* out, hipStream_t s, int n_frames,
                                 size_t src_frame_bytes, size_t out_frame_elems) {
    if (w % 8 == 0 && stride_bytes % 16 == 0 && src_frame_bytes % 16 == 0 && out_frame_elems % 8 == 0 &&
        (reinterpret_cast<uintptr_t>(bgr16) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0) {
        constexpr int R = 2;
        dim3 g8((w / 8 + 255) / 256, (h + R - 1) / R, n_frames);
        bgr16_to_grey8_x8_kernel<R><<<g8, 256, 0, s>>>((const uint16_t*)bgr16, stride_bytes / 2, w, h, out, src_frame_bytes / 2, out_frame_elems);
        return hipGetLastError();
    }
    if (w % 4 == 0 && stride_bytes % 8 == 0 && src_frame_bytes % 8 == 0 && out_frame_elems % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(bgr16) & 7) == 0 && (reinterpret_cast<uintptr_t>(out) & 3) == 0) {
        dim3 g4((w / 4 + 255) / 256, h, n_frames);
        bgr16_to_grey8_x4_kernel<<<g4, 256, 0, s>>>((const uint16_t*)bgr16, stride_bytes / 2, w, h, out, src_frame_bytes / 2, out_frame_elems);
        return hipGetLastError();
    }
    dim3 grid((w + 255) / 256, h, n_frames);
    bgr16_to_grey8_kernel<<<grid, 256, 0, s>>>((const uint16_t*)bgr16, stride_bytes / 2, w, h, out, src_frame_bytes / 2, out_frame_elems);
    return hipGetLastError();
}

hipError_t launch_grey16_to_8(const uint16_t* src, size_t n, uint8_t* dst, hipStream_t s) {
    const int blocks = (int)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, 8192));
    grey16_to_8_kernel<<<blocks, 256, 0, s>>>(src, n, dst);
    return hipGetLastError();
}

// ---- convert ----------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void convert_kernel(const T* __restrict__ src, size_t n, float alpha,
                                                      float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    for (; i < n; i += step) out[i] = (float)src[i] * alpha;
}

hipError_t launch_convert_f32(const void* src, int depth, size_t n, float alpha, float* out, hipStream_t s) {
    int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    if (blocks < 1) blocks = 1;
    if (depth == 8) convert_kernel<uint8_t><<<blocks, 256, 0, s>>>((const uint8_t*)src, n, alpha, out);
    else if (depth == 16) convert_kernel<uint16_t><<<blocks, 256, 0, s>>>((const uint16_t*)src, n, alpha, out);
    else convert_kernel<float><<<blocks, 256, 0, s>>>((const float*)src, n, alpha, out);
    return hipGetLastError();
}

// ---- fused grey + Gaussian blur ---------------------------------------------------------------
struct GaussTaps { float k[32]; int r; };   // k[i] = weight at distance i from the centre (kernel sizes up to 63: the tiled kernel's LDS tile)

static bool gaussian_taps(int ksize, GaussTaps& t) {
    if (ksize <= 0 || ksize % 2 == 0 || ksize > 63) return false;
    t.r = ksize / 2;
    static const float tab[4][4] = {{1.f, 0, 0, 0}, {0.5f, 0.25f, 0, 0}, {0.375f, 0.25f, 0.0625f, 0},
                                    {0.28125f, 0.21875f, 0.109375f, 0.03125f}};
    for (int i = 0; i < 32; i++) t.k[i] = 0;
    if (ksize <= 7) { for (int i = 0; i <= t.r; i++) t.k[i] = tab[t.r][i]; return true; }
    // cv::getGaussianKernel(n, sigma<=0): sigma = 0.3*((n-1)*0.5-1)+0.8, normalised exp(-x^2/2s^2), double -> f32
    const double sigma = ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8, sc = -0.5 / (sigma * sigma);
    double sum = 0, v[64];
    for (int i = 0; i < ksize; i++) { double x = i - (ksize - 1) * 0.5; v[i] = std::exp(sc * x * x); sum += v[i]; }
    for (int i = 0; i <= t.r; i++) t.k[i] = (float)(v[t.r + i] * (1.0 / sum));
    return true;
}

constexpr int BT_X = 64, BT_Y = 16;

template <typename T, int CN>
__global__ __launch_bounds__(256) void grey_blur_kernel(const T* __restrict__ src, size_t stride, int w, int h,
                                                        GaussTaps taps, float* __restrict__ out, int out_stride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int r = taps.r;
    const int GW = BT_X + 2 * r, GH = BT_Y + 2 * r;
    float* G = lds;                 // [GH][GW]  grey tile with halo
    float* R = lds + GH * GW;       // [GH][BT_X] row-filtered
    const int x0 = blockIdx.x * BT_X, y0 = blockIdx.y * BT_Y;
    const int tid = threadIdx.x;

    for (int i = tid; i < GH * GW; i += 256) {
        const int ty = i / GW, tx = i - ty * GW;
        const int sx = reflect101(x0 - r + tx, w), sy = reflect101(y0 - r + ty, h);
        const T* p = src + (size_t)sy * stride + (size_t)sx * CN;
        float v;
        if constexpr (CN == 1) v = (float)p[0];
        else if constexpr (sizeof(T) == 1) v = (float)grey_u8(p[0], p[1], p[2]);
        else if constexpr (sizeof(T) == 2) v = (float)grey_u16(p[0], p[1], p[2]);
        else v = grey_f32((float)p[0], (float)p[1], (float)p[2]);
        G[i] = v;
    }
    __syncthreads();
    for (int i = tid; i < GH * BT_X; i += 256) {
        const int ty = i / BT_X, tx = i - ty * BT_X;
        const float* g = G + ty * GW + tx + r;
        float s = taps.k[0] * g[0];
        for (int j = 1; j <= r; j++) s += taps.k[j] * (g[-j] + g[j]);
        R[i] = s;
    }
    __syncthreads();
    for (int i = tid; i < BT_Y * BT_X; i += 256) {
        const int ty = i / BT_X, tx = i - ty * BT_X;
        const int x = x0 + tx, y = y0 + ty;
        const float* c = R + (ty + r) * BT_X + tx;
        float s = taps.k[0] * c[0];
        for (int j = 1; j <= r; j++) s += taps.k[j] * (c[-j * BT_X] + c[j * BT_X]);
        if (x < w && y < h) out[(size_t)y * out_stride + x] = s;
    }
}

// ---- fused grey + Gaussian blur, fast path: BGR u8, kernel size 3 / 5 / 7, 4-byte aligned rows ---------------
// Same arithmetic as grey_blur_kernel (same grey formula, same tap order), different data movement:
//  * a 128 x 32 output tile per workgroup; the BGR bytes of 4 pixels are three aligned dword loads (the tile starts
//    4 px left of its first output so that every quad is 12-byte aligned), unpacked in registers;
//  * LDS traffic in float4: the row pass reads 12 floats and writes 4 outputs per item, the column pass keeps a
//    4-wide strip in registers and streams 16-byte stores, 512 contiguous bytes per row and 32 lanes.
constexpr int FB_X = 128, FB_Y = 32, FB_HX = 4;           // outputs per tile; horizontal halo loaded (>= r, multiple of 4)
constexpr int FB_GW = FB_X + 2 * FB_HX;                   // 136 floats per grey tile row

// T = uint8_t (stride in bytes) or uint16_t (16-bit stacks of the hybrid path; stride in elements, rows 4-byte aligned)
template <int R, typename T = uint8_t>
__global__ __launch_bounds__(256) void grey_blur_u8c3_kernel(const T* __restrict__ src, size_t stride, int w, int h,
                                                             GaussTaps taps, float* __restrict__ out, int out_stride) {
    constexpr int GH = FB_Y + 2 * R;
    __shared__ __attribute__((aligned(16))) float G[GH * FB_GW];
    __shared__ __attribute__((aligned(16))) float Rw[GH * FB_X];
    const int x0 = blockIdx.x * FB_X, y0 = blockIdx.y * FB_Y;
    const int tid = threadIdx.x;
    float k[R + 1];
#pragma unroll
    for (int j = 0; j <= R; j++) k[j] = taps.k[j];

    // phase 1: grey tile with halo, one pixel quad per item
    constexpr int QW = FB_GW / 4;                          // 34 quads per row
    for (int i = tid; i < GH * QW; i += 256) {
        const int qy = i / QW, qx = i - qy * QW;
        const int sy = reflect101(y0 - R + qy, h);
        const int sx0 = x0 - FB_HX + 4 * qx;
        const T* row = src + (size_t)sy * stride;
        float4 g;
        if (sx0 >= 0 && sx0 + 3 < w) {
            const uint32_t* p = reinterpret_cast<const uint32_t*>(row + (size_t)sx0 * 3);
            if constexpr (sizeof(T) == 1) {
                const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];   // b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
                g.x = (float)grey_u8(d0 & 255u, (d0 >> 8) & 255u, (d0 >> 16) & 255u);
                g.y = (float)grey_u8(d0 >> 24, d1 & 255u, (d1 >> 8) & 255u);
                g.z = (float)grey_u8((d1 >> 16) & 255u, d1 >> 24, d2 & 255u);
                g.w = (float)grey_u8((d2 >> 8) & 255u, (d2 >> 16) & 255u, d2 >> 24);
            } else {
                const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4], d5 = p[5];   // b0 g0 | r0 b1 | g1 r1 | b2 g2 | r2 b3 | g3 r3
                g.x = (float)grey_u16(d0 & 0xffffu, d0 >> 16, d1 & 0xffffu);
                g.y = (float)grey_u16(d1 >> 16, d2 & 0xffffu, d2 >> 16);
                g.z = (float)grey_u16(d3 & 0xffffu, d3 >> 16, d4 & 0xffffu);
                g.w = (float)grey_u16(d4 >> 16, d5 & 0xffffu, d5 >> 16);
            }
        } else {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const T* q = row + (size_t)reflect101(sx0 + e, w) * 3;
                if constexpr (sizeof(T) == 1) v[e] = (float)grey_u8(q[0], q[1], q[2]);
                else v[e] = (float)grey_u16(q[0], q[1], q[2]);
            }
            g = make_float4(v[0], v[1], v[2], v[3]);
        }
        *reinterpret_cast<float4*>(G + qy * FB_GW + 4 * qx) = g;
    }
    __syncthreads();

    // phase 2: row filter, 4 outputs per item
    for (int i = tid; i < GH * (FB_X / 4); i += 256) {
        const int ty = i / (FB_X / 4), q = i - ty * (FB_X / 4);
        const float* gp = G + ty * FB_GW + 4 * q;           // gp[4 + e] is the centre of output e
        const float4 a = *reinterpret_cast<const float4*>(gp), b = *reinterpret_cast<const float4*>(gp + 4),
                     c = *reinterpret_cast<const float4*>(gp + 8);
        const float f[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float sacc = k[0] * f[4 + e];
#pragma unroll
            for (int j = 1; j <= R; j++) sacc += k[j] * (f[4 + e - j] + f[4 + e + j]);
            o[e] = sacc;
        }
        *reinterpret_cast<float4*>(Rw + ty * FB_X + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();

    // phase 3: column filter; thread = 4-wide strip x 4 output rows
    {
        const int strip = tid & 31, rg = tid >> 5;         // 32 strips, 8 row groups of 4 rows
        float4 win[4 + 2 * R];
#pragma unroll
        for (int t = 0; t < 4 + 2 * R; t++) win[t] = *reinterpret_cast<const float4*>(Rw + (rg * 4 + t) * FB_X + 4 * strip);
        const int x = x0 + 4 * strip;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int y = y0 + rg * 4 + e;
            float4 sacc;
            sacc.x = k[0] * win[e + R].x; sacc.y = k[0] * win[e + R].y; sacc.z = k[0] * win[e + R].z; sacc.w = k[0] * win[e + R].w;
#pragma unroll
            for (int j = 1; j <= R; j++) {
                sacc.x += k[j] * (win[e + R - j].x + win[e + R + j].x);
                sacc.y += k[j] * (win[e + R - j].y + win[e + R + j].y);
                sacc.z += k[j] * (win[e + R - j].z + win[e + R + j].z);
                sacc.w += k[j] * (win[e + R - j].w + win[e + R + j].w);
            }
            if (y < h) {
                float* op = out + (size_t)y * out_stride + x;
                if (x + 3 < w) *reinterpret_cast<float4*>(op) = sacc;
                else {
                    if (x < w) op[0] = sacc.x;
                    if (x + 1 < w) op[1] = sacc.y;
                    if (x + 2 < w) op[2] = sacc.z;
                }
            }
        }
    }
}

// ---- fused grey + Gaussian blur, streaming form: a whole batch of frames in one launch -------------------------------
// The tiled kernel above is bound by its two workgroup barriers and by the occupancy its 38 KB of LDS allows (rocprofv3:
// waves parked 70 % of their cycles, 2.5 TB/s). This one uses no LDS and no barrier: a lane owns a 4-pixel-wide column
// strip and walks down GS_SEG + 2 R rows; per row it loads its own 12 bytes (4 BGR pixels, one aligned dwordx3), makes
// the four grey values, fetches the R neighbours on each side from the adjacent lanes (cross-lane moves; the first and
// last lane of a wave only feed their neighbours, so waves overlap by two quads), row-filters, keeps the last 2 R + 1
// row results in registers and column-filters one output row per step (one float4 store). Rows are prefetched GS_PF
// deep. Same arithmetic in the same order as the tiled kernel: bit-identical planes. Enough waves to fill the chip come
// from batching the frames of a shard (blockIdx.z); single frames keep the tiled kernel.
#ifndef GS_SEG_OVERRIDE
#define GS_SEG_OVERRIDE 32
#endif
#ifndef GS_PF_OVERRIDE
#define GS_PF_OVERRIDE 4
#endif
constexpr int GS_SEG = GS_SEG_OVERRIDE, GS_PF = GS_PF_OVERRIDE, GS_QW = 62;         // output rows per segment, rows in flight, output quads per wave

template <typename T> struct QuadRaw;
template <> struct QuadRaw<uint8_t> { uint32_t d[3]; };
template <> struct QuadRaw<uint16_t> { uint32_t d[6]; };

__device__ __forceinline__ float4 quad_grey(const QuadRaw<uint8_t>& r) {      // b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
    const uint32_t d0 = r.d[0], d1 = r.d[1], d2 = r.d[2];
    return make_float4((float)grey_u8(d0 & 255u, (d0 >> 8) & 255u, (d0 >> 16) & 255u), (float)grey_u8(d0 >> 24, d1 & 255u, (d1 >> 8) & 255u),
                       (float)grey_u8((d1 >> 16) & 255u, d1 >> 24, d2 & 255u), (float)grey_u8((d2 >> 8) & 255u, (d2 >> 16) & 255u, d2 >> 24));
}
__device__ __forceinline__ float4 quad_grey(const QuadRaw<uint16_t>& r) {     // b0 g0 | r0 b1 | g1 r1 | b2 g2 | r2 b3 | g3 r3
    const uint32_t* d = r.d;
    return make_float4((float)grey_u16(d[0] & 0xffffu, d[0] >> 16, d[1] & 0xffffu), (float)grey_u16(d[1] >> 16, d[2] & 0xffffu, d[2] >> 16),
                       (float)grey_u16(d[3] & 0xffffu, d[3] >> 16, d[4] & 0xffffu), (float)grey_u16(d[4] >> 16, d[5] & 0xffffu, d[5] >> 16));
}

struct FrameBatch { const void* const* ptrs; const void* base; size_t frame_bytes; };   // frame z: ptrs[z], or base + z * frame_bytes

template <int R, typename T>
__global__ __launch_bounds__(256) void grey_blur_stream_kernel(FrameBatch fb, size_t stride /* elements of T per row */, int w, int h,
                                                               GaussTaps taps, float* __restrict__ out, int out_stride, size_t out_plane_stride) {
    const int lane = threadIdx.x & 63, wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nq = w >> 2;
    if (wv * GS_QW >= nq) return;                                  // wave-uniform: this wave lies beyond the frame
    const int q = wv * GS_QW + lane - 1;                           // quad owned by this lane
    const int qc = min(max(q, 0), nq - 1);
    const bool writer = (lane >= 1) & (lane <= GS_QW) & (q < nq);
    const int y0 = blockIdx.y * GS_SEG;
    const T* __restrict__ src = fb.ptrs ? (const T*)fb.ptrs[blockIdx.z] : (const T*)((const char*)fb.base + blockIdx.z * fb.frame_bytes);
    float* __restrict__ dst = out + blockIdx.z * out_plane_stride + 4 * (size_t)qc;
    float k[R + 1];
#pragma unroll
    for (int j = 0; j <= R; j++) k[j] = taps.k[j];
    constexpr int NT = GS_SEG + 2 * R;
    auto load_row = [&](int t) {
        const int sy = reflect101(y0 - R + min(t, NT - 1), h);
        QuadRaw<T> r;
        __builtin_memcpy(&r, src + (size_t)sy * stride + (size_t)qc * 12, sizeof(r));     // 4 pixels x 3 channels
        return r;
    };
    QuadRaw<T> pf[GS_PF];
#pragma unroll
    for (int i = 0; i < GS_PF; i++) pf[i] = load_row(i);
    float4 ring[2 * R + 1];
#pragma unroll
    for (int i = 0; i < 2 * R + 1; i++) ring[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < NT; t++) {
        const QuadRaw<T> cur = pf[0];
#pragma unroll
        for (int i = 0; i + 1 < GS_PF; i++) pf[i] = pf[i + 1];
        pf[GS_PF - 1] = load_row(t + GS_PF);
        const float4 g = quad_grey(cur);
        float f[4 + 2 * R];
        f[R] = g.x; f[R + 1] = g.y; f[R + 2] = g.z; f[R + 3] = g.w;
        // neighbours: position -j is component 4 - j of the lane to the left, position 3 + j component j - 1 of the lane to the right
        const float lw = __shfl_up(g.w, 1, 64), lz = __shfl_up(g.z, 1, 64), rx = __shfl_down(g.x, 1, 64), ry = __shfl_down(g.y, 1, 64);
        f[R - 1] = lw; f[R + 4] = rx;
        if (R >= 2) { f[R - 2] = lz; f[R + 5] = ry; }
        if (R >= 3) { f[R - 3] = __shfl_up(g.y, 1, 64); f[R + 6] = __shfl_down(g.z, 1, 64); }
        if (q == 0) {                                              // REFLECT_101 at the left edge: -1 -> 1, -2 -> 2, -3 -> 3
            f[R - 1] = g.y;
            if (R >= 2) f[R - 2] = g.z;
            if (R >= 3) f[R - 3] = g.w;
        }
        if (q == nq - 1) {                                         // right edge: w -> w - 2, w + 1 -> w - 3, w + 2 -> w - 4
            f[R + 4] = g.z;
            if (R >= 2) f[R + 5] = g.y;
            if (R >= 3) f[R + 6] = g.x;
        }
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float sacc = k[0] * f[R + e];
#pragma unroll
            for (int j = 1; j <= R; j++) sacc += k[j] * (f[R + e - j] + f[R + e + j]);
            o[e] = sacc;
        }
#pragma unroll
        for (int i = 0; i < 2 * R; i++) ring[i] = ring[i + 1];
        ring[2 * R] = make_float4(o[0], o[1], o[2], o[3]);
        const int y = y0 + t - 2 * R;
        if (t >= 2 * R && y < h && writer) {
            float4 sacc;
            sacc.x = k[0] * ring[R].x; sacc.y = k[0] * ring[R].y; sacc.z = k[0] * ring[R].z; sacc.w = k[0] * ring[R].w;
#pragma unroll
            for (int j = 1; j <= R; j++) {
                sacc.x += k[j] * (ring[R - j].x + ring[R + j].x);
                sacc.y += k[j] * (ring[R - j].y + ring[R + j].y);
                sacc.z += k[j] * (ring[R - j].z + ring[R + j].z);
                sacc.w += k[j] * (ring[R - j].w + ring[R + j].w);
            }
            *reinterpret_cast<float4*>(dst + (size_t)y * out_stride) = sacc;
        }
    }
}

// GaussianBlur(float(grey(frame))) of n frames in one launch; hipErrorNotSupported when the streaming kernel does not
// apply (the caller then takes launch_grey_blur frame by frame): 8- or 16-bit BGR, kernel size 3 / 5 / 7, width a multiple
// of 4 and >= 8, dword-aligned rows and frames, 16-byte-aligned output rows.
hipError_t launch_grey_blur_batch(const void* const* ptrs_dev, const void* base, size_t frame_bytes, int n, int depth, int w, int h,
                                  size_t stride_bytes, int ksize, float* out, int out_stride, size_t out_plane_stride, hipStream_t s) {
    GaussTaps taps;
    if (!gaussian_taps(ksize, taps)) return hipErrorInvalidValue;
    const int r = taps.r;
    if (n <= 0) return hipSuccess;
    if ((depth != 8 && depth != 16) || r < 1 || r > 3 || w % 4 != 0 || w < 8 || stride_bytes % 4 != 0 || out_stride % 4 != 0 ||
        (reinterpret_cast<uintptr_t>(out) & 15) != 0 || out_plane_stride % 4 != 0 ||
        (!ptrs_dev && ((reinterpret_cast<uintptr_t>(base) & 3) != 0 || frame_bytes % 4 != 0)))
        return hipErrorNotSupported;
    const int waves = (w / 4 + GS_QW - 1) / GS_QW;
    dim3 grid((waves + 3) / 4, (h + GS_SEG - 1) / GS_SEG, n);
    const FrameBatch fb{ptrs_dev, base, frame_bytes};
#define STK_GBS(R, T) grey_blur_stream_kernel<R, T><<<grid, 256, 0, s>>>(fb, stride_bytes / sizeof(T), w, h, taps, out, out_stride, out_plane_stride)
    if (depth == 8) { if (r == 1) STK_GBS(1, uint8_t); else if (r == 2) STK_GBS(2, uint8_t); else STK_GBS(3, uint8_t); }
    else { if (r == 1) STK_GBS(1, uint16_t); else if (r == 2) STK_GBS(2, uint16_t); else STK_GBS(3, uint16_t); }
#undef STK_GBS
    return hipGetLastError();
}

hipError_t launch_grey_blur(const void* src, int depth, int cn, int w, int h, size_t stride_bytes, int ksize,
                            float* out, int out_stride, hipStream_t s) {
    GaussTaps taps;
    if (!gaussian_taps(ksize, taps)) return hipErrorInvalidValue;
    const int r = taps.r;
    const size_t lds_bytes = (size_t)((BT_Y + 2 * r) * (BT_X + 2 * r) + (BT_Y + 2 * r) * BT_X) * sizeof(float);
    if (depth == 8 && cn == 3 && r >= 1 && r <= 3 && stride_bytes % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0 &&
        out_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        dim3 fgrid((w + FB_X - 1) / FB_X, (h + FB_Y - 1) / FB_Y);
        if (r == 1) grey_blur_u8c3_kernel<1><<<fgrid, 256, 0, s>>>((const uint8_t*)src, stride_bytes, w, h, taps, out, out_stride);
        else if (r == 2) grey_blur_u8c3_kernel<2><<<fgrid, 256, 0, s>>>((const uint8_t*)src, stride_bytes, w, h, taps, out, out_stride);
        else grey_blur_u8c3_kernel<3><<<fgrid, 256, 0, s>>>((const uint8_t*)src, stride_bytes, w, h, taps, out, out_stride);
        return hipGetLastError();
    }
    if (depth == 16 && cn == 3 && r >= 1 && r <= 3 && stride_bytes % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0 &&
        out_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        dim3 fgrid((w + FB_X - 1) / FB_X, (h + FB_Y - 1) / FB_Y);
        const uint16_t* s16 = (const uint16_t*)src;
        if (r == 1) grey_blur_u8c3_kernel<1, uint16_t><<<fgrid, 256, 0, s>>>(s16, stride_bytes / 2, w, h, taps, out, out_stride);
        else if (r == 2) grey_blur_u8c3_kernel<2, uint16_t><<<fgrid, 256, 0, s>>>(s16, stride_bytes / 2, w, h, taps, out, out_stride);
        else grey_blur_u8c3_kernel<3, uint16_t><<<fgrid, 256, 0, s>>>(s16, stride_bytes / 2, w, h, taps, out, out_stride);
        return hipGetLastError();
    }
    dim3 grid((w + BT_X - 1) / BT_X, (h + BT_Y - 1) / BT_Y);
    if (depth == 8 && cn == 3) grey_blur_kernel<uint8_t, 3><<<grid, 256, lds_bytes, s>>>((const uint8_t*)src, stride_bytes, w, h, taps, out, out_stride);
    else if (depth == 8 && cn == 1) grey_blur_kernel<uint8_t, 1><<<grid, 256, lds_bytes, s>>>((const uint8_t*)src, stride_bytes, w, h, taps, out, out_stride);
    else if (depth == 16 && cn == 3) grey_blur_kernel<uint16_t, 3><<<grid, 256, lds_bytes, s>>>((const uint16_t*)src, stride_bytes / 2, w, h, taps, out, out_stride);
    else if (depth == 16 && cn == 1) grey_blur_kernel<uint16_t, 1><<<grid, 256, lds_bytes, s>>>((const uint16_t*)src, stride_bytes / 2, w, h, taps, out, out_stride);
    else if (depth == 32 && cn == 3) grey_blur_kernel<float, 3><<<grid, 256, lds_bytes, s>>>((const float*)src, stride_bytes / 4, w, h, taps, out, out_stride);
    else if (depth == 32 && cn == 1) grey_blur_kernel<float, 1><<<grid, 256, lds_bytes, s>>>((const float*)src, stride_bytes / 4, w, h, taps, out, out_stride);
    else if (depth == 8 && cn == 4) grey_blur_kernel<uint8_t, 4><<<grid, 256, lds_bytes, s>>>((const uint8_t*)src, stride_bytes, w, h, taps, out, out_stride);
    else if (depth == 16 && cn == 4) grey_blur_kernel<uint16_t, 4><<<grid, 256, lds_bytes, s>>>((const uint16_t*)src, stride_bytes / 2, w, h, taps, out, out_stride);
    else if (depth == 32 && cn == 4) grey_blur_kernel<float, 4><<<grid, 256, lds_bytes, s>>>((const float*)src, stride_bytes / 4, w, h, taps, out, out_stride);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---- reference planes -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ref_planes_kernel(const float* __restrict__ b, int in_stride, int w, int h,
                                                         float* __restrict__ I, float* __restrict__ gx,
                                                         float* __restrict__ gy, float* __restrict__ gxy, float* __restrict__ igg, int rs) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const float* row = b + (size_t)y * in_stride;
    const float* up = b + (size_t)reflect101(y - 1, h) * in_stride;
    const float* dn = b + (size_t)reflect101(y + 1, h) * in_stride;
    const size_t o = (size_t)y * rs + x;
    I[o] = row[x];
    const float vx = -0.5f * row[reflect101(x - 1, w)] + 0.5f * row[reflect101(x + 1, w)];
    const float vy = -0.5f * up[x] + 0.5f * dn[x];
    gx[o] = vx;
    gy[o] = vy;
    *reinterpret_cast<float2*>(gxy + 2 * o) = make_float2(vx, vy);   // interleaved copy for the row-factorised ECC pass
    igg[3 * o] = row[x]; igg[3 * o + 1] = vx; igg[3 * o + 2] = vy;     // (I, gx, gy): one LDS-DMA per row in the column pass
}

hipError_t launch_ref_planes(const float* blurred, int in_stride, int w, int h, float* I, float* gx, float* gy, float* gxy,
                             float* igg, int ref_stride, hipStream_t s) {
    dim3 grid((w + 255) / 256, h);
    ref_planes_kernel<<<grid, 256, 0, s>>>(blurred, in_stride, w, h, I, gx, gy, gxy, igg, ref_stride);
    return hipGetLastError();
}

// ---- elementwise helpers ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n,
                                                    float sc) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const size_t step = (size_t)gridDim.x * 256 * 4;
    for (; i + 3 < n; i += step) {
        float4 v = *reinterpret_cast<const float4*>(in + i);
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        *reinterpret_cast<float4*>(out + i) = v;
    }
    if (i < n && i + 3 >= n) for (size_t j = i; j < n; j++) out[j] = in[j] * sc;
}

hipError_t launch_scale(const float* in, float* out, size_t n, float sc, hipStream_t s) {
    int blocks = (int)std::min<size_t>((n / 4 + 255) / 256 + 1, 2048);
    scale_kernel<<<blocks, 256, 0, s>>>(in, out, n, sc);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void add_kernel(float* __restrict__ acc, const float* __restrict__ in, size_t n) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const size_t step = (size_t)gridDim.x * 256 * 4;
    for (; i + 3 < n; i += step) {
        float4 a = *reinterpret_cast<float4*>(acc + i);
        const float4 v = *reinterpret_cast<const float4*>(in + i);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        *reinterpret_cast<float4*>(acc + i) = a;
    }
    if (i < n && i + 3 >= n) for (size_t j = i; j < n; j++) acc[j] += in[j];
}

hipError_t launch_add(float* acc, const float* in, size_t n, hipStream_t s) {
    int blocks = (int)std::min<size_t>((n / 4 + 255) / 256 + 1, 2048);
    add_kernel<<<blocks, 256, 0, s>>>(acc, in, n);
    return hipGetLastError();
}

}  // namespace stk
