// kernels_warp_tile.hip — the fold (convert + warpPerspective / warpAffine + add, lib.rs:290-316, 780-814) with the source
// CONVERTED ONCE per tile. Round 4 priced the fold of kernels_warp.hip in issue slots (tools/valu_rates.hip): of its ~108
// slots per pixel and frame, 36 are the twelve byte -> float conversions and their x alpha — and every source pixel is
// converted four times, once per destination pixel that taps it. Here a workgroup owns a 64 x 16 destination tile; per
// frame it finds the tile's source window (the images of the tile's four corners under the frame's map: a homography with
// w > 0 maps the convex tile into the convex hull of those), converts that window to alpha-scaled f32 in LDS — (float)b *
// alpha, the very values the gather kernels form per tap — and every destination pixel then takes its 2 x 2 x 3 taps from
// LDS. Same operations on the same values in the same order per pixel and frame: bit-identical to the generic kernel.
// A frame whose window does not fit (strong rotation or zoom, a tile on the frame's rim, an unaligned source) takes, for that
// tile, the gather route of kernels_warp.hip one frame at a time; the decision is uniform per workgroup and frame.
#include "common.h"

namespace stk {

namespace {

constexpr int TL_W = 64, TL_H = 16, TL_R = 4;            // destination tile; rows per thread (wave w: rows 4 w .. 4 w + 3)
constexpr int TL_SW = 76, TL_SH = 24;                    // staged source window: columns (a multiple of 4) x rows
constexpr int TL_PITCH = TL_SW * 3;                      // floats per staged row

__device__ __forceinline__ int floor_i(float v) { int r; asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(v)); return r; }

// destination (fx, fy) -> source (X, Y): the generic kernel's operations (kernels_warp.hip). SHARED: both quotients through one
// reciprocal chain — the expansion of an IEEE f32 division without its range handling, licensed by WARPFRAME_DIV_IN_RANGE.
template <bool AFFINE, bool SHARED>
__device__ __forceinline__ void map_point(const float* __restrict__ M, float fx, float fy, float& X, float& Y) {
    X = __builtin_fmaf(M[0], fx, __builtin_fmaf(M[1], fy, M[2]));
    Y = __builtin_fmaf(M[3], fx, __builtin_fmaf(M[4], fy, M[5]));
    if constexpr (!AFFINE) {
        const float W = __builtin_fmaf(M[6], fx, __builtin_fmaf(M[7], fy, M[8]));
        if constexpr (SHARED) {
            float r = __builtin_amdgcn_rcpf(W);
            r = __builtin_fmaf(__builtin_fmaf(-W, r, 1.0f), r, r);
            float q = X * r, e = __builtin_fmaf(-W, q, X);
            q = __builtin_fmaf(e, r, q); e = __builtin_fmaf(-W, q, X); X = __builtin_fmaf(e, r, q);
            q = Y * r; e = __builtin_fmaf(-W, q, Y);
            q = __builtin_fmaf(e, r, q); e = __builtin_fmaf(-W, q, Y); Y = __builtin_fmaf(e, r, q);
        } else { X = X / W; Y = Y / W; }
    }
}

// one pixel, one frame, taps gathered from global memory with BORDER_CONSTANT handling: the generic kernel's body
template <typename T>
__device__ __forceinline__ void gather_pixel(const T* __restrict__ src, size_t stride, int sw, int sh, float X, float Y, float alpha,
                                             const float (&bv)[3], float (&sum)[3]) {
    const bool finite = (__builtin_fabsf(X) < 1e9f) & (__builtin_fabsf(Y) < 1e9f);
    const float flx = __builtin_floorf(X), fly = __builtin_floorf(Y);
    const int ix = finite ? (int)flx : -100000, iy = finite ? (int)fly : -100000;
    const float ax = finite ? X - flx : 0.0f, ay = finite ? Y - fly : 0.0f;
    const bool vx0 = (unsigned)ix < (unsigned)sw, vx1 = (unsigned)(ix + 1) < (unsigned)sw;
    const bool vy0 = (unsigned)iy < (unsigned)sh, vy1 = (unsigned)(iy + 1) < (unsigned)sh;
    const int xc0 = min(max(ix, 0), sw - 1), xc1 = min(max(ix + 1, 0), sw - 1);
    const int yc0 = min(max(iy, 0), sh - 1), yc1 = min(max(iy + 1, 0), sh - 1);
    const T* q0 = src + (size_t)yc0 * stride;
    const T* q1 = src + (size_t)yc1 * stride;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float p00 = (vx0 & vy0) ? (float)q0[xc0 * 3 + c] * alpha : bv[c];
        const float p01 = (vx1 & vy0) ? (float)q0[xc1 * 3 + c] * alpha : bv[c];
        const float p10 = (vx0 & vy1) ? (float)q1[xc0 * 3 + c] * alpha : bv[c];
        const float p11 = (vx1 & vy1) ? (float)q1[xc1 * 3 + c] * alpha : bv[c];
        const float t0 = __builtin_fmaf(ax, p01 - p00, p00), t1 = __builtin_fmaf(ax, p11 - p10, p10);
        sum[c] = sum[c] + __builtin_fmaf(ay, t1 - t0, t0);
    }
}

}  // namespace

template <typename T, bool AFFINE>
__global__ __launch_bounds__(256) void warp_accumulate_tile_kernel(WarpArgs a) {
    __shared__ __attribute__((aligned(16))) float tile[TL_SH * TL_PITCH];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * TL_W, y0 = blockIdx.y * TL_H;
    const int x = x0 + lane;
    const int xcl = min(x, a.dw - 1);
    float sum[TL_R][3];
    float fyr[TL_R];
#pragma unroll
    for (int r = 0; r < TL_R; r++) {
        const int y = y0 + wave * TL_R + r;
        fyr[r] = (float)min(y, a.dh - 1);
        const bool live = x < a.dw && y < a.dh;
        const float* ap = a.acc + (size_t)min(y, a.dh - 1) * a.acc_stride + (size_t)xcl * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) sum[r][c] = (a.accumulate && live) ? ap[c] : 0.0f;
    }
    const float fx = (float)xcl;
    const int sw = a.sw, sh = a.sh;
    const size_t stride = a.src_stride;                    // elements of T per source row
    const float alpha = a.alpha;
    const float bv[3] = {a.bv[0], a.bv[1], a.bv[2]};
    // the tile's corners in destination coordinates (clipped to the image): lanes 0 .. 3 of every wave map one each
    const float cfx = (float)((lane & 1) ? min(x0 + TL_W - 1, a.dw - 1) : x0);
    const float cfy = (float)((lane & 2) ? min(y0 + TL_H - 1, a.dh - 1) : y0);

    for (int f = 0; f < a.n_frames; f++) {
        const WarpFrame* fr = a.frames + f;
        const T* __restrict__ src = (const T*)fr->src;
        const int flags = fr->flags;
        const bool shared_div = AFFINE || (flags & WARPFRAME_DIV_IN_RANGE);   // (uniform)
        // ---- the tile's source window ----
        float Xc, Yc;
        if (shared_div) map_point<AFFINE, true>(fr->M, cfx, cfy, Xc, Yc); else map_point<AFFINE, false>(fr->M, cfx, cfy, Xc, Yc);
        const bool cfin = (__builtin_fabsf(Xc) < 1e9f) & (__builtin_fabsf(Yc) < 1e9f);
        const int cix = floor_i(Xc), ciy = floor_i(Yc);
        const bool all_fin = (__ballot(cfin) & 0xfull) == 0xfull;
        const int ix0 = __builtin_amdgcn_readlane(cix, 0), ix1 = __builtin_amdgcn_readlane(cix, 1), ix2 = __builtin_amdgcn_readlane(cix, 2), ix3 = __builtin_amdgcn_readlane(cix, 3);
        const int iy0 = __builtin_amdgcn_readlane(ciy, 0), iy1 = __builtin_amdgcn_readlane(ciy, 1), iy2 = __builtin_amdgcn_readlane(ciy, 2), iy3 = __builtin_amdgcn_readlane(ciy, 3);
        // one pixel of guard on every side: an interior pixel's f32 coordinate may round across an integer its corners' do not
        const int bx0 = (min(min(ix0, ix1), min(ix2, ix3)) - 1) & ~3;               // window origin: a multiple of 4 pixels
        const int bx1 = max(max(ix0, ix1), max(ix2, ix3)) + 2;                       // last column a tap may touch, guard included
        const int by0 = min(min(iy0, iy1), min(iy2, iy3)) - 1, by1 = max(max(iy0, iy1), max(iy2, iy3)) + 2;
        const int ngroups = (bx1 - bx0 + 4) >> 2, nrows = by1 - by0 + 1;
        // (the convex-hull argument needs w > 0 over the tile: WARPFRAME_DIV_IN_RANGE says it keeps its sign over the whole image)
        const bool fits = all_fin && shared_div && (flags & WARPFRAME_SRC_ALIGNED4) && ngroups * 4 <= TL_SW && nrows <= TL_SH &&
                          bx0 >= 0 && bx0 + ngroups * 4 <= sw && by0 >= 0 && by1 <= sh - 1;
        if (fits) {
            // ---- convert the window once: 4 pixels (12 samples) per task ----
            const int ntasks = nrows * ngroups;
            const float inv_ng = 1.0f / (float)ngroups;
            for (int task = threadIdx.x; task < ntasks; task += 256) {
                const int row = (int)(((float)task + 0.5f) * inv_ng);
                const int grp = task - row * ngroups;
                const T* p = src + (size_t)(by0 + row) * stride + (size_t)(bx0 + 4 * grp) * 3;
                float v[12];
                if constexpr (sizeof(T) == 1) {
                    uint32_t d[3];
                    __builtin_memcpy(d, __builtin_assume_aligned(p, 4), 12);
#pragma unroll
                    for (int k = 0; k < 12; k++) v[k] = (float)((d[k >> 2] >> (8 * (k & 3))) & 0xffu) * alpha;
                } else {
                    uint32_t d[6];
                    __builtin_memcpy(d, __builtin_assume_aligned(p, 4), 24);
#pragma unroll
                    for (int k = 0; k < 12; k++) v[k] = (float)((d[k >> 1] >> (16 * (k & 1))) & 0xffffu) * alpha;
                }
                float4* o = reinterpret_cast<float4*>(tile + row * TL_PITCH + grp * 12);
                o[0] = make_float4(v[0], v[1], v[2], v[3]); o[1] = make_float4(v[4], v[5], v[6], v[7]); o[2] = make_float4(v[8], v[9], v[10], v[11]);
            }
            __syncthreads();
            // ---- every destination pixel: 2 x 2 x 3 taps out of the window ----
#pragma unroll
            for (int r = 0; r < TL_R; r++) {
                float X, Y;
                map_point<AFFINE, true>(fr->M, fx, fyr[r], X, Y);
                const int ix = floor_i(X), iy = floor_i(Y);
                const float ax = __builtin_amdgcn_fractf(X), ay = __builtin_amdgcn_fractf(Y);   // (inside the image: X, Y >= 0, fract is X - floor(X) exactly)
                const float* t = tile + (iy - by0) * TL_PITCH + (ix - bx0) * 3;
                const float* b = t + TL_PITCH;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float p00 = t[c], p01 = t[3 + c], p10 = b[c], p11 = b[3 + c];
                    const float t0 = __builtin_fmaf(ax, p01 - p00, p00), t1 = __builtin_fmaf(ax, p11 - p10, p10);
                    sum[r][c] = sum[r][c] + __builtin_fmaf(ay, t1 - t0, t0);
                }
            }
            __syncthreads();                               // the window is overwritten by the next frame's
        } else {
#pragma unroll
            for (int r = 0; r < TL_R; r++) {
                float X, Y;
                if (shared_div && !AFFINE) {
                    // the gather kernels test the range per frame exactly like this (kernels_warp.hip): same quotient bits
                    map_point<AFFINE, true>(fr->M, fx, fyr[r], X, Y);
                } else map_point<AFFINE, false>(fr->M, fx, fyr[r], X, Y);
                gather_pixel<T>(src, stride, sw, sh, X, Y, alpha, bv, sum[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < TL_R; r++) {
        const int y = y0 + wave * TL_R + r;
        if (x < a.dw && y < a.dh) {
            float* ap = a.acc + (size_t)y * a.acc_stride + (size_t)x * 3;
            ap[0] = sum[r][0]; ap[1] = sum[r][1]; ap[2] = sum[r][2];
        }
    }
}

// depth 8 / 16, three channels, exact f32 coordinates, BORDER_CONSTANT (what launch_warp_accumulate's fast paths take)
hipError_t launch_warp_accumulate_tile(const WarpArgs& a, int depth, hipStream_t s) {
    dim3 grid((a.dw + TL_W - 1) / TL_W, (a.dh + TL_H - 1) / TL_H);
    if (depth == 8) {
        if (a.is_affine) warp_accumulate_tile_kernel<uint8_t, true><<<grid, 256, 0, s>>>(a);
        else warp_accumulate_tile_kernel<uint8_t, false><<<grid, 256, 0, s>>>(a);
    } else {
        if (a.is_affine) warp_accumulate_tile_kernel<uint16_t, true><<<grid, 256, 0, s>>>(a);
        else warp_accumulate_tile_kernel<uint16_t, false><<<grid, 256, 0, s>>>(a);
    }
    return hipGetLastError();
}

}  // namespace stk
