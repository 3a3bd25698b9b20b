// kernels_orb.hip — ORB::detect_and_compute (utils.rs:174-183), scale_image's resize(INTER_AREA)
// (utils.rs:186-214) and the brute-force Hamming 2-NN matcher (lib.rs:208-219; row C1) as gfx950 kernels: the production
// path. All integer / byte work on 8-bit pyramids, integer-VALU-bound, no MFMA.
//
//   resize_tables / resize_exact_direct   INTER_LINEAR_EXACT pyramid step from per-geometry tables, no LDS, 4 x 4 pixels per thread
//   resize_area                           INTER_AREA (8-bit and f32 greys), OpenCV's table order
//   fast_nms_tiled(_all)                  FAST-9/16 pre-test + strength + 3x3 NMS + border + histogram + collect, 128 x 32 tiles of ALL levels in one launch
//   fast_threshold / fast_pick / fast_describe / orb_cull   retainBest threshold, short list, Harris + IC moments (a wave per corner), device cull
//   gauss7_fused                          7x7 sigma-2 blur of a whole level (only behind orb_patch_blur = 0)
//   brief_patch / brief                   blur of the sampled window + rotated BRIEF / rBRIEF on a blurred level
//   knn2_hamming                          popcount(xor) over 32 B, two best train rows per query, ties keep the lower index
// The plain per-pixel forms the tiled kernels replaced (and fall back to for tiny or unaligned levels) live in
// kernels_orb_small.hip; the device helpers both files use in orb_device.h.
#include "orb_device.h"

namespace stk {

// ---- pyramid ------------------------------------------------------------------------------------
// Round 3: the pyramid step without a tile. The tiled kernel (kernels_orb_small.hip) is neither VALU- nor HBM-bound (SQ counters: 6 waves per
// SIMD that each live 7 us): table computation in f64 by 160 threads while the others wait, a staged footprint behind a
// barrier, four LDS byte reads per pixel. A first rewrite that kept the tile but read its tables from memory cut the VALU
// count by 40 % and the time by nothing. This one has no LDS and no barrier: a thread makes 4 adjacent pixels of 4 rows; the
// step's tables — (offset | weight << 16) of every column and row, a function of the level sizes only, computed once per
// geometry by resize_tables_kernel with the same f64 lin_coef — give it the first column's source offset; the six source
// bytes its four columns touch in a row (3 steps of <= 1.3 px, + the right-hand tap) come with ONE aligned 12-byte load + v_alignbyte,
// a 64-bit shift per pixel extracts the two taps. A tap the reference clamps (last column / row) has weight 0 or is
// clamped here too (rows), so whatever byte sits behind the row's end is never used. Same integer arithmetic, bit-identical
// (test_orb_table_driven_pyramid_equals_the_per_tile_tables); the final min(., 255) is dropped: the weights of a pixel
// sum to 65536, so (v + 32768) >> 16 <= 255.
__global__ void resize_tables_kernel(int sw, int sh, int dw, int dh, double scale_x, double scale_y, int* __restrict__ xt, int* __restrict__ yt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int o, c0, c1;
    if (i < ((dw + 3) & ~3)) { lin_coef(min(i, dw - 1), sw, scale_x, o, c0, c1); xt[i] = o | (c1 << 16); }    // padded to whole quads
    if (i < ((dh + 3) & ~3)) { lin_coef(min(i, dh - 1), sh, scale_y, o, c0, c1); yt[i] = o | (c1 << 16); }
}

// eight bytes at any address: the three aligned dwords that hold them (ONE 12-byte load: an unaligned 8-byte load costs the
// L1 tag pipeline ~46 look-ups per wave instruction, measured — TCP_TOTAL_CACHE_ACCESSES / SQ_INSTS_VMEM_RD) and two v_alignbyte
struct Dw3 { uint32_t a, b, c; };
__device__ __forceinline__ unsigned long long load8_at(const uint8_t* p) {
    const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3);
    const Dw3 t = *reinterpret_cast<const Dw3*>(p - sh);          // (pointer arithmetic, not an integer round trip: stays a global load)
    const uint32_t lo = __builtin_amdgcn_alignbyte(t.b, t.a, sh), hi = __builtin_amdgcn_alignbyte(t.c, t.b, sh);
    return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(256) void resize_exact_direct_kernel(const uint8_t* __restrict__ src, int sw, int sh,
                                                                  uint8_t* __restrict__ dst, int dw, int dh,
                                                                  const int4* __restrict__ xt, const int4* __restrict__ yt, size_t frame_stride) {
    const int xq = blockIdx.x * 64 + (threadIdx.x & 63), yq = blockIdx.y * 4 + (threadIdx.x >> 6);     // quad column, quad row
    const int x = 4 * xq, y = 4 * yq;
    if (x >= dw || y >= dh) return;
    src += blockIdx.z * frame_stride; dst += blockIdx.z * frame_stride;
    const int4 xv = xt[xq], yv = yt[yq];
    const int xo[4] = {xv.x & 0xffff, xv.y & 0xffff, xv.z & 0xffff, xv.w & 0xffff};
    const uint32_t xc[4] = {(uint32_t)xv.x >> 16, (uint32_t)xv.y >> 16, (uint32_t)xv.z >> 16, (uint32_t)xv.w >> 16};
    const int yo[4] = {yv.x & 0xffff, yv.y & 0xffff, yv.z & 0xffff, yv.w & 0xffff};
    const uint32_t yc[4] = {(uint32_t)yv.x >> 16, (uint32_t)yv.y >> 16, (uint32_t)yv.z >> 16, (uint32_t)yv.w >> 16};
    const int base = xo[0];
    const uint32_t sft[4] = {0u, (uint32_t)(xo[1] - base) * 8u, (uint32_t)(xo[2] - base) * 8u, (uint32_t)(xo[3] - base) * 8u};
    const bool aligned = (dw & 3) == 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (y + k >= dh) break;
        const uint8_t* r0 = src + (size_t)yo[k] * sw + base;
        const uint8_t* r1 = src + (size_t)min(yo[k] + 1, sh - 1) * sw + base;
        const unsigned long long w0 = load8_at(r0), w1 = load8_at(r1);
        const uint32_t cy1 = yc[k], cy0 = 256u - cy1;
        uint32_t out = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t cx1 = xc[e], cx0 = 256u - cx1;
            const uint32_t t0 = (uint32_t)(w0 >> sft[e]), t1 = (uint32_t)(w1 >> sft[e]);
            const uint32_t h0 = cx0 * (t0 & 255u) + cx1 * ((t0 >> 8) & 255u);
            const uint32_t h1 = cx0 * (t1 & 255u) + cx1 * ((t1 >> 8) & 255u);
            const uint32_t v = cy0 * h0 + cy1 * h1 + (1u << 15);
            out |= (v >> 16) << (8 * e);
        }
        uint8_t* op = dst + (size_t)(y + k) * dw + x;
        if (aligned && x + 3 < dw) *reinterpret_cast<uint32_t*>(op) = out;
        else {
            op[0] = (uint8_t)out;
            if (x + 1 < dw) op[1] = (uint8_t)(out >> 8);
            if (x + 2 < dw) op[2] = (uint8_t)(out >> 16);
            if (x + 3 < dw) op[3] = (uint8_t)(out >> 24);
        }
    }
}

// tables of one pyramid step at `tab` (16-byte aligned): ((dw + 3) & ~3) column entries, then ((dh + 3) & ~3) row entries
size_t resize_tables_ints(int dw, int dh) { return (size_t)((dw + 3) & ~3) + (size_t)((dh + 3) & ~3); }
hipError_t launch_resize_tables(int sw, int sh, int dw, int dh, int* tab, hipStream_t s) {
    const double sx = 1.0 / ((double)dw / sw), sy = 1.0 / ((double)dh / sh);
    const int n = std::max((dw + 3) & ~3, (dh + 3) & ~3);
    resize_tables_kernel<<<(n + 255) / 256, 256, 0, s>>>(sw, sh, dw, dh, sx, sy, tab, tab + ((dw + 3) & ~3));
    return hipGetLastError();
}

hipError_t launch_resize_exact(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, hipStream_t s,
                               int n_frames, size_t frame_stride, const int* tab) {
    if (tab) {
        // the 12-byte window of a row may run past the row's end by up to 11 bytes (never used): the caller's buffer must have that
        // slack behind the last row (the pyramid workspace has 64 bytes); offsets and weights share a 32-bit table entry
        const double sx = 1.0 / ((double)dw / sw), sy = 1.0 / ((double)dh / sh);
        if (sw >= 2 && sh >= 2 && sw < 65536 && sh < 65536 && sx <= 1.3 && sy <= 1.3 && (reinterpret_cast<uintptr_t>(tab) & 15) == 0) {
            dim3 grid(((dw + 3) / 4 + 63) / 64, ((dh + 3) / 4 + 3) / 4, n_frames);
            resize_exact_direct_kernel<<<grid, 256, 0, s>>>(src, sw, sh, dst, dw, dh, reinterpret_cast<const int4*>(tab),
                                                            reinterpret_cast<const int4*>(tab + ((dw + 3) & ~3)), frame_stride);
            return hipGetLastError();
        }
    }
    return launch_resize_exact_plain(src, sw, sh, dst, dw, dh, s, n_frames, frame_stride);
}

// ---- scale_image: resize(INTER_AREA) of a grey image, 8-bit or f32 (utils.rs:186-214) -------------------
// One thread per destination pixel; it walks the fractional-coverage cells of its source rectangle in the
// order of OpenCV's computeResizeAreaTab tables (partial left cell, full cells, partial right cell; rows
// combined as sum = beta0*buf0, sum += beta_k*buf_k), so the f32 result is the same bit pattern.
struct AreaSpan { int s1, s2; float a_first, a_full, a_last; bool has_first, has_last; };

__device__ __forceinline__ AreaSpan area_span(int d, int ssize, double scale) {
    AreaSpan sp;
    const double f1 = d * scale, f2 = f1 + scale;
    const double cell = fmin(scale, (double)ssize - f1);
    int s1 = (int)ceil(f1), s2 = (int)floor(f2);
    s2 = min(s2, ssize - 1);
    s1 = min(s1, s2);
    sp.s1 = s1; sp.s2 = s2;
    sp.has_first = (s1 - f1) > 1e-3;
    sp.a_first = (float)((s1 - f1) / cell);
    sp.a_full = (float)(1.0 / cell);
    sp.has_last = (f2 - s2) > 1e-3;
    sp.a_last = (float)(fmin(fmin(f2 - s2, 1.0), cell) / cell);
    return sp;
}

// T = uint8_t: WT = int for integer ratios, cvRound on store; T = float (a 32FC1 grey, e.g. a float TIFF stack): the f32 sum as it is.
// (16-bit greys never get here: findTransformECC takes 8UC1 / 32FC1 only and ORB 8UC1 only, the reference fails on them first.)
template <typename T>
__global__ __launch_bounds__(256) void resize_area_kernel(const T* __restrict__ src, int sw, int sh,
                                                          T* __restrict__ dst, int dw, int dh, double scale_x,
                                                          double scale_y, int isx, int isy) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    constexpr bool U8 = sizeof(T) == 1;
    auto store = [&](float v) {
        if constexpr (U8) dst[(size_t)y * dw + x] = (uint8_t)min(max((int)__builtin_rintf(v), 0), 255);
        else dst[(size_t)y * dw + x] = v;
    };
    if (isx > 0) {   // integer ratios: resizeAreaFast_ [OCV-RECALL: imgproc/src/resize.cpp]
        const T* S = src + (size_t)(y * isy) * sw + (size_t)x * isx;
        if (isx == 2 && isy == 2) {
            // ResizeAreaFastVec's 2 x 2 special case: 8-bit (a + b + c + d + 2) >> 2 — half rounds UP, unlike cvRound; f32: the
            // vector form (a + b) + (c + d), times 0.25 (OpenCV's scalar tail of a row, w mod the vector width, adds in the
            // order ((a + b) + c) + d instead: that last-bit difference on a few right-hand columns is not reproduced)
            if constexpr (U8) dst[(size_t)y * dw + x] = (uint8_t)(((int)S[0] + S[1] + S[sw] + S[sw + 1] + 2) >> 2);
            else dst[(size_t)y * dw + x] = ((S[0] + S[1]) + (S[sw] + S[sw + 1])) * 0.25f;
            return;
        }
        const float sc = 1.f / (float)(isx * isy);
        if constexpr (U8) {
            int sum = 0;
            for (int j = 0; j < isy; j++)
                for (int i = 0; i < isx; i++) sum += S[(size_t)j * sw + i];
            store((float)sum * sc);
        } else {
            // sum += S[ofs[k]] + S[ofs[k + 1]] + S[ofs[k + 2]] + S[ofs[k + 3]] over the cell in row-major order, then one by one
            const int area = isx * isy;
            auto at = [&](int k) { const int j = k / isx; return S[(size_t)j * sw + (k - j * isx)]; };
            float sum = 0.f;
            int k = 0;
            for (; k <= area - 4; k += 4) sum += ((at(k) + at(k + 1)) + at(k + 2)) + at(k + 3);
            for (; k < area; k++) sum += at(k);
            store(sum * sc);
        }
        return;
    }
    const AreaSpan sx = area_span(x, sw, scale_x), sy = area_span(y, sh, scale_y);
    float sum = 0.f;
    bool first_row = true;
    auto row = [&](int yy, float beta) {
        const T* S = src + (size_t)yy * sw;
        float buf = 0.f;
        if (sx.has_first) buf += (float)S[sx.s1 - 1] * sx.a_first;
        for (int xx = sx.s1; xx < sx.s2; xx++) buf += (float)S[xx] * sx.a_full;
        if (sx.has_last) buf += (float)S[sx.s2] * sx.a_last;
        sum = first_row ? beta * buf : sum + beta * buf;
        first_row = false;
    };
    if (sy.has_first) row(sy.s1 - 1, sy.a_first);
    for (int yy = sy.s1; yy < sy.s2; yy++) row(yy, sy.a_full);
    if (sy.has_last) row(sy.s2, sy.a_last);
    store(sum);
}

// resize(INTER_AREA) when the image GROWS in a direction (scale_image makes the SMALLER dimension equal to scale_down, and the
// reference only checks scale_down against the WIDTH — lib.rs:377, 876 — so a landscape stack with height < scale_down < width is
// enlarged): "true area interpolation is only implemented for scale >= 1 in both directions; in other cases it is emulated using
// some variant of bilinear interpolation" [OCV-RECALL: imgproc/src/resize.cpp, resize() with area_mode, resizeGeneric_,
// HResizeLinear / VResizeLinear]: sx = floor(dx * scale), fx = (float)((dx + 1) - (sx + 1) * inv_scale), fx <= 0 ? 0 : fx - floor(fx)
// (an integer enlargement replicates pixels); 8 bit: weights cvRound(w * 2048) as shorts, a row pass in int, then
// ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2; f32: r = s0 * a0 + s1 * a1, d = r0 * b0 + r1 * b1. Columns whose
// right-hand tap would leave the row (dx >= xmax) take the single pixel at full weight.
__device__ __forceinline__ void area_up_coef(int d, int ssize, double scale, double inv_scale, int& s0, float& f, bool& single) {
    int sx = (int)floor((double)d * scale);
    float fx = (float)((double)(d + 1) - (double)(sx + 1) * inv_scale);
    fx = fx <= 0.f ? 0.f : fx - floorf(fx);
    single = sx + 1 >= ssize;
    if (sx >= ssize - 1) { fx = 0.f; sx = ssize - 1; }
    s0 = sx; f = fx;
}
template <typename T>
__global__ __launch_bounds__(256) void resize_area_up_kernel(const T* __restrict__ src, int sw, int sh, T* __restrict__ dst, int dw, int dh,
                                                             double scale_x, double scale_y, double inv_x, double inv_y) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    int sx; float fx; bool single_x;
    area_up_coef(x, sw, scale_x, inv_x, sx, fx, single_x);
    // rows: yofs / beta are filled without the single-tap rule, the row pointers are merely clipped to the image
    const int sy = (int)floor((double)y * scale_y);
    float fy = (float)((double)(y + 1) - (double)(sy + 1) * inv_y);
    fy = fy <= 0.f ? 0.f : fy - floorf(fy);
    const int y0 = min(max(sy, 0), sh - 1), y1 = min(max(sy + 1, 0), sh - 1);
    const T* r0 = src + (size_t)y0 * sw;
    const T* r1 = src + (size_t)y1 * sw;
    const int x1 = min(sx + 1, sw - 1);
    if constexpr (sizeof(T) == 1) {
        const int a0 = (int)(short)__builtin_rintf((1.f - fx) * 2048.f), a1 = (int)(short)__builtin_rintf(fx * 2048.f);
        const int b0 = (int)(short)__builtin_rintf((1.f - fy) * 2048.f), b1 = (int)(short)__builtin_rintf(fy * 2048.f);
        const int h0 = single_x ? (int)r0[sx] * 2048 : (int)r0[sx] * a0 + (int)r0[x1] * a1;
        const int h1 = single_x ? (int)r1[sx] * 2048 : (int)r1[sx] * a0 + (int)r1[x1] * a1;
        dst[(size_t)y * dw + x] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
    } else {
        const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
        const float h0 = single_x ? r0[sx] * 1.f : r0[sx] * a0 + r0[x1] * a1;
        const float h1 = single_x ? r1[sx] * 1.f : r1[sx] * a0 + r1[x1] * a1;
        dst[(size_t)y * dw + x] = h0 * b0 + h1 * b1;
    }
}

hipError_t launch_resize_area(const void* src, int depth, int sw, int sh, void* dst, int dw, int dh, hipStream_t s) {
    if (depth != 8 && depth != 32) return hipErrorInvalidValue;
    dim3 grid((dw + 63) / 64, (dh + 3) / 4);
    const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
    const double scale_x = 1.0 / inv_x, scale_y = 1.0 / inv_y;
    if (!(scale_x >= 1.0 && scale_y >= 1.0)) {
        if (depth == 8) resize_area_up_kernel<uint8_t><<<grid, 256, 0, s>>>((const uint8_t*)src, sw, sh, (uint8_t*)dst, dw, dh, scale_x, scale_y, inv_x, inv_y);
        else resize_area_up_kernel<float><<<grid, 256, 0, s>>>((const float*)src, sw, sh, (float*)dst, dw, dh, scale_x, scale_y, inv_x, inv_y);
        return hipGetLastError();
    }
    const int ix = (int)std::lrint(scale_x), iy = (int)std::lrint(scale_y);
    const bool fast = std::fabs(scale_x - ix) < 2.220446049250313e-16 && std::fabs(scale_y - iy) < 2.220446049250313e-16;
    if (depth == 8) resize_area_kernel<uint8_t><<<grid, 256, 0, s>>>((const uint8_t*)src, sw, sh, (uint8_t*)dst, dw, dh, scale_x, scale_y, fast ? ix : 0, fast ? iy : 0);
    else resize_area_kernel<float><<<grid, 256, 0, s>>>((const float*)src, sw, sh, (float*)dst, dw, dh, scale_x, scale_y, fast ? ix : 0, fast ? iy : 0);
    return hipGetLastError();
}
hipError_t launch_resize_area_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, hipStream_t s) {
    return launch_resize_area(src, 8, sw, sh, dst, dw, dh, s);
}

// ---- FAST -----------------------------------------------------------------------------------------
// Tiled FAST + non-maximum suppression: a 128 x 32 pixel tile in LDS (4 px halo: 1 for the 3x3 NMS neighbourhood
// + 3 for the ring) and three scoring passes with workgroup-level compaction, so that the expensive steps run on densely
// populated wavefronts instead of on every wavefront that contains one candidate:
//   pass 1  every pixel of the 130 x 34 score region: the 4-point compass pre-test              -> list A (LDS)
//   pass 2  list A: brighter / darker ring masks, 9-contiguous-bits test                           -> list B (corners)
//   pass 3  list B: corner strength (min / max over the sixteen 9-arcs) into an LDS score tile
//   pass 4  the 128 x 32 interior: strict 3x3 maximum, runByImageBorder(edge), score histogram, candidate list
// The score map never goes to memory. Same candidates and histogram as fast_score_kernel + fast_nms_kernel (kept for
// tiny levels); the candidate ORDER differs (atomics), which the later stages do not depend on.
constexpr int FT_X = 128, FT_Y = 32, FT_H = 4;
constexpr int FT_TW = FT_X + 2 * FT_H;               // 136 bytes per image-tile row
constexpr int FT_TH = FT_Y + 2 * FT_H;               // 40 rows
constexpr int FT_SW = FT_X + 4, FT_SH = FT_Y + 2;    // score region 130 x 34, stored with a row stride of 132

typedef short s16x2 __attribute__((ext_vector_type(2)));
#ifndef STK_FAST_SWAR
#define STK_FAST_SWAR 1
#endif
__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b) {       // v_pk_sub_i16
    return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_add_i16(uint32_t a, uint32_t b) {       // v_pk_add_i16
    return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ int mbcnt64(unsigned long long mask) {              // set bits of `mask` below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

__device__ __forceinline__ bool has_arc9(uint32_t m) {          // 16-bit circular mask: 9 contiguous ones?
    const uint32_t mm = m | (m << 16);                           // unrolled circle: bit k + 16 == bit k
    const uint32_t a = mm & (mm >> 1), b = a & (a >> 2), c = b & (b >> 4);
    return ((c & (mm >> 8)) & 0xffffu) != 0;
}

#ifndef STK_FAST_PRETEST
#define STK_FAST_PRETEST 1
#endif
#ifdef STK_FAST_TIMING
__device__ unsigned long long g_fast_dbg[16];
#define FAST_TICK(i) do { if (threadIdx.x == 0 && w > 1900 && tile_x == 7 && tile_y == 14 && blockIdx.z == 1) g_fast_dbg[i] = wall_clock64(); } while (0)
extern "C" void stk_debug_fast_timing(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fast_dbg), sizeof(g_fast_dbg)); }
#else
#define FAST_TICK(i) do { } while (0)
#endif
struct __attribute__((aligned(4))) U4a4 { uint32_t x, y, z, w; };   // 16 bytes at a 4-byte aligned address: one global_load_dwordx4
__device__ __forceinline__ void fast_nms_tiled_body(const uint8_t* __restrict__ img, int w, int h, int thr, int edge,
                                                    OrbLevelState* st, OrbCandidate* cand, int cap, OrbBatch bs, int tile_x, int tile_y) {
    __shared__ __attribute__((aligned(16))) uint8_t T[FT_TH * FT_TW];
    __shared__ __attribute__((aligned(16))) uint8_t S[FT_SH * FT_SW];
    __shared__ unsigned short listA[FT_SH * FT_SW], listB[FT_SH * FT_SW];
    __shared__ int nA, nB;
    img += blockIdx.z * bs.pyr; st += blockIdx.z * bs.states; cand += blockIdx.z * bs.cand;
    const int x0 = tile_x * FT_X, y0 = tile_y * FT_Y;
    const int tid = threadIdx.x;
    FAST_TICK(0);
    if (tid == 0) { nA = 0; nB = 0; }
    {
        // The 136 x 40 byte tile in 16-byte pieces (9 per row, the last one half used): a piece is the five aligned dwords
        // that hold it — one 16-byte and one 4-byte load — shifted into place by four v_alignbyte. Round 3: dword by dword
        // (two aligned loads + a shift per dword, six per thread) the tile load was 210 of the kernel's 1 019 VALU instructions
        // per wave, and this kernel is bound by exactly that count (rocprofv3 SQ counters with the later passes cut off:
        // tile 210, compass pass 575, ring masks 179, strength 45, NMS 10). All global loads of a thread are issued before
        // its first LDS store: one memory round trip per tile.
        constexpr int PCS = 9, NIT = (FT_TH * PCS + 255) / 256;          // 360 pieces, 2 per thread
        uint4 v[NIT];
#pragma unroll
        for (int k = 0; k < NIT; k++) {
            const int i = min(tid + 256 * k, FT_TH * PCS - 1);
            const int ty = i / PCS, d = i - ty * PCS;
            const int sy = min(max(y0 - FT_H + ty, 0), h - 1);        // values outside the image are never used
            const int sx0 = x0 - FT_H + 16 * d;
            const uint8_t* row = img + (size_t)sy * w;
            if (sx0 >= 0 && sx0 + 15 < w) {
                const uint8_t* p = row + sx0;
                const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3);
                const uint32_t* q = reinterpret_cast<const uint32_t*>(p - sh);
                const U4a4 a = *reinterpret_cast<const U4a4*>(q);
                const uint32_t e = q[4];                               // (<= 3 bytes behind the piece: the next row, or the pyramid's slack)
                v[k] = make_uint4(__builtin_amdgcn_alignbyte(a.y, a.x, sh), __builtin_amdgcn_alignbyte(a.z, a.y, sh),
                                  __builtin_amdgcn_alignbyte(a.w, a.z, sh), __builtin_amdgcn_alignbyte(e, a.w, sh));
            } else {
                uint32_t b[4] = {0, 0, 0, 0};
#pragma unroll
                for (int e = 0; e < 16; e++) b[e >> 2] |= (uint32_t)row[min(max(sx0 + e, 0), w - 1)] << (8 * (e & 3));
                v[k] = make_uint4(b[0], b[1], b[2], b[3]);
            }
        }
        static_assert((FT_SH * FT_SW) % 8 == 0, "the score tile is cleared in 16- and 8-byte pieces");
        for (int i = tid; i < FT_SH * FT_SW / 16; i += 256) reinterpret_cast<uint4*>(S)[i] = make_uint4(0, 0, 0, 0);
        if (tid == 0 && (FT_SH * FT_SW) % 16) reinterpret_cast<uint2*>(S + (FT_SH * FT_SW / 16) * 16)[0] = make_uint2(0, 0);
#pragma unroll
        for (int k = 0; k < NIT; k++) {
            const int i = tid + 256 * k;
            if (i < FT_TH * PCS) {
                const int ty = i / PCS, d = i - ty * PCS;
                uint2* t = reinterpret_cast<uint2*>(T + ty * FT_TW + 16 * d);       // rows are 8-byte aligned (136 = 17 x 8)
                t[0] = make_uint2(v[k].x, v[k].y);
                if (d < PCS - 1) t[1] = make_uint2(v[k].z, v[k].w);
            }
        }
    }
    __syncthreads();
    FAST_TICK(1);
    // score-region pixel id = sy * FT_SW + sx, sx in [0, 130), sy in [0, 34); image pixel (x0 - 1 + sx, y0 - 1 + sy);
    // its byte in T is at row sy + 3, column sx + 3
    // pass 1: compass pre-test, FOUR pixels per lane. A lane takes one aligned dword of a tile row (4 centre pixels), the
    // dwords 3 rows above / below it and its two row neighbours (east / west taps by v_alignbyte), unpacks bytes to 16-bit
    // pairs (v_perm) and does the eight threshold tests per pixel with packed 16-bit subtractions whose SIGN bits are the
    // comparison results: centre - tap > thr  <=>  thr - (centre - tap) < 0, centre - tap < -thr  <=>  (centre - tap) + thr < 0.
    // "At least two of the four compass taps" is then pure bit logic on the sign bits. Pixels that pass are appended to
    // list A with one LDS atomic per wavefront and step (ballot + mbcnt), not one per pixel.
    {
        constexpr int ROW_DW = FT_TW / 4;                               // 34 dwords per tile row
        const uint32_t T2 = (uint32_t)thr | ((uint32_t)thr << 16);
        const int lane = tid & 63;
        // a tile whose 130 x 34 score region lies inside [3, w - 3) x [3, h - 3): the validity mask depends on dq alone
        const bool interior = x0 - 1 >= 3 && x0 + FT_X + 1 <= w - 3 && y0 - 1 >= 3 && y0 + FT_Y + 1 <= h - 3;
        for (int i0 = tid - lane; i0 < FT_SH * ROW_DW; i0 += 256) {      // wave-uniform trip count
            const int i = i0 + lane;
            const bool act = i < FT_SH * ROW_DW;
            const int sy = act ? i / ROW_DW : 0, dq = act ? i - sy * ROW_DW : 0;
            const uint32_t* R = reinterpret_cast<const uint32_t*>(T + (sy + 3) * FT_TW);
            const uint32_t c0 = R[dq];
            // (dq - 1 = -1 and dq + 1 = ROW_DW read the neighbouring tile rows' edge dwords — rows 3 .. 36 of 40, inside T — and feed
            // invalid pixels only: no clamps)
            const uint32_t cm = R[dq - 1], cp = R[dq + 1];
            const uint32_t nn = reinterpret_cast<const uint32_t*>(T + sy * FT_TW)[dq];
            const uint32_t ss = reinterpret_cast<const uint32_t*>(T + (sy + 6) * FT_TW)[dq];
            const uint32_t ee = __builtin_amdgcn_alignbyte(cp, c0, 3);   // columns +3: bytes c0[3] cp[0] cp[1] cp[2]
            const uint32_t ww = __builtin_amdgcn_alignbyte(c0, cm, 1);   // columns -3: bytes cm[1] cm[2] cm[3] c0[0]
            // A 9-pixel arc of the 16-pixel ring contains one pixel of every opposite pair: N or S, and E or W. So a corner
            // needs (N or S) AND (E or W) beyond the threshold on the same side: a necessary condition that is tighter than
            // "two of the four" (which also admits N+S or E+W alone) and takes 3 instead of 7 logic operations per side. The
            // exact ring test of pass 2 decides either way: same corners, same bits.
            // centre - tap > thr  <=>  tap < centre - thr: the sign of tap - lo; centre - tap < -thr  <=>  the sign of hi - tap
            // (16-bit lanes: all operands are within [-255, 510]). One subtraction per tap and side.
#if STK_FAST_SWAR
            // Round 4: the same eight tests per pixel on PLAIN 32-bit adds. A packed 16-bit subtraction, a v_perm unpack and
            // a v_alignbyte each occupy the SIMD twice as long as a 32-bit add / and / shift (tools/valu_rates.hip), and this
            // pass is the largest part of a kernel bound by vector issue. Pixels 0, 2 of the dword (even bytes) and pixels
            // 1, 3 (odd bytes) go into the two 16-bit halves of a register by and / shift; every operand gets a bias of 1024
            // per half so that no difference goes negative and a 32-bit add never carries from one half into the other:
            //   tap < centre - thr  <=>  tap + (1024 + thr - centre) < 1024  <=>  bit 10 of the half is CLEAR   (dark side)
            //   tap > centre + thr  <=>  (1024 + thr + centre) - tap < 1024  <=>  bit 10 CLEAR                   (bright side)
            // (halves stay within [769, 1534 + thr] for thr <= 255). The corner condition on the inverted bits:
            //   (d0 | d2) & (d1 | d3) | (b0 | b2) & (b1 | b3)  =  ~( ((~d0 & ~d2) | (~d1 & ~d3)) & ((~b0 & ~b2) | (~b1 & ~b3)) ).
            const uint32_t BIAS = 0x04000400u, M8 = 0x00ff00ffu;
            const uint32_t taps[4] = {nn, ee, ss, ww};
            uint32_t r[2];
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const uint32_t C = (half ? c0 >> 8 : c0) & M8;
                const uint32_t A = (BIAS + T2) - C, B = (BIAS + T2) + C;
                uint32_t nd[4], nb[4];                                   // bit 10 / 26 set: the test FAILS for that pixel
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const uint32_t t = (half ? taps[d] >> 8 : taps[d]) & M8;
                    nd[d] = t + A;
                    nb[d] = B - t;
                }
                r[half] = ~(((nd[0] & nd[2]) | (nd[1] & nd[3])) & ((nb[0] & nb[2]) | (nb[1] & nb[3])));
            }
            // bit 10 of r[0]: pixel 0, bit 26: pixel 2; r[1]: pixels 1 and 3 -> bits 0 .. 3
            uint32_t m = ((r[0] >> 10) & 1u) | ((r[1] >> 9) & 2u) | ((r[0] >> 24) & 4u) | ((r[1] >> 23) & 8u);
#else
            uint32_t r[2];
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const uint32_t sel = half ? 0x0c030c02u : 0x0c010c00u;   // bytes (2, 3) or (0, 1) into the low bytes of two 16-bit lanes
                const uint32_t C = __builtin_amdgcn_perm(0u, c0, sel);
                const uint32_t lo = pk_sub_i16(C, T2), hi = pk_add_i16(C, T2);
                uint32_t dk[4], br[4];
                const uint32_t taps[4] = {nn, ee, ss, ww};
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const uint32_t t = __builtin_amdgcn_perm(0u, taps[d], sel);
                    dk[d] = pk_sub_i16(t, lo);                           // sign set: tap < centre - thr (centre darker-side test passes)
                    br[d] = pk_sub_i16(hi, t);                           // sign set: tap > centre + thr
                }
#if STK_FAST_PRETEST == 0
                r[half] = (((dk[0] | dk[1]) & (dk[2] | dk[3])) | (dk[0] & dk[1]) | (dk[2] & dk[3])) |
                          (((br[0] | br[1]) & (br[2] | br[3])) | (br[0] & br[1]) | (br[2] & br[3]));
#else
                r[half] = ((dk[0] | dk[2]) & (dk[1] | dk[3])) | ((br[0] | br[2]) & (br[1] | br[3]));
#endif
            }
            // sign bits of r[0] (pixels 0, 1) and r[1] (pixels 2, 3) -> bits 0 .. 3
            const uint32_t sg = ((r[0] & 0x80008000u) >> 15) | ((r[1] & 0x80008000u) >> 13);      // bits 0, 16 | 2, 18
            uint32_t m = (sg | (sg >> 15)) & 0xfu;
#endif
            // pixel k of this dword: sx = 4 dq - 3 + k in [0, FT_X + 2), image x = x0 - 4 + 4 dq + k in [3, w - 3), y likewise
            uint32_t vm;
            if (interior) vm = act ? (dq == 0 ? 0x8u : dq == ROW_DW - 1 ? 0x1u : 0xfu) : 0u;          // wave-uniform branch
            else {
                const int xq = x0 - 4 + 4 * dq, y = y0 - 1 + sy;
                const int klo = max(max(3 - 4 * dq, 3 - xq), 0), khi = min(min(FT_X + 5 - 4 * dq, w - 3 - xq), 4);
                vm = (act && y >= 3 && y < h - 3 && khi > klo) ? ((1u << khi) - 1u) & ~((1u << klo) - 1u) : 0u;
            }
            m &= vm;
            const unsigned long long b0 = __ballot(m & 1u), b1 = __ballot(m & 2u), b2 = __ballot(m & 4u), b3 = __ballot(m & 8u);
            const int n0 = __popcll(b0), n1 = __popcll(b1), n2 = __popcll(b2), n3 = __popcll(b3);
            if (n0 + n1 + n2 + n3) {                                      // wave-uniform
                int base = 0;
                if (lane == 0) base = atomicAdd(&nA, n0 + n1 + n2 + n3);
                base = __builtin_amdgcn_readfirstlane(base);
                const int id = sy * FT_SW + 4 * dq - 3;
                if (m & 1u) listA[base + mbcnt64(b0)] = (unsigned short)id;
                if (m & 2u) listA[base + n0 + mbcnt64(b1)] = (unsigned short)(id + 1);
                if (m & 4u) listA[base + n0 + n1 + mbcnt64(b2)] = (unsigned short)(id + 2);
                if (m & 8u) listA[base + n0 + n1 + n2 + mbcnt64(b3)] = (unsigned short)(id + 3);
            }
        }
    }
    __syncthreads();
    FAST_TICK(2);
    // pass 2: ring masks, 9 contiguous
    const int cntA = nA;
    for (int i = tid; i < cntA; i += 256) {
        const int id = listA[i], sy = id / FT_SW, sx = id - sy * FT_SW;
        const uint8_t* p = T + (sy + 3) * FT_TW + sx + 3;
#if STK_FAST_SWAR
        // the sixteen ring tests on plain adds, two ring pixels (k and k + 8) per register, biased like pass 1: bit 10 / 26 of
        // ring + (1024 + thr - centre) is CLEAR where centre - ring > thr, of (1024 + thr + centre) - ring where centre - ring < -thr
        const uint32_t v = p[0];
        const uint32_t A2 = (1024u + (uint32_t)thr - v) * 0x00010001u, B2 = (1024u + (uint32_t)thr + v) * 0x00010001u;
        const int ro[16] = {3 * FT_TW, 3 * FT_TW + 1, 2 * FT_TW + 2, FT_TW + 3, 3, -FT_TW + 3, -2 * FT_TW + 2, -3 * FT_TW + 1,
                            -3 * FT_TW, -3 * FT_TW - 1, -2 * FT_TW - 2, -FT_TW - 3, -3, FT_TW - 3, 2 * FT_TW - 2, 3 * FT_TW - 1};   // fast_ring's order
        uint32_t accd = 0, accb = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t reg = (uint32_t)p[ro[j]] | ((uint32_t)p[ro[j + 8]] << 16);
            accd |= ((reg + A2) >> (10 - j)) & (0x00010001u << j);
            accb |= ((B2 - reg) >> (10 - j)) & (0x00010001u << j);
        }
        const uint32_t md = ~((accd & 0xffu) | ((accd >> 8) & 0xff00u)) & 0xffffu, mb = ~((accb & 0xffu) | ((accb >> 8) & 0xff00u)) & 0xffffu;
#else
        int d[16];
        fast_ring(p, FT_TW, (int)p[0], d);
        uint32_t md = 0, mb = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) { md |= (uint32_t)(d[k] > thr) << k; mb |= (uint32_t)(d[k] < -thr) << k; }
#endif
        if (has_arc9(md) || has_arc9(mb)) listB[atomicAdd(&nB, 1)] = (unsigned short)id;
    }
    __syncthreads();
    FAST_TICK(3);
    // pass 3: strength of the corners
    const int cntB = nB;
    for (int i = tid; i < cntB; i += 256) {
        const int id = listB[i], sy = id / FT_SW, sx = id - sy * FT_SW;
        const uint8_t* p = T + (sy + 3) * FT_TW + sx + 3;
        int d[16];
        fast_ring(p, FT_TW, (int)p[0], d);
        S[id] = (uint8_t)fast_strength(d, thr);
    }
    __syncthreads();
    FAST_TICK(4);
#ifdef STK_FAST_TIMING
    if (threadIdx.x == 0 && w > 1900 && tile_x == 7 && tile_y == 14 && blockIdx.z == 1) { g_fast_dbg[8] = cntA; g_fast_dbg[9] = cntB; }
#endif
    // pass 4: strict 3x3 maxima of the interior -> histogram + candidate list (only corners can be maxima: walk list B)
    for (int i = tid; i < cntB; i += 256) {
        const int id = listB[i], sy = id / FT_SW, sx = id - sy * FT_SW;
        if (sx < 1 || sx > FT_X || sy < 1 || sy > FT_Y) continue;     // halo pixel: another tile's interior
        const int x = x0 - 1 + sx, y = y0 - 1 + sy;
        if (x < edge || x >= w - edge || y < edge || y >= h - edge) continue;
        const uint8_t* c = S + id;
        const int sc = c[0];
        if (!sc) continue;
        if (sc > c[-1] && sc > c[1] && sc > c[-FT_SW - 1] && sc > c[-FT_SW] && sc > c[-FT_SW + 1] && sc > c[FT_SW - 1] && sc > c[FT_SW] &&
            sc > c[FT_SW + 1]) {
            atomicAdd(&st->hist[sc], 1);
            const int o = atomicAdd(&st->n_cand, 1);
            if (o < cap) { cand[o].xy = x | (y << 16); cand[o].score = sc; }
        }
    }
    FAST_TICK(5);
}

__global__ __launch_bounds__(256) void fast_nms_tiled_kernel(const uint8_t* __restrict__ img, int w, int h, int thr, int edge,
                                                             OrbLevelState* st, OrbCandidate* cand, int cap, OrbBatch bs) {
    fast_nms_tiled_body(img, w, h, thr, edge, st, cand, cap, bs, (int)blockIdx.x, (int)blockIdx.y);
}

// All pyramid levels of a batch in ONE launch: blockIdx.x runs over the tiles of level 0, then level 1, ... (big levels
// first, so the small ones fill the tail of the launch instead of each leaving most of the chip idle), blockIdx.z = frame.
__global__ __launch_bounds__(256) void fast_nms_tiled_all_kernel(const uint8_t* __restrict__ pyr, OrbLevelTable L, int thr, int edge,
                                                                 OrbLevelState* st, OrbCandidate* cand, OrbBatch bs) {
    const int t = (int)blockIdx.x;
    int l = 0;
#pragma unroll
    for (int k = 1; k < ORB_LEVELS; k++) l += t >= L.tile_ofs[k];
    const int local = t - L.tile_ofs[l];
    const int ty = local / L.tiles_x[l], tx = local - ty * L.tiles_x[l];
    fast_nms_tiled_body(pyr + L.pyr_ofs[l], L.w[l], L.h[l], thr, edge, st + l, cand + L.cand_ofs[l], L.cand_cap[l], bs, tx, ty);
}


// retainBest(2 n_l) by FAST score: the largest score s whose "count of candidates >= s" reaches `keep`. One wavefront per
// (frame, level): lane L owns bins 4 L .. 4 L + 3, an inclusive suffix sum over the lanes gives every bin its count from
// the top, and the answer is the largest bin whose suffix count reaches `keep` (1 if none does) — what the serial scan
// from 255 downwards returns.
__device__ __forceinline__ void fast_threshold_body(OrbLevelState* st, int keep) {
    const int lane = threadIdx.x;
    const int4 h4 = *reinterpret_cast<const int4*>(st->hist + 4 * lane);
    const int mine = h4.x + h4.y + h4.z + h4.w;
    int suffix = mine;                                   // sum over lanes >= this one
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_down(suffix, o, 64);
        if (lane + o < 64) suffix += v;
    }
    const int above = suffix - mine;                     // candidates in bins of higher lanes
    int best = 0;                                        // largest qualifying bin of this lane (0: none)
    int cum = above + h4.w;
    if (cum >= keep) best = 4 * lane + 3;
    else { cum += h4.z; if (cum >= keep) best = 4 * lane + 2; else { cum += h4.y; if (cum >= keep) best = 4 * lane + 1; else { cum += h4.x; if (cum >= keep) best = 4 * lane; } } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o, 64));
    if (lane == 0) {
        st->threshold = keep > 0 ? max(best, 1) : 256;    // bin 0 is never a candidate score
        st->n_sel = 0;
    }
}
__global__ __launch_bounds__(64) void fast_threshold_kernel(OrbLevelState* st, int keep, OrbBatch bs) {
    fast_threshold_body(st + blockIdx.x * bs.states, keep);
}
__global__ __launch_bounds__(64) void fast_threshold_all_kernel(OrbLevelState* st, OrbLevelTable L, OrbBatch bs) {   // grid (levels, frames)
    fast_threshold_body(st + blockIdx.y * bs.states + blockIdx.x, L.keep[blockIdx.x]);
}

// Short list: candidates >= threshold, then the Harris response (blockSize 7, Sobel-like 3x3 on the 8-bit level,
// integer sums) and the IC moments of each. Two kernels: fast_pick compacts (cheap, one lane per candidate), then
// fast_describe gives every short-listed corner a whole wavefront — its 49 Harris positions and 31 patch rows are
// spread over the lanes and reduced with integer adds, which are exact in any order (same values as a serial loop).
__device__ __forceinline__ void fast_pick_body(OrbLevelState* st, const OrbCandidate* __restrict__ cand, int cap, OrbSelected* sel, int sel_cap) {
    const int n = min(st->n_cand, cap);
    for (int i = blockIdx.x * 64 + threadIdx.x; i < n; i += gridDim.x * 64) {
        const OrbCandidate c = cand[i];
        if (c.score < st->threshold) continue;
        const int o = atomicAdd(&st->n_sel, 1);
        if (o < sel_cap) { sel[o].xy = c.xy; sel[o].score = c.score; }
    }
}
__global__ __launch_bounds__(64) void fast_pick_kernel(OrbLevelState* st, const OrbCandidate* __restrict__ cand, int cap,
                                                       OrbSelected* sel, int sel_cap, OrbBatch bs) {
    fast_pick_body(st + blockIdx.y * bs.states, cand + blockIdx.y * bs.cand, cap, sel + blockIdx.y * bs.sel, sel_cap);
}
__global__ __launch_bounds__(64) void fast_pick_all_kernel(OrbLevelState* st, const OrbCandidate* __restrict__ cand, OrbLevelTable L,
                                                           OrbSelected* sel, int sel_cap, OrbBatch bs) {     // grid (blocks, levels, frames)
    const int l = blockIdx.y;
    fast_pick_body(st + blockIdx.z * bs.states + l, cand + blockIdx.z * bs.cand + L.cand_ofs[l], L.cand_cap[l],
                   sel + blockIdx.z * bs.sel + (size_t)l * sel_cap, sel_cap);
}

__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void fast_describe_body(const uint8_t* __restrict__ img, int w, const OrbLevelState* st, OrbSelected* sel,
                                                   int sel_cap, const OrbUmax& um) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = min(st->n_sel, sel_cap);
    for (int o = blockIdx.x * 4 + wave; o < n; o += gridDim.x * 4) {
        const int xy = sel[o].xy, x = xy & 0xffff, y = xy >> 16;
        const uint8_t* p0 = img + (size_t)y * w + x;
        int a = 0, b = 0, cc = 0;
        if (lane < 49) {
            const int dy = lane / 7 - 3, dx = lane % 7 - 3;
            const uint8_t* p = p0 + dy * w + dx;
            const int Ix = ((int)p[1] - (int)p[-1]) * 2 + ((int)p[-w + 1] - (int)p[-w - 1]) + ((int)p[w + 1] - (int)p[w - 1]);
            const int Iy = ((int)p[w] - (int)p[-w]) * 2 + ((int)p[w - 1] - (int)p[-w - 1]) + ((int)p[w + 1] - (int)p[-w + 1]);
            a = Ix * Ix; b = Iy * Iy; cc = Ix * Iy;
        }
        a = wave_sum_i32(a); b = wave_sum_i32(b); cc = wave_sum_i32(cc);
        // IC moments over the circular patch (integer sums: any order gives the same bits). Round 3: a lane is a COLUMN u in
        // [-15, 16] of the upper (v = -15 .. 0) or lower (v = 1 .. 16) half of the patch; its 16 pixels are loaded
        // unconditionally, all in flight at once (a short-listed corner is >= 31 px from every border, so the 32 x 32 block is
        // inside the level), and masked afterwards by the disc: bit |v| of `disc` = (|u| <= umax[|v|]). Before, a lane per row
        // walked its row alone, and a first column version looked umax up per step: either way ~30 DEPENDENT global loads per
        // corner, 24 us per corner-wave, 200 us per 64 frames for 0.2 MB of pixels per frame.
        int m01 = 0, m10 = 0;
        {
            const int half = lane >> 5, u = (lane & 31) - 15;
            const int au = u < 0 ? -u : u;
            unsigned disc = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) disc |= (unsigned)(au <= um.u[j]) << j;
            const uint8_t* col = p0 + (half ? w : -15 * w) + u;            // v = 1 or v = -15
            int px[16];
#pragma unroll
            for (int i = 0; i < 16; i++) px[i] = col[i * w];
            int cs = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int v = half ? i + 1 : i - 15, av = v < 0 ? -v : v;      // av = 16 (lower half's last row): outside the disc
                const int t = (av < 16 && ((disc >> av) & 1u)) ? px[i] : 0;
                cs += t; m01 += v * t;
            }
            m10 = u * cs;
        }
        m01 = wave_sum_i32(m01); m10 = wave_sum_i32(m10);
        if (lane == 0) {
            const float scale = 1.f / ((1 << 2) * 7 * 255.f);
            const float scale4 = scale * scale * scale * scale;
            const float fa = (float)a, fb = (float)b, fc = (float)cc;
            sel[o].harris = (fa * fb - fc * fc - 0.04f * (fa + fb) * (fa + fb)) * scale4;
            sel[o].m01 = m01; sel[o].m10 = m10;
        }
    }
}
__global__ __launch_bounds__(256) void fast_describe_kernel(const uint8_t* __restrict__ img, int w, const OrbLevelState* st,
                                                            OrbSelected* sel, int sel_cap, OrbUmax um, OrbBatch bs) {
    fast_describe_body(img + blockIdx.y * bs.pyr, w, st + blockIdx.y * bs.states, sel + blockIdx.y * bs.sel, sel_cap, um);
}
__global__ __launch_bounds__(256) void fast_describe_all_kernel(const uint8_t* __restrict__ pyr, OrbLevelTable L, const OrbLevelState* st,
                                                                OrbSelected* sel, int sel_cap, OrbUmax um, OrbBatch bs) {   // grid (blocks, levels, frames)
    const int l = blockIdx.y;
    fast_describe_body(pyr + blockIdx.z * bs.pyr + L.pyr_ofs[l], L.w[l], st + blockIdx.z * bs.states + l,
                       sel + blockIdx.z * bs.sel + (size_t)l * sel_cap, sel_cap, um);
}

// The Harris cull on the device (round 3; it was ~150 us of host work per lane, on the critical path of a stack's last lane).
// One workgroup per (level, frame): the short list (<= ORB_CULL_MAX entries, 2 n_l + ties) is sorted by the key
// (harris descending, y, x) — one 64-bit integer: the float's order-preserving image, complemented, above y << 16 | x —
// with a bitonic network in LDS; KeyPointsFilter::retainBest keeps the n_l best and everything that ties with the n_l-th,
// which in sorted order is a prefix; the prefix goes out in that order, which is the order the host gave them. Harris
// responses are finite and never -0 (a difference of products times a positive scale), so the integer order IS the
// comparator's order. A level with more than ORB_CULL_MAX short-listed or ORB_KEEP_PACK kept corners is flagged (-1) and
// culled by the host as before.
__device__ __forceinline__ unsigned long long cull_key(float harris, int xy) {
    unsigned hb = __float_as_uint(harris);
    if ((hb << 1) == 0) hb = 0;                                   // (-0 cannot occur; keep the order total anyway)
    const unsigned asc = (hb & 0x80000000u) ? ~hb : (hb | 0x80000000u);
    return ((unsigned long long)(~asc) << 32) | (unsigned)(((xy >> 16) << 16) | (xy & 0xffff));
}

__global__ __launch_bounds__(256) void orb_cull_all_kernel(const OrbLevelState* __restrict__ st, const OrbSelected* __restrict__ sel, int sel_cap,
                                                           OrbLevelTable L, OrbKept* __restrict__ kept, int* __restrict__ kept_cnt, OrbBatch bs) {
    const int l = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    st += f * bs.states + l;
    sel += f * bs.sel + (size_t)l * sel_cap;
    kept += ((size_t)f * ORB_LEVELS + l) * ORB_KEEP_PACK;
    const int n = st->n_sel, keep = L.keep[l] / 2;                // keep[] is the short list's 2 n_l
    if (n > ORB_CULL_MAX || n > sel_cap) { if (tid == 0) kept_cnt[f * ORB_LEVELS + l] = -1; return; }
    __shared__ unsigned long long key[ORB_CULL_MAX];
    __shared__ unsigned short idx[ORB_CULL_MAX];
    __shared__ int cnt_s;
    for (int i = tid; i < ORB_CULL_MAX; i += 256) {
        key[i] = i < n ? cull_key(sel[i].harris, sel[i].xy) : ~0ull;
        idx[i] = (unsigned short)i;
    }
    __syncthreads();
    for (int k = 2; k <= ORB_CULL_MAX; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < ORB_CULL_MAX; i += 256) {
                const int p = i ^ j;
                if (p > i) {
                    const bool up = (i & k) == 0;
                    const unsigned long long a = key[i], b = key[p];
                    if ((a > b) == up) { key[i] = b; key[p] = a; const unsigned short t = idx[i]; idx[i] = idx[p]; idx[p] = t; }
                }
            }
            __syncthreads();
        }
    // the kept prefix: everything when the list is not longer than n_l, else n_l + the ties with the n_l-th response
    if (tid == 0) cnt_s = keep <= 0 ? 0 : min(n, keep);
    __syncthreads();
    if (keep > 0 && n > keep) {
        const unsigned thr_hi = (unsigned)(key[keep - 1] >> 32);
        for (int i = keep + tid; i < n; i += 256)
            if ((unsigned)(key[i] >> 32) == thr_hi) atomicAdd(&cnt_s, 1);          // ties are contiguous behind the n_l-th
    }
    __syncthreads();
    const int cnt = cnt_s;
    if (cnt > ORB_KEEP_PACK) { if (tid == 0) kept_cnt[f * ORB_LEVELS + l] = -1; return; }
    for (int i = tid; i < cnt; i += 256) {
        const OrbSelected e = sel[idx[i]];
        kept[i] = OrbKept{e.xy, e.harris, e.m01, e.m10};
    }
    if (tid == 0) kept_cnt[f * ORB_LEVELS + l] = cnt;
}

// FAST + short list of ALL levels of a batch in four launches (instead of four per level). Returns hipErrorNotSupported when
// a level does not qualify for the tiled kernel (tiny or unaligned levels): the caller then goes level by level.
hipError_t launch_fast_all(const uint8_t* pyr, const OrbLevelTable& L, int thr, int edge, OrbLevelState* st, OrbCandidate* cand,
                           OrbSelected* sel, int sel_cap, const OrbUmax& um, hipStream_t s, int n_frames, size_t pyr_stride,
                           size_t states_stride, size_t cand_stride, size_t sel_stride, OrbKept* kept, int* kept_cnt) {
    const OrbBatch bs{pyr_stride, states_stride, cand_stride, sel_stride};
    if ((reinterpret_cast<uintptr_t>(pyr) & 3) != 0 || (pyr_stride & 3) != 0) return hipErrorNotSupported;
    for (int l = 0; l < ORB_LEVELS; l++)
        if (L.w[l] < 16 || L.h[l] < 8 || (L.pyr_ofs[l] & 3) != 0) return hipErrorNotSupported;
    if (L.tile_ofs[ORB_LEVELS] > 0)
        fast_nms_tiled_all_kernel<<<dim3(L.tile_ofs[ORB_LEVELS], 1, n_frames), 256, 0, s>>>(pyr, L, thr, edge, st, cand, bs);
    fast_threshold_all_kernel<<<dim3(ORB_LEVELS, n_frames), 64, 0, s>>>(st, L, bs);
    fast_pick_all_kernel<<<dim3(64, ORB_LEVELS, n_frames), 64, 0, s>>>(st, cand, L, sel, sel_cap, bs);
    fast_describe_all_kernel<<<dim3(32, ORB_LEVELS, n_frames), 256, 0, s>>>(pyr, L, st, sel, sel_cap, um, bs);
    if (kept && kept_cnt) orb_cull_all_kernel<<<dim3(ORB_LEVELS, n_frames), 256, 0, s>>>(st, sel, sel_cap, L, kept, kept_cnt, bs);
    return hipGetLastError();
}

hipError_t launch_fast_level(const uint8_t* img, int w, int h, int thr, int edge, int keep, uint8_t* score,
                             OrbLevelState* st, OrbCandidate* cand, int cap, OrbSelected* sel, int sel_cap,
                             const OrbUmax& um, hipStream_t s, int n_frames, size_t pyr_stride, size_t states_stride,
                             size_t cand_stride, size_t sel_stride) {
    const OrbBatch bs{pyr_stride, states_stride, cand_stride, sel_stride};
    if (w >= 16 && h >= 8 && (reinterpret_cast<uintptr_t>(img) & 3) == 0 && (pyr_stride & 3) == 0) {
        if (w > 2 * edge && h > 2 * edge) {               // otherwise runByImageBorder leaves nothing: no candidates at all
            dim3 tgrid((w + FT_X - 1) / FT_X, (h + FT_Y - 1) / FT_Y, n_frames);
            fast_nms_tiled_kernel<<<tgrid, 256, 0, s>>>(img, w, h, thr, edge, st, cand, cap, bs);
        }
    } else {
        const hipError_t e = launch_fast_score_nms_plain(img, w, h, thr, edge, score, st, cand, cap, s, n_frames, bs);
        if (e != hipSuccess) return e;
    }
    fast_threshold_kernel<<<n_frames, 64, 0, s>>>(st, keep, bs);
    fast_pick_kernel<<<dim3(std::min((cap + 63) / 64, 64), n_frames), 64, 0, s>>>(st, cand, cap, sel, sel_cap, bs);
    fast_describe_kernel<<<dim3(128, n_frames), 256, 0, s>>>(img, w, st, sel, sel_cap, um, bs);
    return hipGetLastError();
}



// Fused 7x7 blur: a 128 x 32 output tile per workgroup. The 8-bit source tile with its halo goes to LDS through
// aligned dword loads (level rows are not dword-aligned in general: two aligned dwords + v_alignbyte give the four
// bytes at any offset), the row pass writes an f32 tile to LDS, the column pass rounds to 8 bits. Same operation
// order as the two-kernel form (kernels_orb_small.hip, kept for images narrower than the halo), so the result is bit-identical.
constexpr int G7_X = 128, G7_Y = 32, G7_HX = 4, G7_R = 3;
typedef float g7f2 __attribute__((ext_vector_type(2)));
constexpr int G7_TW = G7_X + 2 * G7_HX;              // 136 bytes per tile row
constexpr int G7_TH = G7_Y + 2 * G7_R;               // 38 rows

// Row filter of four adjacent outputs from the twelve bytes d0 d1 d2 (output e is centred on byte 4 + e): acc = k0 * s[x-3];
// acc += k_i * s[x-3+i] — the generic float row filter's operation order, a multiply and an add per tap (no contraction).
// Outputs (0, 1) and (2, 3) go through v_pk_mul_f32 / v_pk_add_f32 as register pairs; the taps of output e + 1 are those of
// output e shifted by one, so every byte is converted into an even and an odd pair layout (24 conversions instead of 12)
// and all operands are aligned pairs: 28 packed operations instead of 52 scalar ones.
__device__ __forceinline__ float4 g7_row_quad(uint32_t d0, uint32_t d1, uint32_t d2, const Gauss7& k) {
#define G7_B(n) ((float)(((n) < 4 ? d0 >> (8 * (n)) : (n) < 8 ? d1 >> (8 * ((n) - 4)) : d2 >> (8 * ((n) - 8))) & 255u))
    // pe[j] = (byte 1+2j, byte 2+2j), po[j] = (byte 2+2j, byte 3+2j): tap t of outputs (0, 1) is (byte 1+t, byte 2+t)
    const g7f2 pe[5] = {g7f2{G7_B(1), G7_B(2)}, g7f2{G7_B(3), G7_B(4)}, g7f2{G7_B(5), G7_B(6)}, g7f2{G7_B(7), G7_B(8)}, g7f2{G7_B(9), G7_B(10)}};
    const g7f2 po[4] = {g7f2{G7_B(2), G7_B(3)}, g7f2{G7_B(4), G7_B(5)}, g7f2{G7_B(6), G7_B(7)}, g7f2{G7_B(8), G7_B(9)}};
#undef G7_B
    // outputs (0, 1): taps (1+t, 2+t) = t even: pe[t/2], t odd: po[(t-1)/2]; outputs (2, 3): taps (3+t, 4+t) = t even: pe[1+t/2], t odd: po[1+(t-1)/2]
    g7f2 a01 = g7f2{k.k[0], k.k[0]} * pe[0], a23 = g7f2{k.k[0], k.k[0]} * pe[1];
#pragma unroll
    for (int t = 1; t < 7; t++) {
        const g7f2 kt = {k.k[t], k.k[t]};
        a01 = a01 + kt * ((t & 1) ? po[(t - 1) / 2] : pe[t / 2]);
        a23 = a23 + kt * ((t & 1) ? po[1 + (t - 1) / 2] : pe[1 + t / 2]);
    }
    return make_float4(a01.x, a01.y, a23.x, a23.y);
}

__global__ __launch_bounds__(256) void gauss7_fused_kernel(const uint8_t* __restrict__ src, int w, int h, Gauss7 k,
                                                           uint8_t* __restrict__ dst, size_t pyr_stride) {
    __shared__ __attribute__((aligned(16))) uint8_t T[G7_TH * G7_TW];
    __shared__ __attribute__((aligned(16))) float R[G7_TH * G7_X];
    src += blockIdx.z * pyr_stride; dst += blockIdx.z * pyr_stride;
    const int x0 = blockIdx.x * G7_X, y0 = blockIdx.y * G7_Y;
    const int tid = threadIdx.x;
    // phase 1: source tile (+ halo) into LDS, 4 bytes per item
    for (int i = tid; i < G7_TH * (G7_TW / 4); i += 256) {
        const int ty = i / (G7_TW / 4), d = i - ty * (G7_TW / 4);
        const int sy = refl101(y0 - G7_R + ty, h);
        const int sx0 = x0 - G7_HX + 4 * d;
        const uint8_t* row = src + (size_t)sy * w;
        uint32_t v;
        if (sx0 >= 0 && sx0 + 3 < w) v = load4_unaligned(row + sx0);
        else {
            v = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) v |= (uint32_t)row[refl101(sx0 + e, w)] << (8 * e);
        }
        *reinterpret_cast<uint32_t*>(T + ty * G7_TW + 4 * d) = v;
    }
    __syncthreads();
    // phase 2: row filter, 4 outputs per item (g7_row_quad; round 3: the kernel is bound by exactly this arithmetic)
    for (int i = tid; i < G7_TH * (G7_X / 4); i += 256) {
        const int ty = i / (G7_X / 4), q = i - ty * (G7_X / 4);
        const uint32_t* tp = reinterpret_cast<const uint32_t*>(T + ty * G7_TW + 4 * q);   // bytes 4q .. 4q+11; centre of e at 4q+4+e
        *reinterpret_cast<float4*>(R + ty * G7_X + 4 * q) = g7_row_quad(tp[0], tp[1], tp[2], k);
    }
    __syncthreads();
    // phase 3: column filter; thread = 4-wide strip x 4 output rows; (x, y) and (z, w) as packed pairs
    {
        const int strip = tid & 31, rg = tid >> 5;
        g7f2 wlo[4 + 2 * G7_R], whi[4 + 2 * G7_R];
#pragma unroll
        for (int t = 0; t < 4 + 2 * G7_R; t++) {
            const float4 v = *reinterpret_cast<const float4*>(R + (rg * 4 + t) * G7_X + 4 * strip);
            wlo[t] = g7f2{v.x, v.y}; whi[t] = g7f2{v.z, v.w};
        }
        const int x = x0 + 4 * strip;
        const bool aligned = (w & 3) == 0;                 // then every row of the level is dword-aligned (level bases are)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int y = y0 + rg * 4 + e;
            const g7f2 k3 = {k.k[3], k.k[3]};
            g7f2 alo = k3 * wlo[e + G7_R], ahi = k3 * whi[e + G7_R];
#pragma unroll
            for (int t = 1; t <= 3; t++) {
                const g7f2 kt = {k.k[3 + t], k.k[3 + t]};
                alo = alo + kt * (wlo[e + G7_R - t] + wlo[e + G7_R + t]);
                ahi = ahi + kt * (whi[e + G7_R - t] + whi[e + G7_R + t]);
            }
            if (y < h && x < w) {
                const int r0 = min(max((int)__builtin_rintf(alo.x), 0), 255), r1 = min(max((int)__builtin_rintf(alo.y), 0), 255);
                const int r2 = min(max((int)__builtin_rintf(ahi.x), 0), 255), r3 = min(max((int)__builtin_rintf(ahi.y), 0), 255);
                uint8_t* op = dst + (size_t)y * w + x;
                if (aligned && x + 3 < w) *reinterpret_cast<uint32_t*>(op) = (uint32_t)r0 | ((uint32_t)r1 << 8) | ((uint32_t)r2 << 16) | ((uint32_t)r3 << 24);
                else {
                    op[0] = (uint8_t)r0;
                    if (x + 1 < w) op[1] = (uint8_t)r1;
                    if (x + 2 < w) op[2] = (uint8_t)r2;
                    if (x + 3 < w) op[3] = (uint8_t)r3;
                }
            }
        }
    }
}

// (A streaming, LDS-free form of this blur — the design of grey_blur_stream_kernel — was built and measured in round 2:
// 0.26 ms SLOWER per 64 x 1080p batch. Unlike the grey + 5x5 template blur, the 8-bit 7x7 blur is bound by its arithmetic
// (10 byte conversions and 28 multiply-adds per 4 pixels and row), not by barriers, so the tiled kernel stays.)
hipError_t launch_gauss7(const uint8_t* src, int w, int h, const Gauss7& k, float* tmp, uint8_t* dst, hipStream_t s,
                         int n_frames, size_t pyr_stride, size_t tmp_stride) {
    if (w >= 8 && h >= 4 && (reinterpret_cast<uintptr_t>(src) & 3) == 0 && (pyr_stride & 3) == 0) {
        dim3 fgrid((w + G7_X - 1) / G7_X, (h + G7_Y - 1) / G7_Y, n_frames);
        gauss7_fused_kernel<<<fgrid, 256, 0, s>>>(src, w, h, k, dst, pyr_stride);
        return hipGetLastError();
    }
    return launch_gauss7_plain(src, w, h, k, tmp, dst, s, n_frames, pyr_stride, tmp_stride);
}

// ---- rotated BRIEF ------------------------------------------------------------------------------------
__constant__ signed char c_orb_pattern[1024];

hipError_t upload_orb_pattern(const signed char* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(c_orb_pattern), p, 1024);
}

// one lane per descriptor byte: grid = n_keypoints blocks of 32 threads
__global__ __launch_bounds__(32) void brief_kernel(const uint8_t* __restrict__ pyr_blur, OrbPyramid pyr, size_t pyr_stride,
                                                   const OrbFinalKeypoint* __restrict__ kps, uint8_t* __restrict__ desc) {
    const OrbFinalKeypoint kp = kps[blockIdx.x];
    const int l = kp.level, w = pyr.w[l];
    const uint8_t* center = pyr_blur + kp.frame * pyr_stride + pyr.ofs[l] + (size_t)kp.cy * w + kp.cx;
    const float a = kp.cos_a, b = kp.sin_a;
    const int i = threadIdx.x;
    int val = 0;
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
        const signed char* p = c_orb_pattern + (i * 8 + bit) * 4;
        const float px0 = (float)p[0], py0 = (float)p[1], px1 = (float)p[2], py1 = (float)p[3];
        const float x0 = px0 * a - py0 * b, y0 = px0 * b + py0 * a;
        const float x1 = px1 * a - py1 * b, y1 = px1 * b + py1 * a;
        const int t0 = center[(int)__builtin_rintf(y0) * w + (int)__builtin_rintf(x0)];
        const int t1 = center[(int)__builtin_rintf(y1) * w + (int)__builtin_rintf(x1)];
        val |= (t0 < t1) << bit;
    }
    desc[(size_t)kp.row * 32 + i] = (uint8_t)val;
}

// Round 3: blur only what the descriptor reads. ORB blurs every pyramid level (GaussianBlur 7x7, sigma 2) and then samples
// 512 points per keypoint within 18.4 px of its centre (|pattern| <= 13 on either axis, rotated): 500 keypoints touch
// 500 x 39 x 39 blurred pixels of the 27 Mpx a 4K pyramid holds. The blurred value of a pixel does not depend on who
// computes it (fixed operation order per pixel, see gauss7_rows/cols), so ONE wave per keypoint row-filters a 45 x 40
// window of the UNBLURRED level into LDS (the same g7_row_quad as the whole-level kernel) and column-filters, rounds and
// compares only at the 512 sample points: bit-identical descriptors (test_gpu_stages: patch blur == whole-level blur),
// and the whole-level blur — the second largest kernel of the keypoint path — is gone.
constexpr int BP_R = 19;                              // sample coordinates lie in [-BP_R, BP_R] (18.38 rounded, + 1 of margin)
constexpr int BP_ROWS = 2 * (BP_R + G7_R) + 1;        // 45 window rows: cy - 22 .. cy + 22
constexpr int BP_TW = 48;                             // window bytes per row: cx - 23 .. cx + 24 (a halo of 4 like G7_HX)
constexpr int BP_Q = 10;                              // 10 quads = 40 row-filtered columns: cx - 19 .. cx + 20
constexpr int BP_RW = 4 * BP_Q;

__global__ __launch_bounds__(64) void brief_patch_kernel(const uint8_t* __restrict__ pyr_img, OrbPyramid pyr, size_t pyr_stride,
                                                         const OrbFinalKeypoint* __restrict__ kps, Gauss7 k, uint8_t* __restrict__ desc) {
    __shared__ __attribute__((aligned(16))) uint8_t T[BP_ROWS * BP_TW];
    __shared__ __attribute__((aligned(16))) float R[BP_ROWS * BP_RW];
    const OrbFinalKeypoint kp = kps[blockIdx.x];
    const int l = kp.level, w = pyr.w[l], h = pyr.h[l];
    const uint8_t* img = pyr_img + kp.frame * pyr_stride + pyr.ofs[l];
    const int lane = threadIdx.x;
    const int wx0 = kp.cx - BP_R - G7_HX, wy0 = kp.cy - BP_R - G7_R;
    // keypoints keep 31 px from the border (runByImageBorder), so the window is inside the level; the reflecting path is
    // there for a caller who hands over other keypoints
    const bool inside = wx0 >= 0 && wx0 + BP_TW <= w && wy0 >= 0 && wy0 + BP_ROWS <= h;
    for (int i = lane; i < BP_ROWS * (BP_TW / 4); i += 64) {
        const int ty = i / (BP_TW / 4), d = i - ty * (BP_TW / 4);
        uint32_t v;
        if (inside) v = load4_unaligned(img + (size_t)(wy0 + ty) * w + wx0 + 4 * d);
        else {
            const uint8_t* row = img + (size_t)refl101(wy0 + ty, h) * w;
            v = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) v |= (uint32_t)row[refl101(wx0 + 4 * d + e, w)] << (8 * e);
        }
        *reinterpret_cast<uint32_t*>(T + ty * BP_TW + 4 * d) = v;
    }
    __syncthreads();
    for (int i = lane; i < BP_ROWS * BP_Q; i += 64) {
        const int ty = i / BP_Q, q = i - ty * BP_Q;
        const uint32_t* tp = reinterpret_cast<const uint32_t*>(T + ty * BP_TW + 4 * q);
        *reinterpret_cast<float4*>(R + ty * BP_RW + 4 * q) = g7_row_quad(tp[0], tp[1], tp[2], k);
    }
    __syncthreads();
    // lane = (descriptor byte, half): four bits = eight sample points each
    const float a = kp.cos_a, b = kp.sin_a;
    const int byte = lane >> 1, half = lane & 1;
    int val = 0;
#pragma unroll
    for (int bit = 0; bit < 4; bit++) {
        const signed char* p = c_orb_pattern + (byte * 8 + half * 4 + bit) * 4;
        int t[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float px = (float)p[2 * e], py = (float)p[2 * e + 1];
            const float x = px * a - py * b, y = px * b + py * a;
            const int ix = min(max((int)__builtin_rintf(x), -BP_R), BP_R), iy = min(max((int)__builtin_rintf(y), -BP_R), BP_R);
            const float* c = R + (iy + BP_R + G7_R) * BP_RW + ix + BP_R;
            float acc = k.k[3] * c[0];
#pragma unroll
            for (int i = 1; i <= 3; i++) acc += k.k[3 + i] * (c[-i * BP_RW] + c[i * BP_RW]);
            t[e] = min(max((int)__builtin_rintf(acc), 0), 255);
        }
        val |= (t[0] < t[1]) << bit;
    }
    const int other = __shfl_xor(val, 1);
    if (half == 0) desc[(size_t)kp.row * 32 + byte] = (uint8_t)(val | (other << 4));
}

hipError_t launch_brief_patch(const uint8_t* pyr_img, const OrbPyramid& pyr, const OrbFinalKeypoint* kps, int n, const Gauss7& k,
                              uint8_t* desc, hipStream_t s, size_t pyr_stride) {
    if (n <= 0) return hipSuccess;
    brief_patch_kernel<<<n, 64, 0, s>>>(pyr_img, pyr, pyr_stride, kps, k, desc);
    return hipGetLastError();
}

hipError_t launch_brief(const uint8_t* pyr_blur, const OrbPyramid& pyr, const OrbFinalKeypoint* kps, int n, uint8_t* desc,
                        hipStream_t s, size_t pyr_stride) {
    if (n <= 0) return hipSuccess;
    brief_kernel<<<n, 32, 0, s>>>(pyr_blur, pyr, pyr_stride, kps, desc);
    return hipGetLastError();
}

// ---- brute-force Hamming 2-NN ---------------------------------------------------------------------------
// blockIdx.y = train set (frame): its rows start `train_stride` rows after the previous set's, it has
// train_counts[blockIdx.y] rows (nt when train_counts is null), and its result block is out + blockIdx.y * nq * 4.
// Eight lanes per query, each over every eighth train row (round 3: one lane per query walked all rows alone — 120 waves of
// a 500-step serial loop on a 1 024-SIMD device, 100 us per 16 frames, all of it on the critical path of a stack). A
// candidate is the key (distance << 16 | train index): BFMatcher's order — smaller distance first, the earlier train
// row on ties (a strict '<' in its sequential scan) — is the integer order of the keys, so the two smallest keys of the
// eight partial scans, merged, are exactly the sequential scan's (best, second).
constexpr int KNN_P = 8;
static_assert(ORB_KNN_MAX_TRAIN <= 65536, "the train index must fit the low half of the key");
__global__ __launch_bounds__(64) void knn2_hamming_kernel(const uint8_t* __restrict__ query, int nq,
                                                          const uint8_t* __restrict__ train, int nt, int* __restrict__ out,
                                                          const int* __restrict__ train_counts, size_t train_stride) {
    const int q = blockIdx.x * (64 / KNN_P) + threadIdx.x / KNN_P, part = threadIdx.x % KNN_P;
    train += blockIdx.y * train_stride * 32; out += (size_t)blockIdx.y * nq * 4;
    if (train_counts) nt = train_counts[blockIdx.y];
    const int qc = min(q, nq - 1);                       // lanes past the last query scan along (the shuffles below want all lanes)
    const uint4* qa = reinterpret_cast<const uint4*>(query + (size_t)qc * 32);
    const uint4 q0 = qa[0], q1 = qa[1];
    constexpr unsigned NONE = 0x7fffffffu;
    unsigned k0 = NONE, k1 = NONE;
    for (int t = part; t < nt; t += KNN_P) {
        const uint4* ta = reinterpret_cast<const uint4*>(train + (size_t)t * 32);
        const uint4 t0 = ta[0], t1 = ta[1];
        const unsigned d = __popc(q0.x ^ t0.x) + __popc(q0.y ^ t0.y) + __popc(q0.z ^ t0.z) + __popc(q0.w ^ t0.w) +
                           __popc(q1.x ^ t1.x) + __popc(q1.y ^ t1.y) + __popc(q1.z ^ t1.z) + __popc(q1.w ^ t1.w);
        const unsigned key = (d << 16) | (unsigned)t;
        k1 = min(k1, max(k0, key));
        k0 = min(k0, key);
    }
#pragma unroll
    for (int m = 1; m < KNN_P; m <<= 1) {
        const unsigned o0 = __shfl_xor(k0, m), o1 = __shfl_xor(k1, m);
        k1 = min(max(k0, o0), min(k1, o1));
        k0 = min(k0, o0);
    }
    if (part == 0 && q < nq) {
        const int4 r = {k0 != NONE ? (int)(k0 & 0xffffu) : -1, k0 != NONE ? (int)(k0 >> 16) : -1,
                        k1 != NONE ? (int)(k1 & 0xffffu) : -1, k1 != NONE ? (int)(k1 >> 16) : -1};
        *reinterpret_cast<int4*>(out + (size_t)q * 4) = r;
    }
}

hipError_t launch_knn2_hamming(const uint8_t* query, int nq, const uint8_t* train, int nt, int* out, hipStream_t s,
                               int n_sets, const int* train_counts, size_t train_stride) {
    if (nq <= 0 || n_sets <= 0) return hipSuccess;
    if (nt > ORB_KNN_MAX_TRAIN) return hipErrorInvalidValue;
    const int per_block = 64 / KNN_P;
    knn2_hamming_kernel<<<dim3((nq + per_block - 1) / per_block, n_sets), 64, 0, s>>>(query, nq, train, nt, out, train_counts, train_stride);
    return hipGetLastError();
}

}  // namespace stk
