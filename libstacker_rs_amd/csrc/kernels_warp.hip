// kernels_warp.hip — warpPerspective / warpAffine (INTER_LINEAR) fused with the u8/u16 -> f32
// convert (x 1/255) and with the running f32 accumulator:
//     acc(p) (+)= sum over frames f of  bilinear( convert(frame_f), Minv_f . p )
// Reference: utils.rs:133 (convert), lib.rs:290-299 / 780-803 (warp), lib.rs:306-316 / 807-814 (add).
// The reference materialises a converted f32 frame (A2), a warped f32 frame (F1) and a fresh sum
// (G1) per frame: 12+12+12+12+12 B/px/channel-triple of traffic. Here one thread owns one
// destination pixel, keeps its three accumulator channels in registers across ALL frames of the
// launch and touches HBM for: the source taps (u8: ~3 B/px/frame, gathered, L2-friendly because the
// maps are near-identity) + one 12-byte accumulator read + one 12-byte accumulator write.
// NOT HBM-bound: 3 bytes per pixel and frame carry ~65 vector instructions (~108 issue slots at gfx950's rates, twelve
// byte conversions alone are 24: tools/valu_rates.hip) — the vector pipe is the limit. No LDS needed for
// the gather (footprints of neighbouring lanes overlap in L1/L2).
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
            ax = finite ? X - flx : 0.0f; ay = finite ? Y - fly : 0.0f;
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

// -----------------------------------------------------------------------------------------------
// Fast path for the production configuration: BGR u8 source, BORDER_CONSTANT, exact f32 coordinates.
// Same arithmetic as the generic kernel (bit-identical results). The kernel is bound by VALU issue, not by HBM (3 B
// of source per pixel and frame against ~60 instructions of coordinate, unpack and lerp arithmetic), so the work of
// round 2 went into the instruction count:
//   * X / W and Y / W share ONE v_rcp_f32 + Newton step and then run the exact fma chain the compiler's IEEE division
//     expands to (q = n r; e = n - d q; q += e r; e = n - d q; q += e r): the same bits as two `/` for a W in the normal
//     range — which the interior predicate requires — at 8 instructions (3 shared + 5 that pack into v_pk_*) instead of 22;
//   * whether the 4 taps of ALL frames of the group are inside the frame is voted per wave BEFORE the loads are issued:
//     interior waves (all but the frame's rim) load from the raw coordinates — no clamps, no end-of-buffer back-off, no
//     per-tap border selects; rim waves take the general path below with the same coordinates;
//   * the two horizontally adjacent taps of a row are 6 contiguous bytes -> ONE unaligned 8-byte load, bytes converted
//     with v_cvt_f32_ubyteN (extract + convert in one instruction);
//   * the frame loop is unrolled by WU: all 2*WU loads of a group are issued before the first is consumed;
//   * the accumulator (12 B/px) is read once (if accumulating) and written once per launch, whatever the frame count.
// Round 3, same bits again, ~70 -> ~50 VALU instructions per pixel and frame:
//   * the range test that licenses the shared reciprocal chain is made ONCE per frame on the host (warp_fold: W, X, Y are
//     affine in (x, y), so their extremes over the destination rectangle sit at its corners) and reaches the kernel as a
//     scalar flag; only frames that fail it take the per-pixel test;
//   * the twelve taps are converted straight out of the loaded dwords (v_cvt_f32_ubyte0..3, no shifts) into register
//     PAIRS — (B, G) of a tap, and R of the two rows — so that the x alpha multiplies (12 -> 6 v_pk_mul_f32), the
//     horizontal and vertical lerps (18 -> 10: v_pk_add_f32 with a negated operand + v_pk_fma_f32) and the running sums
//     (3 -> 2) are packed. Per component these are the generic kernel's operations in the generic kernel's order.
// -----------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t* p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

struct Tap12 { uint32_t a, b, c; };   // 12 bytes of a row: u8 kernel: an aligned window; u16 kernel: B0 G0 | R0 B1 | G1 R1

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32

// n / d for both components, correctly rounded for d and the quotients in the normal range: the compiler's own expansion
// of an IEEE f32 division without the v_div_scale / v_div_fixup range handling, the reciprocal chain shared by both
// quotients and the five dependent steps as packed instructions.
__device__ __forceinline__ f32x2 div2_shared(f32x2 n, float d) {
    float r = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e0, r, r);
    const f32x2 r2 = {r, r}, md = {-d, -d};
    f32x2 q = n * r2;
    f32x2 e = pk_fma(md, q, n);
    q = pk_fma(e, r2, q);
    e = pk_fma(md, q, n);
    return pk_fma(e, r2, q);
}

// WX: waves of a workgroup side by side along x (tile = 64 WX x 4 / WX pixels); WU: frames in flight per lane
template <bool AFFINE, int WX, int WU>
__global__ __launch_bounds__(256) void warp_accumulate_u8c3_kernel(WarpArgs a) {
    const int wave = threadIdx.x >> 6;
    const int x = (blockIdx.x * WX + (wave % WX)) * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * (4 / WX) + wave / WX;
    if (x >= a.dw || y >= a.dh) return;
    float* accp = a.acc + (size_t)y * a.acc_stride + (size_t)x * 3;
    f32x2 s01 = {0.f, 0.f};                   // (B, G) running sums as a register pair, R apart
    float s2 = 0.f;
    if (a.accumulate) { s01.x = accp[0]; s01.y = accp[1]; s2 = accp[2]; }
    const float fx = (float)x, fy = (float)y;
    const int sw = a.sw, sh = a.sh;
    const int stride32 = (int)a.src_stride;
    const float alpha = a.alpha, b0 = a.bv[0], b1 = a.bv[1], b2 = a.bv[2];

#define STK_CH(d, sft) ((float)(((d) >> (sft)) & 0xffu) * alpha)
#define STK_UB(d, k) ((float)(((d) >> (8 * (k))) & 0xffu))                           /* v_cvt_f32_ubyte<k> */
#define STK_LERP(p00, p01, p10, p11)                                                   \
    __builtin_fmaf(ay[u], __builtin_fmaf(ax[u], (p11) - (p10), (p10)) - __builtin_fmaf(ax[u], (p01) - (p00), (p00)), \
                   __builtin_fmaf(ax[u], (p01) - (p00), (p00)))
    for (int f0 = 0; f0 < a.n_frames; f0 += WU) {
        float ax[WU], ay[WU], Xs[WU], Ys[WU];
        int ix[WU], iy[WU];
        bool interior = true;
#pragma unroll
        for (int u = 0; u < WU; u++) {
            const WarpFrame* fr = a.frames + min(f0 + u, a.n_frames - 1);
            // (X, Y) as one packed pair: fma(M0, x, fma(M1, y, M2)) and fma(M3, x, fma(M4, y, M5)) — the generic kernel's operations
            bool fin = true;                              // false: X or Y is NaN / inf / absurdly large
            f32x2 XY = pk_fma(f32x2{fr->M[0], fr->M[3]}, f32x2{fx, fx}, pk_fma(f32x2{fr->M[1], fr->M[4]}, f32x2{fy, fy}, f32x2{fr->M[2], fr->M[5]}));
            if (!AFFINE) {
                const float W = __builtin_fmaf(fr->M[6], fx, __builtin_fmaf(fr->M[7], fy, fr->M[8]));
                // |W| in [2^-40, 2^40] (it is ~1 for any real homography) and |X|, |Y| < 2^40: the range in which the IEEE
                // expansion applies no scaling, so the shared chain returns the same bits
                if (fr->flags & WARPFRAME_DIV_IN_RANGE) XY = div2_shared(XY, W);     // decided per frame on the host (uniform branch)
                else {
                    const float aw = __builtin_fabsf(W);
                    // (compares, not max: a NaN operand must fail the test, v_max would drop it)
                    const bool safe = (aw < 1.0995116e12f) & (__builtin_fabsf(XY.x) < 1.0995116e12f) & (__builtin_fabsf(XY.y) < 1.0995116e12f) &
                                      (aw > 9.094947e-13f);
                    if (__all(safe)) XY = div2_shared(XY, W);   // finite operands in range: finite quotients
                    else {
                        XY.x = XY.x / W; XY.y = XY.y / W;
                        fin = (__builtin_fabsf(XY.x) < 1e9f) & (__builtin_fabsf(XY.y) < 1e9f);
                    }
                }
            } else {
                fin = (__builtin_fabsf(XY.x) < 1e9f) & (__builtin_fabsf(XY.y) < 1e9f);
            }
            const float X = XY.x, Y = XY.y;
            Xs[u] = X; Ys[u] = Y;
            const f32x2 fl = {__builtin_floorf(X), __builtin_floorf(Y)};
            ix[u] = (int)fl.x; iy[u] = (int)fl.y;         // saturating conversion; NaN -> 0, caught by the finite test
            const f32x2 fr2 = XY - fl;
            ax[u] = fr2.x; ay[u] = fr2.y;
            // all four taps inside the frame with a row to spare below (the 8-byte load of the last pixel pair of the last
            // row would run 2 bytes past the frame; rows sh-2 and sh-1 are left to the rim path). The int conversion saturates,
            // so huge coordinates fail the unsigned tests by themselves.
            interior &= fin & ((unsigned)ix[u] < (unsigned)(sw - 1)) & ((unsigned)iy[u] < (unsigned)(sh - 2));
        }
        if (__all(interior)) {
            uint64_t raw0[WU], raw1[WU];
#pragma unroll
            for (int u = 0; u < WU; u++) {
                const uint8_t* __restrict__ src = (const uint8_t*)a.frames[min(f0 + u, a.n_frames - 1)].src;
                // one frame is < 2 GiB (checked by the launcher): 32-bit offsets on the frame's uniform base pointer
                unsigned o = (unsigned)(__mul24(iy[u], stride32) + ix[u] * 3);      // rows < 2^24, row stride < 2^24 bytes
                const uint8_t* __restrict__ src1 = src + (unsigned)stride32;        // uniform: the second row's base stays in SGPRs
                if (a.frames[min(f0 + u, a.n_frames - 1)].flags & WARPFRAME_SRC_ALIGNED4) {
                    // The L1's address pipeline, not VALU issue, is what bounds this kernel (round 3: 20 % fewer VALU
                    // instructions changed nothing): an 8-byte gather at a 3-byte lane stride straddles dword boundaries in
                    // most lanes. A 12-byte window from the dword-aligned address below covers the 6 bytes wherever they
                    // start (offset 0..3), and v_alignbyte moves them into place: 1.23 -> 1.01 ms per 64 4K frames. The
                    // window ends at most 11 bytes behind `o`: inside the frame, the interior test keeps a row to spare.
                    const unsigned oa = o & ~3u, sh = o & 3u;
                    Tap12 t0, t1;
                    __builtin_memcpy(&t0, __builtin_assume_aligned(src + oa, 4), 12);
                    __builtin_memcpy(&t1, __builtin_assume_aligned(src1 + oa, 4), 12);
                    // (the flag also says that the row stride is a multiple of 4: both rows' windows are dword-aligned)
                    const uint32_t l0 = __builtin_amdgcn_alignbyte(t0.b, t0.a, sh), h0 = __builtin_amdgcn_alignbyte(t0.c, t0.b, sh);
                    const uint32_t l1 = __builtin_amdgcn_alignbyte(t1.b, t1.a, sh), h1 = __builtin_amdgcn_alignbyte(t1.c, t1.b, sh);
                    raw0[u] = (uint64_t)l0 | ((uint64_t)h0 << 32);
                    raw1[u] = (uint64_t)l1 | ((uint64_t)h1 << 32);
                } else {
                    raw0[u] = load_u64_unaligned(src + o);
                    raw1[u] = load_u64_unaligned(src1 + o);
                }
            }
#pragma unroll
            for (int u = 0; u < WU; u++) {
                if (f0 + u < a.n_frames) {
                    // a row's two taps are bytes B0 G0 R0 B1 | G1 R1 . . of the (lo, hi) dwords
                    const uint32_t l0 = (uint32_t)raw0[u], h0 = (uint32_t)(raw0[u] >> 32);
                    const uint32_t l1 = (uint32_t)raw1[u], h1 = (uint32_t)(raw1[u] >> 32);
                    const f32x2 al2 = {alpha, alpha}, ax2 = {ax[u], ax[u]}, ay2 = {ay[u], ay[u]};
                    const f32x2 bg00 = f32x2{STK_UB(l0, 0), STK_UB(l0, 1)} * al2, bg01 = f32x2{STK_UB(l0, 3), STK_UB(h0, 0)} * al2;
                    const f32x2 bg10 = f32x2{STK_UB(l1, 0), STK_UB(l1, 1)} * al2, bg11 = f32x2{STK_UB(l1, 3), STK_UB(h1, 0)} * al2;
                    const f32x2 rl = f32x2{STK_UB(l0, 2), STK_UB(l1, 2)} * al2;       // R of the left tap, rows (iy, iy + 1)
                    const f32x2 rr = f32x2{STK_UB(h0, 1), STK_UB(h1, 1)} * al2;       // R of the right tap
                    const f32x2 t0 = pk_fma(ax2, bg01 - bg00, bg00), t1 = pk_fma(ax2, bg11 - bg10, bg10);
                    const f32x2 tr = pk_fma(ax2, rr - rl, rl);                        // (row iy, row iy + 1)
                    s01 = s01 + pk_fma(ay2, t1 - t0, t0);
                    s2 = s2 + __builtin_fmaf(ay[u], tr.y - tr.x, tr.x);
                }
            }
            continue;
        }
        // rim waves: clamped loads, per-tap border selects
#pragma unroll
        for (int u = 0; u < WU; u++) {
            if (f0 + u < a.n_frames) {
                const uint8_t* __restrict__ src = (const uint8_t*)a.frames[f0 + u].src;
                const bool finite = (__builtin_fabsf(Xs[u]) < 1e9f) & (__builtin_fabsf(Ys[u]) < 1e9f);   // false for NaN / inf
                const int jx = finite ? ix[u] : -100000, jy = finite ? iy[u] : -100000;
                if (!finite) { ax[u] = 0.0f; ay[u] = 0.0f; }
                const int xb = min(max(jx, 0), sw - 2);
                const int ox = jx - xb;            // 0 normal, -1 left tap outside, 1 right tap outside, else both outside
                const bool vy0 = (unsigned)jy < (unsigned)sh, vy1 = (unsigned)(jy + 1) < (unsigned)sh;
                const int yb0 = min(max(jy, 0), sh - 1), yb1 = min(max(jy + 1, 0), sh - 1);
                // an 8-byte load at the last pixel pair of the last row would run 2 bytes past the frame: back off, shift
                const int back0 = (yb0 == sh - 1 && xb == sw - 2) ? 2 : 0;
                const int back1 = (yb1 == sh - 1 && xb == sw - 2) ? 2 : 0;
                const uint64_t r0 = load_u64_unaligned(src + (unsigned)(yb0 * stride32 + xb * 3 - back0)) >> (back0 * 8);
                const uint64_t r1 = load_u64_unaligned(src + (unsigned)(yb1 * stride32 + xb * 3 - back1)) >> (back1 * 8);
                // left tap = bytes 0..2, right tap = bytes 3..5 of the pair starting at column xb
                const uint32_t a0 = (uint32_t)r0, a1 = (uint32_t)(r0 >> 24);
                const uint32_t c0 = (uint32_t)r1, c1 = (uint32_t)(r1 >> 24);
                // which dword serves tap x0 = ix and tap x1 = ix+1 (column offset from xb: ox, ox+1)
                const bool l_ok = (ox == 0) | (ox == 1), r_ok = (ox == 0) | (ox == -1);
                const uint32_t tl0 = ox == 0 ? a0 : a1, tr0 = ox == 0 ? a1 : a0;
                const uint32_t tl1 = ox == 0 ? c0 : c1, tr1 = ox == 0 ? c1 : c0;
                const bool v00 = l_ok & vy0, v01 = r_ok & vy0, v10 = l_ok & vy1, v11 = r_ok & vy1;
                {
                    const float p00 = v00 ? STK_CH(tl0, 0) : b0, p01 = v01 ? STK_CH(tr0, 0) : b0;
                    const float p10 = v10 ? STK_CH(tl1, 0) : b0, p11 = v11 ? STK_CH(tr1, 0) : b0;
                    s01.x = s01.x + STK_LERP(p00, p01, p10, p11);
                }
                {
                    const float p00 = v00 ? STK_CH(tl0, 8) : b1, p01 = v01 ? STK_CH(tr0, 8) : b1;
                    const float p10 = v10 ? STK_CH(tl1, 8) : b1, p11 = v11 ? STK_CH(tr1, 8) : b1;
                    s01.y = s01.y + STK_LERP(p00, p01, p10, p11);
                }
                {
                    const float p00 = v00 ? STK_CH(tl0, 16) : b2, p01 = v01 ? STK_CH(tr0, 16) : b2;
                    const float p10 = v10 ? STK_CH(tl1, 16) : b2, p11 = v11 ? STK_CH(tr1, 16) : b2;
                    s2 = s2 + STK_LERP(p00, p01, p10, p11);
                }
            }
        }
    }
#undef STK_CH
#undef STK_UB
#undef STK_LERP
    accp[0] = s01.x; accp[1] = s01.y; accp[2] = s2;
}

// 16-bit BGR fast path (16-bit stacks: stk_hybrid_match): the two horizontally adjacent taps of a row are 12 contiguous
// bytes -> one 12-byte load; waves whose footprints are all interior take it, border waves take per-tap conditional loads.
// Same operation sequence as the generic kernel.
__device__ __forceinline__ Tap12 load_tap12(const uint8_t* p) {
    Tap12 t;
    __builtin_memcpy(&t, p, 12);
    return t;
}

template <bool AFFINE>
__global__ __launch_bounds__(256) void warp_accumulate_u16c3_kernel(WarpArgs a) {
    constexpr int WU = 4;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.dw || y >= a.dh) return;
    float* accp = a.acc + (size_t)y * a.acc_stride + (size_t)x * 3;
    float s[3] = {0.f, 0.f, 0.f};
    if (a.accumulate) { s[0] = accp[0]; s[1] = accp[1]; s[2] = accp[2]; }
    const float fx = (float)x, fy = (float)y;
    const int sw = a.sw, sh = a.sh;
    const int stride_el = (int)a.src_stride;               // row stride in 16-bit elements
    const float alpha = a.alpha;
    const float bv[3] = {a.bv[0], a.bv[1], a.bv[2]};

    for (int f0 = 0; f0 < a.n_frames; f0 += WU) {
        float ax[WU], ay[WU];
        int ix[WU], iy[WU];
        bool interior = true;
#pragma unroll
        for (int u = 0; u < WU; u++) {
            const WarpFrame* fr = a.frames + min(f0 + u, a.n_frames - 1);
            // same coordinate arithmetic as the u8 kernel: packed (X, Y), one reciprocal chain shared by X / W and Y / W
            f32x2 XY = pk_fma(f32x2{fr->M[0], fr->M[3]}, f32x2{fx, fx}, pk_fma(f32x2{fr->M[1], fr->M[4]}, f32x2{fy, fy}, f32x2{fr->M[2], fr->M[5]}));
            if (!AFFINE) {
                const float W = __builtin_fmaf(fr->M[6], fx, __builtin_fmaf(fr->M[7], fy, fr->M[8]));
                if (fr->flags & WARPFRAME_DIV_IN_RANGE) XY = div2_shared(XY, W);     // decided per frame on the host (uniform branch)
                else {
                    const float aw = __builtin_fabsf(W);
                    const bool safe = (aw < 1.0995116e12f) & (__builtin_fabsf(XY.x) < 1.0995116e12f) & (__builtin_fabsf(XY.y) < 1.0995116e12f) &
                                      (aw > 9.094947e-13f);                         // compares: a NaN operand fails
                    if (__all(safe)) XY = div2_shared(XY, W);
                    else { XY.x = XY.x / W; XY.y = XY.y / W; }
                }
            }
            const float X = XY.x, Y = XY.y;
            const bool finite = (__builtin_fabsf(X) < 1e9f) & (__builtin_fabsf(Y) < 1e9f);
            const float flx = __builtin_floorf(X), fly = __builtin_floorf(Y);
            ix[u] = finite ? (int)flx : -100000; iy[u] = finite ? (int)fly : -100000;
            ax[u] = finite ? X - flx : 0.0f; ay[u] = finite ? Y - fly : 0.0f;
            // (a row to spare below, as in the u8 kernel: the aligned 16-byte window of the last pixel pair of the last row
            // would end 4 bytes past the frame; rows sh-2 and sh-1 are left to the rim path)
            interior &= ((unsigned)ix[u] < (unsigned)(sw - 1)) & ((unsigned)iy[u] < (unsigned)(sh - 2));
        }
#define STK_LERP16(p00, p01, p10, p11)                                                   \
    __builtin_fmaf(ay[u], __builtin_fmaf(ax[u], (p11) - (p10), (p10)) - __builtin_fmaf(ax[u], (p01) - (p00), (p00)), \
                   __builtin_fmaf(ax[u], (p01) - (p00), (p00)))
        if (__all(interior)) {
            Tap12 r0[WU], r1[WU];
#pragma unroll
            for (int u = 0; u < WU; u++) {
                const uint8_t* src = (const uint8_t*)a.frames[min(f0 + u, a.n_frames - 1)].src;
                const unsigned o = (unsigned)(__mul24(iy[u], stride_el) + ix[u] * 3) * 2u;
                const uint8_t* src1 = src + (unsigned)stride_el * 2u;               // uniform second-row base
                if (a.frames[min(f0 + u, a.n_frames - 1)].flags & WARPFRAME_SRC_ALIGNED4) {
                    // as in the u8 kernel: a dword-aligned 16-byte window instead of a 12-byte gather at a 6-byte lane stride
                    // that starts on an odd word in half of the lanes; v_alignbyte shifts by 0 or 2 bytes
                    const unsigned oa = o & ~3u, sh = o & 3u;
                    uint32_t t0[4], t1[4];
                    __builtin_memcpy(t0, __builtin_assume_aligned(src + oa, 4), 16);
                    __builtin_memcpy(t1, __builtin_assume_aligned(src1 + oa, 4), 16);
                    r0[u] = Tap12{__builtin_amdgcn_alignbyte(t0[1], t0[0], sh), __builtin_amdgcn_alignbyte(t0[2], t0[1], sh), __builtin_amdgcn_alignbyte(t0[3], t0[2], sh)};
                    r1[u] = Tap12{__builtin_amdgcn_alignbyte(t1[1], t1[0], sh), __builtin_amdgcn_alignbyte(t1[2], t1[1], sh), __builtin_amdgcn_alignbyte(t1[3], t1[2], sh)};
                } else {
                    r0[u] = load_tap12(src + o);
                    r1[u] = load_tap12(src1 + o);
                }
            }
#pragma unroll
            for (int u = 0; u < WU; u++) {
                if (f0 + u < a.n_frames) {
                    // same pairing as the u8 kernel: (B, G) of a tap and R of the two rows go through v_pk_mul / v_pk_fma
                    const f32x2 al2 = {alpha, alpha}, ax2 = {ax[u], ax[u]}, ay2 = {ay[u], ay[u]};
#define STK_LO16(d) ((float)((d) & 0xffffu))
#define STK_HI16(d) ((float)((d) >> 16))
                    const f32x2 bg00 = f32x2{STK_LO16(r0[u].a), STK_HI16(r0[u].a)} * al2, bg01 = f32x2{STK_HI16(r0[u].b), STK_LO16(r0[u].c)} * al2;
                    const f32x2 bg10 = f32x2{STK_LO16(r1[u].a), STK_HI16(r1[u].a)} * al2, bg11 = f32x2{STK_HI16(r1[u].b), STK_LO16(r1[u].c)} * al2;
                    const f32x2 rl = f32x2{STK_LO16(r0[u].b), STK_LO16(r1[u].b)} * al2;
                    const f32x2 rr = f32x2{STK_HI16(r0[u].c), STK_HI16(r1[u].c)} * al2;
#undef STK_LO16
#undef STK_HI16
                    const f32x2 t0 = pk_fma(ax2, bg01 - bg00, bg00), t1 = pk_fma(ax2, bg11 - bg10, bg10);
                    const f32x2 tr = pk_fma(ax2, rr - rl, rl);
                    const f32x2 vbg = pk_fma(ay2, t1 - t0, t0);
                    s[0] = s[0] + vbg.x; s[1] = s[1] + vbg.y;
                    s[2] = s[2] + __builtin_fmaf(ay[u], tr.y - tr.x, tr.x);
                }
            }
            continue;
        }
#pragma unroll
        for (int u = 0; u < WU; u++) {
            if (f0 + u < a.n_frames) {
                const uint16_t* src = (const uint16_t*)a.frames[f0 + u].src;
                const int x0 = ix[u], y0 = iy[u];
                const bool vx0 = (unsigned)x0 < (unsigned)sw, vx1 = (unsigned)(x0 + 1) < (unsigned)sw;
                const bool vy0 = (unsigned)y0 < (unsigned)sh, vy1 = (unsigned)(y0 + 1) < (unsigned)sh;
                const int xc0 = min(max(x0, 0), sw - 1), xc1 = min(max(x0 + 1, 0), sw - 1);
                const int yc0 = min(max(y0, 0), sh - 1), yc1 = min(max(y0 + 1, 0), sh - 1);
                const uint16_t* q0 = src + (size_t)yc0 * stride_el;
                const uint16_t* q1 = src + (size_t)yc1 * stride_el;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float p00 = (vx0 & vy0) ? (float)q0[xc0 * 3 + c] * alpha : bv[c];
                    const float p01 = (vx1 & vy0) ? (float)q0[xc1 * 3 + c] * alpha : bv[c];
                    const float p10 = (vx0 & vy1) ? (float)q1[xc0 * 3 + c] * alpha : bv[c];
                    const float p11 = (vx1 & vy1) ? (float)q1[xc1 * 3 + c] * alpha : bv[c];
                    s[c] = s[c] + STK_LERP16(p00, p01, p10, p11);
                }
            }
        }
#undef STK_LERP16
    }
    accp[0] = s[0]; accp[1] = s[1]; accp[2] = s[2];
}

// One thread per fold entry. A frame whose ECC failed (status != 0) gets the identity: the host reports the failure and
// discards the sum, the table only has to be harmless.
__global__ void warp_frames_from_ecc_kernel(const EccFrameResult* __restrict__ results, const void* const* __restrict__ src_ptrs,
                                            int n_templates, int add_reference, int is_affine, int w, int h, size_t src_row_bytes,
                                            WarpFrame* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_out = n_templates + (add_reference ? 1 : 0);
    if (i >= n_out) return;
    const int t = add_reference ? i - 1 : i;                // template index, -1: the reference frame itself
    double M[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (t >= 0 && results[t].status == 0) {
        for (int k = 0; k < 9; k++) M[k] = (double)results[t].warp[k];
        if (is_affine) { M[6] = 0; M[7] = 0; M[8] = 1; }
    }
    WarpFrame wf;
    warp_frame_make(wf, src_ptrs[t + 1], M, is_affine);
    wf.flags = warp_frame_flags(wf.src, wf.M, src_row_bytes, w, h, is_affine);
    out[i] = wf;
}

hipError_t launch_warp_frames_from_ecc(const EccFrameResult* results, const void* const* src_ptrs, int n_templates, int add_reference,
                                       int is_affine, int w, int h, size_t src_row_bytes, WarpFrame* out, hipStream_t s) {
    const int n_out = n_templates + (add_reference ? 1 : 0);
    if (n_out <= 0) return hipSuccess;
    warp_frames_from_ecc_kernel<<<(n_out + 63) / 64, 64, 0, s>>>(results, src_ptrs, n_templates, add_reference, is_affine, w, h, src_row_bytes, out);
    return hipGetLastError();
}

hipError_t launch_warp_accumulate(const WarpArgs& a, int depth, hipStream_t s) {
    dim3 grid((a.dw + 63) / 64, (a.dh + 3) / 4);
    // (sh >= 2: the kernel's interior bound is (unsigned)(sh - 2); a one-row frame takes the generic kernel)
    if (depth == 8 && a.cn == 3 && a.subpixel_bits == 0 && a.border_mode == STK_BORDER_CONSTANT && a.sw >= 2 && a.sh >= 2 &&
        (size_t)a.sw * a.sh * 3 >= 16 && a.src_stride * (size_t)a.sh < ((size_t)1 << 31) && a.src_stride < (1u << 23) && a.sh < (1 << 23)) {
        const int v = a.tune & 0xff;                // tuning: bits 0-1 tile shape, bits 4-5 frames in flight
#define STK_U8C3(WX, WU)                                                                                   \
        do {                                                                                               \
            dim3 g((a.dw + 64 * WX - 1) / (64 * WX), (a.dh + 4 / WX - 1) / (4 / WX));                      \
            if (a.is_affine) warp_accumulate_u8c3_kernel<true, WX, WU><<<g, 256, 0, s>>>(a);              \
            else warp_accumulate_u8c3_kernel<false, WX, WU><<<g, 256, 0, s>>>(a);                         \
        } while (0)
        switch (v) {
            case 0x01: STK_U8C3(2, 4); break;
            case 0x02: STK_U8C3(4, 4); break;
            case 0x10: STK_U8C3(1, 8); break;
            case 0x11: STK_U8C3(2, 8); break;
            case 0x12: STK_U8C3(4, 8); break;
            case 0x20: STK_U8C3(1, 2); break;
            case 0x22: STK_U8C3(4, 2); break;
            default: STK_U8C3(1, 4); break;
        }
#undef STK_U8C3
        return hipGetLastError();
    }
    if (depth == 16 && a.cn == 3 && a.subpixel_bits == 0 && a.border_mode == STK_BORDER_CONSTANT && a.sw >= 2 && a.sh >= 2 &&
        a.src_stride * 2 * (size_t)a.sh < ((size_t)1 << 31) && a.src_stride < (1u << 23) && a.sh < (1 << 23)) {
        if (a.is_affine) warp_accumulate_u16c3_kernel<true><<<grid, 256, 0, s>>>(a);
        else warp_accumulate_u16c3_kernel<false><<<grid, 256, 0, s>>>(a);
        return hipGetLastError();
    }
#define STK_WARP_CASE(T, CN) warp_accumulate_kernel<T, CN><<<grid, 256, 0, s>>>(a)
    if (depth == 8 && a.cn == 3) STK_WARP_CASE(uint8_t, 3);
    else if (depth == 8 && a.cn == 1) STK_WARP_CASE(uint8_t, 1);
    else if (depth == 16 && a.cn == 3) STK_WARP_CASE(uint16_t, 3);
    else if (depth == 16 && a.cn == 1) STK_WARP_CASE(uint16_t, 1);
    else if (depth == 32 && a.cn == 3) STK_WARP_CASE(float, 3);
    else if (depth == 32 && a.cn == 1) STK_WARP_CASE(float, 1);
    else if (depth == 8 && a.cn == 4) STK_WARP_CASE(uint8_t, 4);        // BGRA: the alpha plane is warped and summed like a colour (the
    else if (depth == 16 && a.cn == 4) STK_WARP_CASE(uint16_t, 4);      // reference converts, warps and adds whatever imread returned)
    else if (depth == 32 && a.cn == 4) STK_WARP_CASE(float, 4);
    else return hipErrorInvalidValue;
#undef STK_WARP_CASE
    return hipGetLastError();
}

}  // namespace stk
