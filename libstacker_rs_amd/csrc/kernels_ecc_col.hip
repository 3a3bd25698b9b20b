// kernels_ecc_col.hip — the column-walking iteration pass of findTransformECC (lib.rs:769-777; algorithm SURVEY.md
// 8a-E*), every motion model. Same moment sums and the same partials layout as the direct kernels in kernels_ecc.hip.
// Compiled with -fno-slp-vectorize (Makefile): every pair is written out as float2 here, and the vectoriser's own
// pairings cost register shuffles.
#include "ecc_pixel.h"
#include <type_traits>

namespace stk {

#ifndef STK_COL_WG
#define STK_COL_WG 4
#endif
// frame-0 rows kept ahead of the row being fetched by the LDS ring: EccIterArgs::ring_lookahead, 5 in production (the debug
// option "ecc_ring_lookahead" lowers it so that the run-time check fires and a test can see the fallback work)

// ---------------------------------------------------------------------------------------------------
// The row-walking pass this one replaced ran ~97 VALU instructions per pixel (95 in the loop, the rest in row-end and
// block-end reductions) with the VALU pipe saturated. What this version removes, all of it arithmetic:
//   * A wave owns a COLUMN strip — 64 adjacent x, a run of consecutive rows — so X is a per-lane constant and Y a
//     scalar: X never enters the loop. The lane accumulates Y-moments (sum q, sum q*Y, sum q*Y^2 with scalar
//     multipliers) and the powers of X are applied once, when the strip is flushed. The template address is a scalar
//     base plus a constant lane offset (no per-pixel address arithmetic), the loop counters live in SGPRs.
//   * J.u, J.v, J.m are factorised the same way: J = (a, b, t) (x) (X, Y, 1), so a lane keeps sum c*w and sum c*w*Y for
//     c in {a, b, t}, w in {u, v, m}: 18 accumulators instead of 24, no J vector at all.
//   * Everything that comes in pairs is written as v_pk_* on float2: (gx, gy) bilinear (vertical blend first, so the
//     two taps of a row pair up as loaded), (ja, jb), the six products, all accumulators.
//   * A strip whose four corners map well inside the frame-0 image (the image of a convex set under a homography with
//     w > 0 is convex) takes a loop without the mask: no compares, no selects, no clamps — bit-identical results to
//     the masked loop, because with m = 1 every masked expression reduces to the unmasked one exactly.
//   * ONE cross-lane reduction per strip instead of one per row plus one per block: 66 per-lane values go through a
//     lane-transposing fold (each level halves the number of registers: v_permlane32_swap / v_permlane16_swap for lane
//     bits 5 and 4, DPP row_ror / quad_perm and ds_swizzle below), ~200 instructions instead of ~460 + 2 x 144.
// ~60 VALU instructions per pixel in the unmasked loop. Work units are (column strip, row) pairs in column-major order,
// split evenly over the 4 x nb waves of the frame, so a frame's summation partition still depends on its size only
// (shard-invariant bits). What limits it now, and what was tried on top: DESIGN.md 4.1.
// ---------------------------------------------------------------------------------------------------
// (the LDS-DMA blocks below set m0 and say so in their clobber lists; clang warns that m0 is a reserved register)
#pragma clang diagnostic ignored "-Winline-asm"

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32
__device__ __forceinline__ f32x2 bc2(float v) { return f32x2{v, v}; }
// a product the vectoriser must not pair up by shuffling its operands into new register pairs (two moves more than
// the two multiplies it saves): the results land in adjacent registers and feed v_pk_* directly
__device__ __forceinline__ float mul_opaque(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// the compiler's own builtins for these two drop the second result (ROCm 7.2: r[0] + r[1] comes out as r[0] + r[0]),
// hence inline assembly; the s_nop covers the VALU-write -> permlane-swap-read hazard the assembler cannot see.
// v_permlane32_swap a, b: a' = {a[0:31], b[0:31]}, b' = {a[32:63], b[32:63]}; permlane16: the same per pair of 16-lane rows.
__device__ __forceinline__ void lane_swap32(float& a, float& b) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void lane_swap16(float& a, float& b) { asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// One level of the fold over lane bit BIT: of two registers, the lanes with the bit clear keep `lo`, those with it
// set keep `hi`, each adding its partner's copy of the same register. Afterwards a lane holds one of the two sums.
template <int BIT>
__device__ __forceinline__ float fold_pair(float lo, float hi, bool bit) {
    if constexpr (BIT == 5) { lane_swap32(lo, hi); return lo + hi; }
    else if constexpr (BIT == 4) { lane_swap16(lo, hi); return lo + hi; }
    else {
        const float keep = bit ? hi : lo, send = bit ? lo : hi;
        if constexpr (BIT == 3) return keep + dpp_move<0x128>(send);                                              // row_ror:8
        else if constexpr (BIT == 2) return keep + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, send), 0x101F));   // lane ^ 4
        else if constexpr (BIT == 1) return keep + dpp_move<0x4E>(send);                                          // quad_perm [2,3,0,1]
        else return keep + dpp_move<0xB1>(send);                                                                   // quad_perm [1,0,3,2]
    }
}
// v[0 .. N): per-lane values. Returns with v[0] (and v[1] for N > 64) holding the 64-lane total of value
// k = 64 * r + bitreverse6(lane): six levels, n -> ceil(n / 2) registers each, order of additions fixed.
template <int N>
__device__ __forceinline__ void lane_transpose_sum(float (&v)[N], int lane) {
    static_assert(N <= 128, "two result registers at most");
    constexpr int n1 = (N + 1) / 2, n2 = (n1 + 1) / 2, n3 = (n2 + 1) / 2, n4 = (n3 + 1) / 2, n5 = (n4 + 1) / 2, n6 = (n5 + 1) / 2;
#pragma unroll
    for (int o = 0; o < n1; o++) v[o] = fold_pair<5>(v[2 * o], 2 * o + 1 < N ? v[2 * o + 1] : 0.f, lane & 32);
#pragma unroll
    for (int o = 0; o < n2; o++) v[o] = fold_pair<4>(v[2 * o], 2 * o + 1 < n1 ? v[2 * o + 1] : 0.f, lane & 16);
#pragma unroll
    for (int o = 0; o < n3; o++) v[o] = fold_pair<3>(v[2 * o], 2 * o + 1 < n2 ? v[2 * o + 1] : 0.f, lane & 8);
#pragma unroll
    for (int o = 0; o < n4; o++) v[o] = fold_pair<2>(v[2 * o], 2 * o + 1 < n3 ? v[2 * o + 1] : 0.f, lane & 4);
#pragma unroll
    for (int o = 0; o < n5; o++) v[o] = fold_pair<1>(v[2 * o], 2 * o + 1 < n4 ? v[2 * o + 1] : 0.f, lane & 2);
#pragma unroll
    for (int o = 0; o < n6; o++) v[o] = fold_pair<0>(v[2 * o], 2 * o + 1 < n5 ? v[2 * o + 1] : 0.f, lane & 1);
}

// one tap row of a pixel: I at (ix, ix + 1) and (gx, gy) at (ix, ix + 1)
struct ColTaps {
    f32x2_a4 i;
    f32x4_a8 g;
};
// a row in flight besides its taps: source coordinate, its fractional part, 1/w, the template sample
struct ColRow {
    f32x2 s, frac;
    float rw, tval;
};
// The per-wave LDS ring of frame-0 rows (see run_ring in the kernel): LK + 1 slots of LW pixels — LW floats of I, then
// LW (gx, gy) pairs — and a small ring of template rows. 4 waves x 9 232 B = 36.9 KB per workgroup, four workgroups per CU.
constexpr int LW = 76;                        // window width in pixels (64 + the spread of a strip's source columns)
constexpr int LROW = LW * 12;                 // bytes per slot
constexpr int LG = LW * 4;                    // offset of the (gx, gy) pairs inside a slot
constexpr int LK = 8;                         // rows in the ring (a power of two); slot LK duplicates slot 0's successor role
constexpr int LT = 4;                         // template rows in flight
constexpr int LWAVE = (LK + 1) * LROW + LT * 256;
static_assert(4 * (4 * LWAVE + 4 * 66 * 8) <= 160 * 1024, "four workgroups (rings + the block reduction's 4 x 66 doubles) must fit a CU's 160 KB of LDS");
static_assert(LROW / 4 + 1 < 256 && LROW / 8 + 1 < 256, "the lower tap row is addressed through ds_read2's 8-bit offset");

struct ColBlend {              // the bilinear samples of a pixel: I, (gx, gy)
    float Iw;
    f32x2 gw;
};

// MOTION: the homography runs the factorised accumulation described above; translation / euclidean / affine have 15 /
// 21 / 45 sums, nothing to factorise, and keep one plain accumulator per sum (the Jacobian is formed per pixel from the
// lane's constant X and the row's Y) — same strips, same ring, same fold. Their warps carry (0, 0, 1) in the last row, so
// the projective coordinate code serves them unchanged (1/w is exactly 1).
template <int MOTION>
__global__ __launch_bounds__(256, STK_COL_WG) void ecc_iter_col_kernel(EccIterArgs a) {
    constexpr bool HOMOGRAPHY = MOTION == STK_MOTION_HOMOGRAPHY;
    constexpr int P = MotionTraits<MOTION>::P, NH = P * (P + 1) / 2, NS = NH + 3 * P + 6;
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
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    __shared__ __attribute__((aligned(16))) char ring_all[4 * LWAVE];
    char* const ring = ring_all + wave * LWAVE;          // this wave's ring; nothing in it is shared between waves

    const int rs = a.ref.stride;
    const int corner = REF_PAD * rs + REF_PAD;
    const char* __restrict__ Ib = reinterpret_cast<const char*>(a.ref.I - corner);
    const char* __restrict__ Gb = reinterpret_cast<const char*>(a.ref.gxy - 2 * (size_t)corner);
    const char* __restrict__ Ib1 = Ib + (size_t)rs * 4;
    const char* __restrict__ Gb1 = Gb + (size_t)rs * 8;

    // accumulators of the current strip (f32, per lane)
    f32x2 hq[3][3];                  // [products (aa,bb) (at,bt) (ab,tt)][power of Y]
    f32x2 m0ab[3], m1ab[3];          // (a.w, b.w) for w = u, v, m; m1: times Y
    f32x2 m0t, m1t;                  // (t.u, t.v)
    float m0tm, m1tm;                // t.m
    float s_mf, s_x;                 // sum m, sum um.v
    f32x2 s_uv, s_sq;                // (sum um, sum v), (sum um.u, sum v.v)
    float accp[HOMOGRAPHY ? 1 : NS];  // the other motions: one accumulator per sum, in the order of the partials
    auto clear = [&]() {
        if constexpr (HOMOGRAPHY) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                hq[k][0] = hq[k][1] = hq[k][2] = bc2(0.f);
                m0ab[k] = m1ab[k] = bc2(0.f);
            }
            m0t = m1t = s_uv = s_sq = bc2(0.f);
            m0tm = m1tm = s_mf = s_x = 0.f;
        } else {
#pragma unroll
            for (int k = 0; k < NS; k++) accp[k] = 0.f;
        }
    };
    clear();
    double dacc0 = 0.0, dacc1 = 0.0;  // lane L: totals of sum number bitreverse6(L) and 64 + bitreverse6(L)

    // this wave's run of (column, row) units, column-major
    const int g = region * 4 + wave;
    int u = g * a.units_q + min(g, a.units_r);
    const int uend = u + a.units_q + (g < a.units_r ? 1 : 0);
    bool no_ring = false;                     // set for one pass of the loop: the strip failed the ring's run-time check
    while (u < uend) {
        const int u_strip = u;
        const int col = u / a.th;
        const int y0 = u - col * a.th;
        const int y1 = min(a.th, y0 + (uend - u));
        u += y1 - y0;
        const int x = col * 64 + lane;
        const bool active = x < a.tw;
        const int xc = min(x, a.tw - 1);
        unsigned xoff = (unsigned)xc << 2;
        const float fx = (float)xc;
        const f32x2 colXY = pk_fma(f32x2{c.m0, c.m3}, bc2(fx), f32x2{c.m2, c.m5});
        const float colW = __builtin_fmaf(c.m6, fx, c.m8);    // m22 == 1 is guaranteed by the launcher (den == w)

        // all four corners of the strip at least 0.05 px inside [0, W-1] x [0, H-1] and w >= 1/4 there:
        // every pixel of the strip is inside the mask and no tap leaves the image
        bool fast = col * 64 + 63 < a.tw;
        float cpx[4], cpy[4];                                   // source coordinates of the corners (x0,y0) (x63,y0) (x0,yl) (x63,yl)
        {
            const float cx[2] = {(float)(col * 64), (float)(col * 64 + 63)}, cy[2] = {(float)y0, (float)(y1 - 1)};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float X = __builtin_fmaf(c.m1, cy[k >> 1], __builtin_fmaf(c.m0, cx[k & 1], c.m2));
                const float Y = __builtin_fmaf(c.m4, cy[k >> 1], __builtin_fmaf(c.m3, cx[k & 1], c.m5));
                const float W = __builtin_fmaf(c.m7, cy[k >> 1], __builtin_fmaf(c.m6, cx[k & 1], c.m8));
                const float r = __builtin_amdgcn_rcpf(W);
                const float px = X * r, py = Y * r;
                cpx[k] = px; cpy[k] = py;
                fast = fast & (W >= 0.25f) & (px >= 0.05f) & (px <= c.mxw - 0.05f) & (py >= 0.05f) & (py <= c.mxh - 0.05f);
            }
        }
        fast = __builtin_amdgcn_readfirstlane((int)fast) != 0;
        // The ring path needs more: the strip's source columns inside a window of LW pixels, the source row rising by
        // 0.6 .. 1.4 per template row (at most two new rows per step, five rows of lookahead suffice) and differing by
        // less than a row across the 64 lanes (the ring holds LK rows). ONE pixel / row of guard on every side: the
        // bounds come from the strip's corners and from lanes 0 and 63 of each row, and an interior lane's coordinate can
        // round across an integer that the end lanes' do not reach (seen: one frame of 255 differing in 3 % of the runs,
        // a lane reading a row that was still in flight).
        const float sxmin = __builtin_fminf(__builtin_fminf(cpx[0], cpx[1]), __builtin_fminf(cpx[2], cpx[3]));
        const float sxmax = __builtin_fmaxf(__builtin_fmaxf(cpx[0], cpx[1]), __builtin_fmaxf(cpx[2], cpx[3]));
        const int xb = __builtin_amdgcn_readfirstlane(((int)__builtin_floorf(sxmin) - 2) & ~3);   // window origin, 16-byte aligned
        bool ringable = fast & (a.ring != 0) & !no_ring & ((int)__builtin_floorf(sxmax) + 3 - xb <= LW - 1) &
                        (__builtin_fabsf(cpy[1] - cpy[0]) <= 0.9f) & (__builtin_fabsf(cpy[3] - cpy[2]) <= 0.9f);
        {
            const float n = (float)(y1 - 1 - y0);
            const float d0 = cpy[2] - cpy[0], d1 = cpy[3] - cpy[1];
            ringable = ringable & (y1 - y0 >= 8) & (d0 >= 0.6f * n) & (d0 <= 1.4f * n) & (d1 >= 0.6f * n) & (d1 <= 1.4f * n);
        }
        ringable = __builtin_amdgcn_readfirstlane((int)ringable) != 0;

        // source coordinate of this lane's pixel in row y: (sx, sy), 1/w, floor
        auto coords = [&](float fy, f32x2& sxy, float& rw, f32x2& fl) {
            const f32x2 XY = pk_fma(f32x2{c.m1, c.m4}, bc2(fy), colXY);
            rw = __builtin_amdgcn_rcpf(__builtin_fmaf(c.m7, fy, colW));
            sxy = XY * bc2(rw);                           // hatX = -X'/den and hatY = -Y'/den are exactly -sx, -sy (den == w)
            fl = f32x2{__builtin_floorf(sxy.x), __builtin_floorf(sxy.y)};
        };
        auto blend = [&](const ColRow& co, const ColTaps& top, const ColTaps& bot, ColBlend& bl) {
            const float ax = co.frac.x, ay = co.frac.y;
            // bilinear taps, the vertical blend first: the two taps of a row are adjacent in memory, so the
            // row pairs go through v_pk_* as loaded
            const f32x2 i0 = top.i, i1 = bot.i;
            const f32x2 iv = pk_fma(bc2(ay), i1 - i0, i0);
            bl.Iw = __builtin_fmaf(ax, iv.y - iv.x, iv.x);
            f32x2 g0a = top.g.lo, g0b = top.g.hi, g1a = bot.g.lo, g1b = bot.g.hi;
            asm("" : "+v"(g0a), "+v"(g0b));                // (the compiler would re-join the halves and subtract four scalars)
            const f32x2 vl = pk_fma(bc2(ay), g1a - g0a, g0a), vr = pk_fma(bc2(ay), g1b - g0b, g0b);
            bl.gw = pk_fma(bc2(ax), vr - vl, vl);                   // (gxw, gyw)
        };
        auto accumulate = [&](auto fast_tag, auto gather_tag, const ColRow& co, const ColBlend& bl, int y) {
            constexpr bool FAST = decltype(fast_tag)::value;
            if constexpr (!FAST) { if (!active) return; }
            if constexpr (!HOMOGRAPHY) {
                const float fy = (float)y, sx = co.s.x, sy = co.s.y, Iw = bl.Iw, gxw = bl.gw.x, gyw = bl.gw.y;
                bool inside = true;
                if constexpr (!FAST) {
                    inside = (sx > 0.0f) & (sx < c.mxw) & (sy > 0.0f) & (sy < c.mxh);
                    if (!inside) {
                        const float rx = __builtin_rintf(sx), ry = __builtin_rintf(sy);
                        inside = (rx >= 0.0f) & (rx <= c.mxw) & (ry >= 0.0f) & (ry <= c.mxh);
                        const bool edge = (__builtin_fabsf(sx + 0.5f) < 0.01f) | (__builtin_fabsf(sx - (c.mxw + 0.5f)) < 0.01f) |
                                          (__builtin_fabsf(sy + 0.5f) < 0.01f) | (__builtin_fabsf(sy - (c.mxh + 0.5f)) < 0.01f);
                        if (edge) inside = nearest_inside_exact<MOTION>(x, y, sl->warp, c.iw, c.ih);
                    }
                }
                const float mf = inside ? 1.0f : 0.0f;
                float J[P];
                if constexpr (MOTION == STK_MOTION_AFFINE) {
                    J[0] = gxw * fx; J[1] = gyw * fx; J[2] = gxw * fy; J[3] = gyw * fy; J[4] = gxw; J[5] = gyw;
                } else if constexpr (MOTION == STK_MOTION_EUCLIDEAN) {
                    const float ex = -(fx * c.m3) - (fy * c.m0);     // h0 = m00 (cos), h1 = m10 (sin)
                    const float ey = (fx * c.m0) - (fy * c.m3);
                    J[0] = gxw * ex + gyw * ey; J[1] = gxw; J[2] = gyw;
                } else {
                    J[0] = gxw; J[1] = gyw;
                }
                const float u = inside ? Iw - c.cI : Iw;
                const float v = inside ? co.tval - c.cT : 0.0f;
                int idx = 0;
#pragma unroll
                for (int k = 0; k < P; k++)
#pragma unroll
                    for (int l = k; l < P; l++) { accp[idx] = __builtin_fmaf(J[k], J[l], accp[idx]); idx++; }
#pragma unroll
                for (int k = 0; k < P; k++) {
                    accp[NH + k] = __builtin_fmaf(J[k], u, accp[NH + k]);
                    accp[NH + P + k] = __builtin_fmaf(J[k], v, accp[NH + P + k]);
                    accp[NH + 2 * P + k] = __builtin_fmaf(J[k], mf, accp[NH + 2 * P + k]);
                }
                const float um = u * mf;
                accp[NH + 3 * P + 0] += mf;
                accp[NH + 3 * P + 1] += um;
                accp[NH + 3 * P + 2] = __builtin_fmaf(um, u, accp[NH + 3 * P + 2]);
                accp[NH + 3 * P + 3] += v;
                accp[NH + 3 * P + 4] = __builtin_fmaf(v, v, accp[NH + 3 * P + 4]);
                accp[NH + 3 * P + 5] = __builtin_fmaf(um, v, accp[NH + 3 * P + 5]);
                return;
            }
            // (Y, Y^2) as a real register pair: a broadcast half-pair would leave its other half to the register
            // allocator, and when that is the target of a load in flight the compiler waits for the load
            const float fy = (float)y;
            f32x2 fyv = {fy, fy * fy};
            if constexpr (decltype(gather_tag)::value) asm("" : "+v"(fyv));
            const f32x2 FY = bc2(fyv.x), FYY = bc2(fyv.y);
            const f32x2 sxy = co.s;
            const float rw = co.rw, Iw = bl.Iw;
            const f32x2 gw = bl.gw;
            const f32x2 jab = gw * bc2(rw);                       // (ja, jb)
            const f32x2 sj = sxy * jab;
            const f32x2 JT = -sj - f32x2{sj.y, sj.x};               // hatX*ja + hatY*jb, in both halves
            const float jt = JT.x;
            const f32x2 P0 = jab * jab, P1 = jab * JT, P2 = {mul_opaque(jab.x, jab.y), mul_opaque(jt, jt)};
            hq[0][0] += P0; hq[0][1] = pk_fma(P0, FY, hq[0][1]); hq[0][2] = pk_fma(P0, FYY, hq[0][2]);
            hq[1][0] += P1; hq[1][1] = pk_fma(P1, FY, hq[1][1]); hq[1][2] = pk_fma(P1, FYY, hq[1][2]);
            hq[2][0] += P2; hq[2][1] = pk_fma(P2, FY, hq[2][1]); hq[2][2] = pk_fma(P2, FYY, hq[2][2]);
            const f32x2 cuv = f32x2{Iw, co.tval} - f32x2{c.cI, c.cT};   // centred samples
            f32x2 uv, umv, Am;
            float tm;
            if constexpr (FAST) { uv = cuv; umv = cuv; Am = jab; tm = jt; }
            else {
                const float sx = sxy.x, sy = sxy.y;
                bool inside = (sx > 0.0f) & (sx < c.mxw) & (sy > 0.0f) & (sy < c.mxh);
                if (!inside) {
                    const float rx = __builtin_rintf(sx), ry = __builtin_rintf(sy);
                    inside = (rx >= 0.0f) & (rx <= c.mxw) & (ry >= 0.0f) & (ry <= c.mxh);
                    const bool edge = (__builtin_fabsf(sx + 0.5f) < 0.01f) | (__builtin_fabsf(sx - (c.mxw + 0.5f)) < 0.01f) |
                                      (__builtin_fabsf(sy + 0.5f) < 0.01f) | (__builtin_fabsf(sy - (c.mxh + 0.5f)) < 0.01f);
                    if (edge) inside = nearest_inside_exact<MOTION>(x, y, sl->warp, c.iw, c.ih);
                }
                const float mf = inside ? 1.0f : 0.0f;
                uv = f32x2{inside ? cuv.x : Iw, inside ? cuv.y : 0.0f};
                umv = f32x2{uv.x * mf, uv.y};
                Am = jab * bc2(mf); tm = jt * mf;
                s_mf += mf;
            }
            const f32x2 Au = jab * bc2(uv.x), Av = jab * bc2(uv.y), Tuv = JT * uv;
            m0ab[0] += Au; m0ab[1] += Av; m0ab[2] += Am; m0t += Tuv; m0tm += tm;
            m1ab[0] = pk_fma(Au, FY, m1ab[0]); m1ab[1] = pk_fma(Av, FY, m1ab[1]); m1ab[2] = pk_fma(Am, FY, m1ab[2]);
            m1t = pk_fma(Tuv, FY, m1t); m1tm = __builtin_fmaf(tm, fy, m1tm);
            s_uv += umv; s_sq = pk_fma(umv, uv, s_sq); s_x = __builtin_fmaf(umv.x, uv.y, s_x);
        };
        auto run = [&](auto fast_tag) {
            constexpr bool FAST = decltype(fast_tag)::value;
            // Stage A of row y: coordinates, then the loads of the template sample and the 2 x 2 taps of the three planes
            // (five load instructions; what was tried instead is listed in DESIGN.md 4.1).
            auto issue = [&](int y, ColRow& co, ColTaps& top, ColTaps& bot) {
                const int yc = min(y, y1 - 1);                 // past the strip end: a harmless repeat, never used
                asm volatile("" : "+v"(xoff));                 // keeps (row base) + (lane offset) in the saddr + voffset form
                co.tval = *(const float*)((const char*)(T + (size_t)yc * a.templ_row_stride) + xoff);
                f32x2 fl;
                coords((float)yc, co.s, co.rw, fl);
                co.frac = co.s - fl;
                if constexpr (!FAST) {
                    // clamp into the zero border with one v_med3_f32 each; NaN -> -2 (all taps zero)
                    fl = f32x2{__builtin_amdgcn_fmed3f(fl.x, -2.0f, c.fiw), __builtin_amdgcn_fmed3f(fl.y, -2.0f, c.fih)};
                }
                const int ix = (int)fl.x, iy = (int)fl.y;
                const unsigned bo = (unsigned)(__mul24(iy, rs) + ix + corner) << 2;     // byte offset of the upper-left tap in the I plane
                top.i = *(const f32x2_a4*)(Ib + bo); bot.i = *(const f32x2_a4*)(Ib1 + bo);
                top.g = *(const f32x4_a8*)(Gb + 2u * bo); bot.g = *(const f32x4_a8*)(Gb1 + 2u * bo);
            };
            // Two rows in flight: the loads of row y+1 are issued before the arithmetic of row y.
            ColTaps ta, tb, ua, ub;
            ColRow ca, cb;
            ColBlend bl;
            issue(y0, ca, ta, ua);
            for (int y = y0; y < y1; y += 2) {
                issue(y + 1, cb, tb, ub);
                blend(ca, ta, ua, bl); accumulate(fast_tag, std::true_type{}, ca, bl, y);
                issue(y + 2, ca, ta, ua);
                if (y + 1 < y1) { blend(cb, tb, ub, bl); accumulate(fast_tag, std::true_type{}, cb, bl, y + 1); }
            }
            if constexpr (FAST && HOMOGRAPHY) s_mf += (float)(y1 - y0);
        };
        // ---- frame-0 rows through a per-wave LDS ring ----
        // The loop above is bound by the L1's tag pipeline: the 2 x 2 taps of a wave are overlapping 8- and 16-byte
        // pieces, 81 tag look-ups per row of 64 pixels against one look-up per clock (TCP_TOTAL_CACHE_ACCESSES, DESIGN.md
        // 4.1). Here every frame-0 row segment the strip needs is fetched ONCE, as aligned 16-byte pieces, by LDS-DMA
        // (global_load_lds_dwordx4: no registers held while in flight) into a ring of LK rows private to the wave — no
        // barrier anywhere — and the taps are ds_read2 with per-lane addresses; the template sample comes the same way
        // through a four-row ring. 30 tag look-ups per row instead of 81.
        // Pipeline per template row y: wait until only the previous step's transfers are in flight -> coordinates and tap
        // reads of row y+1 -> DMA of the frame-0 rows row y+3 will read and of template row y+3 -> arithmetic of row y.
        auto run_ring = [&]() {
            const float* const gI = a.ref.I + xb;
            const float* const gG = a.ref.gxy + 2 * (ptrdiff_t)xb;
            const unsigned l16 = (unsigned)lane * 16u;
            const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;   // the ring's LDS byte address
            // LDS-DMA in inline assembly (m0 = LDS address of the wave's first lane, 16 or 4 bytes per lane): the builtin makes
            // the compiler wait for EVERY outstanding transfer before any LDS read, which caps the prefetch at one row.
            // Here the waits are explicit (wait_keep); the "memory" clobber and the ring operand keep the LDS reads on
            // their side of each transfer and wait. The loader's state is a handful of running scalars (next row's
            // global addresses, its slot), because the scalar unit is shared by the CU's four SIMDs and every
            // instruction of this bookkeeping competes with the other waves' (65 scalar instructions per row in the
            // first version: the scalar unit was 72 % busy).
            const char* pI = nullptr;                            // frame-0 row loaded + 1: I at column xb, (gx, gy) at column xb
            const char* pG = nullptr;
            unsigned dst = ring_lds;                             // LDS address of slot (loaded + 1) % LK
            int issued = 0;                                      // transfers issued in the current step
            int loaded = 0;                                      // last frame-0 row in the ring (or on its way)
            const unsigned long long lanesI = (1ull << (LW / 4)) - 1, lanesG = (1ull << (LW / 2)) - 1;
            auto dma_row = [&]() {                               // the next frame-0 row, columns xb .. xb + LW - 1, into its slot
                unsigned long long saved;
                if (dst == ring_lds) {                           // slot 0 ... and behind the last slot, so that "the row below" is always the next slot
                    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %2\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
                                 "s_add_u32 m0, %1, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
                                 "s_mov_b64 exec, %3\n\ts_add_u32 m0, %1, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %6\n\t"
                                 "s_add_u32 m0, %1, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %6\n\ts_mov_b64 exec, %0"
                                 : "=&s"(saved) : "s"(dst), "s"(lanesI), "s"(lanesG), "v"(l16), "s"(pI), "s"(pG),
                                   "n"(LK * LROW), "n"(LG), "n"(LK * LROW + LG), "r"(ring) : "memory", "m0", "scc");
                    issued += 4;
                } else {
                    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %2\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
                                 "s_mov_b64 exec, %3\n\ts_add_u32 m0, %1, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %6\n\ts_mov_b64 exec, %0"
                                 : "=&s"(saved) : "s"(dst), "s"(lanesI), "s"(lanesG), "v"(l16), "s"(pI), "s"(pG), "n"(LG), "r"(ring)
                                 : "memory", "m0", "scc");
                    issued += 2;
                }
                pI += (size_t)rs * 4; pG += (size_t)rs * 8;
                dst = dst + LROW == ring_lds + LK * LROW ? ring_lds : dst + LROW;
                loaded++;
            };
            const char* pT = nullptr;                            // template row of the next dma_templ, this strip's first column
            unsigned dstT = ring_lds + (LK + 1) * LROW;
            int yT = 0;
            auto dma_templ = [&]() {                             // the next template row (the last one repeats), this lane's pixel
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2"
                             : : "s"(dstT), "v"(xoff), "s"(pT), "r"(ring) : "memory", "m0");
                issued += 1;
                if (yT < y1 - 1) pT += (size_t)a.templ_row_stride * 4;
                yT++;
                dstT = ring_lds + (LK + 1) * LROW + (unsigned)(yT & (LT - 1)) * 256u;
            };
            // wait until only the `keep` newest transfers are in flight (they complete in order); keep is 1 + 2 * rows (+ 2)
            auto wait_keep = [&](int keep) {
                const int k = __builtin_amdgcn_readfirstlane(keep);
                if (k == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                else if (k == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else if (k == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            };
            // coordinates of row y, its taps and template sample out of the ring; returns the lanes' extreme source rows
            auto fetch = [&](int y, ColRow& co, ColTaps& top, ColTaps& bot, int& ilo, int& ihi) {
                const int yc = min(y, y1 - 1);
                f32x2 fl;
                coords((float)yc, co.s, co.rw, fl);
                co.frac = co.s - fl;
                const int ix = (int)fl.x, iy = (int)fl.y;
                const int i0 = __builtin_amdgcn_readlane(iy, 0), i1 = __builtin_amdgcn_readlane(iy, 63);
                ilo = min(i0, i1); ihi = max(i0, i1);
                const int so = __mul24(iy & (LK - 1), LROW), dx = ix - xb;
                const float* const pi = (const float*)(ring + so + dx * 4);
                const f32x2_a4* const pg = (const f32x2_a4*)(ring + so + LG + dx * 8);
                top.i = f32x2_a4{pi[0], pi[1]}; bot.i = f32x2_a4{pi[LROW / 4], pi[LROW / 4 + 1]};
                top.g.lo = pg[0]; top.g.hi = pg[1]; bot.g.lo = pg[LROW / 8]; bot.g.hi = pg[LROW / 8 + 1];
                co.tval = ((const float*)(ring + (LK + 1) * LROW + (yc & (LT - 1)) * 256))[lane];
            };
            ColTaps ta, ua, tb, ub;
            ColRow ca, cb;
            ColBlend bl;
            int ilo, ihi;
            {   // fill: the rows of the first template row plus the lookahead, three template rows
                f32x2 s0, fl0; float rw0;
                coords((float)y0, s0, rw0, fl0);
                const int iy = (int)fl0.y;
                const int i0 = __builtin_amdgcn_readlane(iy, 0), i1 = __builtin_amdgcn_readlane(iy, 63);
                loaded = min(i0, i1) - 2;                         // the first row loaded is the guard row below the lowest one
                pI = (const char*)(gI + (ptrdiff_t)(loaded + 1) * rs); pG = (const char*)(gG + 2 * (ptrdiff_t)(loaded + 1) * rs);
                dst = ring_lds + (unsigned)((loaded + 1) & (LK - 1)) * LROW;
                const int want = max(i0, i1) + a.ring_lookahead;
                while (loaded < want) dma_row();
                pT = (const char*)(T + (size_t)y0 * a.templ_row_stride); yT = y0;
                dstT = ring_lds + (LK + 1) * LROW + (unsigned)(yT & (LT - 1)) * 256u;
                dma_templ(); dma_templ(); dma_templ();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                fetch(y0, ca, ta, ua, ilo, ihi);
            }
            // A transfer issued in step y has landed when step y+2 starts (wait_keep leaves only step y+1's in flight),
            // so step y fetches what row y+3 will read: frame-0 rows up to ihi(y+1) + 5 (the source row rises by at most
            // 1.4 per template row: ihi(y+3) + 1 for the lower taps + 1 of guard <= ihi(y+1) + 5) and template row y+3.
            // Nothing a later fetch needs is overwritten: the rows in flight reach back to ihi(y+1) - 3 at most, a fetch
            // reads from ilo - 1 (guard) and the end lanes of a row are at most one row apart.
            // The bounds that make this safe were derived from the strip's corners before the loop; they are also CHECKED, on
            // scalars, row by row: what fetch() is about to read must have landed (`safe`) and must not have been
            // overwritten by anything issued since. On a violation run_ring returns true: the strip's accumulators are
            // private to the strip, so the caller clears them and redoes the strip through the gather loop (bit-identical
            // by construction) and counts the event (stk_timing.ecc_ring_fallbacks; 0 on every BASELINE stack).
            int prev_issued = 1;                                 // (nothing is in flight before the first step)
            int safe = loaded, before_prev = loaded, violated = 0;
            auto step = [&](ColRow& cur, ColTaps& tcur, ColTaps& ucur, ColRow& nxt, ColTaps& tnxt, ColTaps& unxt, int y) {
                wait_keep(prev_issued);
                safe = before_prev;                               // everything issued before the previous step's transfers has landed
                fetch(y + 1, nxt, tnxt, unxt, ilo, ihi);
                violated |= (safe - (ihi + 2)) | (ilo - 1 + LK - 1 - loaded);   // a sign bit: read (guard row included) before it landed / after it was overwritten
                before_prev = loaded;
                issued = 0;
                if (loaded < ihi + a.ring_lookahead) dma_row();
                if (loaded < ihi + a.ring_lookahead) dma_row();
                dma_templ();
                prev_issued = issued;
                blend(cur, tcur, ucur, bl); accumulate(std::true_type{}, std::false_type{}, cur, bl, y);
            };
            for (int y = y0; y < y1; y += 2) {
                step(ca, ta, ua, cb, tb, ub, y);
                if (y + 1 < y1) step(cb, tb, ub, ca, ta, ua, y + 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // nothing may land in the ring after the strip (it is reused)
            if constexpr (HOMOGRAPHY) s_mf += (float)(y1 - y0);
            return violated < 0;
        };
        if (ringable) {
            if (run_ring()) {
                // the ring's bounds did not hold on this strip (a local source-row rate the corner test cannot see): take
                // the strip again, from the top of the loop, through the gather route
                clear();
                u = u_strip;
                no_ring = true;
                if (lane == 0) atomicAdd(a.ring_fallbacks, 1);
                continue;
            }
        } else if (fast) run(std::true_type{}); else run(std::false_type{});
        no_ring = false;

        // flush the strip: apply the powers of X, sum over the 64 lanes, add to the f64 totals
        {
            float v[NS];
            if constexpr (!HOMOGRAPHY) {
#pragma unroll
                for (int k = 0; k < NS; k++) v[k] = accp[k];
            } else {
            const float fxx = fx * fx;
            int idx = 0;
#pragma unroll
            for (int i = 0; i < P; i++)
#pragma unroll
                for (int j = i; j < P; j++) {
                    const int ci = i < 6 ? i % 3 : i - 6, cj = j < 6 ? j % 3 : j - 6;      // 0 a, 1 b, 2 t
                    const int lo = ci < cj ? ci : cj, hi = ci < cj ? cj : ci;
                    // (aa,bb) -> hq[0], (at,bt) -> hq[1], (ab,tt) -> hq[2]
                    const int reg = lo == hi ? (lo == 2 ? 2 : 0) : (hi == 2 ? 1 : 2);
                    const int half = lo == hi ? (lo == 0 ? 0 : 1) : (hi == 2 ? lo : 0);
                    const int xpow = (i < 3) + (j < 3), ypow = (i >= 3 && i < 6) + (j >= 3 && j < 6);
                    const float m = hq[reg][ypow][half];
                    v[idx++] = xpow == 0 ? m : xpow == 1 ? m * fx : m * fxx;
                }
#pragma unroll
            for (int w = 0; w < 3; w++)
#pragma unroll
                for (int i = 0; i < P; i++) {
                    const int ci = i < 6 ? i % 3 : i - 6;
                    const float z0 = ci == 2 ? (w == 2 ? m0tm : m0t[w]) : m0ab[w][ci];
                    const float z1 = ci == 2 ? (w == 2 ? m1tm : m1t[w]) : m1ab[w][ci];
                    v[NH + w * P + i] = i < 3 ? z0 * fx : i < 6 ? z1 : z0;
                }
            v[NH + 3 * P + 0] = s_mf; v[NH + 3 * P + 1] = s_uv.x; v[NH + 3 * P + 2] = s_sq.x;
            v[NH + 3 * P + 3] = s_uv.y; v[NH + 3 * P + 4] = s_sq.y; v[NH + 3 * P + 5] = s_x;
            }
            lane_transpose_sum<NS>(v, lane);
            dacc0 += (double)v[0];
            if constexpr (NS > 64) dacc1 += (double)v[1];
            clear();
        }
    }

    __shared__ double red[4][NS];
    const int k0 = (int)(__builtin_bitreverse32((unsigned)lane) >> 26);
    if (k0 < NS) red[wave][k0] = dacc0;
    if (k0 + 64 < NS) red[wave][k0 + 64] = dacc1;
    __syncthreads();
    if (threadIdx.x < NS) {
        const int k = threadIdx.x;
        const double s = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
        a.partials[((size_t)slot * NS + k) * a.nb + region] = s;   // [slot][sum][block]
    }
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
