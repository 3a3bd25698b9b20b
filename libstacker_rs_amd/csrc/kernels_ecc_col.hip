// kernels_ecc_col.hip — the column-walking iteration pass of findTransformECC (lib.rs:769-777; algorithm SURVEY.md
// 8a-E*), every motion model. Same moment sums and the same partials layout as the direct kernels in kernels_ecc.hip.
// Compiled with -fno-slp-vectorize (Makefile): the arithmetic is written operation by operation and must stay that way
// (a packed f32 instruction costs the SIMD two plain ones, and the vectoriser's pairings cost register shuffles on top).
#include "ecc_pixel.h"
#include <type_traits>

namespace stk {

#ifndef STK_COL_WG
#define STK_COL_WG 4
#endif
#ifndef STK_COL_RELAXED
#define STK_COL_RELAXED 1       // 1: a template row may stay in flight for three steps (wait_keep); 0: two, like a frame-0 row
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
//   * (Round 2 wrote everything that comes in pairs as v_pk_* on float2; round 4 measured that a packed instruction
//     occupies the SIMD exactly as long as the two plain ones it replaces, see below.)
//   * A strip whose four corners map well inside the frame-0 image (the image of a convex set under a homography with
//     w > 0 is convex) takes a loop without the mask: no compares, no selects, no clamps — bit-identical results to
//     the masked loop, because with m = 1 every masked expression reduces to the unmasked one exactly.
//   * ONE cross-lane reduction per strip instead of one per row plus one per block: 66 per-lane values go through a
//     lane-transposing fold (each level halves the number of registers: v_permlane32_swap / v_permlane16_swap for lane
//     bits 5 and 4, DPP row_ror / quad_perm and ds_swizzle below), ~200 instructions instead of ~460 + 2 x 144.
// Work units are (column strip, row) pairs in column-major order, split evenly over the 4 x nb waves of the frame, so a
// frame's summation partition still depends on its size only (shard-invariant bits).
//
// Round 4: what an instruction costs was measured (tools/valu_rates.hip, profiles/r04/valu_rates.txt) instead of counted.
// With two or more waves per SIMD a plain f32 add / mul / fma occupies the SIMD for ~2.2 cycles and EVERY v_pk_*_f32 for
// ~4.2: pairing never bought throughput on this chip, only the moves that build the pairs, so the per-pixel arithmetic
// is written out operation by operation again (the same operations, the same bits). And the pass turned out to be bound
// as much by the SCALAR unit as by the vector pipe (filler probes: a scalar instruction per row costs what a vector
// instruction does; 50 scalar instructions + 8 branches per row against 97 vector ones): the ring's bookkeeping is what
// this version cuts — one LDS-DMA per frame-0 row from an interleaved (I, gx, gy) plane with every lane active (no exec
// juggling, half the pointer arithmetic), template slots fixed at compile time by unrolling four rows, one compare on the
// common path of the counted wait, no clamps of the row index (the strip's last row is peeled off).
// ---------------------------------------------------------------------------------------------------
// (the LDS-DMA blocks below set m0 and say so in their clobber lists; clang warns that m0 is a reserved register)
#pragma clang diagnostic ignored "-Winline-asm"

typedef float f32x2 __attribute__((ext_vector_type(2)));
// (int)floor(v) in one instruction (v_floor_f32 + v_cvt_i32_f32 are two half-rate instructions)
__device__ __forceinline__ int floor_to_int(float v) { int r; asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(v)); return r; }

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

// one tap row of a pixel: (I, gx, gy) at ix and at ix + 1 — six independent registers (a vector type would tie them to
// consecutive ones and cost moves when they arrive from different loads)
struct ColTaps {
    float i0, x0, y0, i1, x1, y1;
};
// a row in flight besides its taps: source coordinate, its fractional part, 1/w
struct ColRow {
    float sx, sy, ax, ay;
    float rw;
};
// The per-wave LDS ring of frame-0 rows (see run_ring in the kernel): LK + 1 slots of one LDS-DMA each — 64 lanes x 16
// bytes of the interleaved (I, gx, gy) plane = 85 1/3 pixels — and a ring of LT template rows. 4 waves x 10 240 B = 40 KB
// per workgroup, four workgroups per CU = all of its 160 KB (the block reduction at the end re-uses the rings).
constexpr int LROW = 1024;                    // bytes per slot
constexpr int LWP = 85;                       // whole pixels in a slot (window width: 64 + the spread of a strip's source columns)
constexpr int LK = 8;                         // rows in the ring (a power of two); slot LK repeats slot 0, so "the row below" is always the next slot
constexpr int LT = 4;                         // template rows in flight
constexpr int LTOFF = (LK + 1) * LROW;        // offset of the template ring
constexpr int LWAVE = LTOFF + LT * 256;
static_assert(4 * 4 * LWAVE <= 160 * 1024, "four workgroups of four rings must fit the 160 KB of LDS of a CU");
static_assert(4 * LWAVE >= 4 * 66 * 8, "the block reduction (4 x 66 doubles) re-uses the rings");
static_assert(LROW == 1 << 10 && LK == 8, "locate() shifts by 10 and masks with LK - 1");

struct ColBlend {              // the bilinear samples of a pixel: I, gx, gy
    float Iw, gxw, gyw;
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
#ifdef STK_COL_PROBE_T0                                              // (timing probe: every slot reads template 0 — the template stream then comes out of the L2)
    const float* __restrict__ T = a.templates;
#else
    const float* __restrict__ T = a.templates + (size_t)frame * a.templ_plane_stride;
#endif
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    __shared__ __attribute__((aligned(16))) char ring_all[4 * LWAVE];
    char* const ring = ring_all + wave * LWAVE;          // this wave's ring; nothing in it is shared between waves

    const int rs = a.ref.stride;
    const int corner = REF_PAD * rs + REF_PAD;
    const char* __restrict__ Ib = reinterpret_cast<const char*>(a.ref.I - corner);
    const char* __restrict__ Gb = reinterpret_cast<const char*>(a.ref.gxy - 2 * (size_t)corner);

    // accumulators of the current strip (f32, per lane)
    float hs[6][3];                  // homography: [product aa bb at bt ab tt][power of Y]
    float ms0[9], ms1[9];            // [3 w + c]: c.w for c = a, b, t and w = u, v, m; ms1: times Y
    float s_mf, s_x, s_u, s_v, s_uu, s_vv;
    float accp[HOMOGRAPHY ? 1 : NS];  // the other motions: one accumulator per sum, in the order of the partials
    auto clear = [&]() {
        if constexpr (HOMOGRAPHY) {
#pragma unroll
            for (int k = 0; k < 6; k++) hs[k][0] = hs[k][1] = hs[k][2] = 0.f;
#pragma unroll
            for (int k = 0; k < 9; k++) ms0[k] = ms1[k] = 0.f;
            s_mf = s_x = s_u = s_v = s_uu = s_vv = 0.f;
        } else {
#pragma unroll
            for (int k = 0; k < NS; k++) accp[k] = 0.f;
        }
    };
    clear();
    // lane L: f64 totals of sum number bitreverse6(L) and 64 + bitreverse6(L) over the wave's strips. They live in private
    // memory (volatile: two loads and two stores per strip), not in four registers the row loop has no room for: the
    // compiler's own choice of what to spill landed inside that loop, where a scratch load stalls the LDS-DMA pipeline.
    volatile double dacc[2];
    dacc[0] = 0.0; dacc[1] = 0.0;

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
        const float colX = __builtin_fmaf(c.m0, fx, c.m2), colY = __builtin_fmaf(c.m3, fx, c.m5);
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
        // The ring path needs more: the strip's source columns inside a window of LWP pixels, the source row rising by
        // 0.6 .. 1.4 per template row (at most two new rows per step, five rows of lookahead suffice) and differing by
        // less than a row across the 64 lanes (the ring holds LK rows). ONE pixel / row of guard on every side: the
        // bounds come from the strip's corners and from lanes 0 and 63 of each row, and an interior lane's coordinate can
        // round across an integer that the end lanes' do not reach (seen: one frame of 255 differing in 3 % of the runs,
        // a lane reading a row that was still in flight).
        const float sxmin = __builtin_fminf(__builtin_fminf(cpx[0], cpx[1]), __builtin_fminf(cpx[2], cpx[3]));
        const float sxmax = __builtin_fmaxf(__builtin_fmaxf(cpx[0], cpx[1]), __builtin_fmaxf(cpx[2], cpx[3]));
        const int xb = __builtin_amdgcn_readfirstlane(((int)__builtin_floorf(sxmin) - 2) & ~3);   // window origin: a multiple of 4 pixels = 48 bytes
        bool ringable = fast & (a.ring != 0) & !no_ring & ((int)__builtin_floorf(sxmax) + 3 - xb <= LWP - 1) &
                        (__builtin_fabsf(cpy[1] - cpy[0]) <= 0.9f) & (__builtin_fabsf(cpy[3] - cpy[2]) <= 0.9f);
        {
            const float n = (float)(y1 - 1 - y0);
            const float d0 = cpy[2] - cpy[0], d1 = cpy[3] - cpy[1];
            ringable = ringable & (y1 - y0 >= 8) & (d0 >= 0.6f * n) & (d0 <= 1.4f * n) & (d1 >= 0.6f * n) & (d1 <= 1.4f * n);
        }
        ringable = __builtin_amdgcn_readfirstlane((int)ringable) != 0;

        // source coordinate of this lane's pixel in row Y = fy: (sx, sy), 1/w
        auto coords = [&](float fy, float& sx, float& sy, float& rw) {
            const float X = __builtin_fmaf(c.m1, fy, colX), Y = __builtin_fmaf(c.m4, fy, colY);
            rw = __builtin_amdgcn_rcpf(__builtin_fmaf(c.m7, fy, colW));
            sx = X * rw; sy = Y * rw;                     // hatX = -X'/den and hatY = -Y'/den are exactly -sx, -sy (den == w)
        };
        auto blend = [&](const ColRow& co, const ColTaps& top, const ColTaps& bot, ColBlend& bl) {
            const float ax = co.ax, ay = co.ay;
            // bilinear taps, the vertical blend first (the order the round-2 kernel fixed; every operation on its own)
            const float il = __builtin_fmaf(ay, bot.i0 - top.i0, top.i0), ir = __builtin_fmaf(ay, bot.i1 - top.i1, top.i1);
            bl.Iw = __builtin_fmaf(ax, ir - il, il);
            const float xl = __builtin_fmaf(ay, bot.x0 - top.x0, top.x0), yl = __builtin_fmaf(ay, bot.y0 - top.y0, top.y0);
            const float xr = __builtin_fmaf(ay, bot.x1 - top.x1, top.x1), yr = __builtin_fmaf(ay, bot.y1 - top.y1, top.y1);
            bl.gxw = __builtin_fmaf(ax, xr - xl, xl); bl.gyw = __builtin_fmaf(ax, yr - yl, yl);
        };
        auto accumulate = [&](auto fast_tag, const ColRow& co, const ColBlend& bl, float tval, float fy, int y) {
            constexpr bool FAST = decltype(fast_tag)::value;
            if constexpr (!FAST) { if (!active) return; }
            if constexpr (!HOMOGRAPHY) {
                const float sx = co.sx, sy = co.sy, Iw = bl.Iw, gxw = bl.gxw, gyw = bl.gyw;
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
                const float v = inside ? tval - c.cT : 0.0f;
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
            } else {
                const float fyy = fy * fy;
                const float sx = co.sx, sy = co.sy, rw = co.rw, Iw = bl.Iw;
                const float ja = bl.gxw * rw, jb = bl.gyw * rw;
                const float jt = -(sx * ja) - (sy * jb);            // hatX*ja + hatY*jb
                const float pr[6] = {ja * ja, jb * jb, ja * jt, jb * jt, ja * jb, jt * jt};
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    hs[k][0] += pr[k]; hs[k][1] = __builtin_fmaf(pr[k], fy, hs[k][1]); hs[k][2] = __builtin_fmaf(pr[k], fyy, hs[k][2]);
                }
                const float cu = Iw - c.cI, cv = tval - c.cT;     // centred samples
                float uu, vv, um, mf = 1.0f;
                if constexpr (FAST) { uu = cu; vv = cv; um = cu; }
                else {
                    bool inside = (sx > 0.0f) & (sx < c.mxw) & (sy > 0.0f) & (sy < c.mxh);
                    if (!inside) {
                        const float rx = __builtin_rintf(sx), ry = __builtin_rintf(sy);
                        inside = (rx >= 0.0f) & (rx <= c.mxw) & (ry >= 0.0f) & (ry <= c.mxh);
                        const bool edge = (__builtin_fabsf(sx + 0.5f) < 0.01f) | (__builtin_fabsf(sx - (c.mxw + 0.5f)) < 0.01f) |
                                          (__builtin_fabsf(sy + 0.5f) < 0.01f) | (__builtin_fabsf(sy - (c.mxh + 0.5f)) < 0.01f);
                        if (edge) inside = nearest_inside_exact<MOTION>(x, y, sl->warp, c.iw, c.ih);
                    }
                    mf = inside ? 1.0f : 0.0f;
                    uu = inside ? cu : Iw; vv = inside ? cv : 0.0f;
                    um = uu * mf;
                    s_mf += mf;
                }
                const float w[3] = {uu, vv, mf};
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    // (FAST: mf is the constant 1 and the products with it fold away)
                    const float ca = ja * w[k], cb = jb * w[k], ct = jt * w[k];
                    ms0[3 * k + 0] += ca; ms0[3 * k + 1] += cb; ms0[3 * k + 2] += ct;
                    ms1[3 * k + 0] = __builtin_fmaf(ca, fy, ms1[3 * k + 0]); ms1[3 * k + 1] = __builtin_fmaf(cb, fy, ms1[3 * k + 1]);
                    ms1[3 * k + 2] = __builtin_fmaf(ct, fy, ms1[3 * k + 2]);
                }
                s_u += um; s_v += vv;
                s_uu = __builtin_fmaf(um, uu, s_uu); s_vv = __builtin_fmaf(vv, vv, s_vv); s_x = __builtin_fmaf(um, vv, s_x);
            }
        };
        auto run = [&](auto fast_tag) {
            constexpr bool FAST = decltype(fast_tag)::value;
            // Stage A of row y: coordinates, then the loads of the template sample and the 2 x 2 taps of the three planes
            // (five load instructions; what was tried instead is listed in DESIGN.md 4.1).
            auto issue = [&](int y, ColRow& co, ColTaps& top, ColTaps& bot, float& tval) {
                const int yc = min(y, y1 - 1);                 // past the strip end: a harmless repeat, never used
                asm volatile("" : "+v"(xoff));                 // keeps (row base) + (lane offset) in the saddr + voffset form
                tval = *(const float*)((const char*)(T + (size_t)yc * a.templ_row_stride) + xoff);
                coords((float)yc, co.sx, co.sy, co.rw);
                float flx = __builtin_floorf(co.sx), fly = __builtin_floorf(co.sy);
                co.ax = co.sx - flx; co.ay = co.sy - fly;
                if constexpr (!FAST) {
                    // clamp into the zero border with one v_med3_f32 each; NaN -> -2 (all taps zero)
                    flx = __builtin_amdgcn_fmed3f(flx, -2.0f, c.fiw); fly = __builtin_amdgcn_fmed3f(fly, -2.0f, c.fih);
                }
                const int ix = (int)flx, iy = (int)fly;
                const unsigned bo = (unsigned)(__mul24(iy, rs) + ix + corner) << 2;     // byte offset of the upper-left tap in the I plane
                const unsigned bo1 = bo + ((unsigned)rs << 2);                          // ... and of the lower-left one
                const f32x2_a4 ti = *(const f32x2_a4*)(Ib + bo), bi = *(const f32x2_a4*)(Ib + bo1);
                const f32x4_a8 tg = *(const f32x4_a8*)(Gb + 2u * bo), bg = *(const f32x4_a8*)(Gb + 2u * bo1);
                top = ColTaps{ti.x, tg.x, tg.y, ti.y, tg.z, tg.w};
                bot = ColTaps{bi.x, bg.x, bg.y, bi.y, bg.z, bg.w};
            };
            // Two rows in flight: the loads of row y+1 are issued before the arithmetic of row y.
            ColTaps ta, tb, ua, ub;
            ColRow ca, cb;
            ColBlend bl;
            float va, vb;
            issue(y0, ca, ta, ua, va);
            for (int y = y0; y < y1; y += 2) {
                issue(y + 1, cb, tb, ub, vb);
                blend(ca, ta, ua, bl); accumulate(fast_tag, ca, bl, va, (float)y, y);
                issue(y + 2, ca, ta, ua, va);
                if (y + 1 < y1) { blend(cb, tb, ub, bl); accumulate(fast_tag, cb, bl, vb, (float)(y + 1), y + 1); }
            }
            if constexpr (FAST && HOMOGRAPHY) s_mf += (float)(y1 - y0);
        };
        // ---- frame-0 rows through a per-wave LDS ring ----
        // The gather loop above is bound by the L1's tag pipeline: the 2 x 2 taps of a wave are overlapping 8- and 16-byte
        // pieces, 81 tag look-ups per row of 64 pixels against one look-up per clock (TCP_TOTAL_CACHE_ACCESSES, DESIGN.md
        // 4.1). Here every frame-0 row segment the strip needs is fetched ONCE by LDS-DMA (global_load_lds_dwordx4: no
        // registers held while in flight, every lane active, one instruction per row of the interleaved plane) into a ring
        // of LK rows private to the wave — no barrier anywhere — and the taps are ds_read2_b32 with per-lane addresses;
        // the template sample comes the same way through a four-row ring.
        // Pipeline per template row j (counted from the strip's first row): wait until only the previous step's transfers
        // are in flight -> coordinates and tap reads of row j+1 -> DMA of the frame-0 rows row j+3 will read and of
        // template row j+3 -> arithmetic of row j.
        auto run_ring = [&]() {
            const unsigned l16 = (unsigned)lane * 16u;
            const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;   // the ring's LDS byte address
            // LDS-DMA in inline assembly (m0 = LDS address of the wave's first lane, 16 or 4 bytes per lane): the builtin makes
            // the compiler wait for EVERY outstanding transfer before any LDS read, which caps the prefetch at one row.
            // Here the waits are explicit (wait_keep); the "memory" clobber and the ring operand keep the LDS reads on
            // their side of each transfer and wait. The loader's state is a handful of running scalars, because the scalar
            // unit is shared by the CU's four SIMDs and every instruction of this bookkeeping competes with the other waves'.
            const size_t row_bytes = (size_t)rs * 12;
            const char* pR = nullptr;                            // frame-0 row loaded + 1 of the interleaved plane, at column xb
            unsigned dst = ring_lds;                             // LDS address of slot (loaded + 1) % LK
            int issued = 0;                                      // transfers issued in the current step
            int loaded = 0;                                      // last frame-0 row in the ring (or on its way)
            auto dma_row = [&]() {                               // the next frame-0 row, LROW bytes from column xb on, into its slot
#ifndef STK_COL_PROBE_NO_R                                        // (timing probe: no frame-0 transfers at all — results are garbage)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                             : : "s"(dst), "v"(l16), "s"(pR), "r"(ring) : "memory", "m0");
                issued += 1;
                if (dst == ring_lds) {                           // slot 0 ... and behind the last slot, so that "the row below" is always the next slot
                    asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                                 : : "s"(dst), "v"(l16), "s"(pR), "n"(LK * LROW), "r"(ring) : "memory", "m0", "scc");
                    issued += 1;
                }
#endif
                pR += row_bytes;
                dst = dst + LROW == ring_lds + LK * LROW ? ring_lds : dst + LROW;
                loaded++;
            };
            // One register serves the template ring both ways: tr = LDS byte address of this lane's sample in template slot 0
            // — the ds_read address — and, with the scalar base lowered by the ring's address, the per-lane offset of the DMA.
            const unsigned tr = ring_lds + LTOFF + (unsigned)lane * 4u;
            const char* pT = nullptr;                            // template row of the next dma_templ: pixel (this strip's column 0) minus (tr - 4 lane)
            auto dma_templ = [&](auto slot_tag) {                // the next template row, this lane's pixel, into ring slot SLOT
                constexpr int SLOT = decltype(slot_tag)::value;
#ifndef STK_COL_PROBE_NO_T                                        // (timing probe: no template transfers — results are garbage)
#ifdef STK_COL_PROBE_T14                                          // (timing probe: three template rows of four come from template 0, i.e. out of the L2)
                const char* const src = SLOT == 0 ? pT : pT - (size_t)frame * a.templ_plane_stride * 4;
#else
                const char* const src = pT;
#endif
                asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2"
                             : : "s"(ring_lds), "v"(tr), "s"(src), "n"(LTOFF + SLOT * 256), "r"(ring) : "memory", "m0", "scc");
                issued += 1;
#endif
#ifdef STK_COL_PROBE_TSEQ                                         // (timing probe: strip-major template addresses — a wave's rows are consecutive 256-byte pieces)
                pT += 256;
#else
                pT += (size_t)a.templ_row_stride * 4;            // (rows past the strip's end are fetched and never read: the buffer has the room)
#endif
            };
            typedef __attribute__((address_space(3))) const float lds_f;
            auto templ_sample = [&](auto slot_tag) {             // this lane's sample of the template row in ring slot SLOT
                constexpr int SLOT = decltype(slot_tag)::value;
                return ((lds_f*)(size_t)tr)[SLOT * 64];
            };
            // Wait until only the previous step's transfers AND the template row of the step before it are in flight (they
            // complete in order, and a step issues its template row last): prev = 1 (template row) + frame-0 rows (0, 1, 2)
            // + 1 if one of them went into slot 0 = 1 .. 4, and 2 nine times out of ten. The frame-0 rows come out of the
            // L2 and have two steps to land; the template row comes from HBM, is read three steps after its issue, and
            // gets all three (round 4: with `prev` alone in flight the waves spent a third of their time in this wait).
            auto wait_keep = [&](auto relaxed_tag, int prev) {
                constexpr int X = decltype(relaxed_tag)::value ? 1 : 0;     // the extra transfer allowed in flight
                asm volatile("s_cmp_eq_u32 %0, 2\n\t"
                             "s_cbranch_scc1 2f\n\t"
                             "s_cmp_lt_u32 %0, 2\n\t"
                             "s_cbranch_scc1 1f\n\t"
                             "s_cmp_eq_u32 %0, 3\n\t"
                             "s_cbranch_scc1 3f\n\t"
                             "s_waitcnt vmcnt(%2)\n\t"
                             "s_branch 9f\n"
                             "3:\n\ts_waitcnt vmcnt(%3)\n\t"
                             "s_branch 9f\n"
                             "1:\n\ts_waitcnt vmcnt(%5)\n\t"
                             "s_branch 9f\n"
                             "2:\n\ts_waitcnt vmcnt(%4)\n"
                             "9:" : : "s"(prev), "n"(0), "n"(4 + X), "n"(3 + X), "n"(2 + X), "n"(1 + X) : "memory", "scc");
            };
            // coordinates of the row after `prev` (Y + 1), its taps and its template sample (ring slot SLOT) out of the ring;
            // returns the lanes' extreme source rows
            const unsigned ring_k = ring_lds - (unsigned)xb * 12u;
            // coordinates of the row Y = fy; returns the LDS address of its upper-left tap and the lanes' extreme source rows
            auto locate = [&](float fy, ColRow& co, int& ilo, int& ihi) {
                coords(fy, co.sx, co.sy, co.rw);
                // (inside the image: the coordinates are positive, v_fract_f32 is s - floor(s) exactly)
                co.ax = __builtin_amdgcn_fractf(co.sx); co.ay = __builtin_amdgcn_fractf(co.sy);
                const int ix = floor_to_int(co.sx), iy = floor_to_int(co.sy);
                const int i0 = __builtin_amdgcn_readlane(iy, 0), i1 = __builtin_amdgcn_readlane(iy, 63);
                ilo = min(i0, i1); ihi = max(i0, i1);
                // ring_k + (iy mod LK) * LROW + 12 ix in three instructions (left to itself the compiler takes six)
                unsigned row, at;
                asm("v_lshl_add_u32 %0, %1, 10, %2" : "=v"(row) : "v"(iy & (LK - 1)), "s"(ring_k));
                asm("v_mad_u32_u24 %0, %1, 12, %2" : "=v"(at) : "v"(ix), "v"(row));
                return at;
            };
            // the 2 x 2 taps at LDS address `at`: (I, gx, gy) of pixel ix, then of pixel ix + 1; the row below: + LROW
            auto read_taps = [&](unsigned at, ColTaps& top, ColTaps& bot) {
                unsigned ab = at + LROW;
                asm("" : "+v"(ab));                               // (ONE address for the lower row: 1024 is beyond ds_read2's 8-bit offsets)
                lds_f* const pt = (lds_f*)(size_t)at;
                lds_f* const pb = (lds_f*)(size_t)ab;
                top = ColTaps{pt[0], pt[1], pt[2], pt[3], pt[4], pt[5]};
                bot = ColTaps{pb[0], pb[1], pb[2], pb[3], pb[4], pb[5]};
            };
            // ONE set of tap registers: a row's taps are blended (12 registers -> 3) before the next row's are read into the
            // same registers, and the reads then have the whole accumulation of the current row to land in. (Two sets in
            // flight, as the gather loop keeps them, do not fit beside the 42 accumulators: the compiler spilled lane
            // constants and reloaded them inside this loop, and a scratch load there stalls the DMA pipeline.)
            ColTaps top, bot;
            ColRow ca, cb;
            ColBlend bl;
            int ilo, ihi;
            const int la = a.ring_lookahead;
            {   // fill: the rows of the first template row plus the lookahead, three template rows
                float sx0, sy0, rw0;
                coords((float)y0, sx0, sy0, rw0);
                const int iy = floor_to_int(sy0);
                const int i0 = __builtin_amdgcn_readlane(iy, 0), i1 = __builtin_amdgcn_readlane(iy, 63);
                loaded = min(i0, i1) - 2;                         // the first row loaded is the guard row below the lowest one
                pR = (const char*)(a.ref.igg + 3 * ((ptrdiff_t)(loaded + 1) * rs + xb));
                dst = ring_lds + (unsigned)((loaded + 1) & (LK - 1)) * LROW;
                const int want = max(i0, i1) + la;
                while (loaded < want) dma_row();
#ifdef STK_COL_PROBE_TSEQ
                pT = (const char*)(T + ((size_t)col * a.th + y0) * 64) - (ring_lds + LTOFF);
#else
                pT = (const char*)(T + (size_t)y0 * a.templ_row_stride + col * 64) - (ring_lds + LTOFF);
#endif
                dma_templ(std::integral_constant<int, 0>{}); dma_templ(std::integral_constant<int, 1>{}); dma_templ(std::integral_constant<int, 2>{});
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                read_taps(locate((float)y0, ca, ilo, ihi), top, bot);
            }
            // A transfer issued in step j has landed when step j+2 starts (wait_keep leaves only step j+1's in flight),
            // so step j fetches what row j+3 will read: frame-0 rows up to ihi(j+1) + 5 (the source row rises by at most
            // 1.4 per template row: ihi(j+3) + 1 for the lower taps + 1 of guard <= ihi(j+1) + 5) and template row j+3.
            // Nothing a later fetch needs is overwritten: the rows in flight reach back to ihi(j+1) - 3 at most, a fetch
            // reads from ilo - 1 (guard) and the end lanes of a row are at most one row apart.
            // The bounds that make this safe were derived from the strip's corners before the loop; they are also CHECKED, on
            // scalars, row by row: what fetch() is about to read must have landed (`safe`) and must not have been
            // overwritten by anything issued since. On a violation run_ring returns true: the strip's accumulators are
            // private to the strip, so the caller clears them and redoes the strip through the gather loop (bit-identical
            // by construction) and counts the event (stk_timing.ecc_ring_fallbacks; 0 on every BASELINE stack).
            int prev_issued = 1;                                 // (nothing is in flight before the first step)
            int safe = loaded, before_prev = loaded, violated = 0;
            // row j is in `cur`; SLOT = j & 3 is its template slot, the one refilled (for row j + 3) is (j + 3) & 3
            float fy = (float)y0;                                // Y of the row in `cur` (exact: an integer below 2^24)
            auto step = [&](auto slot_tag, ColRow& cur, ColRow& nxt) {
                constexpr int SLOT = decltype(slot_tag)::value;
                wait_keep(std::integral_constant<bool, STK_COL_RELAXED != 0>{}, prev_issued);
                safe = before_prev;                               // everything issued before the previous step's transfers has landed
                const float tval = templ_sample(slot_tag);        // (landed two steps ago; its slot is refilled in the next step)
                unsigned at = locate(fy + 1.0f, nxt, ilo, ihi);
                blend(cur, top, bot, bl);
                // the taps are consumed: their registers take the next row's. (The empty statement ties the read address to
                // the blend's results, or the scheduler hoists the reads above the blend and needs a second register set.)
                asm volatile("" : "+v"(at) : "v"(bl.Iw), "v"(bl.gxw), "v"(bl.gyw));
                read_taps(at, top, bot);
                violated |= (safe - (ihi + 2)) | (ilo - 1 + LK - 1 - loaded);   // a sign bit: read (guard row included) before it landed / after it was overwritten
                before_prev = loaded;
                issued = 0;
                if (loaded < ihi + la) dma_row();
                if (loaded < ihi + la) dma_row();
                dma_templ(std::integral_constant<int, (SLOT + 3) & 3>{});
                prev_issued = issued;
#if defined(STK_COL_PAD_VALU) || defined(STK_COL_PAD_SALU) || defined(STK_COL_PAD_LDS)
                {   // sensitivity probes (tools/ab_build.sh ... -DSTK_COL_PAD_VALU=16): independent filler work per row
                    float pad0 = 1.f, pad1 = 2.f; int spad = 0;
#ifdef STK_COL_PAD_VALU
#pragma unroll
                    for (int k = 0; k < STK_COL_PAD_VALU; k += 2) asm volatile("v_fmac_f32 %0, %2, %2\n\tv_fmac_f32 %1, %2, %2" : "+v"(pad0), "+v"(pad1) : "v"(cur.rw));
#endif
#ifdef STK_COL_PAD_SALU
#pragma unroll
                    for (int k = 0; k < STK_COL_PAD_SALU; k++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(spad) : : "scc");
#endif
#ifdef STK_COL_PAD_LDS
#pragma unroll
                    for (int k = 0; k < STK_COL_PAD_LDS; k++) asm volatile("ds_read_b32 %0, %1" : "=v"(pad0) : "v"(l16) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                    asm volatile("" : : "v"(pad0), "v"(pad1), "s"(spad));
                }
#endif
                accumulate(std::true_type{}, cur, bl, tval, fy, 0);
                fy += 1.0f;
            };
            auto last = [&](auto slot_tag, ColRow& cur) {        // the strip's last row: nothing left to fetch
                // (its template row was the last transfer of the step before the previous one: the steps' wait allows that
                // one in flight; here only the previous step's transfers may be)
                wait_keep(std::false_type{}, prev_issued);
                const float tval = templ_sample(slot_tag);
                blend(cur, top, bot, bl); accumulate(std::true_type{}, cur, bl, tval, fy, 0);
            };
            using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
            using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;
            const int n_steps = y1 - y0 - 1;                     // rows that have a successor to fetch
            int j = 0;
            for (; j + 4 <= n_steps; j += 4) {
                step(S0{}, ca, cb);
                step(S1{}, cb, ca);
                step(S2{}, ca, cb);
                step(S3{}, cb, ca);
            }
            if (j < n_steps) {
                step(S0{}, ca, cb);
                if (j + 1 < n_steps) {
                    step(S1{}, cb, ca);
                    if (j + 2 < n_steps) { step(S2{}, ca, cb); last(S3{}, cb); }
                    else last(S2{}, ca);
                } else last(S1{}, cb);
            } else last(S0{}, ca);
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
            // (X again from the lane number, behind a compiler barrier: the homography's row loop has no use for it and
            // should not carry it in a register)
            int lane_again = (int)(threadIdx.x & 63);
            asm volatile("" : "+v"(lane_again));
            const float fx = (float)min(col * 64 + lane_again, a.tw - 1);
            const float fxx = fx * fx;
            int idx = 0;
#pragma unroll
            for (int i = 0; i < P; i++)
#pragma unroll
                for (int j = i; j < P; j++) {
                    const int ci = i < 6 ? i % 3 : i - 6, cj = j < 6 ? j % 3 : j - 6;      // 0 a, 1 b, 2 t
                    const int lo = ci < cj ? ci : cj, hi = ci < cj ? cj : ci;
                    // product index: aa 0, bb 1, at 2, bt 3, ab 4, tt 5
                    const int prod = lo == hi ? (lo == 0 ? 0 : lo == 1 ? 1 : 5) : (hi == 2 ? 2 + lo : 4);
                    const int xpow = (i < 3) + (j < 3), ypow = (i >= 3 && i < 6) + (j >= 3 && j < 6);
                    const float m = hs[prod][ypow];
                    v[idx++] = xpow == 0 ? m : xpow == 1 ? m * fx : m * fxx;
                }
#pragma unroll
            for (int w = 0; w < 3; w++)
#pragma unroll
                for (int i = 0; i < P; i++) {
                    const int ci = i < 6 ? i % 3 : i - 6;
                    const float z0 = ms0[3 * w + ci], z1 = ms1[3 * w + ci];
                    v[NH + w * P + i] = i < 3 ? z0 * fx : i < 6 ? z1 : z0;
                }
            v[NH + 3 * P + 0] = s_mf; v[NH + 3 * P + 1] = s_u; v[NH + 3 * P + 2] = s_uu;
            v[NH + 3 * P + 3] = s_v; v[NH + 3 * P + 4] = s_vv; v[NH + 3 * P + 5] = s_x;
            }
            lane_transpose_sum<NS>(v, lane);
            dacc[0] = dacc[0] + (double)v[0];
            if constexpr (NS > 64) dacc[1] = dacc[1] + (double)v[1];
            clear();
        }
    }

    // the block's four waves: their rings are idle now and hold the 4 x NS doubles of the reduction
    __syncthreads();
    double (*red)[NS] = reinterpret_cast<double (*)[NS]>(ring_all);
    const int k0 = (int)(__builtin_bitreverse32((unsigned)lane) >> 26);
    if (k0 < NS) red[wave][k0] = dacc[0];
    if (k0 + 64 < NS) red[wave][k0 + 64] = dacc[1];
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
