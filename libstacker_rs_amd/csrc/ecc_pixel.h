// ecc_pixel.h — what the ECC iteration kernels share: tap types, the exact INTER_NEAREST mask test, the per-slot
// constants. Included by kernels_ecc.hip and kernels_ecc_col.hip.
#pragma once
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

}  // namespace stk
