// orb_device.h — device helpers shared by kernels_orb.hip (the production ORB path) and kernels_orb_small.hip (the plain
// per-pixel forms that serve levels too small or too unaligned for the tiled kernels, and double as their cross-checks).
#pragma once
#include "common.h"
#include "keypoint.h"

namespace stk {

// Batched over frames: blockIdx.z (or the named grid dimension) is the frame; per-frame arrays sit `*_stride`
// elements apart (OrbBatch). n_frames = 1 and zero strides give the single-image form.
struct OrbBatch { size_t pyr, states, cand, sel; };

__device__ __forceinline__ int refl101(int p, int len) {
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// four bytes at any address through aligned dword loads (level rows are not dword-aligned in general)
__device__ __forceinline__ uint32_t load4_unaligned(const uint8_t* p) {
    const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3);
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p - sh);   // pointer arithmetic keeps the address space: global_load, not flat_load
    const uint32_t lo = q[0];
    if (sh == 0) return lo;
    return __builtin_amdgcn_alignbyte(q[1], lo, sh);   // bytes sh .. sh+3 of the aligned pair
}

// INTER_LINEAR_EXACT: offset and 8.8 weights of destination index d (f64, as resize_bitExact computes them)
__device__ __forceinline__ void lin_coef(int d, int src, double scale, int& ofs, int& c0, int& c1) {
    const double fval = scale * ((double)d + 0.5) - 0.5;
    const int ival = (int)floor(fval);
    if (ival >= 0 && src > 1) {
        if (ival < src - 1) { ofs = ival; c1 = (int)__builtin_rint((fval - (double)ival) * 256.0); c0 = 256 - c1; }
        else { ofs = src - 1; c0 = 256; c1 = 0; }
    } else { ofs = 0; c0 = 256; c1 = 0; }
}

// ---- FAST-9/16 on one pixel ------------------------------------------------------------------------
__device__ __forceinline__ void fast_ring(const uint8_t* __restrict__ p, int stride, int v, int (&d)[16]) {
    d[0] = v - p[3 * stride];        d[1] = v - p[3 * stride + 1];   d[2] = v - p[2 * stride + 2];   d[3] = v - p[stride + 3];
    d[4] = v - p[3];                 d[5] = v - p[-stride + 3];      d[6] = v - p[-2 * stride + 2];  d[7] = v - p[-3 * stride + 1];
    d[8] = v - p[-3 * stride];       d[9] = v - p[-3 * stride - 1];  d[10] = v - p[-2 * stride - 2]; d[11] = v - p[-stride - 3];
    d[12] = v - p[-3];               d[13] = v - p[stride - 3];      d[14] = v - p[2 * stride - 2];  d[15] = v - p[3 * stride - 1];
}

// corner strength: the largest t for which the pixel is still a FAST-9 corner at threshold t (0 if none above thr)
__device__ __forceinline__ int fast_strength(const int (&d)[16], int thr) {
    int best = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int mn = d[k], mx = d[k];
#pragma unroll
        for (int j = 1; j < 9; j++) { mn = min(mn, d[(k + j) & 15]); mx = max(mx, d[(k + j) & 15]); }
        best = max(best, max(mn, -mx));
    }
    return best > thr ? best - 1 : 0;
}

__device__ __forceinline__ int fast_score_at(const uint8_t* __restrict__ p, int stride, int thr) {
    const int v = p[0];
    // high-speed rejection: any 9 contiguous ring pixels contain at least two of the four compass points
    const int n0 = v - p[3 * stride], n4 = v - p[3], n8 = v - p[-3 * stride], n12 = v - p[-3];
    const int dark = (n0 > thr) + (n4 > thr) + (n8 > thr) + (n12 > thr);
    const int bright = (n0 < -thr) + (n4 < -thr) + (n8 < -thr) + (n12 < -thr);
    if (dark < 2 && bright < 2) return 0;
    int d[16];
    fast_ring(p, stride, v, d);
    return fast_strength(d, thr);
}

// the plain forms (kernels_orb_small.hip)
hipError_t launch_resize_exact_plain(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, hipStream_t s, int n_frames, size_t frame_stride);
hipError_t launch_fast_score_nms_plain(const uint8_t* img, int w, int h, int thr, int edge, uint8_t* score, OrbLevelState* st, OrbCandidate* cand,
                                       int cap, hipStream_t s, int n_frames, const OrbBatch& bs);
hipError_t launch_gauss7_plain(const uint8_t* src, int w, int h, const Gauss7& k, float* tmp, uint8_t* dst, hipStream_t s, int n_frames,
                               size_t pyr_stride, size_t tmp_stride);

}  // namespace stk
