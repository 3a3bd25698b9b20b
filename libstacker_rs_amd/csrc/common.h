// common.h — shared declarations of the HIP engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/stacker.h"

namespace stk {

// ---------------------------------------------------------------------------------------------
// Device data layout (all in HBM, owned by the context's workspace)
//
//  * frames          interleaved BGR u8/u16/f32 as handed in (OpenCV Mat layout).
//  * reference planes (ECC "input" = frame 0, shared by every frame of the stack):
//        f32 planes  I (blurred grey), gx, gy, an interleaved (gx, gy) copy and an interleaved (I, gx, gy) copy, each (H + 2 REF_PAD) x ref_stride
//        with a REF_PAD-pixel ZERO border on every side, so a bilinear footprint with BORDER_CONSTANT 0 is
//        unconditional loads after clamping the integer coordinate to [-2, W] x [-2, H].
//  * templates       one blurred-grey f32 plane per moving frame, row stride rounded up to a
//        multiple of 4 floats so each lane streams aligned 16-byte quads.
//  * accumulator     f32 W*H*3 running sum (the Rayon fold accumulator, lib.rs:306-316, 807-814).
// ---------------------------------------------------------------------------------------------

// Zero border of the frame-0 planes, on every side. >= 2 makes a bilinear footprint unconditional after
// clamping to [-2, W] x [-2, H]; 24 (a multiple of 4: rows stay 16-byte aligned) additionally lets the
// tiled ECC kernel copy whole 72 x 22 footprints of border tiles with unclamped 16-byte LDS-DMA pieces.
constexpr int REF_PAD = 24;

struct RefPlanes {
    const float* I;   // pointer to pixel (0,0) inside the padded plane
    const float* gx;
    const float* gy;
    const float* gxy; // (gx, gy) interleaved, same padded geometry (2 floats per pixel)
    const float* igg; // (I, gx, gy) interleaved, same padded geometry (3 floats per pixel): what the LDS ring of the column pass streams
    int stride;       // floats per padded row
    int w, h;
};

// One ECC "slot": a frame currently being iterated. Lives in device memory and is advanced
// entirely on the device (solve kernel); the host only polls EccQueue::frames_done.
struct EccSlot {
    int   frame;        // index into the template array, -1 = idle
    int   iter;         // iterations executed so far
    float warp[9];      // current map, row-major 3x3
    float cI, cT;       // centring offsets (previous iteration's f32 means) for the moment sums
    double rho, last_rho;
};

struct EccFrameResult {
    float  warp[9];
    int    iters;
    int    status;      // 0 ok, 1 NaN, 2 lambda_d <= 0, 3 not run
    double rho;
};

struct EccQueue {
    int next_frame;     // next template index to hand to a free slot
    int n_frames;       // number of templates
    int frames_done;
    int ready;          // templates [0, ready) exist; raised by the prep stream while frames are still arriving over PCIe
    int ring_fallbacks; // strips of the iteration pass that left the LDS ring for the gather loop at run time (stk_timing.ecc_ring_fallbacks)
#ifdef STK_SOLVE_TIMING
    long long dbg[16];   // wall_clock64 phase deltas of the last solve of slot 0 (10 ns ticks), debug builds only
#endif
};

struct EccCriteria {
    int    n_iter;      // COUNT ? max_count : 200
    double eps;         // EPS ? epsilon : -1
};

// number of moment sums produced per slot and iteration for P warp parameters
__host__ __device__ constexpr int ecc_nsums(int P) { return P * (P + 1) / 2 + 3 * P + 6; }
constexpr int ECC_MAX_SUMS = ecc_nsums(8);   // 66

struct EccIterArgs {
    RefPlanes ref;
    const float* templates;      // n_frames planes
    size_t templ_plane_stride;   // floats between consecutive templates
    int templ_row_stride;        // floats per template row (multiple of 4)
    int tw, th;
    EccSlot* slots;
    int n_slots;                 // slots iterated by this launch: slot0 .. slot0 + n_slots - 1
    int nb;                      // blocks per slot (multiple of 8)
    double* partials;            // [all slots][nsums][nb]
    double* sums;                // [all slots][ECC_MAX_SUMS]: the reduced sums, stage 1 -> stage 2 of the solve kernel
    int* tickets;                // [all slots]: arrival counter of the solve kernel's stage-1 workgroups (self-resetting)
    int slot0;                   // first slot of this launch (0: all slots in one launch)
    int ring;                    // column-walking pass: 1 = frame-0 rows through the per-wave LDS ring where a strip allows it (option ecc_ring)
    int ring_lookahead;          // frame-0 rows the ring keeps ahead of the row being fetched: 5; lower only to provoke the fallback (option ecc_ring_lookahead)
    int* ring_fallbacks;         // device counter: strips whose ring bounds failed the run-time check and were redone by the gather loop (EccQueue::ring_fallbacks)
    int units_q, units_r;        // column-walking pass: (column strip, row) units per wave and the remainder (set by launch_ecc_iter)
};

struct WarpFrame {
    const void* src;
    float  M[9];                 // destination -> source map, f32 (subpixel_bits == 0)
    int    flags;                // WARPFRAME_*: set by warp_fold for the destination rectangle of the launch
    double Md[9];                // same in double (classic quantised path)
};
// over the whole destination rectangle |W| lies in [2^-36, 2^36] and |X|, |Y| (before the division) below 2^36: the range
// in which an IEEE f32 division applies no scaling, so X / W and Y / W may share one reciprocal chain (kernels_warp.hip)
constexpr int WARPFRAME_DIV_IN_RANGE = 1;
// the frame's base address and its row stride are multiples of 4: the u8 fast path may gather dword-aligned 12-byte windows
constexpr int WARPFRAME_SRC_ALIGNED4 = 2;

// 3x3 inverse by the adjugate in double (cv::invert on a 3x3 CV_64F) and invertAffineTransform: what warpPerspective /
// warpAffine do to the forward matrix before they map destination pixels (lib.rs:290-299, 780-803). Host and device run
// the same operations (no contraction: -ffp-contract=off), so a warp frame built on either side has the same bits.
__host__ __device__ inline void warp_invert3x3(const double* m, double* o) {
    double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (d == 0.0) { for (int i = 0; i < 9; i++) o[i] = 0; return; }
    d = 1.0 / d;
    double t[9] = {(m[4] * m[8] - m[5] * m[7]) * d, (m[2] * m[7] - m[1] * m[8]) * d, (m[1] * m[5] - m[2] * m[4]) * d,
                   (m[5] * m[6] - m[3] * m[8]) * d, (m[0] * m[8] - m[2] * m[6]) * d, (m[2] * m[3] - m[0] * m[5]) * d,
                   (m[3] * m[7] - m[4] * m[6]) * d, (m[1] * m[6] - m[0] * m[7]) * d, (m[0] * m[4] - m[1] * m[3]) * d};
    for (int i = 0; i < 9; i++) o[i] = t[i];
}
__host__ __device__ inline void warp_invert_affine(const double* m, double* o) {
    double D = m[0] * m[4] - m[1] * m[3];
    D = D != 0 ? 1.0 / D : 0;
    const double A11 = m[4] * D, A22 = m[0] * D, A12 = -m[1] * D, A21 = -m[3] * D;
    o[0] = A11; o[1] = A12; o[2] = -A11 * m[2] - A12 * m[5];
    o[3] = A21; o[4] = A22; o[5] = -A21 * m[2] - A22 * m[5];
    o[6] = 0; o[7] = 0; o[8] = 1;
}
// WARPFRAME_* flags of a frame for a w x h destination. Range flag: W, X, Y are affine in (x, y), so their extremes over
// [0, w-1] x [0, h-1] sit at the corners; the kernels' own per-pixel bounds are 2^-40 < |W| and |W|, |X|, |Y| < 2^40; the
// corners are tested in double against 2^-36 / 2^36, which leaves the f32 rounding of the kernels' fma chains (relative
// 1e-7) far inside the margin. A NaN or infinite entry fails every comparison: the flag stays clear and the kernel tests
// pixel by pixel.
__host__ __device__ inline int warp_frame_flags(const void* src, const float* M, size_t src_row_bytes, int w, int h, int is_affine) {
    int flags = (((unsigned long long)(size_t)src | (unsigned long long)src_row_bytes) & 3) == 0 ? WARPFRAME_SRC_ALIGNED4 : 0;
    if (is_affine) return flags;
    bool ok = true;
    double wsign = 0;
    const double cx[2] = {0.0, (double)(w - 1)}, cy[2] = {0.0, (double)(h - 1)};
    for (int k = 0; k < 4 && ok; k++) {
        const double x = cx[k & 1], y = cy[k >> 1];
        const double X = (double)M[0] * x + (double)M[1] * y + (double)M[2];
        const double Y = (double)M[3] * x + (double)M[4] * y + (double)M[5];
        const double W = (double)M[6] * x + (double)M[7] * y + (double)M[8];
        const double lim = 68719476736.0;       // 2^36
        const double aW = W < 0 ? -W : W, aX = X < 0 ? -X : X, aY = Y < 0 ? -Y : Y;
        ok = aW > 1.0 / lim && aW < lim && aX < lim && aY < lim;
        if (k == 0) wsign = W; else ok = ok && (W > 0) == (wsign > 0);       // no zero crossing of W inside the rectangle
    }
    return ok ? flags | WARPFRAME_DIV_IN_RANGE : flags;
}
__host__ __device__ inline void warp_frame_make(WarpFrame& wf, const void* src, const double* M, int is_affine) {
    double inv[9];
    if (is_affine) warp_invert_affine(M, inv); else warp_invert3x3(M, inv);
    wf.src = src;
    wf.flags = 0;
    for (int k = 0; k < 9; k++) { wf.Md[k] = inv[k]; wf.M[k] = (float)inv[k]; }
}

struct WarpArgs {
    const WarpFrame* frames;
    int n_frames;
    int sw, sh, cn;
    size_t src_stride;           // elements per source row
    float alpha;
    int border_mode;
    float bv[4];
    float* acc;
    int dw, dh;
    size_t acc_stride;           // floats per accumulator row
    int accumulate;              // 0: overwrite, 1: acc += sum of warped frames
    int is_affine;
    int subpixel_bits;           // 0 or 5
    int tune;                    // launch shape of the u8 fast path (option "warp_tune")
};

// ---- kernel launchers (defined in the .hip files) -------------------------------------------
hipError_t launch_grey(const void* bgr, int depth, int w, int h, size_t stride_bytes, void* out, hipStream_t s,
                       int n_frames = 1, size_t src_frame_bytes = 0, size_t out_frame_elems = 0, int cn = 3 /* 3 BGR, 4 BGRA */);
hipError_t launch_convert_f32(const void* src, int depth, size_t n, float alpha, float* out, hipStream_t s);
// BGR (cn==3) or grey (cn==1) image -> GaussianBlur(float(grey), ksize) f32 plane with row stride out_stride
hipError_t launch_grey16_to_8(const uint16_t* src, size_t n, uint8_t* dst, hipStream_t s);
hipError_t launch_bgr16_to_grey8(const void* bgr16, int w, int h, size_t stride_bytes, uint8_t* out, hipStream_t s, int n_frames = 1,
                                 size_t src_frame_bytes = 0, size_t out_frame_elems = 0);
hipError_t launch_grey_blur(const void* src, int depth, int cn, int w, int h, size_t stride_bytes, int ksize,
                            float* out, int out_stride, hipStream_t s);
// the same for n BGR frames (u8 / u16) in one launch (frame z: ptrs_dev[z], or base + z * frame_bytes when ptrs_dev is null),
// plane z written at out + z * out_plane_stride; hipErrorNotSupported if the streaming kernel does not apply
hipError_t launch_grey_blur_batch(const void* const* ptrs_dev, const void* base, size_t frame_bytes, int n, int depth, int w, int h,
                                  size_t stride_bytes, int ksize, float* out, int out_stride, size_t out_plane_stride, hipStream_t s);
// blurred plane (stride in_stride) -> padded I/gx/gy planes
hipError_t launch_ref_planes(const float* blurred, int in_stride, int w, int h, float* I, float* gx, float* gy, float* gxy,
                             float* igg, int ref_stride, hipStream_t s);
// variant: 3 = production kernels, 0 = direct cross-check version
hipError_t launch_ecc_iter_col(const EccIterArgs& a, int motion, hipStream_t s);   // kernels_ecc_col.hip; a.units_q / units_r set
hipError_t launch_ecc_iter(const EccIterArgs& a, int motion, int variant, hipStream_t s);
hipError_t launch_ecc_solve(const EccIterArgs& a, int motion, EccCriteria crit, EccQueue* queue,
                            EccFrameResult* results, hipStream_t s, const float* init_warps = nullptr);
hipError_t launch_sharpness(const void* grey, int depth, int w, int h, int metric, int ksize, void* partials, int n_blocks,
                            hipStream_t s);
hipError_t launch_ecc_init(EccSlot* slots, int n_slots, int* tickets, EccQueue* queue, int n_frames, EccFrameResult* results,
                           const float* init_warps /* n_frames*9 or null */, hipStream_t s, int ready0 = -1 /* -1: all */);
hipError_t launch_ecc_set_ready(EccQueue* queue, int ready, hipStream_t s);
hipError_t launch_warp_accumulate(const WarpArgs& a, int depth, hipStream_t s);
// the fold's frame table straight from the ECC results, on the device: entry 0 = the reference frame under the identity
// (if add_reference), then template k under results[k].warp — what the host loop of ecc_shard_impl builds, bit for bit
hipError_t launch_warp_frames_from_ecc(const EccFrameResult* results, const void* const* src_ptrs /* n_templates + 1, device */,
                                       int n_templates, int add_reference, int is_affine, int w, int h, size_t src_row_bytes,
                                       WarpFrame* out, hipStream_t s);
hipError_t launch_scale(const float* in, float* out, size_t n, float scale, hipStream_t s);
hipError_t launch_add(float* acc, const float* in, size_t n, hipStream_t s);

}  // namespace stk
