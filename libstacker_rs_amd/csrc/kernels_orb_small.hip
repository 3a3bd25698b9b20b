// kernels_orb_small.hip — the plain, one-thread-per-pixel forms of three ORB stages: the INTER_LINEAR_EXACT pyramid step
// (per pixel, and tiled with per-tile coefficient tables), FAST score + 3x3 non-maximum suppression through a score plane,
// and the separable 7x7 blur through an f32 plane. They are what the first round ran everywhere; since rounds 2-3 the
// production path (kernels_orb.hip) uses the table-driven pyramid step, the tiled FAST + NMS of all levels in one launch and
// the blur of the sampled patch, and comes here only for levels the tiled kernels do not take (narrower than 16 / 8 pixels,
// not dword-aligned) or when an option asks for them as a cross-check: every production kernel is tested bit for bit against
// the plain form it replaced (tests/test_gpu_keypoint.py, test_gpu_stages.py).
#include "orb_device.h"

namespace stk {

// ---- pyramid step, per pixel ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_exact_kernel(const uint8_t* __restrict__ src, int sw, int sh,
                                                           uint8_t* __restrict__ dst, int dw, int dh,
                                                           double scale_x, double scale_y, size_t frame_stride) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    src += blockIdx.z * frame_stride; dst += blockIdx.z * frame_stride;      // batched over frames
    int ox, cx0, cx1, oy, cy0, cy1;
    lin_coef(x, sw, scale_x, ox, cx0, cx1);
    lin_coef(y, sh, scale_y, oy, cy0, cy1);
    const int ox1 = min(ox + 1, sw - 1), oy1 = min(oy + 1, sh - 1);
    const uint8_t* r0 = src + (size_t)oy * sw;
    const uint8_t* r1 = src + (size_t)oy1 * sw;
    const uint32_t h0 = (uint32_t)cx0 * r0[ox] + (uint32_t)cx1 * r0[ox1];
    const uint32_t h1 = (uint32_t)cx0 * r1[ox] + (uint32_t)cx1 * r1[ox1];
    const uint32_t v = (uint32_t)cy0 * h0 + (uint32_t)cy1 * h1;
    dst[(size_t)y * dw + x] = (uint8_t)min((v + (1u << 15)) >> 16, 255u);
}

// Tiled pyramid step: a 128 x 32 output tile per workgroup. The 8.8 coefficients of the tile's 128 columns and 32 rows
// are computed once (in double, as above) into LDS, the source footprint of the tile goes to LDS through aligned dword
// loads, and every output is two table reads + four LDS byte reads. Same arithmetic, bit-identical.
constexpr int RT_X = 128, RT_Y = 32;
constexpr int RT_SW = 176, RT_SH = 48;               // source footprint capacity (scale <= 1.3 plus the +1 tap)

__global__ __launch_bounds__(256) void resize_exact_tiled_kernel(const uint8_t* __restrict__ src, int sw, int sh,
                                                                 uint8_t* __restrict__ dst, int dw, int dh,
                                                                 double scale_x, double scale_y, size_t frame_stride) {
    __shared__ __attribute__((aligned(16))) uint8_t T[RT_SH * RT_SW];
    __shared__ int xo[RT_X], xc[RT_X], yo[RT_Y], yc[RT_Y];
    src += blockIdx.z * frame_stride; dst += blockIdx.z * frame_stride;
    const int x0 = blockIdx.x * RT_X, y0 = blockIdx.y * RT_Y;
    const int tid = threadIdx.x;
    if (tid < RT_X) {
        int o, c0, c1;
        lin_coef(min(x0 + tid, dw - 1), sw, scale_x, o, c0, c1);
        xo[tid] = o; xc[tid] = c1;
    } else if (tid < RT_X + RT_Y) {
        int o, c0, c1;
        lin_coef(min(y0 + tid - RT_X, dh - 1), sh, scale_y, o, c0, c1);
        yo[tid - RT_X] = o; yc[tid - RT_X] = c1;
    }
    __syncthreads();
    const int sx_lo = xo[0] & ~3, sy_lo = yo[0];
    const int nx = min(dw - x0, RT_X), ny = min(dh - y0, RT_Y);
    const int sx_hi = min(xo[nx - 1] + 1, sw - 1), sy_hi = min(yo[ny - 1] + 1, sh - 1);
    const int tw4 = (sx_hi - sx_lo) / 4 + 1, th = sy_hi - sy_lo + 1;          // dwords per row, rows
    if (tw4 * 4 > RT_SW || th > RT_SH) {                                          // never for pyramid steps; keep it correct anyway
        for (int i = tid; i < nx * ny; i += 256) {
            const int tx = i % nx, ty = i / nx;
            const int ox = xo[tx], cx1 = xc[tx], cx0 = 256 - cx1, oy = yo[ty], cy1 = yc[ty], cy0 = 256 - cy1;
            const int ox1 = min(ox + 1, sw - 1), oy1 = min(oy + 1, sh - 1);
            const uint8_t* r0 = src + (size_t)oy * sw;
            const uint8_t* r1 = src + (size_t)oy1 * sw;
            const uint32_t h0 = (uint32_t)cx0 * r0[ox] + (uint32_t)cx1 * r0[ox1], h1 = (uint32_t)cx0 * r1[ox] + (uint32_t)cx1 * r1[ox1];
            const uint32_t v = (uint32_t)cy0 * h0 + (uint32_t)cy1 * h1;
            dst[(size_t)(y0 + ty) * dw + x0 + tx] = (uint8_t)min((v + (1u << 15)) >> 16, 255u);
        }
        return;
    }
    for (int i = tid; i < th * tw4; i += 256) {
        const int ty = i / tw4, d = i - ty * tw4;
        const int gx = sx_lo + 4 * d;
        const uint8_t* row = src + (size_t)(sy_lo + ty) * sw;
        uint32_t v = 0;
        if (gx + 3 < sw) v = load4_unaligned(row + gx);
        else {
#pragma unroll
            for (int e = 0; e < 4; e++) v |= (uint32_t)row[min(gx + e, sw - 1)] << (8 * e);
        }
        *reinterpret_cast<uint32_t*>(T + ty * RT_SW + 4 * d) = v;
    }
    __syncthreads();
    // thread = 4 adjacent columns x 4 rows (rows tid/32 + 8k); the columns' offsets and weights are read and prepared once
    // for the four rows (round 3: they were re-read from LDS for every output)
    const int q = tid & 31, rg = tid >> 5;
    const bool aligned = (dw & 3) == 0;
    int cx[4], o0[4], o1[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int tx = 4 * q + e;
        const int ox = xo[tx];
        cx[e] = xc[tx]; o0[e] = ox - sx_lo; o1[e] = min(ox + 1, sw - 1) - sx_lo;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int ty = rg + 8 * k;
        const int y = y0 + ty;
        if (y >= dh) continue;
        const int oy = yo[ty], cy1 = yc[ty], cy0 = 256 - cy1;
        const uint8_t* r0 = T + (oy - sy_lo) * RT_SW;
        const uint8_t* r1 = T + (min(oy + 1, sh - 1) - sy_lo) * RT_SW;
        uint32_t out = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int cx1 = cx[e], cx0 = 256 - cx1;
            const uint32_t h0 = (uint32_t)cx0 * r0[o0[e]] + (uint32_t)cx1 * r0[o1[e]];
            const uint32_t h1 = (uint32_t)cx0 * r1[o0[e]] + (uint32_t)cx1 * r1[o1[e]];
            const uint32_t v = (uint32_t)cy0 * h0 + (uint32_t)cy1 * h1;
            out |= min((v + (1u << 15)) >> 16, 255u) << (8 * e);
        }
        const int x = x0 + 4 * q;
        if (x >= dw) continue;
        uint8_t* op = dst + (size_t)y * dw + x;
        if (aligned && x + 3 < dw) *reinterpret_cast<uint32_t*>(op) = out;
        else {
            op[0] = (uint8_t)out;
            if (x + 1 < dw) op[1] = (uint8_t)(out >> 8);
            if (x + 2 < dw) op[2] = (uint8_t)(out >> 16);
            if (x + 3 < dw) op[3] = (uint8_t)(out >> 24);
        }
    }
}

hipError_t launch_resize_exact_plain(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, hipStream_t s, int n_frames, size_t frame_stride) {
    const double sx = 1.0 / ((double)dw / sw), sy = 1.0 / ((double)dh / sh);
    if (sw >= 8 && sh >= 2 && sx <= 1.3 && sy <= 1.3 && (reinterpret_cast<uintptr_t>(src) & 3) == 0 && (frame_stride & 3) == 0) {
        dim3 tgrid((dw + RT_X - 1) / RT_X, (dh + RT_Y - 1) / RT_Y, n_frames);
        resize_exact_tiled_kernel<<<tgrid, 256, 0, s>>>(src, sw, sh, dst, dw, dh, sx, sy, frame_stride);
        return hipGetLastError();
    }
    dim3 grid((dw + 63) / 64, (dh + 3) / 4, n_frames);
    resize_exact_kernel<<<grid, 256, 0, s>>>(src, sw, sh, dst, dw, dh, sx, sy, frame_stride);
    return hipGetLastError();
}

// ---- FAST score plane + 3x3 non-maximum suppression ------------------------------------------------------
__global__ __launch_bounds__(256) void fast_score_kernel(const uint8_t* __restrict__ img, int w, int h, int thr,
                                                         uint8_t* __restrict__ score, OrbBatch bs) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    img += blockIdx.z * bs.pyr; score += blockIdx.z * bs.pyr;
    int s = 0;
    if (x >= 3 && x < w - 3 && y >= 3 && y < h - 3) s = fast_score_at(img + (size_t)y * w + x, w, thr);
    score[(size_t)y * w + x] = (uint8_t)s;
}

__global__ __launch_bounds__(256) void fast_nms_kernel(const uint8_t* __restrict__ score, int w, int h, int edge,
                                                       OrbLevelState* st, OrbCandidate* cand, int cap, OrbBatch bs) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63) + edge, y = blockIdx.y * 4 + (threadIdx.x >> 6) + edge;
    if (x >= w - edge || y >= h - edge) return;
    score += blockIdx.z * bs.pyr; st += blockIdx.z * bs.states; cand += blockIdx.z * bs.cand;
    const uint8_t* c = score + (size_t)y * w + x;
    const int s = c[0];
    if (!s) return;
    if (s > c[-1] && s > c[1] && s > c[-w - 1] && s > c[-w] && s > c[-w + 1] && s > c[w - 1] && s > c[w] && s > c[w + 1]) {
        atomicAdd(&st->hist[s], 1);
        const int i = atomicAdd(&st->n_cand, 1);
        if (i < cap) { cand[i].xy = x | (y << 16); cand[i].score = s; }
    }
}

hipError_t launch_fast_score_nms_plain(const uint8_t* img, int w, int h, int thr, int edge, uint8_t* score, OrbLevelState* st, OrbCandidate* cand,
                                       int cap, hipStream_t s, int n_frames, const OrbBatch& bs) {
    dim3 grid((w + 63) / 64, (h + 3) / 4, n_frames);
    fast_score_kernel<<<grid, 256, 0, s>>>(img, w, h, thr, score, bs);
    if (w > 2 * edge && h > 2 * edge) {               // otherwise runByImageBorder leaves nothing: no candidates at all
        dim3 g2((w - 2 * edge + 63) / 64, (h - 2 * edge + 3) / 4, n_frames);
        fast_nms_kernel<<<g2, 256, 0, s>>>(score, w, h, edge, st, cand, cap, bs);
    }
    return hipGetLastError();
}

// ---- 7x7 Gaussian, separable through an f32 plane ------------------------------------------------------------
// ---- 7x7 Gaussian on 8-bit levels -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void gauss7_rows_kernel(const uint8_t* __restrict__ src, int w, int h, Gauss7 k,
                                                          float* __restrict__ tmp, size_t pyr_stride, size_t tmp_stride) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    src += blockIdx.z * pyr_stride; tmp += blockIdx.z * tmp_stride;
    const uint8_t* s = src + (size_t)y * w;
    float acc = k.k[0] * (float)s[refl101(x - 3, w)];
#pragma unroll
    for (int i = 1; i < 7; i++) acc += k.k[i] * (float)s[refl101(x - 3 + i, w)];
    tmp[(size_t)y * w + x] = acc;
}

__global__ __launch_bounds__(256) void gauss7_cols_kernel(const float* __restrict__ tmp, int w, int h, Gauss7 k,
                                                          uint8_t* __restrict__ dst, size_t pyr_stride, size_t tmp_stride) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    tmp += blockIdx.z * tmp_stride; dst += blockIdx.z * pyr_stride;
    float acc = k.k[3] * tmp[(size_t)y * w + x];
#pragma unroll
    for (int i = 1; i <= 3; i++)
        acc += k.k[3 + i] * (tmp[(size_t)refl101(y - i, h) * w + x] + tmp[(size_t)refl101(y + i, h) * w + x]);
    const int r = (int)__builtin_rintf(acc);
    dst[(size_t)y * w + x] = (uint8_t)min(max(r, 0), 255);
}

hipError_t launch_gauss7_plain(const uint8_t* src, int w, int h, const Gauss7& k, float* tmp, uint8_t* dst, hipStream_t s, int n_frames,
                               size_t pyr_stride, size_t tmp_stride) {
    dim3 grid((w + 63) / 64, (h + 3) / 4, n_frames);
    gauss7_rows_kernel<<<grid, 256, 0, s>>>(src, w, h, k, tmp, pyr_stride, tmp_stride);
    gauss7_cols_kernel<<<grid, 256, 0, s>>>(tmp, w, h, k, dst, pyr_stride, tmp_stride);
    return hipGetLastError();
}

}  // namespace stk
