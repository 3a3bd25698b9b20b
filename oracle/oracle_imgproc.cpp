// oracle_imgproc.cpp — CPU restatement of the per-pixel OpenCV entry points on the
// hot path: cvtColor(BGR2GRAY), convertTo, GaussianBlur, the ECC gradient filter,
// warpPerspective / warpAffine (INTER_LINEAR), cv::add, MatExpr division.
// TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle_common.h).
//
// Reference call sites: utils.rs:128-144 (imread -> convert -> cvt_color),
// lib.rs:290-299 / 780-803 (warp), lib.rs:306-316 / 807-814 (accumulate),
// lib.rs:339-345 / 836-839 (normalise).
#include "oracle_common.h"

using namespace orc;

extern "C" {

// A3: cvt_color(BGR2GRAY) on the integer image, utils.rs:136-142.
// 8U: (B*3735 + G*19235 + R*9798 + 2^14) >> 15 ; 16U: (B*1868 + G*9617 + R*4899 + 2^13) >> 14 ;
// 32F: B*0.114f + G*0.587f + R*0.299f.
int orc_grey(const void* bgr, int depth, int w, int h, size_t stride_bytes, void* out) {
    if (stride_bytes == 0) stride_bytes = (size_t)w * 3 * (depth / 8);
    for (int y = 0; y < h; y++) {
        const uint8_t* row = (const uint8_t*)bgr + (size_t)y * stride_bytes;
        if (depth == 8) {
            uint8_t* o = (uint8_t*)out + (size_t)y * w;
            for (int x = 0; x < w; x++) {
                unsigned b = row[3 * x], g = row[3 * x + 1], r = row[3 * x + 2];
                o[x] = (uint8_t)((b * 3735u + g * 19235u + r * 9798u + (1u << 14)) >> 15);
            }
        } else if (depth == 16) {
            const uint16_t* s = (const uint16_t*)row;
            uint16_t* o = (uint16_t*)out + (size_t)y * w;
            for (int x = 0; x < w; x++) {
                unsigned b = s[3 * x], g = s[3 * x + 1], r = s[3 * x + 2];
                o[x] = (uint16_t)((b * 1868u + g * 9617u + r * 4899u + (1u << 13)) >> 14);
            }
        } else if (depth == 32) {
            const float* s = (const float*)row;
            float* o = (float*)out + (size_t)y * w;
            for (int x = 0; x < w; x++)
                o[x] = s[3 * x] * 0.114f + s[3 * x + 1] * 0.587f + s[3 * x + 2] * 0.299f;
        } else return 3;
    }
    return 0;
}

// A2: Mat::convert_to(CV_32F, alpha, 0) utils.rs:133 -> dst = (float)src * (float)alpha.
int orc_convert_f32(const void* src, int depth, size_t n, double alpha, float* out) {
    const float a = (float)alpha;
    if (depth == 8) { const uint8_t* s = (const uint8_t*)src; for (size_t i = 0; i < n; i++) out[i] = (float)s[i] * a; }
    else if (depth == 16) { const uint16_t* s = (const uint16_t*)src; for (size_t i = 0; i < n; i++) out[i] = (float)s[i] * a; }
    else if (depth == 32) { const float* s = (const float*)src; for (size_t i = 0; i < n; i++) out[i] = s[i] * a; }
    else return 3;
    return 0;
}

// cv::getGaussianKernel(n, sigma<=0, CV_32F): fixed taps for n<=7, else sigma = 0.3*((n-1)*0.5-1)+0.8.
int orc_gaussian_kernel(int n, float* k) {
    if (n <= 0 || n % 2 == 0) return 3;
    static const float tab[4][7] = {
        {1.f},
        {0.25f, 0.5f, 0.25f},
        {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
        {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
    if (n <= 7) { for (int i = 0; i < n; i++) k[i] = tab[n >> 1][i]; return 0; }
    double sigma = ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2x = -0.5 / (sigma * sigma), sum = 0;
    std::vector<double> t(n);
    for (int i = 0; i < n; i++) { double x = i - (n - 1) * 0.5; t[i] = std::exp(scale2x * x * x); sum += t[i]; }
    sum = 1.0 / sum;
    for (int i = 0; i < n; i++) k[i] = (float)(t[i] * sum);
    return 0;
}

// GaussianBlur(src(float), g x g, sigma 0, BORDER_REFLECT_101) as sepFilter2D: symmetric row
// filter then symmetric column filter, f32 throughout. src is u8 or f32 single channel.
int orc_gaussian_blur_f32(const void* src, int depth, int w, int h, int ksize, float* out) {
    std::vector<float> k(ksize);
    if (orc_gaussian_kernel(ksize, k.data())) return 3;
    const int r = ksize / 2;
    std::vector<float> tmp((size_t)w * h);
    std::vector<float> line(w);
    for (int y = 0; y < h; y++) {
        if (depth == 8) { const uint8_t* s = (const uint8_t*)src + (size_t)y * w; for (int x = 0; x < w; x++) line[x] = (float)s[x]; }
        else if (depth == 32) { const float* s = (const float*)src + (size_t)y * w; for (int x = 0; x < w; x++) line[x] = s[x]; }
        else return 3;
        float* t = tmp.data() + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            float s = k[r] * line[x];
            for (int i = 1; i <= r; i++) {
                int xl = border_interpolate(x - i, w, BORDER_REFLECT_101);
                int xr = border_interpolate(x + i, w, BORDER_REFLECT_101);
                s += k[r + i] * (line[xl] + line[xr]);
            }
            t[x] = s;
        }
    }
    for (int y = 0; y < h; y++) {
        float* o = out + (size_t)y * w;
        const float* c = tmp.data() + (size_t)y * w;
        for (int x = 0; x < w; x++) o[x] = k[r] * c[x];
        for (int i = 1; i <= r; i++) {
            const float* a = tmp.data() + (size_t)border_interpolate(y - i, h, BORDER_REFLECT_101) * w;
            const float* b = tmp.data() + (size_t)border_interpolate(y + i, h, BORDER_REFLECT_101) * w;
            for (int x = 0; x < w; x++) o[x] += k[r + i] * (a[x] + b[x]);
        }
    }
    return 0;
}

// ECC step 5: filter2D(img, [-0.5 0 0.5]) horizontally and vertically, BORDER_REFLECT_101.
int orc_gradients(const float* img, int w, int h, float* gx, float* gy) {
    for (int y = 0; y < h; y++) {
        const float* s = img + (size_t)y * w;
        const float* su = img + (size_t)border_interpolate(y - 1, h, BORDER_REFLECT_101) * w;
        const float* sd = img + (size_t)border_interpolate(y + 1, h, BORDER_REFLECT_101) * w;
        for (int x = 0; x < w; x++) {
            int xl = border_interpolate(x - 1, w, BORDER_REFLECT_101);
            int xr = border_interpolate(x + 1, w, BORDER_REFLECT_101);
            gx[(size_t)y * w + x] = -0.5f * s[xl] + 0.5f * s[xr];
            gy[(size_t)y * w + x] = -0.5f * su[x] + 0.5f * sd[x];
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------
// F1/F2: warpPerspective / warpAffine, INTER_LINEAR. `Minv` maps destination -> source
// (callers without WARP_INVERSE_MAP invert first, orc_warp_frame below does).
//
// subpixel_bits == 0: OpenCV >= 4.11 kernels (imgproc/src/warp_kernels.simd.hpp): matrix cast
//   to f32, sx = fma(M0,x,fma(M1,y,M2)) / w in f32, ix = floor(sx), ax = sx - ix,
//   v0 = fma(ax, p01-p00, p00); v1 = fma(ax, p11-p10, p10); out = fma(ay, v1-v0, v0).
// subpixel_bits == 5: classic remap path: double coordinates scaled by 32, cvRound, 32x32
//   table of float weights, out = p00*w00 + p01*w01 + p10*w10 + p11*w11.
// Source samples are converted on the fly: p = (float)src * (float)alpha (A2 fused).
// ---------------------------------------------------------------------------------------
static inline float fetch(const void* src, int depth, size_t idx, float a) {
    if (depth == 8) return (float)((const uint8_t*)src)[idx] * a;
    if (depth == 16) return (float)((const uint16_t*)src)[idx] * a;
    return ((const float*)src)[idx] * a;
}

int orc_warp(const void* src, int depth, int sw, int sh, int cn, size_t src_stride_bytes,
             const double* Minv, int is_affine, int border_mode, const double* border_value,
             double alpha, int subpixel_bits, int dw, int dh, float* dst, int accumulate) {
    if (border_mode == BORDER_TRANSPARENT) return 3;
    const size_t esz = depth / 8;
    if (src_stride_bytes == 0) src_stride_bytes = (size_t)sw * cn * esz;
    const size_t sstride = src_stride_bytes / esz;  // in elements
    const float a = (float)alpha;
    float bv[4] = {0, 0, 0, 0};
    for (int c = 0; c < cn && c < 4; c++) bv[c] = border_value ? (float)border_value[c] : 0.f;
    float M[9];
    for (int i = 0; i < 9; i++) M[i] = (float)Minv[i];
    float wtab[32][2];
    for (int i = 0; i < 32; i++) { float x = i * (1.f / 32); wtab[i][0] = 1.f - x; wtab[i][1] = x; }

    #pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; y++) {
        for (int x = 0; x < dw; x++) {
            int ix, iy; float ax, ay; int qx = 0, qy = 0;
            bool finite = true;
            if (subpixel_bits == 0) {
                float fx = (float)x, fy = (float)y;
                float X = std::fmaf(M[0], fx, std::fmaf(M[1], fy, M[2]));
                float Y = std::fmaf(M[3], fx, std::fmaf(M[4], fy, M[5]));
                if (!is_affine) {
                    float W = std::fmaf(M[6], fx, std::fmaf(M[7], fy, M[8]));
                    X = X / W; Y = Y / W;
                }
                finite = std::isfinite(X) && std::isfinite(Y) && std::fabs(X) < 1e9f && std::fabs(Y) < 1e9f;
                float flx = std::floor(X), fly = std::floor(Y);
                ix = finite ? (int)flx : -100000; iy = finite ? (int)fly : -100000;
                ax = finite ? X - flx : 0.f; ay = finite ? Y - fly : 0.f;
            } else {
                const double S = 32.0;
                double X, Y;
                if (is_affine) {
                    // WarpAffineInvoker fixed point: AB_BITS 10, rounding delta AB_SCALE/32/2.
                    int adx = sat_int(Minv[0] * x * 1024), bdx = sat_int(Minv[3] * x * 1024);
                    int X0 = sat_int((Minv[1] * y + Minv[2]) * 1024) + 16;
                    int Y0 = sat_int((Minv[4] * y + Minv[5]) * 1024) + 16;
                    int Xi = (X0 + adx) >> 5, Yi = (Y0 + bdx) >> 5;
                    ix = Xi >> 5; iy = Yi >> 5; qx = Xi & 31; qy = Yi & 31;
                } else {
                    double W = Minv[6] * x + Minv[7] * y + Minv[8];
                    W = W != 0 ? S / W : 0;
                    X = std::max(-2147483648.0, std::min(2147483647.0, (Minv[0] * x + Minv[1] * y + Minv[2]) * W));
                    Y = std::max(-2147483648.0, std::min(2147483647.0, (Minv[3] * x + Minv[4] * y + Minv[5]) * W));
                    int Xi = sat_int(X), Yi = sat_int(Y);
                    ix = Xi >> 5; iy = Yi >> 5; qx = Xi & 31; qy = Yi & 31;
                }
                ax = ay = 0;
            }
            int x0 = border_interpolate(ix, sw, border_mode), x1 = border_interpolate(ix + 1, sw, border_mode);
            int y0 = border_interpolate(iy, sh, border_mode), y1 = border_interpolate(iy + 1, sh, border_mode);
            if (!finite) x0 = x1 = y0 = y1 = -1;
            if (border_mode != BORDER_CONSTANT && !finite) { x0 = x1 = y0 = y1 = 0; }
            for (int c = 0; c < cn; c++) {
                float p00 = (x0 >= 0 && y0 >= 0) ? fetch(src, depth, (size_t)y0 * sstride + (size_t)x0 * cn + c, a) : bv[c];
                float p01 = (x1 >= 0 && y0 >= 0) ? fetch(src, depth, (size_t)y0 * sstride + (size_t)x1 * cn + c, a) : bv[c];
                float p10 = (x0 >= 0 && y1 >= 0) ? fetch(src, depth, (size_t)y1 * sstride + (size_t)x0 * cn + c, a) : bv[c];
                float p11 = (x1 >= 0 && y1 >= 0) ? fetch(src, depth, (size_t)y1 * sstride + (size_t)x1 * cn + c, a) : bv[c];
                float v;
                if (subpixel_bits == 0) {
                    float v0 = std::fmaf(ax, p01 - p00, p00);
                    float v1 = std::fmaf(ax, p11 - p10, p10);
                    v = std::fmaf(ay, v1 - v0, v0);
                } else {
                    float w00 = wtab[qy][0] * wtab[qx][0], w01 = wtab[qy][0] * wtab[qx][1];
                    float w10 = wtab[qy][1] * wtab[qx][0], w11 = wtab[qy][1] * wtab[qx][1];
                    v = p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11;
                }
                float* d = dst + ((size_t)y * dw + x) * cn + c;
                *d = accumulate ? *d + v : v;   // G1: cv::add f32
            }
        }
    }
    return 0;
}

// warp_perspective / warp_affine as libstacker calls them (no WARP_INVERSE_MAP):
// M (forward, frame_i -> frame_0) is inverted in double first.
int orc_warp_frame(const void* src, int depth, int w, int h, int cn, size_t stride_bytes,
                   const double* M, int is_affine, int border_mode, const double* border_value,
                   double alpha, int subpixel_bits, float* dst, int accumulate) {
    double Minv[9];
    if (is_affine) invert_affine(M, Minv);
    else invert3x3(M, Minv);
    return orc_warp(src, depth, w, h, cn, stride_bytes, Minv, is_affine, border_mode, border_value,
                    alpha, subpixel_bits, w, h, dst, accumulate);
}

// the same with a destination of its own size: warp_perspective(src, M, dsize = (dw, dh)) — keypoint_match warps every frame
// into the FIRST frame's size, whatever its own (lib.rs:290-299)
int orc_warp_frame_sized(const void* src, int depth, int sw, int sh, int cn, size_t stride_bytes,
                         const double* M, int is_affine, int border_mode, const double* border_value,
                         double alpha, int subpixel_bits, float* dst, int dw, int dh, int accumulate) {
    double Minv[9];
    if (is_affine) invert_affine(M, Minv);
    else invert3x3(M, Minv);
    return orc_warp(src, depth, sw, sh, cn, stride_bytes, Minv, is_affine, border_mode, border_value,
                    alpha, subpixel_bits, dw, dh, dst, accumulate);
}

// G1: acc = acc + img (cv::add, f32).
int orc_add(float* acc, const float* img, size_t n) {
    for (size_t i = 0; i < n; i++) acc[i] = acc[i] + img[i];
    return 0;
}

// G2: img / (n as f64) -> img * (float)(1.0/n)   lib.rs:342, 837.
int orc_scale(const float* in, size_t n, double divisor, float* out) {
    const float s = (float)(1.0 / divisor);
    for (size_t i = 0; i < n; i++) out[i] = in[i] * s;
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// scale_image (utils.rs:186-214): aspect-preserving resize(INTER_AREA) so that the SMALLER
// dimension becomes `scale_down` (despite the "width" in the caller's parameter name).
// ---------------------------------------------------------------------------------------
extern "C" int orc_scaled_size(int w, int h, float scale_down, int* nw, int* nh) {
    const double sf = w < h ? (double)scale_down / (double)w : (double)scale_down / (double)h;
    *nw = (int)((double)w * sf);        // `as i32`: truncation
    *nh = (int)((double)h * sf);
    return (*nw > 0 && *nh > 0) ? 0 : 3;
}

namespace {
struct DecimateAlpha { int si, di; float alpha; };
// imgproc/src/resize.cpp computeResizeAreaTab [OCV-RECALL]
int area_tab(int ssize, int dsize, double scale, std::vector<DecimateAlpha>& tab) {
    tab.clear();
    for (int dx = 0; dx < dsize; dx++) {
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cellWidth = std::min(scale, ssize - fsx1);
        int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) tab.push_back({sx1 - 1, dx, (float)((sx1 - fsx1) / cellWidth)});
        for (int sx = sx1; sx < sx2; sx++) tab.push_back({sx, dx, (float)(1.0 / cellWidth)});
        if (fsx2 - sx2 > 1e-3) tab.push_back({sx2, dx, (float)(std::min(std::min(fsx2 - sx2, 1.), cellWidth) / cellWidth)});
    }
    return (int)tab.size();
}
}  // namespace

// resize(src, dsize, INTER_AREA) for a single-channel image, 8-bit or f32 [OCV-RECALL:
// imgproc/src/resize.cpp]. Integer ratios take resizeAreaFast_: the 2 x 2 case is ResizeAreaFastVec's (a + b + c + d + 2) >> 2
// for 8 bit (half rounds up) and the vector form ((a + b) + (c + d)) * 0.25f for f32 (its scalar tail of w mod the vector
// width adds ((a + b) + c) + d: a last-bit difference on a few right-hand columns, not reproduced); other integer ratios sum
// the cell (int for 8 bit; f32 in groups of four in row-major order for f32) and store saturate_cast<T>(sum * (1.f / area)).
// Otherwise the fractional-coverage tables, f32 accumulation in table order, rows combined as sum = beta0*buf0;
// sum += beta_k*buf_k.
namespace {
template <typename T> inline T area_store(float v);
template <> inline uint8_t area_store<uint8_t>(float v) { return (uint8_t)std::min(std::max(cv_round(v), 0), 255); }
template <> inline float area_store<float>(float v) { return v; }

template <typename T>
int resize_area_t(const T* src, int sw, int sh, T* dst, int dw, int dh) {
    if (dw <= 0 || dh <= 0) return 3;
    constexpr bool U8 = sizeof(T) == 1;
    const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
    const double scale_x = 1.0 / inv_x, scale_y = 1.0 / inv_y;
    if (!(scale_x >= 1.0 && scale_y >= 1.0)) {
        // The image grows in a direction: "true area interpolation is only implemented for scale >= 1; in other cases it is
        // emulated using some variant of bilinear interpolation" — resize() with area_mode, then resizeGeneric_ with
        // HResizeLinear / VResizeLinear [OCV-RECALL]. Tables first, as OpenCV fills them.
        std::vector<int> xofs(dw), yofs(dh);
        std::vector<float> fxs(dw), fys(dh);
        int xmax = dw;
        for (int dx = 0; dx < dw; dx++) {
            int sx = cv_floor(dx * scale_x);
            float fx = (float)((dx + 1) - (sx + 1) * inv_x);
            fx = fx <= 0 ? 0.f : fx - (float)cv_floor(fx);
            if (sx + 1 >= sw) { xmax = std::min(xmax, dx); if (sx >= sw - 1) { fx = 0; sx = sw - 1; } }
            xofs[dx] = sx; fxs[dx] = fx;
        }
        for (int dy = 0; dy < dh; dy++) {
            const int sy = cv_floor(dy * scale_y);
            float fy = (float)((dy + 1) - (sy + 1) * inv_y);
            fy = fy <= 0 ? 0.f : fy - (float)cv_floor(fy);
            yofs[dy] = sy; fys[dy] = fy;
        }
        auto clip = [](int v, int n) { return v < 0 ? 0 : v >= n ? n - 1 : v; };
        for (int dy = 0; dy < dh; dy++) {
            const T* S0 = src + (size_t)clip(yofs[dy], sh) * sw;
            const T* S1 = src + (size_t)clip(yofs[dy] + 1, sh) * sw;
            for (int dx = 0; dx < dw; dx++) {
                const int sx = xofs[dx], sx1 = std::min(sx + 1, sw - 1);
                if (U8) {
                    const int a0 = (short)cv_round((1.f - fxs[dx]) * 2048), a1 = (short)cv_round(fxs[dx] * 2048);
                    const int b0 = (short)cv_round((1.f - fys[dy]) * 2048), b1 = (short)cv_round(fys[dy] * 2048);
                    const int h0 = dx >= xmax ? (int)S0[sx] * 2048 : (int)S0[sx] * a0 + (int)S0[sx1] * a1;
                    const int h1 = dx >= xmax ? (int)S1[sx] * 2048 : (int)S1[sx] * a0 + (int)S1[sx1] * a1;
                    dst[(size_t)dy * dw + dx] = (T)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
                } else {
                    const float a0 = 1.f - fxs[dx], a1 = fxs[dx], b0 = 1.f - fys[dy], b1 = fys[dy];
                    const float h0 = dx >= xmax ? (float)S0[sx] * 1.f : (float)S0[sx] * a0 + (float)S0[sx1] * a1;
                    const float h1 = dx >= xmax ? (float)S1[sx] * 1.f : (float)S1[sx] * a0 + (float)S1[sx1] * a1;
                    dst[(size_t)dy * dw + dx] = (T)(h0 * b0 + h1 * b1);
                }
            }
        }
        return 0;
    }
    const int isx = sat_int(scale_x), isy = sat_int(scale_y);
    if (std::fabs(scale_x - isx) < DBL_EPSILON && std::fabs(scale_y - isy) < DBL_EPSILON) {
        const float sc = 1.f / (isx * isy);
        const int area = isx * isy;
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++) {
                const T* S = src + (size_t)(y * isy) * sw + (size_t)x * isx;
                T& D = dst[(size_t)y * dw + x];
                if (isx == 2 && isy == 2) {
                    if (U8) D = (T)(((int)S[0] + (int)S[1] + (int)S[sw] + (int)S[sw + 1] + 2) >> 2);
                    else D = (T)((((float)S[0] + (float)S[1]) + ((float)S[sw] + (float)S[sw + 1])) * 0.25f);
                    continue;
                }
                if (U8) {
                    int sum = 0;
                    for (int k = 0; k < area; k++) sum += (int)S[(size_t)(k / isx) * sw + k % isx];
                    D = area_store<T>((float)sum * sc);
                } else {
                    auto at = [&](int k) { return (float)S[(size_t)(k / isx) * sw + k % isx]; };
                    float sum = 0.f;
                    int k = 0;
                    for (; k <= area - 4; k += 4) sum += ((at(k) + at(k + 1)) + at(k + 2)) + at(k + 3);
                    for (; k < area; k++) sum += at(k);
                    D = area_store<T>(sum * sc);
                }
            }
        return 0;
    }
    std::vector<DecimateAlpha> xt, yt;
    area_tab(sw, dw, scale_x, xt);
    area_tab(sh, dh, scale_y, yt);
    std::vector<float> buf(dw), sum(dw);
    int prev_dy = yt.empty() ? 0 : yt[0].di;
    std::fill(sum.begin(), sum.end(), 0.f);
    for (size_t j = 0; j < yt.size(); j++) {
        const float beta = yt[j].alpha;
        const int dy = yt[j].di, sy = yt[j].si;
        const T* S = src + (size_t)sy * sw;
        std::fill(buf.begin(), buf.end(), 0.f);
        for (const DecimateAlpha& t : xt) buf[t.di] += (float)S[t.si] * t.alpha;
        if (dy != prev_dy) {
            T* D = dst + (size_t)prev_dy * dw;
            for (int dx = 0; dx < dw; dx++) { D[dx] = area_store<T>(sum[dx]); sum[dx] = beta * buf[dx]; }
            prev_dy = dy;
        } else {
            for (int dx = 0; dx < dw; dx++) sum[dx] += beta * buf[dx];
        }
    }
    T* D = dst + (size_t)prev_dy * dw;
    for (int dx = 0; dx < dw; dx++) D[dx] = area_store<T>(sum[dx]);
    return 0;
}
}  // namespace

extern "C" int orc_resize_area_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
    return resize_area_t<uint8_t>(src, sw, sh, dst, dw, dh);
}
extern "C" int orc_resize_area_f32(const float* src, int sw, int sh, float* dst, int dw, int dh) {
    return resize_area_t<float>(src, sw, sh, dst, dw, dh);
}
