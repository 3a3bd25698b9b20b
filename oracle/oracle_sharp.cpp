// oracle_sharp.cpp — CPU restatement of the reference's four sharpness metrics (lib.rs:1030-1166).
// TEST INFRASTRUCTURE ONLY / PARITY UNPINNED: see oracle_common.h.
//
// Every metric filters a single-channel image into CV_64F and reduces it with cv::mean / cv::meanStdDev:
//   LAPM  lib.rs:1032-1071  sepFilter2D(kernelX = [-1 2 -1], kernelY = getGaussianKernel(3) = [.25 .5 .25]) and
//                           its transpose, BORDER_DEFAULT (REFLECT_101); mean(|Lx| + |Ly|)
//   LAPV  lib.rs:1075-1091  Laplacian(ksize 3) = filter2D with [2 0 2; 0 -8 0; 2 0 2], BORDER_REPLICATE;
//                           meanStdDev -> sigma * sigma
//   TENG  lib.rs:1103-1147  Sobel(dx) and Sobel(dy), ksize in {1,3,5,7} (getDerivKernels' integer kernels, no
//                           normalisation), REFLECT_101; mean(gx^2 + gy^2)
//   GLVN  lib.rs:1151-1166  meanStdDev of the image as f64; sigma^2 / max(mean, DBL_EPSILON)
// cv::mean and cv::meanStdDev scale the f64 sums by the reciprocal 1./N (not a division); stddev is
// sqrt(max(sqsum * scale - mean * mean, 0)). For 8-bit input every filtered value is an integer (or a multiple of
// 1/4 for LAPM), so the sums are exact in f64 in any order and the result is order-independent.
#include "oracle_common.h"

using namespace orc;

namespace {

struct Img { int w, h; std::vector<double> v; double at(int x, int y) const { return v[(size_t)y * w + x]; } };

// separable filter, f64 buffer: row pass (along x) with kx, then column pass with ky, anchored at the centre
void sep_filter(const Img& s, const std::vector<double>& kx, const std::vector<double>& ky, int border, std::vector<double>& out) {
    const int w = s.w, h = s.h, rx = (int)kx.size() / 2, ry = (int)ky.size() / 2;
    std::vector<double> rowf((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double acc = 0;
            for (int i = 0; i < (int)kx.size(); i++) acc += kx[i] * s.at(border_interpolate(x + i - rx, w, border), y);
            rowf[(size_t)y * w + x] = acc;
        }
    out.assign((size_t)w * h, 0.0);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double acc = 0;
            for (int j = 0; j < (int)ky.size(); j++) acc += ky[j] * rowf[(size_t)border_interpolate(y + j - ry, h, border) * w + x];
            out[(size_t)y * w + x] = acc;
        }
}

// getDerivKernels(dx, dy, ksize) without normalisation: the smoothing and the first-derivative kernel
void sobel_kernels(int ksize, std::vector<double>& smooth, std::vector<double>& deriv) {
    switch (ksize) {
        case 1: smooth = {1}; deriv = {-1, 0, 1}; break;                         // ksize 1: a 3x1 / 1x3 kernel
        case 3: smooth = {1, 2, 1}; deriv = {-1, 0, 1}; break;
        case 5: smooth = {1, 4, 6, 4, 1}; deriv = {-1, -2, 0, 2, 1}; break;
        default: smooth = {1, 6, 15, 20, 15, 6, 1}; deriv = {-1, -4, -5, 0, 5, 4, 1}; break;
    }
}

void mean_std(const std::vector<double>& v, double& mean, double& sigma) {
    double s = 0, sq = 0;
    for (double e : v) { s += e; sq += e * e; }
    const double scale = v.empty() ? 0. : 1. / (double)v.size();
    mean = s * scale;
    sigma = std::sqrt(std::max(sq * scale - mean * mean, 0.));
}

}  // namespace

extern "C" {

// metric: 0 LAPM, 1 LAPV, 2 TENG (ksize), 3 GLVN. depth 8 or 32 (f32). Returns 0, or 2 for invalid parameters.
int orc_sharpness(const void* img, int depth, int w, int h, int metric, int ksize, double* out) {
    if (!img || !out || w <= 0 || h <= 0 || (depth != 8 && depth != 32) || metric < 0 || metric > 3) return 2;
    if (metric == 2 && ksize != 1 && ksize != 3 && ksize != 5 && ksize != 7) return 2;     // lib.rs:1105-1109
    Img s{w, h, std::vector<double>((size_t)w * h)};
    for (size_t i = 0; i < s.v.size(); i++) s.v[i] = depth == 8 ? (double)((const uint8_t*)img)[i] : (double)((const float*)img)[i];
    const double scale = 1. / ((double)w * h);
    if (metric == 0) {
        const std::vector<double> m = {-1, 2, -1}, g = {0.25, 0.5, 0.25};
        std::vector<double> lx, ly;
        sep_filter(s, m, g, BORDER_REFLECT_101, lx);
        sep_filter(s, g, m, BORDER_REFLECT_101, ly);
        double sum = 0;
        for (size_t i = 0; i < lx.size(); i++) sum += std::fabs(lx[i]) + std::fabs(ly[i]);
        *out = sum * scale;
    } else if (metric == 1) {
        std::vector<double> lap((size_t)w * h);
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const int xm = border_interpolate(x - 1, w, BORDER_REPLICATE), xp = border_interpolate(x + 1, w, BORDER_REPLICATE);
                const int ym = border_interpolate(y - 1, h, BORDER_REPLICATE), yp = border_interpolate(y + 1, h, BORDER_REPLICATE);
                lap[(size_t)y * w + x] = 2 * s.at(xm, ym) + 2 * s.at(xp, ym) - 8 * s.at(x, y) + 2 * s.at(xm, yp) + 2 * s.at(xp, yp);
            }
        double mean, sigma;
        mean_std(lap, mean, sigma);
        *out = sigma * sigma;
    } else if (metric == 2) {
        std::vector<double> sm, dv, gx, gy;
        sobel_kernels(ksize, sm, dv);
        sep_filter(s, dv, sm, BORDER_REFLECT_101, gx);      // d/dx: derivative along x, smoothing along y
        sep_filter(s, sm, dv, BORDER_REFLECT_101, gy);
        double sum = 0;
        for (size_t i = 0; i < gx.size(); i++) sum += gx[i] * gx[i] + gy[i] * gy[i];
        *out = sum * scale;
    } else {
        double mean, sigma;
        mean_std(s.v, mean, sigma);
        *out = (sigma * sigma) / std::max(mean, DBL_EPSILON);
    }
    return 0;
}

}  // extern "C"
