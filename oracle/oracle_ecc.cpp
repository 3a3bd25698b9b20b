// oracle_ecc.cpp — CPU restatement of cv::findTransformECC (OpenCV 4.12 video/src/ecc.cpp) as
// libstacker calls it (lib.rs:769-777), plus the whole ecc_match() driver (lib.rs:719-847).
// TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle_common.h). Algorithm: SURVEY.md §8a-E*.
#include "oracle_common.h"
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

extern "C" {
int orc_grey(const void* bgr, int depth, int w, int h, size_t stride_bytes, void* out);
int orc_gaussian_blur_f32(const void* src, int depth, int w, int h, int ksize, float* out);
int orc_gradients(const float* img, int w, int h, float* gx, float* gy);
int orc_warp_frame(const void* src, int depth, int w, int h, int cn, size_t stride_bytes,
                   const double* M, int is_affine, int border_mode, const double* border_value,
                   double alpha, int subpixel_bits, float* dst, int accumulate);
int orc_scale(const float* in, size_t n, double divisor, float* out);
int orc_scaled_size(int w, int h, float scale_down, int* nw, int* nh);
int orc_resize_area_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh);
int orc_resize_area_f32(const float* src, int sw, int sh, float* dst, int dw, int dh);
}

namespace {

// hal::LU32f / LU64f (core/src/matrix_decomp.cpp LUImpl): partial pivoting on [A|b], in place.
template <typename T>
int lu_impl(T* A, int astep, int m, T* b, int bstep, int n, T eps) {
    int p = 1;
    for (int i = 0; i < m; i++) {
        int k = i;
        for (int j = i + 1; j < m; j++)
            if (std::abs(A[j * astep + i]) > std::abs(A[k * astep + i])) k = j;
        if (std::abs(A[k * astep + i]) < eps) return 0;
        if (k != i) {
            for (int j = i; j < m; j++) std::swap(A[i * astep + j], A[k * astep + j]);
            if (b) for (int j = 0; j < n; j++) std::swap(b[i * bstep + j], b[k * bstep + j]);
            p = -p;
        }
        T d = -1 / A[i * astep + i];
        for (int j = i + 1; j < m; j++) {
            T alpha = A[j * astep + i] * d;
            for (int kk = i + 1; kk < m; kk++) A[j * astep + kk] += alpha * A[i * astep + kk];
            if (b) for (int kk = 0; kk < n; kk++) b[j * bstep + kk] += alpha * b[i * bstep + kk];
        }
    }
    if (b) {
        for (int i = m - 1; i >= 0; i--)
            for (int j = 0; j < n; j++) {
                T s = b[i * bstep + j];
                for (int k = i + 1; k < m; k++) s -= A[i * astep + k] * b[k * bstep + j];
                b[i * bstep + j] = s / A[i * astep + i];
            }
    }
    return p;
}

// Mat::inv(DECOMP_LU) for an n x n CV_32F matrix (core/src/lapack.cpp cv::invert):
// n==2, n==3 closed forms evaluated in double; otherwise LU32f on [A | I]; singular -> zeros.
void invert_f32(const float* S, int n, float* D) {
    if (n == 2) {
        double d = (double)S[0] * S[3] - (double)S[1] * S[2];
        if (d != 0.) {
            d = 1. / d;
            double t0 = S[0] * d, t1 = S[3] * d;
            D[3] = (float)t0; D[0] = (float)t1;
            t0 = -S[1] * d; t1 = -S[2] * d;
            D[1] = (float)t0; D[2] = (float)t1;
        } else for (int i = 0; i < 4; i++) D[i] = 0;
        return;
    }
    if (n == 3) {
        auto s = [&](int y, int x) { return (double)S[y * 3 + x]; };
        double d = s(0,0) * (s(1,1) * s(2,2) - s(1,2) * s(2,1)) - s(0,1) * (s(1,0) * s(2,2) - s(1,2) * s(2,0)) +
                   s(0,2) * (s(1,0) * s(2,1) - s(1,1) * s(2,0));
        if (d != 0.) {
            d = 1. / d;
            D[0] = (float)((s(1,1) * s(2,2) - s(1,2) * s(2,1)) * d);
            D[1] = (float)((s(0,2) * s(2,1) - s(0,1) * s(2,2)) * d);
            D[2] = (float)((s(0,1) * s(1,2) - s(0,2) * s(1,1)) * d);
            D[3] = (float)((s(1,2) * s(2,0) - s(1,0) * s(2,2)) * d);
            D[4] = (float)((s(0,0) * s(2,2) - s(0,2) * s(2,0)) * d);
            D[5] = (float)((s(0,2) * s(1,0) - s(0,0) * s(1,2)) * d);
            D[6] = (float)((s(1,0) * s(2,1) - s(1,1) * s(2,0)) * d);
            D[7] = (float)((s(0,1) * s(2,0) - s(0,0) * s(2,1)) * d);
            D[8] = (float)((s(0,0) * s(1,1) - s(0,1) * s(1,0)) * d);
        } else for (int i = 0; i < 9; i++) D[i] = 0;
        return;
    }
    float A[64];
    for (int i = 0; i < n * n; i++) A[i] = S[i];
    for (int i = 0; i < n * n; i++) D[i] = 0;
    for (int i = 0; i < n; i++) D[i * n + i] = 1.f;
    if (lu_impl<float>(A, n, n, D, n, n, FLT_EPSILON * 10) == 0)
        for (int i = 0; i < n * n; i++) D[i] = 0;
}

struct Planes { std::vector<float> img, gx, gy; int w, h; };

// bilinear sample of an f32 plane exactly as orc_warp's subpixel_bits==0 branch with border 0
inline float sample(const float* p, int w, int h, int ix, int iy, float ax, float ay) {
    auto at = [&](int x, int y) -> float { return ((unsigned)x < (unsigned)w && (unsigned)y < (unsigned)h) ? p[(size_t)y * w + x] : 0.f; };
    float p00 = at(ix, iy), p01 = at(ix + 1, iy), p10 = at(ix, iy + 1), p11 = at(ix + 1, iy + 1);
    float v0 = std::fmaf(ax, p01 - p00, p00);
    float v1 = std::fmaf(ax, p11 - p10, p10);
    return std::fmaf(ay, v1 - v0, v0);
}

}  // namespace

extern "C" {

// Prepared, frame-0-derived state shared by every frame of a stack.
struct orc_ecc_input {
    Planes p;
};

orc_ecc_input* orc_ecc_prepare_input(const void* input, int depth, int w, int h, int gauss) {
    orc_ecc_input* s = new orc_ecc_input();
    s->p.w = w; s->p.h = h;
    s->p.img.resize((size_t)w * h); s->p.gx.resize((size_t)w * h); s->p.gy.resize((size_t)w * h);
    if (orc_gaussian_blur_f32(input, depth, w, h, gauss, s->p.img.data())) { delete s; return nullptr; }
    // preMask is all ones (no input mask; blurred*0.5/0.95 rounds to 1), so the products are no-ops.
    orc_gradients(s->p.img.data(), w, h, s->p.gx.data(), s->p.gy.data());
    return s;
}
void orc_ecc_free_input(orc_ecc_input* s) { delete s; }

// The iteration loop, given the blurred template and the prepared input planes.
// warp: 9 floats row-major (2x3 motions: rows 0,1; row 2 ignored). Returns 0 ok, 1 NaN,
// 2 lambda_d <= 0 ("correlation is going to be minimized"), 3 bad arguments.
int orc_ecc_run(const float* templ, int tw, int th, const orc_ecc_input* in, float* warp,
                int motion, int has_count, int max_count, int has_eps, double eps,
                double* rho_out, int* iters_out) {
    if (!has_count && !has_eps) return 3;             // CV_Assert(criteria.type & (COUNT|EPS))
    const int nIter = has_count ? max_count : 200;
    const double term_eps = has_eps ? eps : -1;
    const int P = motion == MOTION_HOMOGRAPHY ? 8 : motion == MOTION_AFFINE ? 6 : motion == MOTION_EUCLIDEAN ? 3 : 2;
    const int iw = in->p.w, ih = in->p.h;
    const size_t N = (size_t)tw * th;
    std::vector<float> Iw(N), Gx(N), Gy(N);
    std::vector<uint8_t> mask(N);
    float* m = warp;  // m[0..8]
    if (motion != MOTION_HOMOGRAPHY) { m[6] = 0; m[7] = 0; m[8] = 1; }

    double rho = -1, last_rho = -term_eps;
    int it = 0;
    for (int i = 1; i <= nIter && std::fabs(rho - last_rho) >= term_eps; i++) {
        it = i;
        // (a) warps with WARP_INVERSE_MAP: map used directly. Bilinear planes through the f32
        // kernels; mask through the classic INTER_NEAREST path (double / fixed point).
        double n_mask = 0, sI = 0, sII = 0, sT = 0, sTT = 0;
        double Md[9]; for (int k = 0; k < 9; k++) Md[k] = m[k];
        #pragma omp parallel for schedule(static) reduction(+ : n_mask, sI, sII, sT, sTT)
        for (int y = 0; y < th; y++) {
            for (int x = 0; x < tw; x++) {
                float fx = (float)x, fy = (float)y;
                float X = std::fmaf(m[0], fx, std::fmaf(m[1], fy, m[2]));
                float Y = std::fmaf(m[3], fx, std::fmaf(m[4], fy, m[5]));
                int mx, my;
                if (motion == MOTION_HOMOGRAPHY) {
                    float W = std::fmaf(m[6], fx, std::fmaf(m[7], fy, m[8]));
                    X = X / W; Y = Y / W;
                    double Wd = Md[6] * x + Md[7] * y + Md[8];
                    Wd = Wd != 0 ? 1.0 / Wd : 0;
                    double fX = std::max(-2147483648.0, std::min(2147483647.0, (Md[0] * x + Md[1] * y + Md[2]) * Wd));
                    double fY = std::max(-2147483648.0, std::min(2147483647.0, (Md[3] * x + Md[4] * y + Md[5]) * Wd));
                    mx = sat_int(fX); my = sat_int(fY);
                } else {
                    int adx = sat_int(Md[0] * x * 1024), bdx = sat_int(Md[3] * x * 1024);
                    int X0 = sat_int((Md[1] * y + Md[2]) * 1024) + 512;
                    int Y0 = sat_int((Md[4] * y + Md[5]) * 1024) + 512;
                    mx = (X0 + adx) >> 10; my = (Y0 + bdx) >> 10;
                }
                bool fin = std::isfinite(X) && std::isfinite(Y) && std::fabs(X) < 1e9f && std::fabs(Y) < 1e9f;
                float flx = std::floor(X), fly = std::floor(Y);
                int ix = fin ? (int)flx : -100000, iy = fin ? (int)fly : -100000;
                float ax = X - flx, ay = Y - fly;
                size_t idx = (size_t)y * tw + x;
                Iw[idx] = sample(in->p.img.data(), iw, ih, ix, iy, ax, ay);
                Gx[idx] = sample(in->p.gx.data(), iw, ih, ix, iy, ax, ay);
                Gy[idx] = sample(in->p.gy.data(), iw, ih, ix, iy, ax, ay);
                uint8_t mk = ((unsigned)mx < (unsigned)iw && (unsigned)my < (unsigned)ih) ? 1 : 0;
                mask[idx] = mk;
                if (mk) {
                    double v = Iw[idx], t = templ[idx];
                    n_mask += 1; sI += v; sII += v * v; sT += t; sTT += t * t;
                }
            }
        }
        // (b) meanStdDev with mask (double accumulators)
        double imgMean = n_mask > 0 ? sI / n_mask : 0, tmpMean = n_mask > 0 ? sT / n_mask : 0;
        double imgVar = n_mask > 0 ? std::max(sII / n_mask - imgMean * imgMean, 0.) : 0;
        double tmpVar = n_mask > 0 ? std::max(sTT / n_mask - tmpMean * tmpMean, 0.) : 0;
        double imgStd = std::sqrt(imgVar), tmpStd = std::sqrt(tmpVar);
        // (d)
        const double tmpNorm = std::sqrt(n_mask * tmpStd * tmpStd);
        const double imgNorm = std::sqrt(n_mask * imgStd * imgStd);
        const float imgMeanF = (float)imgMean, tmpMeanF = (float)tmpMean;

        // (c,e,f,g,h) zero-mean images, Jacobian, Hessian, projections in one sweep
        double H[64]; double ip[8], tp[8]; double corr = 0;
        for (int k = 0; k < 64; k++) H[k] = 0;
        for (int k = 0; k < 8; k++) ip[k] = tp[k] = 0;
        const float h0 = m[0], h1 = m[3], h2 = m[6], h3 = m[1], h4 = m[4], h5 = m[7], h6 = m[2], h7 = m[5];
        #pragma omp parallel
        {
            double Hl[64], ipl[8], tpl[8], corrl = 0;
            for (int k = 0; k < 64; k++) Hl[k] = 0;
            for (int k = 0; k < 8; k++) ipl[k] = tpl[k] = 0;
            #pragma omp for schedule(static) nowait
            for (int y = 0; y < th; y++) {
                for (int x = 0; x < tw; x++) {
                    size_t idx = (size_t)y * tw + x;
                    const float Xg = (float)x, Yg = (float)y;
                    const float gx = Gx[idx], gy = Gy[idx];
                    float J[8];
                    if (motion == MOTION_HOMOGRAPHY) {
                        float den = Xg * h2 + Yg * h5 + 1.0f;
                        float hatX = (-Xg * h0 - Yg * h3 - h6) / den;
                        float hatY = (-Xg * h1 - Yg * h4 - h7) / den;
                        float a = gx / den, b = gy / den;
                        float t = hatX * a + hatY * b;
                        J[0] = a * Xg; J[1] = b * Xg; J[2] = t * Xg;
                        J[3] = a * Yg; J[4] = b * Yg; J[5] = t * Yg;
                        J[6] = a; J[7] = b;
                    } else if (motion == MOTION_AFFINE) {
                        J[0] = gx * Xg; J[1] = gy * Xg; J[2] = gx * Yg; J[3] = gy * Yg; J[4] = gx; J[5] = gy;
                    } else if (motion == MOTION_EUCLIDEAN) {
                        float hatX = -(Xg * h1) - (Yg * h0);
                        float hatY = (Xg * h0) - (Yg * h1);
                        J[0] = gx * hatX + gy * hatY; J[1] = gx; J[2] = gy;
                    } else {
                        J[0] = gx; J[1] = gy;
                    }
                    const uint8_t mk = mask[idx];
                    const float iz = mk ? Iw[idx] - imgMeanF : Iw[idx];
                    const float tz = mk ? templ[idx] - tmpMeanF : 0.f;
                    corrl += (double)tz * iz;
                    for (int k = 0; k < P; k++) {
                        ipl[k] += (double)J[k] * iz;
                        tpl[k] += (double)J[k] * tz;
                        for (int l = k; l < P; l++) Hl[k * 8 + l] += (double)J[k] * J[l];
                    }
                }
            }
            #pragma omp critical
            {
                for (int k = 0; k < 64; k++) H[k] += Hl[k];
                for (int k = 0; k < 8; k++) { ip[k] += ipl[k]; tp[k] += tpl[k]; }
                corr += corrl;
            }
        }
        float Hf[64], Hinv[64], ipf[8], tpf[8];
        for (int k = 0; k < P; k++) {
            for (int l = k; l < P; l++) { Hf[k * P + l] = (float)H[k * 8 + l]; Hf[l * P + k] = Hf[k * P + l]; }
            ipf[k] = (float)ip[k]; tpf[k] = (float)tp[k];
        }
        invert_f32(Hf, P, Hinv);

        const double correlation = corr;
        last_rho = rho;
        rho = correlation / (imgNorm * tmpNorm);
        if (std::isnan(rho)) { if (rho_out) *rho_out = rho; if (iters_out) *iters_out = it; return 1; }

        // (i) imageProjectionHessian = hessianInv * imageProjection (f32 gemm)
        float iph[8];
        for (int k = 0; k < P; k++) { float s = 0; for (int l = 0; l < P; l++) s += Hinv[k * P + l] * ipf[l]; iph[k] = s; }
        double dot_ip = 0, dot_tp = 0;
        for (int k = 0; k < P; k++) { dot_ip += (double)ipf[k] * iph[k]; dot_tp += (double)tpf[k] * iph[k]; }
        const double lambda_n = imgNorm * imgNorm - dot_ip;
        const double lambda_d = correlation - dot_tp;
        if (lambda_d <= 0.0) { rho = -1; if (rho_out) *rho_out = rho; if (iters_out) *iters_out = it; return 2; }
        const double lambda = lambda_n / lambda_d;

        // (j) error = lambda*templateZM - imageWarped (f32 per pixel), projected onto J
        double ep[8]; for (int k = 0; k < 8; k++) ep[k] = 0;
        const float lamf = (float)lambda;
        #pragma omp parallel
        {
            double epl[8]; for (int k = 0; k < 8; k++) epl[k] = 0;
            #pragma omp for schedule(static) nowait
            for (int y = 0; y < th; y++) {
                for (int x = 0; x < tw; x++) {
                    size_t idx = (size_t)y * tw + x;
                    const float Xg = (float)x, Yg = (float)y;
                    const float gx = Gx[idx], gy = Gy[idx];
                    float J[8];
                    if (motion == MOTION_HOMOGRAPHY) {
                        float den = Xg * h2 + Yg * h5 + 1.0f;
                        float hatX = (-Xg * h0 - Yg * h3 - h6) / den;
                        float hatY = (-Xg * h1 - Yg * h4 - h7) / den;
                        float a = gx / den, b = gy / den;
                        float t = hatX * a + hatY * b;
                        J[0] = a * Xg; J[1] = b * Xg; J[2] = t * Xg;
                        J[3] = a * Yg; J[4] = b * Yg; J[5] = t * Yg;
                        J[6] = a; J[7] = b;
                    } else if (motion == MOTION_AFFINE) {
                        J[0] = gx * Xg; J[1] = gy * Xg; J[2] = gx * Yg; J[3] = gy * Yg; J[4] = gx; J[5] = gy;
                    } else if (motion == MOTION_EUCLIDEAN) {
                        float hatX = -(Xg * h1) - (Yg * h0);
                        float hatY = (Xg * h0) - (Yg * h1);
                        J[0] = gx * hatX + gy * hatY; J[1] = gx; J[2] = gy;
                    } else { J[0] = gx; J[1] = gy; }
                    const uint8_t mk = mask[idx];
                    const float iz = mk ? Iw[idx] - imgMeanF : Iw[idx];
                    const float tz = mk ? templ[idx] - tmpMeanF : 0.f;
                    const float e = lamf * tz - iz;
                    for (int k = 0; k < P; k++) epl[k] += (double)J[k] * e;
                }
            }
            #pragma omp critical
            { for (int k = 0; k < 8; k++) ep[k] += epl[k]; }
        }
        float epf[8], dp[8];
        for (int k = 0; k < P; k++) epf[k] = (float)ep[k];
        for (int k = 0; k < P; k++) { float s = 0; for (int l = 0; l < P; l++) s += Hinv[k * P + l] * epf[l]; dp[k] = s; }

        // (k) update_warping_matrix_ECC
        if (motion == MOTION_HOMOGRAPHY) {
            m[0] += dp[0]; m[3] += dp[1]; m[6] += dp[2]; m[1] += dp[3]; m[4] += dp[4]; m[7] += dp[5]; m[2] += dp[6]; m[5] += dp[7];
        } else if (motion == MOTION_AFFINE) {
            m[0] += dp[0]; m[3] += dp[1]; m[1] += dp[2]; m[4] += dp[3]; m[2] += dp[4]; m[5] += dp[5];
        } else if (motion == MOTION_TRANSLATION) {
            m[2] += dp[0]; m[5] += dp[1];
        } else {
            double new_theta = (double)dp[0];
            new_theta += std::asin((double)m[3]);
            m[2] += dp[1]; m[5] += dp[2];
            m[0] = m[4] = (float)std::cos(new_theta);
            m[3] = (float)std::sin(new_theta);
            m[1] = -m[3];
        }
    }
    if (rho_out) *rho_out = rho;
    if (iters_out) *iters_out = it;
    return 0;
}

// video::find_transform_ecc(template, input, warp, motion, criteria, noArray(), gaussFiltSize).
int orc_find_transform_ecc(const void* templ, int tw, int th, const void* input, int iw, int ih,
                           int depth, float* warp, int motion, int has_count, int max_count,
                           int has_eps, double eps, int gauss, double* rho_out, int* iters_out) {
    if (depth != 8 && depth != 32) return 3;
    orc_ecc_input* in = orc_ecc_prepare_input(input, depth, iw, ih, gauss);
    if (!in) return 3;
    std::vector<float> tf((size_t)tw * th);
    orc_gaussian_blur_f32(templ, depth, tw, th, gauss, tf.data());
    int rc = orc_ecc_run(tf.data(), tw, th, in, warp, motion, has_count, max_count, has_eps, eps, rho_out, iters_out);
    orc_ecc_free_input(in);
    return rc;
}

// ecc_match_no_scaling, lib.rs:719-847. frames: n pointers to BGR u8 images (w*h*3, tight).
// Frame-parallel with one private accumulator per thread and a final pairwise sum, mirroring
// the Rayon try_fold/try_reduce. warps_out (optional): n*9 floats. Returns 0 or 10+ecc error.
// scale_down > 0 selects ecc_match_scaling_down (lib.rs:849-1028): ECC on INTER_AREA-shrunk greys, then
// the translation rescale (affine family, lib.rs:941-951) or adjust_homography_for_scale_f32 (utils.rs:218-248).
int orc_ecc_match(const void* const* frames, int n, int w, int h, int depth, int motion, int has_count,
                  int max_count, int has_eps, double eps, int gauss, float scale_down, float* out, float* warps_out,
                  int* iters_out, int n_threads) {
    if (n <= 0) return 1;                               // NotEnoughFiles lib.rs:725, 859
    if (!has_count && !has_eps) return 4;
    int ew = w, eh = h;                                 // size the ECC runs at
    if (scale_down > 0) {
        if (scale_down >= (float)w) return 2;           // InvalidParams lib.rs:876-881
        if (scale_down <= 10.0f) return 2;              // InvalidParams lib.rs:883-888
        if (orc_scaled_size(w, h, scale_down, &ew, &eh)) return 2;
    }
    const size_t npx = (size_t)w * h, nel = npx * 3;
    const size_t gsz = depth / 8;
    std::vector<uint8_t> grey0(npx * gsz);
    orc_grey(frames[0], depth, w, h, 0, grey0.data());
    // findTransformECC accepts 8UC1 / 32FC1 only: 16-bit input fails in the reference.
    if (depth == 16) return 4;
    std::vector<uint8_t> small0;
    // (scale_image keeps the grey's depth: resize(INTER_AREA) on 8UC1 or 32FC1, utils.rs:186-214)
    auto shrink = [&](const std::vector<uint8_t>& g, std::vector<uint8_t>& sm) {
        sm.resize((size_t)ew * eh * gsz);
        if (depth == 8) orc_resize_area_u8(g.data(), w, h, sm.data(), ew, eh);
        else orc_resize_area_f32(reinterpret_cast<const float*>(g.data()), w, h, reinterpret_cast<float*>(sm.data()), ew, eh);
    };
    if (scale_down > 0) shrink(grey0, small0);
    orc_ecc_input* in = orc_ecc_prepare_input(scale_down > 0 ? small0.data() : grey0.data(), depth, ew, eh, gauss);
    if (!in) return 4;
#ifdef _OPENMP
    const int T = n_threads > 0 ? n_threads : omp_get_max_threads();
#else
    const int T = 1;
#endif
    std::vector<std::vector<float>> accs(T);
    int err = 0;
    const double alpha = 1.0 / 255.0;
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    #pragma omp parallel for schedule(dynamic, 1) num_threads(T)
    for (int i = 0; i < n; i++) {
#ifdef _OPENMP
        int tid = omp_get_thread_num();
#else
        int tid = 0;
#endif
        if (err) continue;
        std::vector<float>& acc = accs[tid];
        bool fresh = acc.empty();
        if (fresh) acc.assign(nel, 0.f);
        float wm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        int its = 0;
        if (i > 0) {
            std::vector<uint8_t> grey(npx * gsz);
            orc_grey(frames[i], depth, w, h, 0, grey.data());
            std::vector<float> tf((size_t)ew * eh);
            if (scale_down > 0) {
                std::vector<uint8_t> sm;
                shrink(grey, sm);
                orc_gaussian_blur_f32(sm.data(), depth, ew, eh, gauss, tf.data());
            } else orc_gaussian_blur_f32(grey.data(), depth, w, h, gauss, tf.data());
            double rho;
            // inner loops are OpenMP-parallel too; nested regions stay serial by default
            const int rc = orc_ecc_run(tf.data(), ew, eh, in, wm, motion, has_count, max_count, has_eps, eps, &rho, &its);
            if (rc) { err = 10 + rc; continue; }
            if (scale_down > 0) {
                if (motion != MOTION_HOMOGRAPHY) {          // lib.rs:941-951: only the translation column is rescaled
                    wm[2] *= (float)w / (float)ew;
                    wm[5] *= (float)h / (float)eh;
                } else {                                    // utils.rs:229-239 with $type = f32
                    const double sx = (double)w / (double)ew, sy = (double)h / (double)eh;
                    wm[2] *= (float)sx; wm[5] *= (float)sy; wm[6] /= (float)sx; wm[7] /= (float)sy;
                }
            }
        }
        double Md[9]; for (int k = 0; k < 9; k++) Md[k] = wm[k];
        if (i == 0) orc_warp_frame(frames[0], depth, w, h, 3, 0, I3, 1, BORDER_CONSTANT, nullptr, alpha, 0, acc.data(), !fresh ? 1 : 0);
        else orc_warp_frame(frames[i], depth, w, h, 3, 0, Md, motion != MOTION_HOMOGRAPHY, BORDER_CONSTANT, nullptr, alpha, 0, acc.data(), !fresh ? 1 : 0);
        if (warps_out) for (int k = 0; k < 9; k++) warps_out[(size_t)i * 9 + k] = wm[k];
        if (iters_out) iters_out[i] = its;
    }
    orc_ecc_free_input(in);
    if (err) return err;
    std::vector<float> total(nel, 0.f);
    bool first = true;
    for (int t = 0; t < T; t++) {
        if (accs[t].empty()) continue;
        if (first) { total = accs[t]; first = false; }
        else for (size_t k = 0; k < nel; k++) total[k] = total[k] + accs[t][k];
    }
    orc_scale(total.data(), nel, (double)n, out);
    return 0;
}

}  // extern "C"
