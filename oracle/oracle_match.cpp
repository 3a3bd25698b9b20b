// oracle_match.cpp — CPU restatement of the matching / geometry half of keypoint_match
// (lib.rs:146-353): BFMatcher(NORM_HAMMING).knn_match k=2, Lowe ratio + stable sort + truncate,
// calib3d::find_homography (RANSAC / least squares) with the LM refinement, validity checks,
// and the whole driver. TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle_common.h).
// OpenCV sources restated: core/src/batch_distance.cpp, calib3d/src/fundam.cpp, ptsetreg.cpp,
// levmarq.cpp, core/src/lapack.cpp (Jacobi eigen), core/src/rand.cpp (RNG).
#include "oracle_common.h"
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

extern "C" {
int orc_grey(const void* bgr, int depth, int w, int h, size_t stride_bytes, void* out);
int orc_warp_frame(const void* src, int depth, int w, int h, int cn, size_t stride_bytes, const double* M,
                   int is_affine, int border_mode, const double* border_value, double alpha, int subpixel_bits,
                   float* dst, int accumulate);
int orc_warp_frame_sized(const void* src, int depth, int sw, int sh, int cn, size_t stride_bytes, const double* M,
                         int is_affine, int border_mode, const double* border_value, double alpha, int subpixel_bits,
                         float* dst, int dw, int dh, int accumulate);
int orc_scale(const float* in, size_t n, double divisor, float* out);
int orc_scaled_size(int w, int h, float scale_down, int* nw, int* nh);
int orc_resize_area_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh);
int orc_orb_detect_and_compute(const uint8_t* grey, int w, int h, int max_keypoints, float* kp_out,
                               uint8_t* desc_out, int* n_out);
}

namespace {

// ---- cv::RNG (multiply-with-carry) ---------------------------------------------------------------
struct Rng {
    uint64_t state;
    explicit Rng(uint64_t s) : state(s ? s : 0xffffffffull) {}
    unsigned next() { state = (uint64_t)(unsigned)state * 4164903690u + (unsigned)(state >> 32); return (unsigned)state; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// ---- symmetric eigen-decomposition, cv::eigen -> JacobiImpl_<double> ----------------------------------
// eigenvalues descending in W, eigenvectors in the ROWS of V.
void jacobi(double* A, int n, double* W, double* V) {
    const double eps = DBL_EPSILON;
    std::vector<int> indR(n), indC(n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i * n + j] = i == j;
    double mv = 0;
    int m;
    for (int k = 0; k < n; k++) {
        W[k] = A[(n + 1) * k];
        if (k < n - 1) {
            int i; for (m = k + 1, mv = std::abs(A[n * k + m]), i = k + 2; i < n; i++) { double v = std::abs(A[n * k + i]); if (mv < v) mv = v, m = i; }
            indR[k] = m;
        }
        if (k > 0) {
            int i; for (m = 0, mv = std::abs(A[k]), i = 1; i < k; i++) { double v = std::abs(A[n * i + k]); if (mv < v) mv = v, m = i; }
            indC[k] = m;
        }
    }
    if (n > 1) for (int iters = 0, maxIters = n * n * 30; iters < maxIters; iters++) {
        int k, i;
        for (k = 0, mv = std::abs(A[indR[0]]), i = 1; i < n - 1; i++) { double v = std::abs(A[n * i + indR[i]]); if (mv < v) mv = v, k = i; }
        int l = indR[k];
        for (i = 1; i < n; i++) { double v = std::abs(A[n * indC[i] + i]); if (mv < v) mv = v, k = indC[i], l = i; }
        double p = A[n * k + l];
        if (std::abs(p) <= eps) break;
        double y = (W[l] - W[k]) * 0.5;
        double t = std::abs(y) + std::hypot(p, y);
        double s = std::hypot(p, t);
        double c = t / s;
        s = p / s; t = (p / t) * p;
        if (y < 0) s = -s, t = -t;
        A[n * k + l] = 0;
        W[k] -= t; W[l] += t;
        double a0, b0;
#define ROT(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * s, v1 = a0 * s + b0 * c
        for (i = 0; i < k; i++) ROT(A[n * i + k], A[n * i + l]);
        for (i = k + 1; i < l; i++) ROT(A[n * k + i], A[n * i + l]);
        for (i = l + 1; i < n; i++) ROT(A[n * k + i], A[n * l + i]);
        for (i = 0; i < n; i++) ROT(V[n * k + i], V[n * l + i]);
#undef ROT
        for (int j = 0; j < 2; j++) {
            int idx = j == 0 ? k : l;
            if (idx < n - 1) {
                for (m = idx + 1, mv = std::abs(A[n * idx + m]), i = idx + 2; i < n; i++) { double v = std::abs(A[n * idx + i]); if (mv < v) mv = v, m = i; }
                indR[idx] = m;
            }
            if (idx > 0) {
                for (m = 0, mv = std::abs(A[idx]), i = 1; i < idx; i++) { double v = std::abs(A[n * i + idx]); if (mv < v) mv = v, m = i; }
                indC[idx] = m;
            }
        }
    }
    for (int k = 0; k < n - 1; k++) {
        m = k;
        for (int i = k + 1; i < n; i++) if (W[m] < W[i]) m = i;
        if (k != m) { std::swap(W[m], W[k]); for (int i = 0; i < n; i++) std::swap(V[n * m + i], V[n * k + i]); }
    }
}

// cv::solve / cv::invert with DECOMP_EIG on a symmetric n x n system (SVBkSb with the eigenvectors)
void eig_solve(const double* A, int n, const double* b, int nb, double* x) {
    std::vector<double> a(A, A + n * n), w(n), v(n * n);
    jacobi(a.data(), n, w.data(), v.data());
    double thr = 0;
    for (int i = 0; i < n; i++) thr += w[i];
    thr *= DBL_EPSILON * 2;
    for (int i = 0; i < n * nb; i++) x[i] = 0;
    for (int i = 0; i < n; i++) {
        if (std::abs(w[i]) <= thr) continue;
        const double wi = 1.0 / w[i];
        for (int j = 0; j < nb; j++) {
            double s = 0;
            for (int k = 0; k < n; k++) s += v[i * n + k] * b[k * nb + j];
            s *= wi;
            for (int k = 0; k < n; k++) x[k * nb + j] += s * v[i * n + k];
        }
    }
}

struct P2 { float x, y; };

// HomographyEstimatorCallback::runKernel — normalised DLT; M -> m. Returns false when degenerate.
bool dlt(const P2* M, const P2* m, int count, double* H) {
    double LtL[81], W[9], V[81];
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < count; i++) { cmx += m[i].x; cmy += m[i].y; cMx += M[i].x; cMy += M[i].y; }
    cmx /= count; cmy /= count; cMx /= count; cMy /= count;
    for (int i = 0; i < count; i++) {
        smx += std::fabs(m[i].x - cmx); smy += std::fabs(m[i].y - cmy);
        sMx += std::fabs(M[i].x - cMx); sMy += std::fabs(M[i].y - cMy);
    }
    if (std::fabs(smx) < DBL_EPSILON || std::fabs(smy) < DBL_EPSILON || std::fabs(sMx) < DBL_EPSILON || std::fabs(sMy) < DBL_EPSILON) return false;
    smx = count / smx; smy = count / smy; sMx = count / sMx; sMy = count / sMy;
    const double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
    const double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
    for (int i = 0; i < 81; i++) LtL[i] = 0;
    for (int i = 0; i < count; i++) {
        const double x = (m[i].x - cmx) * smx, y = (m[i].y - cmy) * smy;
        const double X = (M[i].x - cMx) * sMx, Y = (M[i].y - cMy) * sMy;
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int j = 0; j < 9; j++) for (int k = j; k < 9; k++) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    for (int j = 0; j < 9; j++) for (int k = 0; k < j; k++) LtL[j * 9 + k] = LtL[k * 9 + j];
    jacobi(LtL, 9, W, V);
    const double* H0 = V + 72;                       // eigenvector of the smallest eigenvalue
    double T[9], R[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += invHnorm[i * 3 + k] * H0[k * 3 + j]; T[i * 3 + j] = s; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += T[i * 3 + k] * Hnorm2[k * 3 + j]; R[i * 3 + j] = s; }
    const double sc = 1. / R[8];
    for (int i = 0; i < 9; i++) H[i] = R[i] * sc;
    return true;
}

bool collinear(const P2* p, int count) {
    const int i = count - 1;
    for (int j = 0; j < i; j++) {
        const double dx1 = p[j].x - p[i].x, dy1 = p[j].y - p[i].y;
        for (int k = 0; k < j; k++) {
            const double dx2 = p[k].x - p[i].x, dy2 = p[k].y - p[i].y;
            if (std::fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2))) return true;
        }
    }
    return false;
}
double det3(const P2& a, const P2& b, const P2& c) {
    const double a00 = a.x, a01 = a.y, a10 = b.x, a11 = b.y, a20 = c.x, a21 = c.y;
    return a00 * (a11 * 1. - 1. * a21) - a01 * (a10 * 1. - 1. * a20) + 1. * (a10 * a21 - a11 * a20);
}
bool check_subset(const P2* s, const P2* d, int count) {
    if (collinear(s, count) || collinear(d, count)) return false;
    if (count == 4) {
        static const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
        int negative = 0;
        for (int i = 0; i < 4; i++) {
            const int* t = tt[i];
            negative += det3(s[t[0]], s[t[1]], s[t[2]]) * det3(d[t[0]], d[t[1]], d[t[2]]) < 0;
        }
        if (negative != 0 && negative != 4) return false;
    }
    return true;
}

int find_inliers(const P2* M, const P2* m, int count, const double* H, double thresh, uint8_t* mask) {
    const float Hf[8] = {(float)H[0], (float)H[1], (float)H[2], (float)H[3], (float)H[4], (float)H[5], (float)H[6], (float)H[7]};
    const float t = (float)(thresh * thresh);
    int nz = 0;
    for (int i = 0; i < count; i++) {
        const float ww = 1.f / (Hf[6] * M[i].x + Hf[7] * M[i].y + 1.f);
        const float dx = (Hf[0] * M[i].x + Hf[1] * M[i].y + Hf[2]) * ww - m[i].x;
        const float dy = (Hf[3] * M[i].x + Hf[4] * M[i].y + Hf[5]) * ww - m[i].y;
        const float e = dx * dx + dy * dy;
        const int f = e <= t;
        mask[i] = (uint8_t)f; nz += f;
    }
    return nz;
}

// HomographyEstimatorCallback::computeError: squared reprojection error in f32
void reproj_errors(const P2* M, const P2* m, int count, const double* H, float* err) {
    const float Hf[8] = {(float)H[0], (float)H[1], (float)H[2], (float)H[3], (float)H[4], (float)H[5], (float)H[6], (float)H[7]};
    for (int i = 0; i < count; i++) {
        const float ww = 1.f / (Hf[6] * M[i].x + Hf[7] * M[i].y + 1.f);
        const float dx = (Hf[0] * M[i].x + Hf[1] * M[i].y + Hf[2]) * ww - m[i].x;
        const float dy = (Hf[3] * M[i].x + Hf[4] * M[i].y + Hf[5]) * ww - m[i].y;
        err[i] = dx * dx + dy * dy;
    }
}

int ransac_update_iters(double p, double ep, int modelPoints, int maxIters) {
    p = std::max(p, 0.); p = std::min(p, 1.);
    ep = std::max(ep, 0.); ep = std::min(ep, 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num = std::log(num); denom = std::log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : cv_round(num / denom);
}

// HomographyRefineCallback::compute
void refine_compute(const P2* M, const P2* m, int count, const double* h, double* err, double* J) {
    for (int i = 0; i < count; i++) {
        const double Mx = M[i].x, My = M[i].y;
        double ww = h[6] * Mx + h[7] * My + 1.;
        ww = std::fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
        const double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
        err[i * 2] = xi - m[i].x; err[i * 2 + 1] = yi - m[i].y;
        if (J) {
            double* j = J + (size_t)i * 16;
            j[0] = Mx * ww; j[1] = My * ww; j[2] = ww; j[3] = j[4] = j[5] = 0.; j[6] = -Mx * ww * xi; j[7] = -My * ww * xi;
            j[8] = j[9] = j[10] = 0.; j[11] = Mx * ww; j[12] = My * ww; j[13] = ww; j[14] = -Mx * ww * yi; j[15] = -My * ww * yi;
        }
    }
}

// LMSolverImpl::run (calib3d/src/levmarq.cpp), 8 parameters, maxIters 10, eps FLT_EPSILON
void lm_refine(const P2* M, const P2* m, int count, double* H) {
    const int lx = 8, maxIters = 10;
    const double epsx = FLT_EPSILON, epsf = FLT_EPSILON;
    std::vector<double> x(H, H + 8), xd(8), r(count * 2), rd(count * 2), J((size_t)count * 16), A(64), Ap(64), v(8), d(8), D(8), tmp(8);
    auto normal = [&]() {
        for (int a = 0; a < 8; a++) {
            for (int b = 0; b < 8; b++) { double s = 0; for (int i = 0; i < count * 2; i++) s += J[(size_t)i * 8 + a] * J[(size_t)i * 8 + b]; A[a * 8 + b] = s; }
            double s = 0; for (int i = 0; i < count * 2; i++) s += J[(size_t)i * 8 + a] * r[i]; v[a] = s;
        }
    };
    auto sq = [](const std::vector<double>& q) { double s = 0; for (double e : q) s += e * e; return s; };
    refine_compute(M, m, count, x.data(), r.data(), J.data());
    double S = sq(r);
    normal();
    for (int i = 0; i < 8; i++) D[i] = A[i * 8 + i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    int iter = 0;
    for (;;) {
        Ap = A;
        for (int i = 0; i < lx; i++) Ap[i * 8 + i] += lambda * D[i];
        eig_solve(Ap.data(), 8, v.data(), 1, d.data());
        for (int i = 0; i < 8; i++) xd[i] = x[i] - d[i];
        refine_compute(M, m, count, xd.data(), rd.data(), nullptr);
        const double Sd = sq(rd);
        for (int i = 0; i < 8; i++) { double s = 0; for (int k = 0; k < 8; k++) s += A[i * 8 + k] * d[k]; tmp[i] = -s + 2 * v[i]; }
        double dS = 0; for (int i = 0; i < 8; i++) dS += d[i] * tmp[i];
        const double R = (S - Sd) / (std::fabs(dS) > DBL_EPSILON ? dS : 1);
        if (R > Rhi) { lambda *= 0.5; if (lambda < lc) lambda = 0; }
        else if (R < Rlo) {
            double t = 0; for (int i = 0; i < 8; i++) t += d[i] * v[i];
            double nu = (Sd - S) / (std::fabs(t) > DBL_EPSILON ? t : 1) + 2;
            nu = std::min(std::max(nu, 2.), 10.);
            if (lambda == 0) {
                double I8[64]; for (int i = 0; i < 64; i++) I8[i] = (i % 9 == 0);
                std::vector<double> inv(64);
                eig_solve(A.data(), 8, I8, 8, inv.data());
                double maxval = DBL_EPSILON;
                for (int i = 0; i < lx; i++) maxval = std::max(maxval, std::abs(inv[i * 8 + i]));
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            std::swap(x, xd);
            refine_compute(M, m, count, x.data(), r.data(), J.data());
            normal();
        }
        iter++;
        double nd = 0, nr = 0;
        for (int i = 0; i < 8; i++) nd = std::max(nd, std::fabs(d[i]));
        for (double e : r) nr = std::max(nr, std::fabs(e));
        const bool proceed = iter < maxIters && nd >= epsx && nr >= epsf;
        if (!proceed) break;
    }
    for (int i = 0; i < 8; i++) H[i] = x[i];
}

}  // namespace

extern "C" {

// BFMatcher(NORM_HAMMING, crossCheck=false).knnMatch(query, k=2): rows {train0, dist0, train1, dist1}
int orc_bf_knn2_hamming(const uint8_t* query, int nq, const uint8_t* train, int nt, int* out) {
    for (int q = 0; q < nq; q++) {
        int idx[2] = {-1, -1};
        int dist[2] = {std::numeric_limits<int>::max(), std::numeric_limits<int>::max()};
        const uint64_t* a = (const uint64_t*)(query + (size_t)q * 32);
        for (int t = 0; t < nt; t++) {
            const uint64_t* b = (const uint64_t*)(train + (size_t)t * 32);
            int d = 0;
            for (int k = 0; k < 4; k++) { uint64_t av, bv; std::memcpy(&av, a + k, 8); std::memcpy(&bv, b + k, 8); d += __builtin_popcountll(av ^ bv); }
            if (d < dist[1]) {
                int k;
                for (k = 0; k >= 0 && dist[k] > d; k--) { idx[k + 1] = idx[k]; dist[k + 1] = dist[k]; }
                idx[k + 1] = t; dist[k + 1] = d;
            }
        }
        out[q * 4 + 0] = idx[0]; out[q * 4 + 1] = idx[0] >= 0 ? dist[0] : -1;
        out[q * 4 + 2] = idx[1]; out[q * 4 + 3] = idx[1] >= 0 ? dist[1] : -1;
    }
    return 0;
}

unsigned orc_rng_next(uint64_t* state) { Rng r(*state); unsigned v = r.next(); *state = r.state; return v; }

// calib3d::find_homography(src, dst, method, thr, mask, maxIters 2000, confidence 0.995).
// Returns 0 and *found = 1 with H (row-major 3x3 double), or *found = 0 (empty Mat); 3 = bad arguments
// (OpenCV would throw: fewer than 4 points / unknown method), 7 = method not restated (LMEDS, RHO).
int orc_find_homography(const float* src_pts, const float* dst_pts, int n, int method, double thr, double* H,
                        uint8_t* mask_out, int* found) {
    *found = 0;
    if (n < 4) return 3;
    if (method != 0 && method != 8 && method != 4) return method == 16 ? 7 : 3;
    if (thr <= 0) thr = 3;
    std::vector<P2> src(n), dst(n);
    for (int i = 0; i < n; i++) { src[i] = {src_pts[2 * i], src_pts[2 * i + 1]}; dst[i] = {dst_pts[2 * i], dst_pts[2 * i + 1]}; }
    std::vector<uint8_t> mask(n, 1);
    bool result = false;
    if (method == 0 || n == 4) {
        result = dlt(src.data(), dst.data(), n, H);
    } else if (method == 4) {
        // LMeDSPointSetRegistrator::run (calib3d/src/ptsetreg.cpp): fixed iteration count from an assumed
        // 45% outlier ratio, model with the least median squared reprojection error, inliers from the robust sigma
        const int modelPoints = 4;
        const int niters = ransac_update_iters(0.995, 0.45, modelPoints, 2000);
        Rng rng((uint64_t)-1);
        std::vector<float> err(n);
        double model[9], bestModel[9], minMedian = DBL_MAX;
        P2 ms1[4], ms2[4];
        for (int iter = 0; iter < niters; iter++) {
            bool got = false;
            for (int attempt = 0; attempt < 1000 && !got; attempt++) {
                int idx[4];
                for (int i = 0; i < modelPoints; i++) {
                    int idx_i;
                    for (idx_i = rng.uniform(0, n); std::find(idx, idx + i, idx_i) != idx + i; idx_i = rng.uniform(0, n)) {}
                    idx[i] = idx_i;
                    ms1[i] = src[idx_i]; ms2[i] = dst[idx_i];
                }
                got = check_subset(ms1, ms2, modelPoints);
            }
            if (!got) { if (iter == 0) { return 0; } break; }
            if (!dlt(ms1, ms2, modelPoints, model)) continue;
            reproj_errors(src.data(), dst.data(), n, model, err.data());
            std::nth_element(err.begin(), err.begin() + n / 2, err.end());
            const double median = err[n / 2];
            if (median < minMedian) { minMedian = median; std::memcpy(bestModel, model, sizeof(model)); }
        }
        if (minMedian < DBL_MAX) {
            double sigma = 2.5 * 1.4826 * (1 + 5. / (n - modelPoints)) * std::sqrt(minMedian);
            sigma = std::max(sigma, 0.001);
            const int good = find_inliers(src.data(), dst.data(), n, bestModel, sigma, mask.data());
            std::memcpy(H, bestModel, sizeof(bestModel));
            result = good >= modelPoints;
        }
    } else {
        const int modelPoints = 4;
        const double confidence = 0.995;
        int niters = 2000, maxGood = 0;
        Rng rng((uint64_t)-1);
        std::vector<uint8_t> cur(n), best(n, 0);
        double model[9], bestModel[9];
        P2 ms1[4], ms2[4];
        for (int iter = 0; iter < niters; iter++) {
            bool got = false;
            for (int attempt = 0; attempt < 1000 && !got; attempt++) {
                int idx[4];
                for (int i = 0; i < modelPoints; i++) {
                    int idx_i;
                    for (idx_i = rng.uniform(0, n); std::find(idx, idx + i, idx_i) != idx + i; idx_i = rng.uniform(0, n)) {}
                    idx[i] = idx_i;
                    ms1[i] = src[idx_i]; ms2[i] = dst[idx_i];
                }
                got = check_subset(ms1, ms2, modelPoints);
            }
            if (!got) { if (iter == 0) { return 0; } break; }
            if (!dlt(ms1, ms2, modelPoints, model)) continue;
            const int good = find_inliers(src.data(), dst.data(), n, model, thr, cur.data());
            if (good > std::max(maxGood, modelPoints - 1)) {
                std::swap(cur, best);
                std::memcpy(bestModel, model, sizeof(model));
                maxGood = good;
                niters = ransac_update_iters(confidence, (double)(n - good) / n, modelPoints, niters);
            }
        }
        if (maxGood > 0) { std::memcpy(H, bestModel, sizeof(bestModel)); mask = best; result = true; }
    }
    if (result && n > 4) {
        std::vector<P2> s2, d2;
        for (int i = 0; i < n; i++) if (mask[i]) { s2.push_back(src[i]); d2.push_back(dst[i]); }
        const int np = (int)s2.size();
        if (np > 0) {
            if (method == 8 || method == 4) dlt(s2.data(), d2.data(), np, H);
            lm_refine(s2.data(), d2.data(), np, H);
        }
    }
    if (result) { *found = 1; if (mask_out) std::memcpy(mask_out, mask.data(), n); }
    else if (mask_out) std::memset(mask_out, 0, n);
    return 0;
}

// keypoint_match_no_scale, lib.rs:146-353, with the DOCUMENTED drop semantics (lib.rs:98): a frame
// whose homography cannot be estimated is skipped and counted; divisor n - dropped.
// Returns 0 ok, 1 NotEnoughFiles, 2 all frames dropped (InvalidParams, lib.rs:324), 4 backend error.
// scale_down > 0 selects keypoint_match_scale_down (lib.rs:355-601): ORB on INTER_AREA-shrunk greys, homography
// estimated there, then adjust_homography_for_scale_f64 (utils.rs:218-248) before the full-size warp.
// ws / hs (n entries each, or null): per-frame sizes of a stack whose frames differ in size — every frame is read and
// described at its own size (lib.rs:200-204) and warped into the first frame's (lib.rs:290-299); (w, h) is then frame 0's.
static int keypoint_match_impl(const void* const* frames, int n, int w, int h, const int* ws, const int* hs, int method, double thr,
                               float keep_ratio, float match_ratio, int border_mode, const double* border_value, float scale_down,
                               float* out, int* dropped_out, double* H_out, int* status_out, int n_threads);

int orc_keypoint_match(const void* const* frames, int n, int w, int h, int method, double thr, float keep_ratio,
                       float match_ratio, int border_mode, const double* border_value, float scale_down, float* out,
                       int* dropped_out, double* H_out, int* status_out, int n_threads) {
    return keypoint_match_impl(frames, n, w, h, nullptr, nullptr, method, thr, keep_ratio, match_ratio, border_mode, border_value,
                               scale_down, out, dropped_out, H_out, status_out, n_threads);
}

int orc_keypoint_match_sized(const void* const* frames, int n, const int* ws, const int* hs, int method, double thr, float keep_ratio,
                             float match_ratio, int border_mode, const double* border_value, float scale_down, float* out,
                             int* dropped_out, double* H_out, int* status_out, int n_threads) {
    if (n <= 0) return 1;
    return keypoint_match_impl(frames, n, ws[0], hs[0], ws, hs, method, thr, keep_ratio, match_ratio, border_mode, border_value,
                               scale_down, out, dropped_out, H_out, status_out, n_threads);
}

static int keypoint_match_impl(const void* const* frames, int n, int w, int h, const int* ws, const int* hs, int method, double thr,
                               float keep_ratio, float match_ratio, int border_mode, const double* border_value, float scale_down,
                               float* out, int* dropped_out, double* H_out, int* status_out, int n_threads) {
    if (n <= 0) return 1;
    const size_t npx = (size_t)w * h, nel = npx * 3;
    int ew = w, eh = h;
    if (scale_down > 0) {
        if (scale_down >= (float)w) return 3;               // InvalidParams lib.rs:377-382
        if (orc_scaled_size(w, h, scale_down, &ew, &eh)) return 3;
    }
    std::vector<uint8_t> g0(npx);
    orc_grey(frames[0], 8, w, h, 0, g0.data());
    if (scale_down > 0) { std::vector<uint8_t> sm((size_t)ew * eh); orc_resize_area_u8(g0.data(), w, h, sm.data(), ew, eh); g0.swap(sm); }
    const int MAXKP = 4096;
    std::vector<float> kp0((size_t)MAXKP * 7);
    std::vector<uint8_t> de0((size_t)MAXKP * 32);
    int n0 = 0;
    orc_orb_detect_and_compute(g0.data(), ew, eh, MAXKP, kp0.data(), de0.data(), &n0);
#ifdef _OPENMP
    const int T = n_threads > 0 ? n_threads : omp_get_max_threads();
#else
    const int T = 1;
#endif
    std::vector<std::vector<float>> accs(T);
    int dropped = 0, err = 0;
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    #pragma omp parallel for schedule(dynamic, 1) num_threads(T)
    for (int i = 0; i < n; i++) {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        double Hm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        int status = 0;
        const int wi = ws ? ws[i] : w, hi = hs ? hs[i] : h;       // this frame's own size
        if (i > 0) {
            std::vector<uint8_t> g((size_t)wi * hi);
            orc_grey(frames[i], 8, wi, hi, 0, g.data());
            // scale_image on THIS frame's grey: its own smaller dimension becomes scale_down (lib.rs:429, utils.rs:186-214)
            int ewi = wi, ehi = hi;
            if (scale_down > 0) {
                if (orc_scaled_size(wi, hi, scale_down, &ewi, &ehi)) { err = 3; continue; }
                std::vector<uint8_t> sm((size_t)ewi * ehi); orc_resize_area_u8(g.data(), wi, hi, sm.data(), ewi, ehi); g.swap(sm);
            }
            std::vector<float> kp((size_t)MAXKP * 7);
            std::vector<uint8_t> de((size_t)MAXKP * 32);
            int nk = 0;
            orc_orb_detect_and_compute(g.data(), ewi, ehi, MAXKP, kp.data(), de.data(), &nk);
            // query = frame-0 descriptors, train = frame-i descriptors (lib.rs:208-219)
            std::vector<int> knn((size_t)std::max(n0, 1) * 4);
            orc_bf_knn2_hamming(de0.data(), n0, de.data(), nk, knn.data());
            struct Mt { int q, t; float d; };
            std::vector<Mt> ms;
            for (int q = 0; q < n0; q++) {
                if (knn[q * 4] < 0 || knn[q * 4 + 2] < 0) continue;                // needs len == 2
                const float d0 = (float)knn[q * 4 + 1], d1 = (float)knn[q * 4 + 3];
                if (d0 < match_ratio * d1) ms.push_back({q, knn[q * 4], d0});
            }
            std::stable_sort(ms.begin(), ms.end(), [](const Mt& a, const Mt& b) { return a.d < b.d; });
            const size_t keep = (size_t)std::round((float)ms.size() * keep_ratio);   // f32 round(): half away from zero
            if (keep < ms.size()) ms.resize(keep);
            if (ms.size() < 5) status = 1;
            else {
                std::vector<float> sp(ms.size() * 2), dp(ms.size() * 2);
                for (size_t k = 0; k < ms.size(); k++) {
                    sp[2 * k] = kp0[(size_t)ms[k].q * 7]; sp[2 * k + 1] = kp0[(size_t)ms[k].q * 7 + 1];     // src_pts: frame 0
                    dp[2 * k] = kp[(size_t)ms[k].t * 7]; dp[2 * k + 1] = kp[(size_t)ms[k].t * 7 + 1];       // dst_pts: frame i
                }
                int found = 0;
                // find_homography(dst_pts, src_pts): maps frame i -> frame 0 (lib.rs:267-269)
                const int rc = orc_find_homography(dp.data(), sp.data(), (int)ms.size(), method, thr, Hm, nullptr, &found);
                if (rc == 7) { err = 4; }
                if (rc != 0 || !found) status = 1;
                else {
                    const double det = Hm[0] * (Hm[4] * Hm[8] - Hm[5] * Hm[7]) - Hm[1] * (Hm[3] * Hm[8] - Hm[5] * Hm[6]) + Hm[2] * (Hm[3] * Hm[7] - Hm[4] * Hm[6]);
                    if (std::fabs(det) < 1e-6) status = 1;
                    else if (scale_down > 0) {              // adjust_homography_for_scale_f64(h_small, this frame's small grey, this frame), utils.rs:229-239
                        const double sx = (double)wi / (double)ewi, sy = (double)hi / (double)ehi;
                        Hm[2] *= sx; Hm[5] *= sy; Hm[6] /= sx; Hm[7] /= sy;
                    }
                }
            }
        }
        if (status_out) status_out[i] = status;
        if (H_out) for (int k = 0; k < 9; k++) H_out[(size_t)i * 9 + k] = Hm[k];
        if (status) {
            #pragma omp atomic
            dropped++;
            continue;
        }
        std::vector<float>& acc = accs[tid];
        const bool fresh = acc.empty();
        if (fresh) acc.assign(nel, 0.f);
        if (i == 0) orc_warp_frame(frames[0], 8, w, h, 3, 0, I3, 1, BORDER_CONSTANT, nullptr, 1.0 / 255.0, 0, acc.data(), fresh ? 0 : 1);
        else orc_warp_frame_sized(frames[i], 8, wi, hi, 3, 0, Hm, 0, border_mode, border_value, 1.0 / 255.0, 0, acc.data(), w, h, fresh ? 0 : 1);
    }
    if (err) return err;
    if (dropped_out) *dropped_out = dropped;
    if (dropped >= n) return 2;
    std::vector<float> total;
    for (int t = 0; t < T; t++) {
        if (accs[t].empty()) continue;
        if (total.empty()) total = accs[t];
        else for (size_t k = 0; k < nel; k++) total[k] = total[k] + accs[t][k];
    }
    orc_scale(total.data(), nel, (double)(n - dropped), out);
    return 0;
}

}  // extern "C"
