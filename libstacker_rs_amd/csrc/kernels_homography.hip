// kernels_homography.hip — the arithmetic of calib3d::findHomography (lib.rs:267-276) on gfx950.
//
// Two kernels, both wave64-shaped (a problem has <= a few hundred correspondences: one wavefront sweeps them in a few
// steps, __ballot turns the inlier predicate of 64 points into the count directly, and the small dense algebra is done
// redundantly by all 64 lanes on wave-uniform values so there is no divergence and no LDS traffic):
//
//   hg_models_kernel   one wavefront per 4-point sample (RANSAC / LMEDS hypothesis), blockIdx.y = problem (frame).
//                      The model through 4 correspondences is computed in closed form (projective-basis construction in
//                      Hartley-normalised coordinates, f64) instead of OpenCV's 9x9 eigen-decomposition: for exactly four
//                      points both give the unique interpolating homography, so the f32 reprojection test that follows
//                      (HomographyEstimatorCallback::computeError, written out operation by operation) sees the same
//                      model to ~1e-13. Output: inlier count (RANSAC) or the median squared error (LMEDS) per sample.
//   hg_refine_kernel   one wavefront per problem: inlier mask of the winning sample's model, normalised DLT over the
//                      inliers (24 moment sums by wave reduction; smallest eigenvector of the 9x9 normal matrix by shifted
//                      inverse iteration on a Cholesky factor), then LMSolver's control flow (<= 10 iterations, 8
//                      parameters, Marquardt damping with the initial diagonal) with 28 moment sums per evaluation and an
//                      8x8 Cholesky solve per step.
//
// Built with -ffp-contract=off: every f32 operation of the reprojection test is individually rounded, like OpenCV's.
#include <cfloat>

#include "homography.h"

namespace stk {
namespace geom {

namespace {

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ inline double wave_max(double v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ inline int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct V3 { double x, y, z; };
__device__ inline V3 cross(const V3& a, const V3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ inline double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Columns of the matrix that sends the canonical projective frame (e1, e2, e3, e1+e2+e3) to (p0, p1, p2, p3), up to scale.
__device__ inline void projective_frame(const V3 p[4], V3 col[3]) {
    const double l0 = dot(p[3], cross(p[1], p[2]));
    const double l1 = dot(p[0], cross(p[3], p[2]));
    const double l2 = dot(p[0], cross(p[1], p[3]));
    col[0] = {p[0].x * l0, p[0].y * l0, p[0].z * l0};
    col[1] = {p[1].x * l1, p[1].y * l1, p[1].z * l1};
    col[2] = {p[2].x * l2, p[2].y * l2, p[2].z * l2};
}

// Homography through four correspondences M_k -> m_k, scaled so that H[8] = 1. False when a coordinate has no spread
// (the case runKernel rejects); a vanishing H[8] yields non-finite entries and therefore no inliers, as in OpenCV.
__device__ inline bool four_point_model(const HgPoint s[4], double H[9]) {
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { cMx += s[k].Mx; cMy += s[k].My; cmx += s[k].mx; cmy += s[k].my; }
    cMx *= 0.25; cMy *= 0.25; cmx *= 0.25; cmy *= 0.25;
    double dMx = 0, dMy = 0, dmx = 0, dmy = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        dMx += fabs(s[k].Mx - cMx); dMy += fabs(s[k].My - cMy);
        dmx += fabs(s[k].mx - cmx); dmy += fabs(s[k].my - cmy);
    }
    if (dMx < DBL_EPSILON || dMy < DBL_EPSILON || dmx < DBL_EPSILON || dmy < DBL_EPSILON) return false;
    const double sMx = 4.0 / dMx, sMy = 4.0 / dMy, smx = 4.0 / dmx, smy = 4.0 / dmy;
    V3 P[4], Q[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        P[k] = {(s[k].Mx - cMx) * sMx, (s[k].My - cMy) * sMy, 1.0};
        Q[k] = {(s[k].mx - cmx) * smx, (s[k].my - cmy) * smy, 1.0};
    }
    V3 a[3], b[3];
    projective_frame(P, a);
    projective_frame(Q, b);
    // Hn = B * adj(A); the rows of adj(A) are the cross products of A's columns
    const V3 r0 = cross(a[1], a[2]), r1 = cross(a[2], a[0]), r2 = cross(a[0], a[1]);
    double Hn[9];
    Hn[0] = b[0].x * r0.x + b[1].x * r1.x + b[2].x * r2.x; Hn[1] = b[0].x * r0.y + b[1].x * r1.y + b[2].x * r2.y; Hn[2] = b[0].x * r0.z + b[1].x * r1.z + b[2].x * r2.z;
    Hn[3] = b[0].y * r0.x + b[1].y * r1.x + b[2].y * r2.x; Hn[4] = b[0].y * r0.y + b[1].y * r1.y + b[2].y * r2.y; Hn[5] = b[0].y * r0.z + b[1].y * r1.z + b[2].y * r2.z;
    Hn[6] = b[0].z * r0.x + b[1].z * r1.x + b[2].z * r2.x; Hn[7] = b[0].z * r0.y + b[1].z * r1.y + b[2].z * r2.y; Hn[8] = b[0].z * r0.z + b[1].z * r1.z + b[2].z * r2.z;
    // undo the normalisation: H = Tm^-1 * Hn * TM with TM = diag(sMx, sMy, 1) * translate(-cM), Tm likewise
    double T[9];
    T[0] = Hn[0] / smx + cmx * Hn[6]; T[1] = Hn[1] / smx + cmx * Hn[7]; T[2] = Hn[2] / smx + cmx * Hn[8];
    T[3] = Hn[3] / smy + cmy * Hn[6]; T[4] = Hn[4] / smy + cmy * Hn[7]; T[5] = Hn[5] / smy + cmy * Hn[8];
    T[6] = Hn[6]; T[7] = Hn[7]; T[8] = Hn[8];
    double R[9];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        R[3 * r + 0] = T[3 * r + 0] * sMx;
        R[3 * r + 1] = T[3 * r + 1] * sMy;
        R[3 * r + 2] = T[3 * r + 2] - T[3 * r + 0] * sMx * cMx - T[3 * r + 1] * sMy * cMy;
    }
    const double sc = 1.0 / R[8];
#pragma unroll
    for (int k = 0; k < 9; k++) H[k] = R[k] * sc;
    return true;
}

// squared reprojection error in f32, the operation sequence of HomographyEstimatorCallback::computeError
__device__ inline float reproj_error(const float Hf[8], const HgPoint& p) {
    const float ww = 1.f / (Hf[6] * p.Mx + Hf[7] * p.My + 1.f);
    const float dx = (Hf[0] * p.Mx + Hf[1] * p.My + Hf[2]) * ww - p.mx;
    const float dy = (Hf[3] * p.Mx + Hf[4] * p.My + Hf[5]) * ww - p.my;
    return dx * dx + dy * dy;
}

__global__ __launch_bounds__(256) void hg_models_kernel(const HgPoint* __restrict__ pts, const HgFrame* __restrict__ frames,
                                                        const HgSample* __restrict__ samples, int lmeds,
                                                        float* __restrict__ err_scratch, int* __restrict__ scores) {
    const HgFrame fr = frames[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int h = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (h >= fr.n_hyp) return;                       // whole wavefronts leave together
    const HgSample q = samples[fr.hyp_ofs + h];
    const HgPoint* P = pts + fr.pt_ofs;
    HgPoint s[4];
#pragma unroll
    for (int k = 0; k < 4; k++) s[k] = P[q.idx[k]];
    double H[9];
    const bool ok = four_point_model(s, H);
    float Hf[8];
#pragma unroll
    for (int k = 0; k < 8; k++) Hf[k] = (float)H[k];
    if (!ok) { if (lane == 0) scores[fr.hyp_ofs + h] = -1; return; }
    if (!lmeds) {
        int count = 0;
        for (int base = 0; base < fr.n; base += 64) {
            const int i = base + lane;
            bool in = false;
            if (i < fr.n) in = reproj_error(Hf, P[i]) <= fr.thr2;
            count += __popcll(__ballot(in));
        }
        if (lane == 0) scores[fr.hyp_ofs + h] = count;
        return;
    }
    // LMEDS: the (n / 2)-th smallest squared error (std::nth_element's element), by bisection on the f32 bit pattern
    float* E = err_scratch + fr.err_ofs + (size_t)h * fr.n;
    for (int i = lane; i < fr.n; i += 64) E[i] = reproj_error(Hf, P[i]);
    __threadfence();
    const int k = fr.n / 2;
    unsigned prefix = 0;
    for (int bit = 31; bit >= 0; bit--) {
        const unsigned cand = prefix | (1u << bit);
        int c = 0;
        for (int i = lane; i < fr.n; i += 64) c += __float_as_uint(E[i]) < cand;
        if (wave_sum_i(c) <= k) prefix = cand;       // the k-th smallest is >= cand
    }
    if (lane == 0) scores[fr.hyp_ofs + h] = (int)prefix;
}

// ---- small dense symmetric solves, fully unrolled so that everything stays in registers -------------------------------
template <int N>
struct Chol {
    double L[N * (N + 1) / 2];                       // lower triangle, row-major packed
    __device__ static constexpr int at(int i, int j) { return i * (i + 1) / 2 + j; }
    // A: full N x N row-major (lower triangle read). False if a pivot is not positive.
    __device__ bool factor(const double* A) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < N; i++) {
#pragma unroll
            for (int j = 0; j <= i; j++) {
                double s = A[i * N + j];
#pragma unroll
                for (int k = 0; k < j; k++) s -= L[at(i, k)] * L[at(j, k)];
                if (j == i) { if (!(s > 0)) { ok = false; s = 1; } L[at(i, i)] = sqrt(s); }
                else L[at(i, j)] = s / L[at(j, j)];
            }
        }
        return ok;
    }
    __device__ void forward(double* b) const {       // L y = b
#pragma unroll
        for (int i = 0; i < N; i++) {
            double s = b[i];
#pragma unroll
            for (int k = 0; k < i; k++) s -= L[at(i, k)] * b[k];
            b[i] = s / L[at(i, i)];
        }
    }
    __device__ void backward(double* b) const {      // L^T x = y
#pragma unroll
        for (int i = N - 1; i >= 0; i--) {
            double s = b[i];
#pragma unroll
            for (int k = i + 1; k < N; k++) s -= L[at(k, i)] * b[k];
            b[i] = s / L[at(i, i)];
        }
    }
    __device__ void solve(double* b) const { forward(b); backward(b); }
};

struct LmSums { double A[64]; double v[8]; double S; double rmax; };

// residuals (and, if jac, J^T J and J^T r) of the reprojection x -> (h M) / w - m over the inliers this lane owns
__device__ inline void lm_evaluate(const HgPoint* P, int n, int lane, unsigned long long mine, const double* h, bool jac, LmSums& o) {
    double aa = 0, ab = 0, ac = 0, bb = 0, bc = 0, cc = 0;                      // (a b c)^T (a b c)
    double xaa = 0, xab = 0, xbb = 0, xac = 0, xbc = 0;                         // xi * ...
    double yaa = 0, yab = 0, ybb = 0, yac = 0, ybc = 0;                         // yi * ...
    double qaa = 0, qab = 0, qbb = 0;                                           // (xi^2 + yi^2) * ...
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0, v6 = 0, v7 = 0, S = 0, rmax = 0;
    for (int j = 0, i = lane; i < n; j++, i += 64) {
        if (!((mine >> j) & 1)) continue;
        const double Mx = P[i].Mx, My = P[i].My;
        double ww = h[6] * Mx + h[7] * My + 1.;
        ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
        const double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
        const double rx = xi - P[i].mx, ry = yi - P[i].my;
        S += rx * rx + ry * ry;
        rmax = fmax(rmax, fmax(fabs(rx), fabs(ry)));
        if (!jac) continue;
        const double a = Mx * ww, b = My * ww, c = ww;
        aa += a * a; ab += a * b; ac += a * c; bb += b * b; bc += b * c; cc += c * c;
        xaa += xi * a * a; xab += xi * a * b; xbb += xi * b * b; xac += xi * a * c; xbc += xi * b * c;
        yaa += yi * a * a; yab += yi * a * b; ybb += yi * b * b; yac += yi * a * c; ybc += yi * b * c;
        const double q = xi * xi + yi * yi;
        qaa += q * a * a; qab += q * a * b; qbb += q * b * b;
        const double t = xi * rx + yi * ry;
        v0 += a * rx; v1 += b * rx; v2 += c * rx; v3 += a * ry; v4 += b * ry; v5 += c * ry; v6 -= a * t; v7 -= b * t;
    }
    o.S = wave_sum(S);
    o.rmax = wave_max(rmax);
    if (!jac) return;
    aa = wave_sum(aa); ab = wave_sum(ab); ac = wave_sum(ac); bb = wave_sum(bb); bc = wave_sum(bc); cc = wave_sum(cc);
    xaa = wave_sum(xaa); xab = wave_sum(xab); xbb = wave_sum(xbb); xac = wave_sum(xac); xbc = wave_sum(xbc);
    yaa = wave_sum(yaa); yab = wave_sum(yab); ybb = wave_sum(ybb); yac = wave_sum(yac); ybc = wave_sum(ybc);
    qaa = wave_sum(qaa); qab = wave_sum(qab); qbb = wave_sum(qbb);
    o.v[0] = wave_sum(v0); o.v[1] = wave_sum(v1); o.v[2] = wave_sum(v2); o.v[3] = wave_sum(v3);
    o.v[4] = wave_sum(v4); o.v[5] = wave_sum(v5); o.v[6] = wave_sum(v6); o.v[7] = wave_sum(v7);
    double* A = o.A;
#pragma unroll
    for (int k = 0; k < 64; k++) A[k] = 0;
    A[0] = aa; A[1] = ab; A[2] = ac; A[9] = bb; A[10] = bc; A[18] = cc;                     // rows 0-2, x residual
    A[27] = aa; A[28] = ab; A[29] = ac; A[36] = bb; A[37] = bc; A[45] = cc;                 // rows 3-5, y residual
    A[6] = -xaa; A[7] = -xab; A[14] = -xab; A[15] = -xbb; A[22] = -xac; A[23] = -xbc;       // rows 0-2 x cols 6-7
    A[30] = -yaa; A[31] = -yab; A[38] = -yab; A[39] = -ybb; A[46] = -yac; A[47] = -ybc;     // rows 3-5 x cols 6-7
    A[54] = qaa; A[55] = qab; A[63] = qbb;
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
        for (int c = r + 1; c < 8; c++) A[c * 8 + r] = A[r * 8 + c];
}

__global__ __launch_bounds__(64) void hg_refine_kernel(const HgPoint* __restrict__ pts, const HgJob* __restrict__ jobs,
                                                       HgResult* __restrict__ results, uint8_t* __restrict__ masks) {
    const HgJob job = jobs[blockIdx.x];
    const int lane = threadIdx.x;
    const HgPoint* P = pts + job.pt_ofs;
    uint8_t* mask = masks + job.pt_ofs;
    HgResult* out = results + blockIdx.x;
    const int n = job.n;
    if (job.mode < 0) {
        for (int i = lane; i < n; i += 64) mask[i] = 0;
        if (lane == 0) { for (int k = 0; k < 9; k++) out->H[k] = 0; out->found = 0; out->n_inliers = 0; out->lm_iterations = 0; out->dlt_degenerate = 0; }
        return;
    }
    double H[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    unsigned long long mine = 0;          // bit j: point lane + 64 j is an inlier
    int np = 0;
    if (job.mode == 1) {
        HgPoint s[4];
#pragma unroll
        for (int k = 0; k < 4; k++) s[k] = P[job.idx[k]];
        four_point_model(s, H);
        float Hf[8];
#pragma unroll
        for (int k = 0; k < 8; k++) Hf[k] = (float)H[k];
        for (int j = 0, i = lane; i < n; j++, i += 64) {
            const bool in = reproj_error(Hf, P[i]) <= job.thr2;
            if (in) mine |= 1ull << j;
            mask[i] = in;
        }
    } else {
        for (int j = 0, i = lane; i < n; j++, i += 64) { mine |= 1ull << j; mask[i] = 1; }
    }
    np = wave_sum_i(__popcll(mine));
    if (job.mode == 1 && np < 4) {        // LMEDS only: fewer inliers than model points -> no result, empty mask
        for (int i = lane; i < n; i += 64) mask[i] = 0;
        if (lane == 0) { for (int k = 0; k < 9; k++) out->H[k] = 0; out->found = 0; out->n_inliers = np; out->lm_iterations = 0; out->dlt_degenerate = 0; }
        return;
    }

    // ---- normalised DLT over the inliers (HomographyEstimatorCallback::runKernel) ----
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0;
    for (int j = 0, i = lane; i < n; j++, i += 64)
        if ((mine >> j) & 1) { cMx += P[i].Mx; cMy += P[i].My; cmx += P[i].mx; cmy += P[i].my; }
    const double inv_np = 1.0 / np;
    cMx = wave_sum(cMx) * inv_np; cMy = wave_sum(cMy) * inv_np; cmx = wave_sum(cmx) * inv_np; cmy = wave_sum(cmy) * inv_np;
    double dMx = 0, dMy = 0, dmx = 0, dmy = 0;
    for (int j = 0, i = lane; i < n; j++, i += 64)
        if ((mine >> j) & 1) { dMx += fabs(P[i].Mx - cMx); dMy += fabs(P[i].My - cMy); dmx += fabs(P[i].mx - cmx); dmy += fabs(P[i].my - cmy); }
    dMx = wave_sum(dMx); dMy = wave_sum(dMy); dmx = wave_sum(dmx); dmy = wave_sum(dmy);
    const bool degenerate = dMx < DBL_EPSILON || dMy < DBL_EPSILON || dmx < DBL_EPSILON || dmy < DBL_EPSILON;
    if (!degenerate) {
        const double sMx = np / dMx, sMy = np / dMy, smx = np / dmx, smy = np / dmy;
        double u[6] = {0, 0, 0, 0, 0, 0}, ux[6] = {0, 0, 0, 0, 0, 0}, uy[6] = {0, 0, 0, 0, 0, 0}, ur[6] = {0, 0, 0, 0, 0, 0};
        for (int j = 0, i = lane; i < n; j++, i += 64) {
            if (!((mine >> j) & 1)) continue;
            const double X = (P[i].Mx - cMx) * sMx, Y = (P[i].My - cMy) * sMy;
            const double x = (P[i].mx - cmx) * smx, y = (P[i].my - cmy) * smy;
            const double m[6] = {X * X, X * Y, X, Y * Y, Y, 1.0};
            const double r = x * x + y * y;
#pragma unroll
            for (int k = 0; k < 6; k++) { u[k] += m[k]; ux[k] += x * m[k]; uy[k] += y * m[k]; ur[k] += r * m[k]; }
        }
#pragma unroll
        for (int k = 0; k < 6; k++) { u[k] = wave_sum(u[k]); ux[k] = wave_sum(ux[k]); uy[k] = wave_sum(uy[k]); ur[k] = wave_sum(ur[k]); }
        // 9x9 normal matrix  [ U 0 -Ux ; 0 U -Uy ; -Ux -Uy Ur ],  each block the symmetric 3x3 {0 1 2; 1 3 4; 2 4 5}
        double N9[81];
#pragma unroll
        for (int k = 0; k < 81; k++) N9[k] = 0;
        const int sym[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
        double trace = 0;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int k = sym[3 * r + c];
                N9[r * 9 + c] = u[k]; N9[(r + 3) * 9 + (c + 3)] = u[k];
                N9[r * 9 + (c + 6)] = -ux[k]; N9[(c + 6) * 9 + r] = -ux[k];
                N9[(r + 3) * 9 + (c + 6)] = -uy[k]; N9[(c + 6) * 9 + (r + 3)] = -uy[k];
                N9[(r + 6) * 9 + (c + 6)] = ur[k];
            }
#pragma unroll
        for (int k = 0; k < 9; k++) trace += N9[k * 10];
        // eigenvector of the smallest eigenvalue: inverse iteration on N9 + shift * I (the shift keeps the factor positive
        // definite when the data are exact and N9 is singular to working precision; it does not move the eigenvectors)
        const double shift = 64.0 * DBL_EPSILON * trace;
#pragma unroll
        for (int k = 0; k < 9; k++) N9[k * 10] += shift;
        Chol<9> ch;
        ch.factor(N9);
        double v[9] = {0.31, -0.22, 0.53, 0.17, 0.41, -0.35, 0.29, -0.13, 0.37};
        for (int it = 0; it < 8; it++) {
            double w[9];
#pragma unroll
            for (int k = 0; k < 9; k++) w[k] = v[k];
            ch.solve(w);
            double nrm = 0;
#pragma unroll
            for (int k = 0; k < 9; k++) nrm += w[k] * w[k];
            nrm = 1.0 / sqrt(nrm);
            double sgn = 0;
#pragma unroll
            for (int k = 0; k < 9; k++) sgn += w[k] * v[k];
            if (sgn < 0) nrm = -nrm;
            double change = 0;
#pragma unroll
            for (int k = 0; k < 9; k++) { const double t = w[k] * nrm; change = fmax(change, fabs(t - v[k])); v[k] = t; }
            if (it > 0 && change < 4 * DBL_EPSILON) break;
        }
        // H = Tm^-1 * V * TM, scaled to H[8] = 1
        double T[9];
        T[0] = v[0] / smx + cmx * v[6]; T[1] = v[1] / smx + cmx * v[7]; T[2] = v[2] / smx + cmx * v[8];
        T[3] = v[3] / smy + cmy * v[6]; T[4] = v[4] / smy + cmy * v[7]; T[5] = v[5] / smy + cmy * v[8];
        T[6] = v[6]; T[7] = v[7]; T[8] = v[8];
        double R[9];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            R[3 * r + 0] = T[3 * r + 0] * sMx;
            R[3 * r + 1] = T[3 * r + 1] * sMy;
            R[3 * r + 2] = T[3 * r + 2] - T[3 * r + 0] * sMx * cMx - T[3 * r + 1] * sMy * cMy;
        }
        const double sc = 1.0 / R[8];
#pragma unroll
        for (int k = 0; k < 9; k++) H[k] = R[k] * sc;
    }
    int found = 1;
    if (degenerate && job.mode == 0) found = 0;       // method 0 / n == 4: runKernel failed -> empty result

    // ---- Levenberg-Marquardt polish over the inliers (LMSolver, <= 10 iterations), only when n > 4 ----
    int lm_iters = 0;
    if (found && n > 4) {
        double x[8];
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = H[k];
        LmSums cur;
        lm_evaluate(P, n, lane, mine, x, true, cur);
        double S = cur.S, rmax = cur.rmax;
        double D[8];
#pragma unroll
        for (int k = 0; k < 8; k++) D[k] = cur.A[k * 9];
        double lambda = 1.0, lc = 0.75;
        for (;;) {
            double Ap[64], d[8];
#pragma unroll
            for (int k = 0; k < 64; k++) Ap[k] = cur.A[k];
#pragma unroll
            for (int k = 0; k < 8; k++) { Ap[k * 9] += lambda * D[k]; d[k] = cur.v[k]; }
            Chol<8> ch;
            if (ch.factor(Ap)) ch.solve(d);
            else {
#pragma unroll
                for (int k = 0; k < 8; k++) d[k] = 0;
            }
            double xd[8];
#pragma unroll
            for (int k = 0; k < 8; k++) xd[k] = x[k] - d[k];
            LmSums trial;
            lm_evaluate(P, n, lane, mine, xd, false, trial);
            const double Sd = trial.S;
            double dS = 0, dv = 0, dmax = 0;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                double Ad = 0;
#pragma unroll
                for (int c = 0; c < 8; c++) Ad += cur.A[r * 8 + c] * d[c];
                dS += d[r] * (2 * cur.v[r] - Ad);
                dv += d[r] * cur.v[r];
                dmax = fmax(dmax, fabs(d[r]));
            }
            const double gain = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
            if (gain > 0.75) { lambda *= 0.5; if (lambda < lc) lambda = 0; }
            else if (gain < 0.25) {
                double nu = (Sd - S) / (fabs(dv) > DBL_EPSILON ? dv : 1) + 2;
                nu = fmin(fmax(nu, 2.), 10.);
                if (lambda == 0) {
                    // 1 / max |diag(A^-1)|: diag of the inverse from the Cholesky factor, (A^-1)_jj = |L^-1 e_j|^2
                    double maxval = DBL_EPSILON;
                    Chol<8> ca;
                    if (ca.factor(cur.A)) {
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            double e[8];
#pragma unroll
                            for (int k = 0; k < 8; k++) e[k] = k == j;
                            ca.forward(e);
                            double s2 = 0;
#pragma unroll
                            for (int k = 0; k < 8; k++) s2 += e[k] * e[k];
                            maxval = fmax(maxval, s2);
                        }
                    }
                    lambda = lc = 1. / maxval;
                    nu *= 0.5;
                }
                lambda *= nu;
            }
            if (Sd < S) {
                S = Sd;
#pragma unroll
                for (int k = 0; k < 8; k++) x[k] = xd[k];
                lm_evaluate(P, n, lane, mine, x, true, cur);
                rmax = cur.rmax;
            }
            lm_iters++;
            if (!(lm_iters < 10 && dmax >= FLT_EPSILON && rmax >= FLT_EPSILON)) break;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) H[k] = x[k];
    }
    if (lane == 0) {
        for (int k = 0; k < 9; k++) out->H[k] = found ? H[k] : 0;
        out->found = found; out->n_inliers = np; out->lm_iterations = lm_iters; out->dlt_degenerate = degenerate;
    }
    if (!found) for (int i = lane; i < n; i += 64) mask[i] = 0;
}

}  // namespace

hipError_t launch_hg_models(const HgPoint* pts, const HgFrame* frames, int n_frames, int max_hyp, const HgSample* samples,
                            int lmeds, float* err_scratch, int* scores, hipStream_t s) {
    if (n_frames <= 0 || max_hyp <= 0) return hipSuccess;
    dim3 grid((max_hyp + 3) / 4, n_frames);
    hipLaunchKernelGGL(hg_models_kernel, grid, dim3(256), 0, s, pts, frames, samples, lmeds, err_scratch, scores);
    return hipGetLastError();
}

hipError_t launch_hg_refine(const HgPoint* pts, const HgJob* jobs, int n_frames, HgResult* results, uint8_t* masks, hipStream_t s) {
    if (n_frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(hg_refine_kernel, dim3(n_frames), dim3(64), 0, s, pts, jobs, results, masks);
    return hipGetLastError();
}

}  // namespace geom
}  // namespace stk
