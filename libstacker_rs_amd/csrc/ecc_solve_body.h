// ecc_solve_body.h — the per-iteration "solve" step of findTransformECC as a device routine (see
// kernels_ecc_solve.hip for the description): the normal equations, the parameter update and the loop control of
// one slot, run by one workgroup. (A variant that ran it inside the other slot group's pixel pass was measured in
// round 1 and removed: no gain at 32 or 256 frames per GPU)
#pragma once
#include "common.h"

#ifdef STK_SOLVE_TIMING
#define STK_TICK(i) do { if (threadIdx.x == 0 && slot == a.slot0) queue->dbg[i] = wall_clock64(); } while (0)
#else
#define STK_TICK(i) do { } while (0)
#endif

namespace stk {

// Claim the next template for a free slot, if one has been prepared already (queue->ready is raised by the prep stream
// while later frames are still crossing PCIe; device-resident stacks start with ready == n_frames). A slot that finds
// nothing stays idle (frame = -1) and asks again at the next solve launch.
__device__ inline void slot_take_next(EccSlot* sl, EccQueue* queue, const float* init_warps) {
    const int avail = min(__hip_atomic_load(&queue->ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT), queue->n_frames);
    int nxt = __hip_atomic_load(&queue->next_frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool got = false;
    while (nxt < avail) {
        const int seen = atomicCAS(&queue->next_frame, nxt, nxt + 1);
        if (seen == nxt) { got = true; break; }
        nxt = seen;
    }
    if (got) {
        sl->frame = nxt;
        sl->iter = 0;
        for (int k = 0; k < 9; k++) sl->warp[k] = init_warps ? init_warps[(size_t)nxt * 9 + k] : ((k % 4 == 0) ? 1.f : 0.f);
        sl->cI = 0; sl->cT = 0;
        sl->rho = -1;
    } else {
        sl->frame = -1;
    }
}

// cv::invert(DECOMP_LU) closed forms for CV_32F 2x2 / 3x3 (evaluated in double), serial.
__device__ inline void invert_small_f32(const float* S, int n, float* D) {
    if (n == 2) {
        double d = (double)S[0] * S[3] - (double)S[1] * S[2];
        if (d != 0.) {
            d = 1. / d;
            D[3] = (float)(S[0] * d); D[0] = (float)(S[3] * d);
            D[1] = (float)(-S[1] * d); D[2] = (float)(-S[2] * d);
        } else { for (int i = 0; i < 4; i++) D[i] = 0; }
        return;
    }
    const double s00 = S[0], s01 = S[1], s02 = S[2], s10 = S[3], s11 = S[4], s12 = S[5], s20 = S[6], s21 = S[7], s22 = S[8];
    double d = s00 * (s11 * s22 - s12 * s21) - s01 * (s10 * s22 - s12 * s20) + s02 * (s10 * s21 - s11 * s20);
    if (d != 0.) {
        d = 1. / d;
        D[0] = (float)((s11 * s22 - s12 * s21) * d); D[1] = (float)((s02 * s21 - s01 * s22) * d);
        D[2] = (float)((s01 * s12 - s02 * s11) * d); D[3] = (float)((s12 * s20 - s10 * s22) * d);
        D[4] = (float)((s00 * s22 - s02 * s20) * d); D[5] = (float)((s02 * s10 - s00 * s12) * d);
        D[6] = (float)((s10 * s21 - s11 * s20) * d); D[7] = (float)((s01 * s20 - s00 * s21) * d);
        D[8] = (float)((s00 * s11 - s01 * s10) * d);
    } else { for (int i = 0; i < 9; i++) D[i] = 0; }
}

// One workgroup of SOLVE_WAVES wavefronts solves one slot. PRE_REDUCED: the block partials were already reduced to
// the 66 sums by the caller's stage 1 (kernels_ecc_solve.hip); otherwise this workgroup reduces them itself.
template <int SOLVE_WAVES, bool PRE_REDUCED = false>
__device__ __forceinline__ void ecc_solve_body(const EccIterArgs& a, int slot, int motion, EccCriteria crit, EccQueue* queue,
                                               EccFrameResult* results, const float* init_warps) {
    EccSlot* sl = a.slots + slot;
    const int frame = sl->frame;
    if (frame < 0) return;
    const int P = motion == STK_MOTION_HOMOGRAPHY ? 8 : motion == STK_MOTION_AFFINE ? 6 : motion == STK_MOTION_EUCLIDEAN ? 3 : 2;
    const int NH = P * (P + 1) / 2, NS = NH + 3 * P + 6;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    __shared__ double S[ECC_MAX_SUMS];
    __shared__ float AB[8][16];          // [A | B] of the LU inverse, B starts as I
    __shared__ float Hinv[64];
    __shared__ float vec[4][8];          // ipf, tpf, iph, epf
    __shared__ double dvec[2][8];        // ipd, tpd

    // ---- 1. reduce block partials: 16 waves, wave w owns sums w, w+16, ...; all loads of a wave are
    //         issued before the first add so the HBM round trips overlap --------------------------------
    // slot state needed at the very end: loaded now (uniform -> scalar loads) so the latency hides behind the reduction
    const double prev_rho = sl->rho;
    const int prev_iter = sl->iter;
    float wm[9];
#pragma unroll
    for (int k = 0; k < 9; k++) wm[k] = sl->warp[k];
    const double* base = a.partials + (size_t)slot * NS * a.nb;
    if constexpr (PRE_REDUCED) {
        if (tid < NS) S[tid] = a.sums[(size_t)slot * ECC_MAX_SUMS + tid];      // stage 1 ran in the caller (kernels_ecc_solve.hip)
    } else {
        constexpr int KR = (ECC_MAX_SUMS + SOLVE_WAVES - 1) / SOLVE_WAVES;     // sums per wave
        constexpr int KB = KR < 6 ? KR : 6;                                     // sums per batch (bounds the registers)
        constexpr int JB = 5;                                                    // partials per lane and sum in flight
        const int nbi = (a.nb + 63) >> 6;
        for (int r0 = 0; r0 < KR; r0 += KB) {
            double acc[KB];
#pragma unroll
            for (int r = 0; r < KB; r++) acc[r] = 0;
            for (int j0 = 0; j0 < nbi; j0 += JB) {
                // all KB x JB loads of the batch are issued before the first add (one memory round trip instead of
                // JB); the adds keep the ascending-block order, out-of-range entries add +0
                double v[KB][JB];
#pragma unroll
                for (int j = 0; j < JB; j++) {
                    const int b = lane + 64 * (j0 + j);
#pragma unroll
                    for (int r = 0; r < KB; r++) {
                        const int k = wave + SOLVE_WAVES * (r0 + r);
                        v[r][j] = (b < a.nb && k < NS) ? base[(size_t)k * a.nb + b] : 0.0;
                    }
                }
#pragma unroll
                for (int j = 0; j < JB; j++)
#pragma unroll
                    for (int r = 0; r < KB; r++) acc[r] += v[r][j];
            }
#pragma unroll
            for (int r = 0; r < KB; r++) {
                double v = acc[r];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                const int k = wave + SOLVE_WAVES * (r0 + r);
                if (lane == 0 && k < NS) S[k] = v;
            }
        }
    }
    __syncthreads();
    STK_TICK(4);

    // ---- 2. statistics (every thread computes the same scalars; no divergence) ------------------
    const double cI = sl->cI, cT = sl->cT;
    const double* ST = S + NH + 3 * P;
    const double n = ST[0];
    const double mu = n > 0 ? ST[1] / n : 0, mv = n > 0 ? ST[3] / n : 0;   // means of the centred samples
    const double imgMean = cI + mu, tmpMean = cT + mv;
    const double imgVar = n > 0 ? fmax(ST[2] / n - mu * mu, 0.) : 0;
    const double tmpVar = n > 0 ? fmax(ST[4] / n - mv * mv, 0.) : 0;
    const double imgStd = sqrt(imgVar), tmpStd = sqrt(tmpVar);
    const double imgNorm = sqrt(n * imgStd * imgStd), tmpNorm = sqrt(n * tmpStd * tmpStd);
    // OpenCV subtracts the means cast to f32 (arithm_op's scalar path); dI/dT are those casts relative
    // to the centring offsets the iteration kernel used.
    const float imgMeanF = (float)imgMean, tmpMeanF = (float)tmpMean;
    const double dI = (double)imgMeanF - cI, dT = (double)tmpMeanF - cT;
    const double correlation = ST[5] - dT * ST[1] - dI * ST[3] + n * dT * dI;

    if (tid < P) {
        const double jm = S[NH + 2 * P + tid];
        const double ipd = S[NH + tid] - dI * jm;          // sum J.(Iw - mean.m)
        const double tpd = S[NH + P + tid] - dT * jm;      // sum J.(T - mean).m
        dvec[0][tid] = ipd; dvec[1][tid] = tpd;
        vec[0][tid] = (float)ipd; vec[1][tid] = (float)tpd;
    }
    // Hessian (upper triangle packed row-major) -> symmetric f32 matrix, augmented with I
    if (tid < P * 2 * P) {
        const int r = tid / (2 * P), c = tid - r * 2 * P;
        float v;
        if (c < P) {
            const int i = min(r, c), j = max(r, c);
            v = (float)S[i * P - i * (i - 1) / 2 + (j - i)];
        } else v = (c - P == r) ? 1.f : 0.f;
        AB[r][c] = v;
    }
    __syncthreads();
    STK_TICK(5);

    // ---- 3. inverse ------------------------------------------------------------------------------
    if (P <= 3) {
        if (tid == 0) {
            float Sm[9];
            for (int r = 0; r < P; r++) for (int c = 0; c < P; c++) Sm[r * P + c] = AB[r][c];
            invert_small_f32(Sm, P, Hinv);
        }
    } else if (wave == 0) {
        // hal::LU32f on [A | I]: one wavefront, each lane owns up to two elements of the P x 2P array.
        // Per pivot step every element is rewritten from a snapshot of the previous state, which is the
        // serial loop's arithmetic element by element (swap rows i,k; row j += (A[j][i] * -1/A[i][i]) * row i).
        const float eps = 1.1920929e-07f * 10;
        const int W2 = 2 * P, NE = P * W2;
        bool singular = false;
        for (int i = 0; i < P && !singular; i++) {
            int k = i;                                             // partial pivoting: strict '>' keeps the first maximum
            float col[8];                                          // column i, all rows: independent LDS reads, one round trip
#pragma unroll
            for (int j = 0; j < 8; j++) col[j] = j < P ? AB[j][i] : 0.f;
            float best = 0.f, piv = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float v = fabsf(col[j]);
                if (j == i) { best = v; piv = col[j]; }
                else if (j > i && j < P && v > best) { best = v; k = j; piv = col[j]; }
            }
            if (best < eps) { singular = true; break; }
            const float d = -1 / piv;
            float nv[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int t = lane + 64 * e;
                const int r = t / W2, c = t - r * W2;
                float v = 0;
                if (t < NE) {
                    const int src = (r == i) ? k : (r == k) ? i : r;   // row that sits in row r after the swap
                    v = AB[src][c];
                    if (r > i && c > i) v = v + (AB[src][i] * d) * AB[k][c];
                }
                nv[e] = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < 2; e++) { const int t = lane + 64 * e; if (t < NE) AB[t / W2][t % W2] = nv[e]; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (!singular) {
            // back substitution, one B column per lane, the solution kept in registers: the U entries are uniform
            // LDS reads that do not depend on the running solution, so they are all in flight at once
            // (same operations in the same order as the serial loop: sacc -= U[i][k] * x[k], then / U[i][i])
            float x[8];
#pragma unroll
            for (int i = 7; i >= 0; i--) {
                x[i] = 0.f;
                if (i < P) {
                    float sacc = AB[i][P + (lane < P ? lane : 0)];
#pragma unroll
                    for (int k = i + 1; k < 8; k++)
                        if (k < P) sacc -= AB[i][k] * x[k];
                    x[i] = sacc / AB[i][i];
                }
            }
            if (lane < P) {
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (i < P) Hinv[i * P + lane] = x[i];
            }
        } else if (lane < P * P) Hinv[lane] = 0.f;
    }
    __syncthreads();
    STK_TICK(6);

    // ---- 4. lambda, parameter update, loop control -------------------------------------------------
    if (tid < P) {                                             // iph
        float sacc = 0;
#pragma unroll
        for (int l = 0; l < 8; l++)
            if (l < P) sacc += Hinv[tid * P + l] * vec[0][l];
        vec[2][tid] = sacc;
    }
    __syncthreads();
    if (tid != 0) return;
    STK_TICK(7);

    const double last_rho = prev_rho;
    double rho = correlation / (imgNorm * tmpNorm);
    const int iter = prev_iter + 1;
    int status = 0;
    bool finished = false;
    if (rho != rho) { status = 1; finished = true; }
    else {
        // (loops over the fixed bound 8 with guards, fully unrolled: the LDS reads are then issued together instead of
        //  one dependent round trip per term — this serial tail was 5 us of the 19 us solve)
        double dot_ip = 0, dot_tp = 0;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < P) { dot_ip += (double)vec[0][k] * vec[2][k]; dot_tp += (double)vec[1][k] * vec[2][k]; }
        const double lambda_n = imgNorm * imgNorm - dot_ip;
        const double lambda_d = correlation - dot_tp;
        if (lambda_d <= 0.0) { rho = -1; status = 2; finished = true; }
        else {
            const float lamf = (float)(lambda_n / lambda_d);
            float epf[8], dp[8];
#pragma unroll
            for (int k = 0; k < 8; k++) epf[k] = k < P ? (float)((double)lamf * dvec[1][k] - dvec[0][k]) : 0.f;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float sacc = 0;
#pragma unroll
                for (int l = 0; l < 8; l++)
                    if (k < P && l < P) sacc += Hinv[k * P + l] * epf[l];
                dp[k] = sacc;
            }
            float* m = wm;
            if (motion == STK_MOTION_HOMOGRAPHY) {
                m[0] += dp[0]; m[3] += dp[1]; m[6] += dp[2]; m[1] += dp[3]; m[4] += dp[4]; m[7] += dp[5]; m[2] += dp[6]; m[5] += dp[7];
            } else if (motion == STK_MOTION_AFFINE) {
                m[0] += dp[0]; m[3] += dp[1]; m[1] += dp[2]; m[4] += dp[3]; m[2] += dp[4]; m[5] += dp[5];
            } else if (motion == STK_MOTION_TRANSLATION) {
                m[2] += dp[0]; m[5] += dp[1];
            } else {
                const double th = (double)dp[0] + asin((double)m[3]);
                m[2] += dp[1]; m[5] += dp[2];
                m[0] = m[4] = (float)cos(th);
                m[3] = (float)sin(th);
                m[1] = -m[3];
            }
#pragma unroll
            for (int k = 0; k < 9; k++) sl->warp[k] = wm[k];
        }
    }
    // for (i = 1; i <= nIter && fabs(rho - last_rho) >= eps; i++): would iteration iter+1 run?
    if (!finished) finished = (iter + 1 > crit.n_iter) || !(fabs(rho - last_rho) >= crit.eps);
    sl->iter = iter;
    sl->last_rho = last_rho;
    sl->rho = rho;
    sl->cI = imgMeanF; sl->cT = tmpMeanF;
    STK_TICK(8);
    if (finished) {
        EccFrameResult* r = results + frame;
        for (int k = 0; k < 9; k++) r->warp[k] = wm[k];
        r->iters = iter; r->status = status; r->rho = rho;
        slot_take_next(sl, queue, init_warps);
        __threadfence();
        atomicAdd(&queue->frames_done, 1);
    }
}


}  // namespace stk
