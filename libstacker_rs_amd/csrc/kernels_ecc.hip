// kernels_ecc.hip — findTransformECC (lib.rs:769-777; algorithm SURVEY.md §8a-E*) as two kernels
// per iteration, with the whole iteration loop resident on the device:
//
//  ecc_iter   ONE fused pass per iteration and slot: per template pixel, projective coordinates ->
//             bilinear gathers of (I, gx, gy) from the zero-padded frame-0 planes -> nearest mask ->
//             Jacobian in registers -> 66 moment sums (36 Hessian, 8 J.Iw, 8 J.T.m, 8 J.m, 6 scalars)
//             accumulated in f32 per lane, reduced with wavefront shuffles, then LDS across the
//             four waves, written as f64 block partials. OpenCV materialises ~150 f32 planes per
//             iteration for the same result; here the algorithmic traffic is 16 B/px.
//  ecc_solve  one block per slot: fixed-order f64 sum of the block partials (deterministic), then
//             the 8x8 normal equations exactly as OpenCV forms them (f32 Hessian, LU inverse in
//             f32, lambda in f64), warp update, convergence test, and — when a frame finishes —
//             hand the slot the next frame from the device-side queue. The host never sees an
//             iteration; it only polls EccQueue::frames_done.
//
// Several frames ("slots") iterate concurrently in one launch. blockIdx is decoded so that the
// blocks working on the SAME image region for different slots share blockIdx % 8, i.e. one XCD
// and its L2: the frame-0 planes they all gather from are fetched from HBM/MALL once per XCD.
#include "common.h"

namespace stk {

typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

template <int MOTION> struct MotionTraits;
template <> struct MotionTraits<STK_MOTION_TRANSLATION> { static constexpr int P = 2; };
template <> struct MotionTraits<STK_MOTION_EUCLIDEAN> { static constexpr int P = 3; };
template <> struct MotionTraits<STK_MOTION_AFFINE> { static constexpr int P = 6; };
template <> struct MotionTraits<STK_MOTION_HOMOGRAPHY> { static constexpr int P = 8; };

__device__ __forceinline__ float bilerp(f32x2_a4 top, f32x2_a4 bot, float ax, float ay) {
    const float v0 = __builtin_fmaf(ax, top.y - top.x, top.x);
    const float v1 = __builtin_fmaf(ax, bot.y - bot.x, bot.x);
    return __builtin_fmaf(ay, v1 - v0, v0);
}

template <int MOTION>
__global__ __launch_bounds__(256) void ecc_iter_kernel(EccIterArgs a) {
    constexpr int P = MotionTraits<MOTION>::P;
    constexpr int NH = P * (P + 1) / 2;
    constexpr int NS = NH + 3 * P + 6;

    // XCD-aware decode: bid = xcd + 8 * (slot + n_slots * (region / 8))
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = bid >> 3;
    const int slot = q % a.n_slots;
    const int region = (q / a.n_slots) * 8 + xcd;

    const EccSlot* sl = a.slots + slot;
    const int frame = sl->frame;
    if (frame < 0) return;                                    // idle slot: whole block leaves

    const float m0 = sl->warp[0], m1 = sl->warp[1], m2 = sl->warp[2];
    const float m3 = sl->warp[3], m4 = sl->warp[4], m5 = sl->warp[5];
    const float m6 = sl->warp[6], m7 = sl->warp[7], m8 = sl->warp[8];
    const float cI = sl->cI, cT = sl->cT;
    const bool den_is_w = (m8 == 1.0f);

    const float* __restrict__ T = a.templates + (size_t)frame * a.templ_plane_stride;
    const float* __restrict__ RI = a.ref.I;
    const float* __restrict__ RX = a.ref.gx;
    const float* __restrict__ RY = a.ref.gy;
    const int rs = a.ref.stride;
    const float fiw = (float)a.ref.w, fih = (float)a.ref.h;
    const float mxw = (float)(a.ref.w - 1), mxh = (float)(a.ref.h - 1);

    float acc[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) acc[k] = 0.f;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qw = (a.tw + 3) >> 2;                           // quads per row

    for (int y = region * 4 + wave; y < a.th; y += a.nb * 4) {
        const float fy = (float)y;
        const float rowX = __builtin_fmaf(m1, fy, m2);
        const float rowY = __builtin_fmaf(m4, fy, m5);
        const float rowW = __builtin_fmaf(m7, fy, m8);
        const float rowD = __builtin_fmaf(m7, fy, 1.0f);
        const float* trow = T + (size_t)y * a.templ_row_stride;
        for (int qx = lane; qx < qw; qx += 64) {
            const float4 t4 = *reinterpret_cast<const float4*>(trow + qx * 4);
            const float tv[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int x = qx * 4 + j;
                if (x < a.tw) {
                    const float fx = (float)x;
                    float sx = __builtin_fmaf(m0, fx, rowX);
                    float sy = __builtin_fmaf(m3, fx, rowY);
                    float rden = 1.0f, hx = 0.0f, hy = 0.0f;      // 1/den, hatX, hatY of the homography Jacobian
                    if constexpr (MOTION == STK_MOTION_HOMOGRAPHY) {
                        const float rw = 1.0f / __builtin_fmaf(m6, fx, rowW);
                        rden = rw;
                        if (!den_is_w) rden = 1.0f / __builtin_fmaf(m6, fx, rowD);   // den = X*h2 + Y*h5 + 1
                        hx = -sx * rden; hy = -sy * rden;
                        sx *= rw; sy *= rw;
                    }
                    const float flx = __builtin_floorf(sx), fly = __builtin_floorf(sy);
                    const float ax = sx - flx, ay = sy - fly;
                    // clamp the integer coordinate into the zero border; NaN -> -2 (all taps zero)
                    const int ix = (int)__builtin_fminf(__builtin_fmaxf(flx, -2.0f), fiw);
                    const int iy = (int)__builtin_fminf(__builtin_fmaxf(fly, -2.0f), fih);
                    const int off = iy * rs + ix;
                    const float Iw = bilerp(*(const f32x2_a4*)(RI + off), *(const f32x2_a4*)(RI + off + rs), ax, ay);
                    const float gxw = bilerp(*(const f32x2_a4*)(RX + off), *(const f32x2_a4*)(RX + off + rs), ax, ay);
                    const float gyw = bilerp(*(const f32x2_a4*)(RY + off), *(const f32x2_a4*)(RY + off + rs), ax, ay);
                    // INTER_NEAREST mask: rounded source coordinate inside the input image
                    const float rx = __builtin_rintf(sx), ry = __builtin_rintf(sy);
                    const bool inside = (rx >= 0.0f) & (rx <= mxw) & (ry >= 0.0f) & (ry <= mxh);
                    const float mf = inside ? 1.0f : 0.0f;

                    float J[P];
                    if constexpr (MOTION == STK_MOTION_HOMOGRAPHY) {
                        const float ja = gxw * rden, jb = gyw * rden;
                        const float jt = hx * ja + hy * jb;
                        J[0] = ja * fx; J[1] = jb * fx; J[2] = jt * fx;
                        J[3] = ja * fy; J[4] = jb * fy; J[5] = jt * fy;
                        J[6] = ja; J[7] = jb;
                    } else if constexpr (MOTION == STK_MOTION_AFFINE) {
                        J[0] = gxw * fx; J[1] = gyw * fx; J[2] = gxw * fy; J[3] = gyw * fy; J[4] = gxw; J[5] = gyw;
                    } else if constexpr (MOTION == STK_MOTION_EUCLIDEAN) {
                        const float ex = -(fx * m3) - (fy * m0);     // h0 = m00 (cos), h1 = m10 (sin)
                        const float ey = (fx * m0) - (fy * m3);
                        J[0] = gxw * ex + gyw * ey; J[1] = gxw; J[2] = gyw;
                    } else {
                        J[0] = gxw; J[1] = gyw;
                    }
                    // centred samples: u = Iw - cI inside the mask (Iw outside), v = (T - cT) inside, 0 outside
                    const float u = inside ? Iw - cI : Iw;
                    const float v = inside ? tv[j] - cT : 0.0f;
                    int idx = 0;
#pragma unroll
                    for (int k = 0; k < P; k++)
#pragma unroll
                        for (int l = k; l < P; l++) { acc[idx] = __builtin_fmaf(J[k], J[l], acc[idx]); idx++; }
#pragma unroll
                    for (int k = 0; k < P; k++) {
                        acc[NH + k] = __builtin_fmaf(J[k], u, acc[NH + k]);
                        acc[NH + P + k] = __builtin_fmaf(J[k], v, acc[NH + P + k]);
                        acc[NH + 2 * P + k] = __builtin_fmaf(J[k], mf, acc[NH + 2 * P + k]);
                    }
                    const float um = u * mf;
                    acc[NH + 3 * P + 0] += mf;
                    acc[NH + 3 * P + 1] += um;
                    acc[NH + 3 * P + 2] = __builtin_fmaf(um, u, acc[NH + 3 * P + 2]);
                    acc[NH + 3 * P + 3] += v;
                    acc[NH + 3 * P + 4] = __builtin_fmaf(v, v, acc[NH + 3 * P + 4]);
                    acc[NH + 3 * P + 5] = __builtin_fmaf(um, v, acc[NH + 3 * P + 5]);
                }
            }
        }
    }

    // wavefront reduction (64 lanes), then the four waves through LDS in f64
#pragma unroll
    for (int k = 0; k < NS; k++) {
        float v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        acc[k] = v;
    }
    __shared__ double red[4][NS];
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NS; k++) red[wave][k] = (double)acc[k];
    }
    __syncthreads();
    if (threadIdx.x < NS) {
        const int k = threadIdx.x;
        const double s = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
        a.partials[((size_t)slot * a.nb + region) * NS + k] = s;
    }
}

hipError_t launch_ecc_iter(const EccIterArgs& a, int motion, hipStream_t s) {
    const int grid = a.nb * a.n_slots;
    switch (motion) {
        case STK_MOTION_HOMOGRAPHY: ecc_iter_kernel<STK_MOTION_HOMOGRAPHY><<<grid, 256, 0, s>>>(a); break;
        case STK_MOTION_AFFINE: ecc_iter_kernel<STK_MOTION_AFFINE><<<grid, 256, 0, s>>>(a); break;
        case STK_MOTION_EUCLIDEAN: ecc_iter_kernel<STK_MOTION_EUCLIDEAN><<<grid, 256, 0, s>>>(a); break;
        case STK_MOTION_TRANSLATION: ecc_iter_kernel<STK_MOTION_TRANSLATION><<<grid, 256, 0, s>>>(a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// solve
// ---------------------------------------------------------------------------------------------

// hal::LU32f on [A | b] with partial pivoting (core/src/matrix_decomp.cpp), eps = FLT_EPSILON*10.
__device__ bool lu_inverse_f32(float* A, int n, float* B) {
    const float eps = 1.1920929e-07f * 10;
    for (int i = 0; i < n; i++) {
        int k = i;
        for (int j = i + 1; j < n; j++)
            if (fabsf(A[j * n + i]) > fabsf(A[k * n + i])) k = j;
        if (fabsf(A[k * n + i]) < eps) return false;
        if (k != i) {
            for (int j = i; j < n; j++) { float t = A[i * n + j]; A[i * n + j] = A[k * n + j]; A[k * n + j] = t; }
            for (int j = 0; j < n; j++) { float t = B[i * n + j]; B[i * n + j] = B[k * n + j]; B[k * n + j] = t; }
        }
        const float d = -1 / A[i * n + i];
        for (int j = i + 1; j < n; j++) {
            const float alpha = A[j * n + i] * d;
            for (int kk = i + 1; kk < n; kk++) A[j * n + kk] += alpha * A[i * n + kk];
            for (int kk = 0; kk < n; kk++) B[j * n + kk] += alpha * B[i * n + kk];
        }
    }
    for (int i = n - 1; i >= 0; i--)
        for (int j = 0; j < n; j++) {
            float s = B[i * n + j];
            for (int k = i + 1; k < n; k++) s -= A[i * n + k] * B[k * n + j];
            B[i * n + j] = s / A[i * n + i];
        }
    return true;
}

// cv::invert(DECOMP_LU) for CV_32F n x n: closed forms in double for n <= 3, LU32f otherwise.
__device__ void invert_f32(const float* S, int n, float* D) {
    if (n == 2) {
        double d = (double)S[0] * S[3] - (double)S[1] * S[2];
        if (d != 0.) {
            d = 1. / d;
            D[3] = (float)(S[0] * d); D[0] = (float)(S[3] * d);
            D[1] = (float)(-S[1] * d); D[2] = (float)(-S[2] * d);
        } else { for (int i = 0; i < 4; i++) D[i] = 0; }
        return;
    }
    if (n == 3) {
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
        return;
    }
    float A[64];
    for (int i = 0; i < n * n; i++) { A[i] = S[i]; D[i] = 0; }
    for (int i = 0; i < n; i++) D[i * n + i] = 1.f;
    if (!lu_inverse_f32(A, n, D))
        for (int i = 0; i < n * n; i++) D[i] = 0;
}

__device__ void slot_take_next(EccSlot* sl, EccQueue* queue, const float* init_warps) {
    const int nxt = atomicAdd(&queue->next_frame, 1);
    if (nxt < queue->n_frames) {
        sl->frame = nxt;
        sl->iter = 0;
        for (int k = 0; k < 9; k++) sl->warp[k] = init_warps ? init_warps[(size_t)nxt * 9 + k] : ((k % 4 == 0) ? 1.f : 0.f);
        sl->cI = 0; sl->cT = 0;
        sl->rho = -1;
    } else {
        sl->frame = -1;
    }
}

__global__ __launch_bounds__(256) void ecc_solve_kernel(EccIterArgs a, int motion, EccCriteria crit, EccQueue* queue,
                                                        EccFrameResult* results, const float* init_warps) {
    const int slot = blockIdx.x;
    EccSlot* sl = a.slots + slot;
    const int frame = sl->frame;
    if (frame < 0) return;
    const int P = motion == STK_MOTION_HOMOGRAPHY ? 8 : motion == STK_MOTION_AFFINE ? 6 : motion == STK_MOTION_EUCLIDEAN ? 3 : 2;
    const int NH = P * (P + 1) / 2, NS = NH + 3 * P + 6;

    __shared__ double part[4][ECC_MAX_SUMS];
    __shared__ double S[ECC_MAX_SUMS];
    const double* base = a.partials + (size_t)slot * a.nb * NS;
    for (int t = threadIdx.x; t < 4 * NS; t += blockDim.x) {
        const int g = t / NS, k = t - g * NS;
        const int b0 = (a.nb * g) / 4, b1 = (a.nb * (g + 1)) / 4;
        double s = 0;
        for (int b = b0; b < b1; b++) s += base[(size_t)b * NS + k];
        part[g][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < NS) {
        const int k = threadIdx.x;
        S[k] = ((part[0][k] + part[1][k]) + part[2][k]) + part[3][k];
    }
    __syncthreads();
    if (threadIdx.x != 0) return;

    // ---- normal equations, as ecc.cpp forms them (SURVEY §8a-E* steps b..k) --------------------
    const double cI = sl->cI, cT = sl->cT;
    const double* ST = S + NH + 3 * P;
    const double n = ST[0];
    const double mu = n > 0 ? ST[1] / n : 0, mv = n > 0 ? ST[3] / n : 0;   // means of centred samples
    const double imgMean = cI + mu, tmpMean = cT + mv;
    const double imgVar = n > 0 ? fmax(ST[2] / n - mu * mu, 0.) : 0;
    const double tmpVar = n > 0 ? fmax(ST[4] / n - mv * mv, 0.) : 0;
    const double imgStd = sqrt(imgVar), tmpStd = sqrt(tmpVar);
    const double imgNorm = sqrt(n * imgStd * imgStd), tmpNorm = sqrt(n * tmpStd * tmpStd);
    // OpenCV subtracts the means cast to f32 (arithm_op scalar path); dI/dT are those casts
    // relative to the centring offsets used while accumulating.
    const float imgMeanF = (float)imgMean, tmpMeanF = (float)tmpMean;
    const double dI = (double)imgMeanF - cI, dT = (double)tmpMeanF - cT;

    float Hf[64], Hinv[64], ipf[8], tpf[8];
    double ipd[8], tpd[8];
    {
        int idx = 0;
        for (int k = 0; k < P; k++)
            for (int l = k; l < P; l++) { Hf[k * P + l] = (float)S[idx]; Hf[l * P + k] = Hf[k * P + l]; idx++; }
    }
    for (int k = 0; k < P; k++) {
        const double jm = S[NH + 2 * P + k];
        ipd[k] = S[NH + k] - dI * jm;           // sum J.(Iw - mean.m)
        tpd[k] = S[NH + P + k] - dT * jm;       // sum J.(T - mean).m
        ipf[k] = (float)ipd[k]; tpf[k] = (float)tpd[k];
    }
    const double correlation = ST[5] - dT * ST[1] - dI * ST[3] + n * dT * dI;
    invert_f32(Hf, P, Hinv);

    const double last_rho = sl->rho;
    double rho = correlation / (imgNorm * tmpNorm);
    const int iter = sl->iter + 1;
    int status = 0;
    bool finished = false;
    float dp[8];
    if (rho != rho) { status = 1; finished = true; }
    else {
        float iph[8];
        for (int k = 0; k < P; k++) { float s = 0; for (int l = 0; l < P; l++) s += Hinv[k * P + l] * ipf[l]; iph[k] = s; }
        double dot_ip = 0, dot_tp = 0;
        for (int k = 0; k < P; k++) { dot_ip += (double)ipf[k] * iph[k]; dot_tp += (double)tpf[k] * iph[k]; }
        const double lambda_n = imgNorm * imgNorm - dot_ip;
        const double lambda_d = correlation - dot_tp;
        if (lambda_d <= 0.0) { rho = -1; status = 2; finished = true; }
        else {
            const float lamf = (float)(lambda_n / lambda_d);
            float epf[8];
            for (int k = 0; k < P; k++) epf[k] = (float)((double)lamf * tpd[k] - ipd[k]);
            for (int k = 0; k < P; k++) { float s = 0; for (int l = 0; l < P; l++) s += Hinv[k * P + l] * epf[l]; dp[k] = s; }
            float* m = sl->warp;
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
        }
    }
    if (!finished) {
        // for (i = 1; i <= nIter && fabs(rho - last_rho) >= eps; i++): would iteration iter+1 run?
        finished = (iter + 1 > crit.n_iter) || !(fabs(rho - last_rho) >= crit.eps);
    }
    sl->iter = iter;
    sl->last_rho = last_rho;
    sl->rho = rho;
    sl->cI = imgMeanF; sl->cT = tmpMeanF;
    if (finished) {
        EccFrameResult* r = results + frame;
        for (int k = 0; k < 9; k++) r->warp[k] = sl->warp[k];
        r->iters = iter; r->status = status; r->rho = rho;
        slot_take_next(sl, queue, init_warps);
        __threadfence();
        atomicAdd(&queue->frames_done, 1);
    }
}

hipError_t launch_ecc_solve(const EccIterArgs& a, int motion, EccCriteria crit, EccQueue* queue,
                            EccFrameResult* results, hipStream_t s) {
    ecc_solve_kernel<<<a.n_slots, 256, 0, s>>>(a, motion, crit, queue, results, nullptr);
    return hipGetLastError();
}

__global__ void ecc_init_kernel(EccSlot* slots, int n_slots, EccQueue* queue, int n_frames, EccFrameResult* results,
                                const float* init_warps) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    queue->next_frame = 0; queue->n_frames = n_frames; queue->frames_done = 0; queue->pad = 0;
    for (int f = 0; f < n_frames; f++) { results[f].status = 3; results[f].iters = 0; results[f].rho = -1; }
    for (int s = 0; s < n_slots; s++) {
        slots[s].last_rho = 0;
        slot_take_next(slots + s, queue, init_warps);
    }
}

hipError_t launch_ecc_init(EccSlot* slots, int n_slots, EccQueue* queue, int n_frames, EccFrameResult* results,
                           const float* init_warps, hipStream_t s) {
    ecc_init_kernel<<<1, 64, 0, s>>>(slots, n_slots, queue, n_frames, results, init_warps);
    return hipGetLastError();
}

}  // namespace stk
