// tools/valu_rates.hip — what one VALU / LDS instruction costs a SIMD on gfx950, measured: a wave runs a long stream of
// INDEPENDENT instructions of one kind (16 accumulators, no memory traffic) on every CU, with 1, 4 and 8 waves per SIMD;
// the figure is wall time (HIP events) x the in-kernel clock (s_memtime / s_memrealtime) / instructions per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/valu_rates tools/valu_rates.hip && tools/bin/valu_rates
// Why: the engine's pixel passes are VALU-bound and were tuned by INSTRUCTION COUNT, a packed-f32
// instruction counted as one. Measured here (profiles/r04/valu_rates.txt): with two or more waves per SIMD a v_fma_f32 /
// v_mul_f32 / v_add_f32 / simple 32-bit integer op occupies the SIMD for ~2.2 cycles, every v_pk_*_f32, conversion,
// floor / fract, three-operand integer op, v_alignbyte, v_perm, DPP move and v_cmp for ~4.2, v_rcp_f32 for ~8.3.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define R4(S) S S S S
#define R16(S) R4(R4(S))

// X(name, mode, text): mode 1 = 32-bit accumulators (%0 acc, %1 %2 f32 inputs, %3 int input, %4 sgpr f32), 2 = register pairs
#define KINDS(X) \
    X(FMA, 1, "v_fma_f32 %0, %1, %2, %0") \
    X(FMA_NEG, 1, "v_fma_f32 %0, -%1, %2, %0") \
    X(FMA_SAME, 1, "v_fma_f32 %0, %1, %1, %0") \
    X(FMA_2SRC, 1, "v_fma_f32 %0, %0, %1, %0") \
    X(MUL_E64, 1, "v_mul_f32_e64 %0, %1, %0") \
    X(ADD_NEG, 1, "v_add_f32_e64 %0, %1, -%0") \
    X(FMAC, 1, "v_fmac_f32 %0, %1, %2") \
    X(MUL, 1, "v_mul_f32 %0, %1, %0") \
    X(ADD, 1, "v_add_f32 %0, %1, %0") \
    X(SUB, 1, "v_sub_f32 %0, %1, %0") \
    X(MAX, 1, "v_max_f32 %0, %1, %0") \
    X(MUL_SGPR, 1, "v_mul_f32 %0, %4, %0") \
    X(MUL_LIT, 1, "v_mul_f32 %0, 0x3b808081, %0") \
    X(FMA_SGPR, 1, "v_fma_f32 %0, %4, %2, %0") \
    X(FMA_LIT, 1, "v_fmaak_f32 %0, %1, %0, 0x3b808081") \
    X(PK_FMA, 2, "v_pk_fma_f32 %0, %1, %2, %0") \
    X(PK_MUL, 2, "v_pk_mul_f32 %0, %1, %0") \
    X(PK_ADD, 2, "v_pk_add_f32 %0, %1, %0") \
    X(PK_FMA_OPSEL, 2, "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]") \
    X(RCP, 1, "v_rcp_f32 %0, %0") \
    X(CVT_UB0, 1, "v_cvt_f32_ubyte0 %0, %0") \
    X(CVT_UB1, 1, "v_cvt_f32_ubyte1 %0, %0") \
    X(CVT_F32_U32, 1, "v_cvt_f32_u32 %0, %0") \
    X(CVT_F32_I32, 1, "v_cvt_f32_i32 %0, %0") \
    X(CVT_I32, 1, "v_cvt_i32_f32 %0, %0") \
    X(CVT_FLR, 1, "v_cvt_flr_i32_f32 %0, %0") \
    X(FLOOR, 1, "v_floor_f32 %0, %0") \
    X(FRACT, 1, "v_fract_f32 %0, %0") \
    X(RNDNE, 1, "v_rndne_f32 %0, %0") \
    X(MED3, 1, "v_med3_f32 %0, %0, %1, %2") \
    X(AND, 1, "v_and_b32 %0, %3, %0") \
    X(OR, 1, "v_or_b32 %0, %3, %0") \
    X(XOR, 1, "v_xor_b32 %0, %3, %0") \
    X(ADD_U32, 1, "v_add_u32 %0, %3, %0") \
    X(SUB_U32, 1, "v_sub_u32 %0, %0, %3") \
    X(LSHLREV, 1, "v_lshlrev_b32 %0, 2, %0") \
    X(LSHRREV, 1, "v_lshrrev_b32 %0, 8, %0") \
    X(ASHRREV, 1, "v_ashrrev_i32 %0, 8, %0") \
    X(MIN_I32, 1, "v_min_i32 %0, %3, %0") \
    X(MAX_U32, 1, "v_max_u32 %0, %3, %0") \
    X(MUL_U24, 1, "v_mul_u32_u24 %0, %3, %0") \
    X(MUL_I24, 1, "v_mul_i32_i24 %0, %3, %0") \
    X(MAD_U24, 1, "v_mad_u32_u24 %0, %0, %3, %3") \
    X(MUL_LO, 1, "v_mul_lo_u32 %0, %0, %3") \
    X(LSHL_ADD, 1, "v_lshl_add_u32 %0, %0, 2, %3") \
    X(ADD3, 1, "v_add3_u32 %0, %0, %3, %3") \
    X(LSHL_OR, 1, "v_lshl_or_b32 %0, %0, 2, %3") \
    X(AND_OR, 1, "v_and_or_b32 %0, %0, %3, %3") \
    X(BFE, 1, "v_bfe_u32 %0, %0, 8, 8") \
    X(BFI, 1, "v_bfi_b32 %0, %3, %0, %3") \
    X(ALIGNBYTE, 1, "v_alignbyte_b32 %0, %1, %0, %3") \
    X(PERM, 1, "v_perm_b32 %0, %0, %1, %3") \
    X(OR_SDWA_B1, 1, "v_or_b32_sdwa %0, %0, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD") \
    X(ADD_SDWA_W1, 1, "v_add_u32_sdwa %0, %0, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD") \
    X(MOV, 1, "v_mov_b32 %0, %1") \
    X(MOV_B64, 2, "v_mov_b64 %0, %1") \
    X(MOV_DPP, 1, "v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf") \
    X(ADD_DPP, 1, "v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
    X(PK_ADD_U16, 1, "v_pk_add_u16 %0, %0, %3") \
    X(PK_SUB_I16, 1, "v_pk_sub_i16 %0, %0, %3") \
    X(PK_MAX_I16, 1, "v_pk_max_i16 %0, %0, %3") \
    X(PK_MIN_U16, 1, "v_pk_min_u16 %0, %0, %3") \
    X(PK_LSHRREV_B16, 1, "v_pk_lshrrev_b16 %0, 1, %0") \
    X(SAD_U8, 1, "v_sad_u8 %0, %0, %3, %3") \
    X(MAX3_U32, 1, "v_max3_u32 %0, %0, %3, %3") \
    X(MIN3_I32, 1, "v_min3_i32 %0, %0, %3, %3") \
    X(BCNT, 1, "v_bcnt_u32_b32 %0, %3, %0") \
    X(MBCNT, 1, "v_mbcnt_lo_u32_b32 %0, %3, %0") \
    X(XAD, 1, "v_xad_u32 %0, %0, %3, %3") \
    X(ADD_CO, 10, "v_add_co_u32 %0, vcc, %3, %0") \
    X(CMP_F32, 3, "v_cmp_lt_f32 vcc, %1, %2") \
    X(CMP_U32, 3, "v_cmp_lt_u32 vcc, %3, %0") \
    X(CMP_SGPRDST, 3, "v_cmp_lt_f32 s[20:21], %1, %2") \
    X(CNDMASK, 1, "v_cndmask_b32 %0, %0, %1, vcc") \
    X(READLANE, 4, "v_readlane_b32 s20, %3, 63") \
    X(READFIRSTLANE, 4, "v_readfirstlane_b32 s20, %3") \
    X(PERMLANE32_SWAP, 1, "v_permlane32_swap_b32 %0, %0") \
    X(DS_SWIZZLE, 5, "ds_swizzle_b32 %0, %0 offset:0x101F") \
    X(DS_READ_B32, 5, "ds_read_b32 %0, %3") \
    X(DS_READ2_B32, 6, "ds_read2_b32 %0, %1 offset1:1") \
    X(DS_READ_B64, 6, "ds_read_b64 %0, %1") \
    X(DS_READ2_B64, 7, "ds_read2_b64 %0, %1 offset1:1") \
    X(DS_READ_B128, 7, "ds_read_b128 %0, %1") \
    X(S_NOP, 8, "s_nop 0") \
    X(S_ADD, 8, "s_add_u32 s20, s20, 1") \
    X(PK_MIX, 9, "")

enum Kind {
#define X(n, m, s) n,
    KINDS(X)
#undef X
    N_KINDS
};
static const char* kind_name[] = {
#define X(n, m, s) s,
    KINDS(X)
#undef X
};

template <int KIND>
__global__ __launch_bounds__(512) void rate_kernel(int iters, float seed, unsigned long long* cycles, float* sink) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    float a[16];
    f32x2 p[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { a[k] = seed + k + threadIdx.x * 1e-3f; p[k] = f32x2{a[k], a[k] + 0.5f}; }
    float x = seed * 1.0001f, y = seed * 0.9999f;
    f32x2 px = {x, y}, py = {y, x};
    f32x4 q4 = {x, y, x, y};
    int ix = (int)threadIdx.x * 16 & 0xff0;       // LDS byte address: 16 bytes per lane, conflict-free for every width below
    float sx = __builtin_amdgcn_readfirstlane(seed);
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x), "v"(y) : "vcc");
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#define OP1(STR, K) asm volatile(STR : "+v"(a[K]) : "v"(x), "v"(y), "v"(ix), "s"(sx));
#define OPV(STR, K) asm volatile(STR : "+v"(a[K]) : "v"(x), "v"(y), "v"(ix), "s"(sx) : "vcc");
#define OPP(STR, K) asm volatile(STR : "+v"(p[K]) : "v"(px), "v"(py), "v"(ix), "s"(sx));
#define OP3(STR, K) asm volatile(STR : : "v"(a[K]), "v"(x), "v"(y), "v"(ix), "s"(sx) : "vcc", "s20", "s21");
#define OP4(STR, K) asm volatile(STR : : "v"(a[K]), "v"(x), "v"(y), "v"(ix), "s"(sx) : "s20");
#define OP6(STR, K) asm volatile(STR : "=v"(p[K]) : "v"(ix), "v"(y), "v"(ix), "s"(sx) : "memory");
#define OP7(STR, K) asm volatile(STR : "=v"(q4) : "v"(ix), "v"(y), "v"(ix), "s"(sx) : "memory");
#define OP8(STR, K) asm volatile(STR : : "v"(x), "v"(x), "v"(y), "v"(ix), "s"(sx) : "s20", "scc");
#define ALL(OP, STR) R4(OP(STR, 0) OP(STR, 1) OP(STR, 2) OP(STR, 3) OP(STR, 4) OP(STR, 5) OP(STR, 6) OP(STR, 7) OP(STR, 8) OP(STR, 9) OP(STR, 10) OP(STR, 11) OP(STR, 12) OP(STR, 13) OP(STR, 14) OP(STR, 15))
        if constexpr (false) {}
#define X(n, m, s)                                                                          \
        else if constexpr (KIND == n) {                                                     \
            if constexpr (m == 1) { ALL(OP1, s) }                                           \
            else if constexpr (m == 2) { ALL(OPP, s) }                                      \
            else if constexpr (m == 3) { ALL(OP3, s) }                                      \
            else if constexpr (m == 4) { ALL(OP4, s) }                                      \
            else if constexpr (m == 5) { ALL(OP1, s) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } \
            else if constexpr (m == 6) { ALL(OP6, s) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } \
            else if constexpr (m == 7) { ALL(OP7, s) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } \
            else if constexpr (m == 8) { ALL(OP8, s) }                                      \
            else if constexpr (m == 10) { ALL(OPV, s) }                                     \
            else {                                                                          \
                R4(OPP("v_pk_fma_f32 %0, %1, %2, %0", 0) OP1("v_fma_f32 %0, %1, %2, %0", 0) OPP("v_pk_fma_f32 %0, %1, %2, %0", 1) OP1("v_fma_f32 %0, %1, %2, %0", 1) \
                   OPP("v_pk_fma_f32 %0, %1, %2, %0", 2) OP1("v_fma_f32 %0, %1, %2, %0", 2) OPP("v_pk_fma_f32 %0, %1, %2, %0", 3) OP1("v_fma_f32 %0, %1, %2, %0", 3) \
                   OPP("v_pk_fma_f32 %0, %1, %2, %0", 4) OP1("v_fma_f32 %0, %1, %2, %0", 4) OPP("v_pk_fma_f32 %0, %1, %2, %0", 5) OP1("v_fma_f32 %0, %1, %2, %0", 5) \
                   OPP("v_pk_fma_f32 %0, %1, %2, %0", 6) OP1("v_fma_f32 %0, %1, %2, %0", 6) OPP("v_pk_fma_f32 %0, %1, %2, %0", 7) OP1("v_fma_f32 %0, %1, %2, %0", 7)) \
            }                                                                               \
        }
        KINDS(X)
#undef X
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { cycles[0] = t1 - t0; cycles[1] = r1 - r0; }
    float s = q4.x + q4.w;
#pragma unroll
    for (int k = 0; k < 16; k++) s += a[k] + p[k].x + p[k].y;
    if (s == 12345.678f) sink[0] = s + ix + lds[ix & 1023];
}

template <int KIND>
static void run_kind(int n_cu, unsigned long long* d_cycles, float* d_sink) {
    const int iters = 4000, per_iter = 64;
    printf("%-44.44s", KIND == PK_MIX ? "v_pk_fma_f32, v_fma_f32 alternating" : kind_name[KIND]);
    for (int wps : {1, 2, 4, 8}) {                     // waves per SIMD: blocks of 256 / 512 threads, 1 / 2 / 4 blocks per CU
        const int threads = wps == 1 ? 256 : 512, blocks = n_cu * (wps <= 2 ? 1 : wps / 2);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        rate_kernel<KIND><<<blocks, threads>>>(10, 1.5f, d_cycles, d_sink);
        (void)hipEventRecord(e0);
        rate_kernel<KIND><<<blocks, threads>>>(iters, 1.5f, d_cycles, d_sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[2];
        (void)hipMemcpy(c, d_cycles, sizeof(c), hipMemcpyDeviceToHost);
        const double mhz = (double)c[0] / (double)c[1] * 100.0;
        // cycles a SIMD spends per wave-instruction = wall time x clock / (instructions per wave x waves per SIMD)
        printf(" %dw %5.2f", wps, (ms - 0.004) * 1e3 * mhz / ((double)iters * per_iter * wps));
        if (wps == 8) printf("  (%4.0f MHz)", mhz);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    printf("\n");
}

template <int K>
static void run_all(int n_cu, unsigned long long* c, float* s) {
    if constexpr (K < N_KINDS) { run_kind<K>(n_cu, c, s); run_all<K + 1>(n_cu, c, s); }
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no GPU\n"); return 1; }
    const int n_cu = prop.multiProcessorCount;
    printf("%s, %d CUs: core-clock cycles one SIMD spends per wave64 instruction, by waves per SIMD (independent instructions)\n", prop.gcnArchName, n_cu);
    unsigned long long* d_cycles; float* d_sink;
    (void)hipMalloc(&d_cycles, 64);
    (void)hipMalloc(&d_sink, 64);
    run_all<0>(n_cu, d_cycles, d_sink);
    return 0;
}
