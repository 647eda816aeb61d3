// issue_costs.hip -- what one SIMD of an MI355X (gfx950) pays per wave64 instruction, for the instruction kinds the two blend
// kernels are made of beyond plain FMAs: selects (v_cndmask with VCC / an SGPR pair), compares (to VCC / to an SGPR pair),
// compare + select pairs, min/max/med3/and as select substitutes, VALU with an SGPR or literal operand, exec masking around an
// instruction, compare + never-taken vcc branch (the ballot early-out), v_readfirstlane, and the LDS read shapes of the list walk.
// tools/valu_rate.hip (round 2) found v_cndmask_b32 at 23 cycles per instruction per SIMD -- ten times an FMA; this program
// checks that and prices the alternatives.  Method as there: every wave runs REPS groups of 8 instructions between s_memtime
// stamps, `w` waves per SIMD on every SIMD of the chip; reported: cycles per instruction per SIMD (first start to last end over
// all waves, s_memrealtime scaled by the measured shader clock) and for one wave.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/issue_costs tools/issue_costs.hip ; run: tools/issue_costs > profiles/r03_issue_costs.txt
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CHECK(x)                                                                                                  \
    do {                                                                                                          \
        hipError_t e_ = (x);                                                                                      \
        if (e_ != hipSuccess) {                                                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                               \
            exit(1);                                                                                              \
        }                                                                                                         \
    } while (0)

constexpr int REPS = 2048;

// operands: %0..%7 = eight independent VGPRs (read-write), %8, %9 = VGPR constants, %10 = SGPR float, %11 = SGPR pair (lane mask)
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(sf), "s"(sm)
#define ALL8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)

enum Kind {
    FMA_V, FMA_S, MUL_LIT, MOV, MAX, MED3, AND, CNDMASK_VCC, CNDMASK_SGPR, CNDMASK_VCC_E64, CMP_VCC, CMP_SGPR, CMP_CNDMASK, CMP_CNDMASK_SGPR, CMPX,
    SAVEEXEC_FMA, CMP_BRANCH, READFIRSTLANE, EXP_DEP, MIN_U32, MAX_I32, BFI, SUB_U32, ASHR, ADD3, AND_OR, MAX3_F32, MIN_F32_LIT, MUL_CLAMP, FMAC, CMP_I32, CMP_CLASS, MOV_DPP, ADD_DPP_ROW, SUB_CO, MED3_I32, LSHL_ADD, MUL_LEGACY, CVT_I32, LDS_B128_BCAST, LDS_B64_BCAST, LDS_U16_BCAST, LDS_B128_LANES, LDS_B32_LANES, KINDS
};
static const char *kind_name[KINDS] = {
    "v_fma_f32 v,v,v,v (reference)",
    "v_fma_f32 with one SGPR operand",
    "v_mul_f32 with a 32-bit literal",
    "v_mov_b32 v,v",
    "v_max_f32",
    "v_med3_f32",
    "v_and_b32",
    "v_cndmask_b32 (VOP2, mask = vcc)",
    "v_cndmask_b32_e64 (mask = SGPR pair)",
    "v_cndmask_b32_e64 (mask = vcc, VOP3 encoding)",
    "v_cmp_lt_f32 -> vcc (VOPC)",
    "v_cmp_lt_f32_e64 -> SGPR pair",
    "v_cmp_lt_f32 -> vcc ; v_cndmask vcc   (pair = 2 instructions)",
    "v_cmp_lt_f32_e64 -> s[..] ; v_cndmask_e64 s[..]   (pair = 2 instructions)",
    "v_cmpx_lt_f32 (writes exec; all lanes stay on)",
    "s_and_saveexec_b64 ; v_fma ; s_or_b64 exec   (triple = 3 instructions)",
    "v_cmp_lt_f32 -> vcc ; s_cbranch_vccnz (never taken)   (pair = 2 instructions)",
    "v_readfirstlane_b32",
    "v_fma -> v_exp_f32 -> v_fma dependent chain (3 instructions per link)",
    "v_min_u32",
    "v_max_i32",
    "v_bfi_b32",
    "v_sub_u32 (no carry out)",
    "v_ashrrev_i32",
    "v_add3_u32",
    "v_and_or_b32",
    "v_max3_f32",
    "v_min_f32 with a literal",
    "v_mul_f32_e64 ... clamp",
    "v_fmac_f32 (VOP2)",
    "v_cmp_lt_i32 -> vcc",
    "v_cmp_class_f32 -> vcc",
    "v_mov_b32_dpp row_shr:1",
    "v_add_f32_dpp row_shr:1 (independent)",
    "v_sub_co_u32 (carry out to vcc)",
    "v_med3_i32",
    "v_lshl_add_u32",
    "v_mul_legacy_f32",
    "v_cvt_i32_f32",
    "ds_read_b128, all lanes one address",
    "ds_read_b64, all lanes one address",
    "ds_read_u16, all lanes one address",
    "ds_read_b128, lane-consecutive addresses (conflict-free)",
    "ds_read_b32, lane-consecutive addresses",
};

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float *out, long long *cycles /* [waves][4] */, float seed)
{
    __shared__ float4 lds[512];
    lds[threadIdx.x] = make_float4(seed, seed, seed, seed);
    lds[threadIdx.x + 256] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0000001f, c = 1e-9f;
    float sf = __builtin_amdgcn_readfirstlane(__float_as_int(seed)) == 0 ? 0.5f : 1.0000001f; // wave-uniform: lives in an SGPR
    unsigned long long sm = __builtin_amdgcn_read_exec() ^ 0x5555555555555555ull;             // a lane mask in an SGPR pair
    int sacc = 0;
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r4 = 0; r4 < REPS; r4 += 4)
#pragma unroll
    for (int r = r4; r < r4 + 4; ++r) {
        if constexpr (KIND == FMA_V) {
#define I(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == FMA_S) {
#define I(n) "v_fma_f32 %" #n ", %" #n ", %10, %9\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == MUL_LIT) {
#define I(n) "v_mul_f32 %" #n ", 0x3f800001, %" #n "\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == MOV) {
#define I(n) "v_mov_b32 %" #n ", %8\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == MAX) {
#define I(n) "v_max_f32 %" #n ", %" #n ", %8\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == MED3) {
#define I(n) "v_med3_f32 %" #n ", %" #n ", %8, %9\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == AND) {
#define I(n) "v_and_b32 %" #n ", %" #n ", %8\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == CNDMASK_VCC) {
#define I(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n\t"
            asm volatile(ALL8(I) OPS : "vcc");
#undef I
        } else if constexpr (KIND == CNDMASK_SGPR) {
#define I(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, %11\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == CNDMASK_VCC_E64) {
#define I(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, vcc\n\t"
            asm volatile(ALL8(I) OPS : "vcc");
#undef I
        } else if constexpr (KIND == CMP_VCC) {
#define I(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n\t"
            asm volatile(ALL8(I) OPS : "vcc");
#undef I
        } else if constexpr (KIND == CMP_SGPR) {
#define I(n) "v_cmp_lt_f32_e64 s[20:21], %" #n ", %8\n\t"
            asm volatile(ALL8(I) OPS : "s20", "s21");
#undef I
        } else if constexpr (KIND == CMP_CNDMASK) { // 4 pairs
#define I(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n\tv_cndmask_b32 %" #n ", %" #n ", %9, vcc\n\t"
            asm volatile(I(0) I(1) I(2) I(3) OPS : "vcc");
#undef I
        } else if constexpr (KIND == CMP_CNDMASK_SGPR) { // 4 pairs, four different SGPR pairs
            asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %8\n\tv_cndmask_b32_e64 %0, %0, %9, s[20:21]\n\t"
                         "v_cmp_lt_f32_e64 s[22:23], %1, %8\n\tv_cndmask_b32_e64 %1, %1, %9, s[22:23]\n\t"
                         "v_cmp_lt_f32_e64 s[24:25], %2, %8\n\tv_cndmask_b32_e64 %2, %2, %9, s[24:25]\n\t"
                         "v_cmp_lt_f32_e64 s[26:27], %3, %8\n\tv_cndmask_b32_e64 %3, %3, %9, s[26:27]\n\t" OPS
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if constexpr (KIND == CMPX) {
#define I(n) "v_cmpx_gt_f32 %" #n ", %9\n\t" /* a > 1e-9 for every lane: exec stays full */
            asm volatile(ALL8(I) OPS : "vcc");
#undef I
        } else if constexpr (KIND == SAVEEXEC_FMA) { // 8 instructions = 2.67 triples; reported per instruction
            asm volatile("s_and_saveexec_b64 s[20:21], %11\n\tv_fma_f32 %0, %0, %8, %9\n\ts_or_b64 exec, exec, s[20:21]\n\t"
                         "s_and_saveexec_b64 s[20:21], %11\n\tv_fma_f32 %1, %1, %8, %9\n\ts_or_b64 exec, exec, s[20:21]\n\t"
                         "s_and_saveexec_b64 s[20:21], %11\n\tv_fma_f32 %2, %2, %8, %9\n\t" OPS
                         : "s20", "s21", "scc");
            asm volatile("s_or_b64 exec, exec, s[20:21]\n\t"
                         "s_and_saveexec_b64 s[20:21], %11\n\tv_fma_f32 %3, %3, %8, %9\n\ts_or_b64 exec, exec, s[20:21]\n\t"
                         "s_and_saveexec_b64 s[20:21], %11\n\tv_fma_f32 %4, %4, %8, %9\n\ts_or_b64 exec, exec, s[20:21]\n\t"
                         "s_and_saveexec_b64 s[20:21], %11\n\tv_fma_f32 %5, %5, %8, %9\n\ts_or_b64 exec, exec, s[20:21]\n\t" OPS
                         : "s20", "s21", "scc");
        } else if constexpr (KIND == CMP_BRANCH) { // 4 pairs; a0..a3 stay > 0.5, so "a < 1e-9" is never true
            asm volatile("v_cmp_lt_f32 vcc, %0, %9\n\ts_cbranch_vccnz 1f\n\t1:\n\t"
                         "v_cmp_lt_f32 vcc, %1, %9\n\ts_cbranch_vccnz 2f\n\t2:\n\t"
                         "v_cmp_lt_f32 vcc, %2, %9\n\ts_cbranch_vccnz 3f\n\t3:\n\t"
                         "v_cmp_lt_f32 vcc, %3, %9\n\ts_cbranch_vccnz 4f\n\t4:\n\t" OPS
                         : "vcc");
        } else if constexpr (KIND == READFIRSTLANE) {
            int t;
            asm volatile("v_readfirstlane_b32 %0, %1\n\tv_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %0, %3\n\tv_readfirstlane_b32 %0, %4\n\t"
                         "v_readfirstlane_b32 %0, %5\n\tv_readfirstlane_b32 %0, %6\n\tv_readfirstlane_b32 %0, %7\n\tv_readfirstlane_b32 %0, %8\n\t"
                         : "=s"(t) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
            sacc += t;
        } else if constexpr (KIND == EXP_DEP) { // 8 instructions: 2.67 links
            asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_exp_f32 %0, %0\n\tv_fma_f32 %0, %0, %9, %8\n\t"
                         "v_fma_f32 %0, %0, %8, %9\n\tv_exp_f32 %0, %0\n\tv_fma_f32 %0, %0, %9, %8\n\t"
                         "v_fma_f32 %0, %0, %8, %9\n\tv_exp_f32 %0, %0\n\t" OPS);
        } else if constexpr (KIND == MIN_U32) {
#define I(n) "v_min_u32 %" #n ", %" #n ", %8\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == MAX_I32) {
#define I(n) "v_max_i32 %" #n ", %" #n ", %8\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == BFI) {
#define I(n) "v_bfi_b32 %" #n ", %8, %" #n ", %9\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == SUB_U32) {
#define I(n) "v_sub_u32 %" #n ", %" #n ", %8\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == ASHR) {
#define I(n) "v_ashrrev_i32 %" #n ", 1, %" #n "\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == ADD3) {
#define I(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == AND_OR) {
#define I(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == MAX3_F32) {
#define I(n) "v_max3_f32 %" #n ", %" #n ", %8, %9\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == MIN_F32_LIT) {
#define I(n) "v_min_f32 %" #n ", 0x3f7d70a4, %" #n "\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == MUL_CLAMP) {
#define I(n) "v_mul_f32_e64 %" #n ", %" #n ", %8 clamp\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == FMAC) {
#define I(n) "v_fmac_f32 %" #n ", %8, %9\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == CMP_I32) {
#define I(n) "v_cmp_lt_i32 vcc, %" #n ", %8\n\t"
            asm volatile(ALL8(I) OPS : "vcc");
#undef I
        } else if constexpr (KIND == CMP_CLASS) {
#define I(n) "v_cmp_class_f32 vcc, %" #n ", %8\n\t"
            asm volatile(ALL8(I) OPS : "vcc");
#undef I
        } else if constexpr (KIND == MOV_DPP) {
#define I(n) "v_mov_b32_dpp %" #n ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == ADD_DPP_ROW) {
#define I(n) "v_add_f32_dpp %" #n ", %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == SUB_CO) {
#define I(n) "v_sub_co_u32 %" #n ", vcc, %" #n ", %8\n\t"
            asm volatile(ALL8(I) OPS : "vcc");
#undef I
        } else if constexpr (KIND == MED3_I32) {
#define I(n) "v_med3_i32 %" #n ", %" #n ", %8, %9\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == LSHL_ADD) {
#define I(n) "v_lshl_add_u32 %" #n ", %" #n ", 1, %9\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == MUL_LEGACY) {
#define I(n) "v_mul_legacy_f32 %" #n ", %" #n ", %8\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == CVT_I32) {
#define I(n) "v_cvt_i32_f32 %" #n ", %8\n\t"
            asm volatile(ALL8(I) OPS);
#undef I
        } else if constexpr (KIND == LDS_B128_BCAST || KIND == LDS_B128_LANES) {
            float4 q0, q1, q2, q3, q4, q5, q6, q7;
            const float4 *p = lds + (r & 31) + (KIND == LDS_B128_LANES ? (threadIdx.x & 63) : 0);
            q0 = p[0]; q1 = p[64]; q2 = p[128]; q3 = p[192]; q4 = p[256]; q5 = p[320]; q6 = p[384]; q7 = p[416];
            asm volatile("" : "+v"(q0.x), "+v"(q1.x), "+v"(q2.x), "+v"(q3.x), "+v"(q4.x), "+v"(q5.x), "+v"(q6.x), "+v"(q7.x));
            a0 += q0.x; a1 += q1.x; a2 += q2.x; a3 += q3.x; a4 += q4.x; a5 += q5.x; a6 += q6.x; a7 += q7.x;
        } else if constexpr (KIND == LDS_B64_BCAST) {
            float2 q0, q1, q2, q3, q4, q5, q6, q7;
            const float2 *p = reinterpret_cast<const float2 *>(lds) + (r & 31);
            q0 = p[0]; q1 = p[64]; q2 = p[128]; q3 = p[192]; q4 = p[256]; q5 = p[320]; q6 = p[384]; q7 = p[416];
            asm volatile("" : "+v"(q0.x), "+v"(q1.x), "+v"(q2.x), "+v"(q3.x), "+v"(q4.x), "+v"(q5.x), "+v"(q6.x), "+v"(q7.x));
            a0 += q0.x; a1 += q1.x; a2 += q2.x; a3 += q3.x; a4 += q4.x; a5 += q5.x; a6 += q6.x; a7 += q7.x;
        } else if constexpr (KIND == LDS_U16_BCAST) {
            int q0, q1, q2, q3, q4, q5, q6, q7;
            const uint16_t *p = reinterpret_cast<const uint16_t *>(lds) + (r & 31);
            q0 = p[0]; q1 = p[64]; q2 = p[128]; q3 = p[192]; q4 = p[256]; q5 = p[320]; q6 = p[384]; q7 = p[416];
            asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));
            sacc += q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7;
        } else if constexpr (KIND == LDS_B32_LANES) {
            float q0, q1, q2, q3, q4, q5, q6, q7;
            const float *p = reinterpret_cast<const float *>(lds) + (r & 31) + (threadIdx.x & 63);
            q0 = p[0]; q1 = p[64]; q2 = p[128]; q3 = p[192]; q4 = p[256]; q5 = p[320]; q6 = p[384]; q7 = p[416];
            asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));
            a0 += q0; a1 += q1; a2 += q2; a3 += q3; a4 += q4; a5 += q5; a6 += q6; a7 += q7;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) {
        cycles[4 * wave] = t0; cycles[4 * wave + 1] = t1; cycles[4 * wave + 2] = r0; cycles[4 * wave + 3] = r1;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)sacc + sf + (float)(sm & 1);
}

template <int KIND>
static void run(int waves_per_simd, float *out, long long *cyc_d, std::vector<long long> &host)
{
    const int cus = 256, blocks = cus * waves_per_simd, nw = blocks * 4;
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc_d, 1.0f); // warm-up
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc_d, 1.0f);
    CHECK(hipDeviceSynchronize());
    host.resize(4 * (size_t)nw);
    CHECK(hipMemcpy(host.data(), cyc_d, 4 * (size_t)nw * sizeof(long long), hipMemcpyDeviceToHost));
    std::vector<double> per_wave(nw), clk(nw);
    long long rmin = host[2], rmax = host[3];
    for (int w = 0; w < nw; ++w) {
        per_wave[w] = (double)(host[4 * w + 1] - host[4 * w]);
        clk[w] = per_wave[w] / ((double)(host[4 * w + 3] - host[4 * w + 2]) / 100e6);
        rmin = std::min(rmin, host[4 * w + 2]);
        rmax = std::max(rmax, host[4 * w + 3]);
    }
    std::sort(per_wave.begin(), per_wave.end());
    std::sort(clk.begin(), clk.end());
    const double med = per_wave[nw / 2], ghz = clk[nw / 2] * 1e-9;
    const double insts = (double)REPS * 8;
    const double span_cycles = (double)(rmax - rmin) / 100e6 * clk[nw / 2];
    printf("  %d waves/SIMD: %6.2f cycles/instruction/SIMD   (one wave: %6.2f; clock %.2f GHz)\n", waves_per_simd,
           span_cycles / (insts * waves_per_simd), med / insts, ghz);
    fflush(stdout);
}

int main()
{
    float *out;
    long long *cyc;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * 8 * 256));
    CHECK(hipMalloc(&cyc, sizeof(long long) * 256 * 8 * 4 * 4));
    std::vector<long long> host;
    printf("# MI355X issue cost per wave64 instruction per SIMD (shader cycles; %d groups of 8 instructions per wave; 256 CUs x 4 SIMDs)\n", REPS);
    printf("# LDS kinds: the loop adds one v_add_f32 per read (not counted); pairs / triples are counted per instruction\n");
#define SWEEP(K)                                                                                                  \
    printf("%s\n", kind_name[K]);                                                                                 \
    for (int w : {1, 4, 8}) run<K>(w, out, cyc, host);
    SWEEP(FMA_V) SWEEP(FMA_S) SWEEP(MUL_LIT) SWEEP(MOV) SWEEP(MAX) SWEEP(MED3) SWEEP(AND)
    SWEEP(CNDMASK_VCC) SWEEP(CNDMASK_SGPR) SWEEP(CNDMASK_VCC_E64) SWEEP(CMP_VCC) SWEEP(CMP_SGPR) SWEEP(CMP_CNDMASK) SWEEP(CMP_CNDMASK_SGPR)
    SWEEP(CMPX) SWEEP(SAVEEXEC_FMA) SWEEP(CMP_BRANCH) SWEEP(READFIRSTLANE) SWEEP(EXP_DEP)
    SWEEP(MIN_U32) SWEEP(MAX_I32) SWEEP(BFI) SWEEP(SUB_U32) SWEEP(ASHR) SWEEP(ADD3) SWEEP(AND_OR) SWEEP(MAX3_F32) SWEEP(MIN_F32_LIT) SWEEP(MUL_CLAMP)
    SWEEP(FMAC) SWEEP(CMP_I32) SWEEP(CMP_CLASS) SWEEP(MOV_DPP) SWEEP(ADD_DPP_ROW) SWEEP(SUB_CO) SWEEP(MED3_I32) SWEEP(LSHL_ADD) SWEEP(MUL_LEGACY) SWEEP(CVT_I32)
    SWEEP(LDS_B128_BCAST) SWEEP(LDS_B64_BCAST) SWEEP(LDS_U16_BCAST) SWEEP(LDS_B128_LANES) SWEEP(LDS_B32_LANES)
    return 0;
}
