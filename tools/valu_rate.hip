// valu_rate.hip -- calibration of the vector-ALU issue roof on MI355X (gfx950) for `roofline_valu` (bench.py):
// how many cycles ONE SIMD needs per wave64 instruction of each kind the blend kernels are made of, as a function of the
// number of resident waves per SIMD.  Every wave runs REPS unrolled groups of 8 instructions of one kind (independent
// registers, or one dependent chain) and brackets them with s_memtime (shader cycles); the grid puts `waves` waves on every
// SIMD of every CU (256-thread workgroups = one wave per SIMD, `waves` workgroups per CU).
//   per wave:  median over waves of (s_memtime end - start) / instructions
//   per SIMD:  (last end - first start over ALL waves, from s_memrealtime (100 MHz, chip-wide) scaled by the measured shader
//              clock) / (instructions per wave * waves per SIMD) -- the figure `roofline_valu` prices instructions at
// build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate tools/valu_rate.hip ; run: tools/valu_rate > profiles/...txt
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

constexpr int REPS = 4096; // groups of 8 instructions per wave (the loop is unrolled 8 groups deep, so its branch is amortised over 64 instructions)

enum Kind { FMA_INDEP, FMA_DEP, MUL_DPP_INDEP, MUL_DPP_SCAN, EXP_INDEP, RCP_INDEP, PK_FMA_INDEP, CNDMASK_INDEP, SALU_INDEP, LDS_BCAST, KINDS };
static const char *kind_name[KINDS] = {"v_fma_f32 independent", "v_fma_f32 dependent chain", "v_mul_f32_dpp row_shr independent",
                                       "v_mul_f32_dpp 6-step wave scan (+s_nop 1 each)", "v_exp_f32 independent", "v_rcp_f32 independent",
                                       "v_pk_fma_f32 independent", "v_cndmask_b32 independent", "s_add_u32 independent (scalar)",
                                       "ds_read_b128 broadcast (same address)"};

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float *out, long long *cycles /* [waves][4]: memtime t0,t1, memrealtime r0,r1 */, float seed)
{
    __shared__ float4 lds[64];
    if (threadIdx.x < 64) lds[threadIdx.x] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0000001f, c = 1e-9f;
    int s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r8 = 0; r8 < REPS; r8 += 8)
#pragma unroll
    for (int r = r8; r < r8 + 8; ++r) {
        if constexpr (KIND == FMA_INDEP) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                         "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == FMA_DEP) {
            asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                         "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2"
                         : "+v"(a0) : "v"(b), "v"(c));
        } else if constexpr (KIND == MUL_DPP_INDEP) {
            asm volatile("v_mul_f32_dpp %0, %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mul_f32_dpp %1, %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mul_f32_dpp %2, %8, %8 row_shr:2 row_mask:0xf bank_mask:0xf\n\tv_mul_f32_dpp %3, %8, %8 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mul_f32_dpp %4, %8, %8 row_shr:8 row_mask:0xf bank_mask:0xf\n\tv_mul_f32_dpp %5, %8, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                         "v_mul_f32_dpp %6, %8, %8 row_bcast:31 row_mask:0xc bank_mask:0xf\n\tv_mul_f32_dpp %7, %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if constexpr (KIND == MUL_DPP_SCAN) { // the scan of blend_bwd_splat.hip: 6 dependent steps (counted as 8 with 2 extra below)
            asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_mul_f32 %0, %1, %0\n\tv_mul_f32 %0, %1, %0"
                         : "+v"(a0) : "v"(b));
        } else if constexpr (KIND == EXP_INDEP) {
            asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                         "v_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == RCP_INDEP) {
            asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\t"
                         "v_rcp_f32 %4, %4\n\tv_rcp_f32 %5, %5\n\tv_rcp_f32 %6, %6\n\tv_rcp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == PK_FMA_INDEP) { // 4 packed instructions on register pairs = 8 scalar FMAs of work
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, bb = {b, b}, cc = {c, c};
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t"
                         "v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(bb), "v"(cc));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if constexpr (KIND == CNDMASK_INDEP) {
            asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\t"
                         "v_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
        } else if constexpr (KIND == SALU_INDEP) {
            asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\t"
                         "s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if constexpr (KIND == LDS_BCAST) {
            float4 q0, q1, q2, q3, q4, q5, q6, q7;
            const float4 *p = lds + (r & 31);
            q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3]; q4 = p[4]; q5 = p[5]; q6 = p[6]; q7 = p[7];
            asm volatile("" : "+v"(q0.x), "+v"(q1.x), "+v"(q2.x), "+v"(q3.x), "+v"(q4.x), "+v"(q5.x), "+v"(q6.x), "+v"(q7.x));
            a0 += q0.x; a1 += q1.x; a2 += q2.x; a3 += q3.x; a4 += q4.x; a5 += q5.x; a6 += q6.x; a7 += q7.x;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) {
        cycles[4 * wave] = t0; cycles[4 * wave + 1] = t1; cycles[4 * wave + 2] = r0; cycles[4 * wave + 3] = r1;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(s0 + s1 + s2 + s3);
}

template <int KIND>
static void run(int waves_per_simd, float *out, long long *cyc_d, std::vector<long long> &host)
{
    const int cus = 256, blocks = cus * waves_per_simd, nw = blocks * 4;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc_d, 1.0f); // warm-up
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc_d, 1.0f);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    host.resize(4 * (size_t)nw);
    CHECK(hipMemcpy(host.data(), cyc_d, 4 * (size_t)nw * sizeof(long long), hipMemcpyDeviceToHost));
    std::vector<double> per_wave(nw), clk(nw);
    long long rmin = host[2], rmax = host[3];
    for (int w = 0; w < nw; ++w) {
        per_wave[w] = (double)(host[4 * w + 1] - host[4 * w]);
        clk[w] = per_wave[w] / ((double)(host[4 * w + 3] - host[4 * w + 2]) / 100e6);   // s_memrealtime ticks at 100 MHz
        rmin = std::min(rmin, host[4 * w + 2]);
        rmax = std::max(rmax, host[4 * w + 3]);
    }
    std::sort(per_wave.begin(), per_wave.end());
    std::sort(clk.begin(), clk.end());
    const double med = per_wave[nw / 2], ghz = clk[nw / 2] * 1e-9;
    const double insts = (double)REPS * 8;
    const double span_cycles = (double)(rmax - rmin) / 100e6 * clk[nw / 2];   // first start .. last end of any wave, in shader cycles
    printf("  %d waves/SIMD: %6.2f cycles per instruction for one wave; chip-wide %6.2f cycles per instruction per SIMD (all waves, first start to last end); "
           "shader clock %.2f GHz; kernel %.1f us\n",
           waves_per_simd, med / insts, span_cycles / (insts * waves_per_simd), ghz, ms * 1e3);
}

int main()
{
    float *out;
    long long *cyc;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * 8 * 256));
    CHECK(hipMalloc(&cyc, sizeof(long long) * 256 * 8 * 4 * 4));
    std::vector<long long> host;
    printf("# MI355X vector-instruction issue cost (s_memtime shader cycles; groups of 8 instructions x %d per wave; 256 CUs x 4 SIMDs)\n", REPS);
    printf("# for v_pk_fma_f32 and the 6-step scan an 'instruction' is 1/8 of the group: one scalar-FMA equivalent / one scan eighth\n");
#define SWEEP(K)                                                                                                  \
    printf("%s\n", kind_name[K]);                                                                                 \
    for (int w : {1, 2, 4, 8}) run<K>(w, out, cyc, host);
    SWEEP(FMA_INDEP)
    SWEEP(FMA_DEP)
    SWEEP(MUL_DPP_INDEP)
    SWEEP(MUL_DPP_SCAN)
    SWEEP(EXP_INDEP)
    SWEEP(RCP_INDEP)
    SWEEP(PK_FMA_INDEP)
    SWEEP(CNDMASK_INDEP)
    SWEEP(SALU_INDEP)
    SWEEP(LDS_BCAST)
    return 0;
}
