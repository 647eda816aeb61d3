// census.hip -- how many waves does an MI355X (gfx950) CU really keep resident for a given kernel shape?
//
// blend_backward_splat_kernel (62 VGPRs, 37 SGPRs, 4608 B of LDS, single-wave workgroups) admits 8 waves
// per SIMD by every documented rule and yet SQ_WAVE_CYCLES and the kernel's own timeline show 5 (20 workgroups per CU, flat).
// This program launches spin kernels of a chosen shape -- threads per workgroup, allocated VGPRs (forced with a clobber of
// the highest register), static + dynamic LDS bytes -- on an oversubscribed grid.  Every wave records HW_ID / XCC_ID and its
// s_memrealtime start and end (100 MHz, chip-wide), and spins for SPIN_US.  The host then counts, per physical CU
// (xcc, se, sh, cu), the largest number of waves alive at one time, and prints the mode / min / max over CUs.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/census tools/census.hip ;  run: tools/census > profiles/r03_census.txt
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <vector>

#define CHECK(x)                                                                                                  \
    do {                                                                                                          \
        hipError_t e_ = (x);                                                                                      \
        if (e_ != hipSuccess) {                                                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                               \
            exit(1);                                                                                              \
        }                                                                                                         \
    } while (0)

struct WaveRec {
    uint32_t hw_id, xcc_id;
    long long t0, t1;
};

constexpr int SPIN_TICKS = 2000; // 20 us of s_memrealtime

// s_getreg simm16 = id | offset << 6 | (size - 1) << 11 ;  HW_REG_HW_ID = 4, HW_REG_XCC_ID = 20 (gfx940+)
#define GETREG(id) __builtin_amdgcn_s_getreg((id) | (0 << 6) | (31 << 11))

template <int VGPRS, int SGPRS, bool BARRIER>
__device__ __forceinline__ void census_body(WaveRec *out, float *sink)
{
    extern __shared__ float dyn_lds[];
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (VGPRS >= 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    else if constexpr (VGPRS >= 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    else if constexpr (VGPRS >= 72) asm volatile("v_mov_b32 v71, 0" ::: "v71");
    else if constexpr (VGPRS >= 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    else if constexpr (VGPRS >= 56) asm volatile("v_mov_b32 v55, 0" ::: "v55");
    if constexpr (SGPRS >= 96) asm volatile("s_mov_b32 s95, 0" ::: "s95");
    else if constexpr (SGPRS >= 80) asm volatile("s_mov_b32 s79, 0" ::: "s79");
    else if constexpr (SGPRS >= 48) asm volatile("s_mov_b32 s47, 0" ::: "s47");
    dyn_lds[threadIdx.x] = (float)threadIdx.x; // the allocation is what matters; touch it anyway
    if constexpr (BARRIER) __syncthreads();
    long long t1;
    do {
        __builtin_amdgcn_s_sleep(8);
        t1 = __builtin_amdgcn_s_memrealtime();
    } while (t1 - t0 < SPIN_TICKS);
    if constexpr (BARRIER) __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        const int wave = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[wave] = WaveRec{GETREG(4), GETREG(20), t0, t1};
    }
    if (dyn_lds[(threadIdx.x + 1) % blockDim.x] == -1.0f) *sink = 1.0f;
}

template <int VGPRS, int SGPRS, bool BARRIER>
__global__ __launch_bounds__(64) void census64(WaveRec *out, float *sink) { census_body<VGPRS, SGPRS, BARRIER>(out, sink); }
template <int VGPRS, int SGPRS, bool BARRIER>
__global__ __launch_bounds__(128) void census128(WaveRec *out, float *sink) { census_body<VGPRS, SGPRS, BARRIER>(out, sink); }
template <int VGPRS, int SGPRS, bool BARRIER>
__global__ __launch_bounds__(256) void census256(WaveRec *out, float *sink) { census_body<VGPRS, SGPRS, BARRIER>(out, sink); }

struct Result {
    int cus, mode, lo, hi;
    double span_us;
};

static Result analyse(const std::vector<WaveRec> &w)
{
    std::map<uint32_t, std::vector<std::pair<long long, int>>> ev; // per CU: (time, +1 / -1)
    long long tmin = w[0].t0, tmax = w[0].t1;
    for (const WaveRec &r : w) {
        // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
        const uint32_t cu = (r.hw_id >> 8) & 0xF, sh = (r.hw_id >> 12) & 1, se = (r.hw_id >> 13) & 7, xcc = r.xcc_id & 0xF;
        const uint32_t key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        ev[key].push_back({r.t0, +1});
        ev[key].push_back({r.t1, -1});
        tmin = std::min(tmin, r.t0);
        tmax = std::max(tmax, r.t1);
    }
    std::map<int, int> hist;
    int lo = 1 << 30, hi = 0;
    for (auto &kv : ev) {
        auto &v = kv.second;
        std::sort(v.begin(), v.end(), [](auto &a, auto &b) { return a.first != b.first ? a.first < b.first : a.second < b.second; });
        int cur = 0, best = 0;
        for (auto &e : v) {
            cur += e.second;
            best = std::max(best, cur);
        }
        hist[best]++;
        lo = std::min(lo, best);
        hi = std::max(hi, best);
    }
    int mode = 0, cnt = 0;
    for (auto &h : hist)
        if (h.second > cnt) cnt = h.second, mode = h.first;
    return Result{(int)ev.size(), mode, lo, hi, (tmax - tmin) / 100.0};
}

template <typename K>
static void run(const char *label, K kernel, int threads, int vgprs, int sgprs, int lds_bytes, WaveRec *d_out, float *d_sink, int wgs)
{
    const int waves = wgs * threads / 64;
    CHECK(hipMemset(d_out, 0, sizeof(WaveRec) * waves));
    hipLaunchKernelGGL(kernel, dim3(wgs), dim3(threads), lds_bytes, 0, d_out, d_sink);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    std::vector<WaveRec> h(waves);
    CHECK(hipMemcpy(h.data(), d_out, sizeof(WaveRec) * waves, hipMemcpyDeviceToHost));
    const Result r = analyse(h);
    printf("%-10s threads/WG %4d  VGPR<= %3d  SGPR<= %3d  LDS %6d B/WG | CUs seen %3d | resident waves per CU: mode %2d (min %2d max %2d) = %.2f per SIMD | WGs per CU %5.1f | span %.0f us\n",
           label, threads, vgprs, sgprs, lds_bytes, r.cus, r.mode, r.lo, r.hi, r.mode / 4.0, r.mode / (threads / 64.0), r.span_us);
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("# %s, %d CUs, maxThreadsPerMultiProcessor %d, sharedMemPerMultiprocessor %zu, maxBlocksPerMultiProcessor %d\n", p.gcnArchName,
           p.multiProcessorCount, p.maxThreadsPerMultiProcessor, p.sharedMemPerMultiprocessor, p.maxBlocksPerMultiProcessor);
    const int max_waves = 256 * 64 * 4;
    WaveRec *d_out;
    float *d_sink;
    CHECK(hipMalloc(&d_out, sizeof(WaveRec) * max_waves));
    CHECK(hipMalloc(&d_sink, 4));
    // grid: 64 waves' worth per CU, i.e. two full rounds at 32 waves per CU
    const int lds_list[] = {64, 1024, 2048, 2560, 3072, 3584, 4096, 4608, 5120, 6144, 8192, 10240, 14960, 16384, 20480};
    printf("## single-wave workgroups, 32 VGPRs, no barrier: LDS sweep\n");
    for (int lds : lds_list) run("wg64", census64<32, 32, false>, 64, 32, 32, lds, d_out, d_sink, 256 * 64);
    printf("## single-wave workgroups, 64 VGPRs (blend_bwd's allocation), with s_barrier\n");
    for (int lds : {64, 4096, 4608, 5120, 8192}) run("wg64+bar", census64<64, 48, true>, 64, 64, 48, lds, d_out, d_sink, 256 * 64);
    printf("## two / four waves per workgroup, 64 VGPRs, barrier: the same LDS per WAVE\n");
    for (int lds : {64, 2048, 4096, 4608, 5120, 8192}) run("wg128+bar", census128<64, 48, true>, 128, 64, 48, 2 * lds, d_out, d_sink, 256 * 32);
    for (int lds : {64, 2048, 3740, 4096, 4608, 5120, 8192}) run("wg256+bar", census256<64, 48, true>, 256, 64, 48, 4 * lds, d_out, d_sink, 256 * 16);
    printf("## register steps (single-wave workgroups, 64 B of LDS)\n");
    run("wg64", census64<56, 32, false>, 64, 56, 32, 64, d_out, d_sink, 256 * 64);
    run("wg64", census64<64, 32, false>, 64, 64, 32, 64, d_out, d_sink, 256 * 64);
    run("wg64", census64<72, 32, false>, 64, 72, 32, 64, d_out, d_sink, 256 * 64);
    run("wg64", census64<96, 32, false>, 64, 96, 32, 64, d_out, d_sink, 256 * 64);
    run("wg64", census64<128, 32, false>, 64, 128, 32, 64, d_out, d_sink, 256 * 64);
    run("wg64", census64<32, 80, false>, 64, 32, 80, 64, d_out, d_sink, 256 * 64);
    run("wg64", census64<32, 96, false>, 64, 32, 96, 64, d_out, d_sink, 256 * 64);
    run("wg256", census256<64, 80, true>, 256, 64, 80, 64, d_out, d_sink, 256 * 16);
    run("wg256", census256<64, 96, true>, 256, 64, 96, 64, d_out, d_sink, 256 * 16);
    return 0;
}
