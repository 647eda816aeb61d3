// What does a hand-written streaming kernel reach on this box?  (The guide quotes 6.29 TB/s for a float4 copy; torch's copy_ gives
// 4.5-5.1 TB/s on this pool, profiles/r03_j_stream_bandwidth_torch_copy.txt.)  float4 copy and a 1-read / 1.3-write mix shaped like
// geom_backward_kernel's traffic (180 B in, 236 B out per element), grid-stride, several grid sizes and loads in flight per thread,
// also with non-temporal loads / stores.   hipcc --offload-arch=gfx950 -O3 tools/copy_bw.hip -o tools/copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t k = i + (size_t)u * 256;
            if (k < n) { if (NT) { typedef float f4 __attribute__((ext_vector_type(4))); const f4 q = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(in + k)); v[u] = make_float4(q.x, q.y, q.z, q.w); } else v[u] = in[k]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t k = i + (size_t)u * 256;
            if (k < n) { if (NT) { typedef float f4 __attribute__((ext_vector_type(4))); f4 q = {v[u].x, v[u].y, v[u].z, v[u].w}; __builtin_nontemporal_store(q, reinterpret_cast<f4 *>(out + k)); } else out[k] = v[u]; }
        }
    }
}
template <int U>
__global__ __launch_bounds__(256) void read_kernel(const float4 *__restrict__ in, float *__restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256 * U;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t k = i + (size_t)u * 256;
            if (k < n) { const float4 v = in[k]; acc += v.x + v.y + v.z + v.w; }
        }
    }
    if (acc == 123.456f) out[0] = acc;
}
template <class F> static float time_ms(F f, int reps = 20)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}
int main()
{
    const size_t bytes = (size_t)416 << 20, n = bytes / 16;
    float4 *in, *out; float *sink;
    hipMalloc(&in, bytes); hipMalloc(&out, bytes); hipMalloc(&sink, 64);
    hipMemset(in, 1, bytes); hipMemset(out, 0, bytes);
    printf("buffers of %zu MB (each far beyond the 256 MB last-level cache together)\n", bytes >> 20);
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        const float t1 = time_ms([&] { hipLaunchKernelGGL((copy_kernel<1, false>), dim3(blocks), dim3(256), 0, 0, in, out, n); });
        const float t4 = time_ms([&] { hipLaunchKernelGGL((copy_kernel<4, false>), dim3(blocks), dim3(256), 0, 0, in, out, n); });
        const float t8 = time_ms([&] { hipLaunchKernelGGL((copy_kernel<8, false>), dim3(blocks), dim3(256), 0, 0, in, out, n); });
        const float n4 = time_ms([&] { hipLaunchKernelGGL((copy_kernel<4, true>), dim3(blocks), dim3(256), 0, 0, in, out, n); });
        const float r4 = time_ms([&] { hipLaunchKernelGGL((read_kernel<4>), dim3(blocks), dim3(256), 0, 0, in, sink, n); });
        printf("%6d workgroups: copy 1 / 4 / 8 float4 in flight per thread %.2f / %.2f / %.2f TB/s, non-temporal (4) %.2f TB/s, read-only (4) %.2f TB/s\n", blocks,
               2.0 * bytes / t1 / 1e9, 2.0 * bytes / t4 / 1e9, 2.0 * bytes / t8 / 1e9, 2.0 * bytes / n4 / 1e9, 1.0 * bytes / r4 / 1e9);
    }
    return 0;
}
