// Does s_barrier on gfx950 wait only for the SURVIVING waves of a workgroup once some have terminated (as the GCN / CDNA ISA
// documents say of S_BARRIER)?  Wave `quitter` of each 256-thread workgroup returns before the loop; the other three go through
// `rounds` barriers, handing a token round-robin through LDS, and report.  If the barrier waited for the dead wave the kernel would
// hang: run under `timeout -k 5 20`.   hipcc --offload-arch=gfx950 -O2 tools/barrier_exit_probe.hip -o tools/barrier_exit_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void probe(int *out, int rounds, int quitter, int late_quit_round)
{
    __shared__ int s_token, s_alive;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) { s_token = 0; s_alive = 0xF; }
    __syncthreads();
    if (wv == quitter && late_quit_round < 0) { if (lane == 0) atomicAnd(&s_alive, ~(1 << wv)); return; }
    int seen = 0;
    for (int r = 0; r < rounds; ++r) {
        if (wv == quitter && r == late_quit_round) { if (lane == 0) atomicAnd(&s_alive, ~(1 << wv)); return; } // leaves mid-way
        __syncthreads();
        const int alive = s_alive;
        seen |= alive;
        if (lane == 0 && wv == (r % 3 == 0 ? (quitter + 1) & 3 : (quitter + 2) & 3)) atomicAdd(&s_token, 1);
        __syncthreads();
    }
    if (lane == 0) out[blockIdx.x * 4 + wv] = s_token * 256 + (seen & 0xFF) * 16 + s_alive;
}
int main()
{
    int *d; hipMalloc(&d, 4096 * 4 * sizeof(int)); hipMemset(d, 0xFF, 4096 * 4 * sizeof(int));
    for (int late : {-1, 5}) {
        hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, 0, d, 64, 2, late);
        hipError_t e = hipDeviceSynchronize();
        int h[16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("quitter leaves %s: sync=%d  block0: %x %x %x %x  block1: %x %x %x %x\n", late < 0 ? "before the loop" : "in round 5", (int)e, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    }
    return 0;
}
