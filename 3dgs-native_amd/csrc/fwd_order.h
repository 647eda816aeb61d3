// fwd_order.h -- the forward blend's tile dispatch order (gsr_internal.h "forward tile order"), made once per frame by a spare
// workgroup of preprocess_kernel.
#pragma once
#include "gsr_internal.h"

// The forward blend's tile order for THIS frame from the costs its waves left LAST frame (gsr_internal.h "forward tile order"), by
// one 256-thread workgroup: the spare one at the end of preprocess_kernel's grid (a kernel of 145 VGPRs and 90 us: the order's 50
// registers and 4 us cost it nothing; at the end of the expansion's grid they took that kernel from 8 to 5 workgroups per CU).
// cost of a tile = the largest of its four waves' costs (a wave's life in ticks, in the `walked` field with `staged` = 0 -- or, with
// GSR_FWD_COST_LIFE=0, entries walked + entries staged / 2); 64 classes on a scale set by the frame's
// largest cost, heaviest class first; inside a class the tiles keep their order (wave by wave).  Any content of fwd_cost -- a
// fresh workspace's garbage included -- yields a permutation of the tiles.
// `lds`: GSR_FO_LDS_INTS ints of the caller's shared memory (the caller's own arrays, idle in this workgroup: no LDS of its own, so
// the host kernel's occupancy is untouched).
#define GSR_FO_LDS_INTS (5 * GSR_FO_CLASSES + 4)
__device__ __forceinline__ void fwd_order_block(const int32_t *__restrict__ fwd_cost, int32_t *__restrict__ fwd_order, int n_tiles, int *lds)
{
    int (*s_cnt)[GSR_FO_CLASSES] = reinterpret_cast<int (*)[GSR_FO_CLASSES]>(lds);
    int *s_base = lds + 4 * GSR_FO_CLASSES, *s_max = lds + 5 * GSR_FO_CLASSES;
    const int tid = threadIdx.x, w = tid >> 6;
    auto cost_of = [&](int t) {
        const int4 c = reinterpret_cast<const int4 *>(fwd_cost)[t];
        auto one = [](int v) { const int walked = (v >> 16) & 0x7FFF, staged = v & 0xFFFF; return walked + (staged >> 1); };
        return max(max(one(c.x), one(c.y)), max(one(c.z), one(c.w)));
    };
    for (int c = tid & 63; c < GSR_FO_CLASSES; c += 64) s_cnt[w][c] = 0;
    constexpr int PER = GSR_FO_MAX_TILES / 256; // tiles per thread, kept in registers: tile = k * 256 + tid
    int cost[PER], mx = 1;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int t = k * 256 + tid;
        cost[k] = t < n_tiles ? cost_of(t) : -1;
        mx = max(mx, cost[k]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, __shfl_xor(mx, d, 64));
    if ((tid & 63) == 0) s_max[w] = mx;
    __syncthreads();
    mx = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    const float scale = (float)(GSR_FO_CLASSES - 1) / (float)mx;
    int cls[PER], rank[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        cls[k] = (GSR_FO_CLASSES - 1) - min(GSR_FO_CLASSES - 1, (int)((float)max(cost[k], 0) * scale)); // 0 = heaviest
        rank[k] = cost[k] >= 0 ? atomicAdd(&s_cnt[w][cls[k]], 1) : 0; // position among this wave's tiles of the class (k ascending)
    }
    __syncthreads();
    static_assert(GSR_FO_CLASSES <= 256 && GSR_FO_CLASSES % 64 == 0, "one thread per class, whole waves");
    int run = 0, incl = 0;
    if (tid < GSR_FO_CLASSES) { // per class: the four waves' counts -> their offsets inside the class; then the classes' first slots
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int c = s_cnt[q][tid]; s_cnt[q][tid] = run; run += c; }
        incl = run;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d, 64);
            if ((tid & 63) >= d) incl += up;
        }
        if ((tid & 63) == 63) s_max[w] = incl; // (s_max is free again: every thread has read it)
    }
    __syncthreads();
    if (tid < GSR_FO_CLASSES) {
        int before = 0;
        for (int q = 0; q < w; ++q) before += s_max[q];
        s_base[tid] = before + incl - run;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (cost[k] >= 0) fwd_order[s_base[cls[k]] + s_cnt[w][cls[k]] + rank[k]] = k * 256 + tid;
}

