// depth_order.hip -- the Gaussians in (depth bits, id) order by SAMPLE SORT: four launches instead of the twelve of the
// four-pass LSD radix sort (scan_sort.hip), for the same result.
//
// Why: at C3 the LSD sort orders 8 MB in 92 us -- twelve launch-bound kernels of 5-9 us each; the bytes are nothing.  The
// items are unique 64-bit keys (depth bits << 32 | id; the reference's order is "stable by depth", i.e. exactly the order of
// these keys, quirk Q13), so any comparison sort gives the same permutation, and sample sort needs no digit passes:
//   S1  splitters   one workgroup takes up to 4094 keys at jittered, evenly spaced positions, sorts them (bitonic in LDS) and
//                   keeps every second one: they cut the key space into NB <= 2048 buckets of about N / NB items WHATEVER
//                   the depth distribution is -- keys are unique, so even a million Gaussians at one depth split evenly (by
//                   id).  A bucket is the sum of two sample spacings (Gamma(2)): the largest of 2048 is about 5.5x the mean
//   S2  count       every item finds its bucket (binary search over the splitters in LDS), bucket ids are kept (uint16),
//                   per-block LDS histograms are added to the global bucket totals
//   S3  scatter     items are moved to their bucket's range, in any order inside it (one global cursor per bucket, one
//                   returning atomic per (block, bucket), LDS atomics for the rank inside the block)
//   S4  bucket sort one workgroup per bucket: bitonic sort of its items in LDS, written to their final places together with
//                   each Gaussian's tile rectangle and tile count (what the LSD sort's last pass carried)
// A bucket that does not fit LDS (DO_CAP = 8192 items = 16.8x the mean at N = 1 M: probability ~ 2047 e^-33.6 33.6 = 2e-10 per
// frame for ids uncorrelated with depth; the jitter is there for ids that ARE periodic in depth) raises a flag; the host then
// redoes the order with the LSD sort (api.hip).  Nothing here is approximate: the output is the sorted order.
#include "gsr_internal.h"

namespace {

constexpr int DO_MAX_SPLIT = 2047; // splitters (NB = splitters + 1 buckets)
constexpr int DO_CHUNK = 4096;     // items per block in count / scatter
constexpr int DO_CAP = 8192;       // items a bucket may hold for the LDS sort (64 KB)
constexpr int DO_OVERSAMPLE = 2;   // samples per splitter

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// bitonic sort of s[0..P) (P a power of two) by all threads of the block; every thread calls it
__device__ __forceinline__ void bitonic_sort_lds(unsigned long long *s, int P, int tid, int nthreads)
{
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int t = tid; t < (P >> 1); t += nthreads) {
                // t-th compare-exchange of this step: partners i and i + j with i's bit j clear
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const unsigned long long a = s[i], b = s[i + j];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { s[i] = b; s[i + j] = a; }
            }
        }
    }
    __syncthreads();
}

// S1: one workgroup.  Also clears the bucket totals, the cursors and the overflow flag.
__global__ __launch_bounds__(1024) void do_splitters_kernel(const uint64_t *__restrict__ items, int64_t n, int ns, uint64_t *__restrict__ split,
                                                            int32_t *__restrict__ totals, int32_t *__restrict__ cursor, int32_t *__restrict__ flag)
{
    __shared__ unsigned long long s[4096];
    const int tid = threadIdx.x;
    const int nsamp = ns * DO_OVERSAMPLE; // <= 4094
    for (int j = tid; j < 4096; j += 1024) {
        unsigned long long v = ~0ull;
        if (j < nsamp) {
            // sample j: one item of the j-th of nsamp equal stretches, at a hashed offset inside it (regular sampling keeps
            // the buckets even; the jitter keeps a scene whose ids are periodic in depth from aliasing with the stride)
            const int64_t lo = (int64_t)j * n / nsamp, hi = (int64_t)(j + 1) * n / nsamp;
            const int64_t len = hi - lo > 0 ? hi - lo : 1;
            v = items[min(n - 1, lo + (int64_t)(hash32((uint32_t)j * 2654435761u + (uint32_t)n) % (uint64_t)len))];
        }
        s[j] = v;
    }
    int P = 64;
    while (P < nsamp) P <<= 1;
    bitonic_sort_lds(s, P, tid, 1024);
    for (int j = tid; j < ns; j += 1024) split[j] = s[DO_OVERSAMPLE * j + DO_OVERSAMPLE - 1];
    for (int j = tid; j <= DO_MAX_SPLIT + 1; j += 1024) { totals[j] = 0; cursor[j] = 0; }
    if (tid == 0) *flag = 0;
}

// bucket of key x = number of splitters < x  (in [0, ns])
__device__ __forceinline__ int bucket_of(const unsigned long long *sp, int ns, unsigned long long x)
{
    int lo = 0;
#pragma unroll
    for (int step = 1024; step >= 1; step >>= 1) {
        const int p = lo + step - 1;
        if (p < ns && sp[p] < x) lo += step;
    }
    return lo;
}

// S2
__global__ __launch_bounds__(256) void do_count_kernel(const uint64_t *__restrict__ items, int64_t n, const uint64_t *__restrict__ split, int ns,
                                                       uint16_t *__restrict__ bucket_id, int32_t *__restrict__ totals)
{
    __shared__ unsigned long long sp[DO_MAX_SPLIT + 1];
    __shared__ int h[DO_MAX_SPLIT + 2];
    const int tid = threadIdx.x;
    for (int j = tid; j < ns; j += 256) sp[j] = split[j];
    for (int j = tid; j <= ns; j += 256) h[j] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * DO_CHUNK;
    unsigned long long key[DO_CHUNK / 256];
#pragma unroll
    for (int r = 0; r < DO_CHUNK / 256; ++r) {
        const int64_t i = base + r * 256 + tid;
        key[r] = i < n ? items[i] : ~0ull;
    }
#pragma unroll
    for (int r = 0; r < DO_CHUNK / 256; ++r) {
        const int64_t i = base + r * 256 + tid;
        if (i < n) {
            const int b = bucket_of(sp, ns, key[r]);
            bucket_id[i] = (uint16_t)b;
            atomicAdd(&h[b], 1);
        }
    }
    __syncthreads();
    for (int j = tid; j <= ns; j += 256)
        if (h[j]) atomicAdd(&totals[j], h[j]);
}

// exclusive scan of v[0..m) (m <= 2048) into out[], by a 256-thread block: 8 consecutive values per thread
__device__ __forceinline__ void block_excl_scan_2048(const int32_t *__restrict__ v, int m, int *out, int *tmp /* [4] */, int tid)
{
    int loc[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int j = tid * 8 + k;
        loc[k] = j < m ? v[j] : 0;
        sum += loc[k];
    }
    // inclusive scan of `sum` over the block
    const int lane = tid & 63, w = tid >> 6;
    int inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(inc, d, 64);
        if (lane >= d) inc += u;
    }
    if (lane == 63) tmp[w] = inc;
    __syncthreads();
    int add = 0;
    for (int k = 0; k < w; ++k) add += tmp[k];
    int run = add + inc - sum;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int j = tid * 8 + k;
        if (j < m) out[j] = run;
        run += loc[k];
    }
    __syncthreads();
}

// S3
__global__ __launch_bounds__(256) void do_scatter_kernel(const uint64_t *__restrict__ items, int64_t n, const uint16_t *__restrict__ bucket_id,
                                                         int ns, const int32_t *__restrict__ totals, int32_t *__restrict__ cursor,
                                                         uint64_t *__restrict__ out)
{
    __shared__ int bases[DO_MAX_SPLIT + 2];  // first position of each bucket
    __shared__ int cnt[DO_MAX_SPLIT + 2];    // this block's items per bucket, then the running rank inside the block
    __shared__ int start[DO_MAX_SPLIT + 2];  // where this block's items of the bucket start inside the bucket
    __shared__ int tmp[4];
    const int tid = threadIdx.x, nb = ns + 1;
    block_excl_scan_2048(totals, nb, bases, tmp, tid);
    for (int j = tid; j < nb; j += 256) cnt[j] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * DO_CHUNK;
    unsigned long long key[DO_CHUNK / 256];
    int bid[DO_CHUNK / 256];
#pragma unroll
    for (int r = 0; r < DO_CHUNK / 256; ++r) {
        const int64_t i = base + r * 256 + tid;
        key[r] = i < n ? items[i] : 0ull;
        bid[r] = i < n ? (int)bucket_id[i] : -1;
    }
#pragma unroll
    for (int r = 0; r < DO_CHUNK / 256; ++r)
        if (bid[r] >= 0) atomicAdd(&cnt[bid[r]], 1);
    __syncthreads();
    for (int j = tid; j < nb; j += 256) {
        const int c = cnt[j];
        start[j] = c ? atomicAdd(&cursor[j], c) : 0;
        cnt[j] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < DO_CHUNK / 256; ++r)
        if (bid[r] >= 0) {
            const int b = bid[r];
            const int rank = atomicAdd(&cnt[b], 1);
            out[(int64_t)bases[b] + start[b] + rank] = key[r];
        }
}

// S4: one workgroup per bucket (1024 threads: the few buckets of several thousand items set the kernel's duration)
constexpr int S4_THREADS = 1024;
__global__ __launch_bounds__(S4_THREADS) void do_bucket_sort_kernel(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, int ns,
                                                             const int32_t *__restrict__ totals, const TileRect *__restrict__ rect,
                                                             TileRect *__restrict__ rect_sorted, int32_t *__restrict__ cnt_sorted,
                                                             int32_t *__restrict__ flag, int cap)
{
    __shared__ unsigned long long s[DO_CAP];
    __shared__ int s_red[S4_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.x;
    const int nb_items = totals[b];
    if (nb_items == 0) return;
    // first position of the bucket = sum of the totals before it
    int part = 0;
    for (int j = tid; j < b; j += S4_THREADS) part += totals[j];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o, 64);
    if (lane == 0) s_red[w] = part;
    __syncthreads();
    int64_t base = 0;
#pragma unroll
    for (int k = 0; k < S4_THREADS / 64; ++k) base += s_red[k];
    if (nb_items > cap) { // does not fit: the host redoes the order with the LSD sort (see the file header)
        if (tid == 0) atomicOr(flag, 1);
        return;
    }
    int P = 64;
    while (P < nb_items) P <<= 1;
    for (int j = tid; j < P; j += S4_THREADS) s[j] = j < nb_items ? in[base + j] : ~0ull;
    bitonic_sort_lds(s, P, tid, S4_THREADS);
    for (int j = tid; j < nb_items; j += S4_THREADS) {
        const unsigned long long k = s[j];
        out[base + j] = k;
        // carry: the Gaussian's tile rectangle and tile count to its sorted position (as the LSD sort's last pass does)
        const unsigned long long q = reinterpret_cast<const unsigned long long *>(rect)[(uint32_t)k];
        reinterpret_cast<unsigned long long *>(rect_sorted)[base + j] = q;
        const int x0 = (int)(q & 0xFFFF), y0 = (int)((q >> 16) & 0xFFFF), x1 = (int)((q >> 32) & 0xFFFF), y1 = (int)(q >> 48);
        cnt_sorted[base + j] = (x1 - x0) * (y1 - y0);
    }
}

// the unsorted items again, as preprocess wrote them (depth bits << 32 | id, 0xFFFFFFFF depth for culled Gaussians): the LSD
// sort is stable by depth FROM ID ORDER, and the sample sort's scatter has permuted its input
__global__ __launch_bounds__(256) void do_rebuild_items_kernel(const float *__restrict__ depths, const int32_t *__restrict__ tiles_touched, int64_t n,
                                                               uint64_t *__restrict__ items)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t dbits = tiles_touched[i] != 0 ? __float_as_uint(depths[i]) : 0xFFFFFFFFu;
    items[i] = ((uint64_t)dbits << 32) | (uint64_t)(uint32_t)i;
}

} // namespace

hipError_t gsr_launch_rebuild_depth_items(const GeomWs &ws, const GsrGeom &g, int64_t N, hipStream_t s)
{
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(do_rebuild_items_kernel, dim3((unsigned)gsr_div_up(N, 256)), dim3(256), 0, s, g.depths, g.tiles_touched, N, ws.depth_item);
    return hipGetLastError();
}

// scratch inside the geom workspace: split [2048] u64, totals [2049] i32, cursor [2049] i32, flag [1] i32, bucket_id [N] u16
hipError_t gsr_launch_depth_order(const GeomWs &ws, int64_t N, hipStream_t s)
{
    if (N <= 0) return hipSuccess;
    int ns = (int)(N / 512);
    if (ns < 1) ns = 1;
    if (ns > DO_MAX_SPLIT) ns = DO_MAX_SPLIT;
    const int cap = (gsr_debug_flags & 256) ? 16 : DO_CAP; // GSR_DEBUG bit 8: make nearly every bucket overflow (tests of the redo path)
    const unsigned blocks = (unsigned)gsr_div_up(N, DO_CHUNK);
    hipLaunchKernelGGL(do_splitters_kernel, dim3(1), dim3(1024), 0, s, ws.depth_item, N, ns, ws.do_split, ws.do_totals, ws.do_cursor, ws.do_flag);
    hipLaunchKernelGGL(do_count_kernel, dim3(blocks), dim3(256), 0, s, ws.depth_item, N, ws.do_split, ns, ws.do_bucket, ws.do_totals);
    hipLaunchKernelGGL(do_scatter_kernel, dim3(blocks), dim3(256), 0, s, ws.depth_item, N, ws.do_bucket, ns, ws.do_totals, ws.do_cursor, ws.sort_tmp);
    hipLaunchKernelGGL(do_bucket_sort_kernel, dim3(ns + 1), dim3(S4_THREADS), 0, s, ws.sort_tmp, ws.depth_item, ns, ws.do_totals, ws.rect, ws.rect_sorted,
                       ws.cnt_sorted, ws.do_flag, cap);
    return hipGetLastError();
}
