// tile_bin.hip -- one-pass stable partition of the D (tile, Gaussian) pairs by tile id, fused with their expansion.
//
// The reference emits one int64 key per (Gaussian, tile) pair and radix-sorts all D of them on 64 bits (forward.py:518-558,
// :791-824); round 1 of this library expanded the depth-sorted Gaussians to (tile, id) items and stable-partitioned those by
// tile id with two radix passes (scan_sort.hip: expand, 2 x (histogram, row scan, scatter), ranges: eight launches that move
// the D items through HBM five times).  Both only need, per pair, its position in the output:
//     position = start of the tile's run + number of pairs of the same tile that precede it in (depth, id) order.
// With the Gaussians already in depth order (gsr_forward_count) the pairs can be visited in exactly that order, so the
// partition is a counting sort with ONE digit -- the whole tile id -- and the pair itself never has to exist in memory:
//   K1  bin_count    a block walks its chunk of S consecutive pairs (expanded on the fly from the depth-sorted rectangles) and
//                    counts them per tile in LDS -> hist[chunk][tile] (uint16)
//   K2  bin_colsum   sums hist over segments of 16 chunks -> segsum[seg][tile]
//   K3  bin_tilescan one block: per-tile totals, exclusive scan over tiles = start of every tile's run -> `ranges` (the
//                    reference's wp_identify_tile_ranges, forward.py:561-586, falls out of the scan) and segbase[seg][tile]
//   K4  bin_base     base[chunk][tile] = segbase + the counts of the chunks before it in its segment: where each chunk's pairs
//                    of each tile start
//   K5  bin_scatter  a block walks its chunk again and writes every pair's Gaussian id straight to point_list[position]:
//                    within a wave the rank among equal tiles comes from wave64 ballots (match-any over the tile-id bits, no
//                    per-item atomics), across the block's four waves from per-wave counts taken in a first sweep.
// Five launches (three of them a few microseconds of work); the D pairs are written once (4 bytes each) and never read back.  The order is the reference's (tile, depth
// bits, id): chunks, waves, 64-pair groups and lanes are all visited in expansion order, which is depth order, with ties in id
// order because the depth sort is stable (quirk Q13).
//
// Chunks are ranges of PAIRS, not of Gaussians, so one screen-filling splat cannot overload a block; a wave finds the
// Gaussian its first pair belongs to in `kidx`, an index the depth-order scan leaves behind (one entry per 1024 pairs).
#include "gsr_internal.h"

namespace {

constexpr int WIN = 64; // Gaussians per window (one per lane)
constexpr int NW = 8;   // waves per block: a wave walks S / NW consecutive pairs
constexpr int NT = NW * 64;

struct Window {
    int off[NW][WIN + 1];     // pair offset of each Gaussian of the wave's current window, [WIN] = offset behind the window
    TileRect rect[NW][WIN];
    uint32_t gid[NW][WIN];
    float inv[NW][WIN];       // 1 / (rectangle width in tiles)
};

// Consecutive chunks write consecutive pieces of every tile's run, so they share cache lines of point_list.  Workgroups are
// dealt round-robin over the 8 XCDs (whose L2s are not coherent with each other): with chunk = blockIdx the eight L2s would
// each hold a few bytes of every line and write them back separately.  This map gives every XCD one contiguous eighth of the
// chunks, taken in order, so a line fills up inside one L2 before it leaves (placement is for speed only: any map is correct).
__device__ __forceinline__ int chunk_of_block(int b, int nchunks)
{
    const int per = (nchunks + 7) >> 3;
    const int c = (b & 7) * per + (b >> 3);
    return c; // may be >= nchunks for the last XCD's tail: the caller skips those
}

// Visit the pairs [j0, j1) in order, 64 at a time: f(pair index, tile id, Gaussian id, active) is called with all 64 lanes
// (inactive lanes carry tile 0), so f may use wave-wide ballots.
template <class F>
__device__ __forceinline__ void walk_pairs(int j0, int j1, const int32_t *__restrict__ kidx, const uint64_t *__restrict__ sorted,
                                           const int32_t *__restrict__ doff, const TileRect *__restrict__ rect, int n, int D, int grid_x,
                                           Window &win, int w, int lane, F f)
{
    if (j0 >= j1) return;
    int kw = kidx[j0 >> 10]; // the Gaussian that holds pair j0 (j0 is a multiple of 1024)
    int j = j0;
    // the next window's rows are fetched while the current one is walked
    int k = kw + lane;
    int n_off = k < n ? doff[k] : D;
    TileRect n_rc = k < n ? rect[k] : TileRect{0, 0, 0, 0};
    uint32_t n_id = k < n ? (uint32_t)sorted[k] : 0u;
    int n_end = kw + WIN < n ? doff[kw + WIN] : D;
    while (j < j1 && kw < n) {
        win.off[w][lane] = n_off;
        win.rect[w][lane] = n_rc;
        win.gid[w][lane] = n_id;
        win.inv[w][lane] = n_rc.x1 > n_rc.x0 ? 1.0f / (float)((int)n_rc.x1 - (int)n_rc.x0) : 0.0f;
        const int wend = min(j1, n_end);
        if (lane == 0) win.off[w][WIN] = n_end;
        kw += WIN;
        k = kw + lane;
        n_off = k < n ? doff[k] : D;
        n_rc = k < n ? rect[k] : TileRect{0, 0, 0, 0};
        n_id = k < n ? (uint32_t)sorted[k] : 0u;
        n_end = kw + WIN < n ? doff[kw + WIN] : D;
        for (int jb = j; jb < wend; jb += 64) {
            const int jj = jb + lane;
            const bool active = jj < wend;
            const int q = active ? jj : jb;
            int lo = 0; // last Gaussian of the window with off <= q (zero-count Gaussians share their successor's offset)
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1)
                if (win.off[w][lo + step] <= q) lo += step;
            const TileRect r = win.rect[w][lo];
            const int t = q - win.off[w][lo];
            const int wd = (int)r.x1 - (int)r.x0;
            // row-major walk of the rectangle: y = t / wd, x = t % wd (reference forward.py:546-548); t < 2^24, so the quotient
            // comes from one float multiply by 1/wd and a +-1 correction
            int y = (int)((float)t * win.inv[w][lo]);
            int x = t - y * wd;
            if (x < 0) { --y; x += wd; }
            else if (x >= wd) { ++y; x -= wd; }
            const int tile = active ? ((int)r.y0 + y) * grid_x + (int)r.x0 + x : 0;
            f(jj, tile, win.gid[w][lo], active);
        }
        j = wend;
    }
}

// K1: per-chunk tile histogram
__global__ __launch_bounds__(NT) void bin_count_kernel(const int32_t *__restrict__ kidx, const uint64_t *__restrict__ sorted,
                                                        const int32_t *__restrict__ doff, const TileRect *__restrict__ rect, int n, int D,
                                                        int grid_x, int tiles, int S, uint16_t *__restrict__ hist)
{
    extern __shared__ int s_dyn[];
    __shared__ Window win;
    int *cnt = s_dyn; // [tiles]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < tiles; t += NT) cnt[t] = 0;
    __syncthreads();
    const int c0 = blockIdx.x * S, q4 = S / NW;
    const int j0 = min(D, c0 + w * q4), j1 = min(D, c0 + (w + 1) * q4);
    walk_pairs(j0, j1, kidx, sorted, doff, rect, n, D, grid_x, win, w, lane, [&](int, int tile, uint32_t, bool active) {
        if (active) atomicAdd(&cnt[tile], 1);
    });
    __syncthreads();
    uint16_t *row = hist + (size_t)blockIdx.x * tiles;
    for (int t = threadIdx.x; t < tiles; t += NT) row[t] = (uint16_t)cnt[t];
}

// K2: along the chunks.  One 1024-thread block per 64 tiles: wave v sums its sixteenth of the chunks for the 64 tiles (lane =
// tile), the sixteen sums are prefixed through LDS, and every wave then rewrites its chunks as exclusive running counts:
// rel[c][t] = number of tile t's pairs in chunks before c.  Also tot[t], the tile's total.
constexpr int K2_WAVES = 16;
constexpr int K2_MAXPER = 64; // chunks per wave held in registers (nchunks <= 1024)
__global__ __launch_bounds__(1024) void bin_prefix_kernel(const uint16_t *__restrict__ hist, int nchunks, int tiles, int32_t *__restrict__ rel,
                                                          int32_t *__restrict__ tot)
{
    __shared__ int s_part[K2_WAVES][64];
    const int lane = threadIdx.x & 63, v = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    const bool ok = t < tiles;
    const int per = (nchunks + K2_WAVES - 1) / K2_WAVES;
    const int c0 = v * per, c1 = min(nchunks, c0 + per);
    int val[K2_MAXPER];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < K2_MAXPER; ++k) {
        const int c = c0 + k;
        val[k] = (ok && c < c1) ? (int)hist[(size_t)c * tiles + t] : 0;
    }
#pragma unroll
    for (int k = 0; k < K2_MAXPER; ++k) sum += val[k];
    s_part[v][lane] = sum;
    __syncthreads();
    int run = 0, total = 0;
#pragma unroll
    for (int k = 0; k < K2_WAVES; ++k) {
        const int pk = s_part[k][lane];
        if (k < v) run += pk;
        total += pk;
    }
    if (v == 0 && ok) tot[t] = total;
#pragma unroll
    for (int k = 0; k < K2_MAXPER; ++k) {
        const int c = c0 + k;
        if (ok && c < c1) rel[(size_t)c * tiles + t] = run;
        run += val[k];
    }
}

// K3: one block: exclusive scan of the tile totals = start of every tile's run -> tile_start, `ranges`
__global__ __launch_bounds__(1024) void bin_tilescan_kernel(const int32_t *__restrict__ tot, int tiles, int32_t *__restrict__ tile_start,
                                                            int32_t *__restrict__ ranges)
{
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int t0 = 0; t0 < tiles; t0 += 1024) {
        const int t = t0 + tid;
        const int v = t < tiles ? tot[t] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int u = __shfl_up(inc, d, 64);
            if (lane >= d) inc += u;
        }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        int add = s_carry;
        for (int k = 0; k < wv; ++k) add += s_w[k];
        const int start = add + inc - v;
        if (t < tiles) {
            tile_start[t] = start;
            ranges[2 * t] = v > 0 ? start : 0;        // untouched tiles read (0, 0) (reference forward.py:561-586)
            ranges[2 * t + 1] = v > 0 ? start + v : 0;
        }
        __syncthreads();
        if (tid == 1023) s_carry = add + inc;
        __syncthreads();
    }
}

// K5: ranks and writes
template <int TILE_BITS>
__global__ __launch_bounds__(NT) void bin_scatter_kernel(const int32_t *__restrict__ kidx, const uint64_t *__restrict__ sorted,
                                                          const int32_t *__restrict__ doff, const TileRect *__restrict__ rect, int n, int D,
                                                          int grid_x, int tiles, int S, int nchunks, const int32_t *__restrict__ rel,
                                                          const int32_t *__restrict__ tile_start, int32_t *__restrict__ point_list)
{
    extern __shared__ int s_dyn[];
    __shared__ Window win;
    const int tp = (tiles + 1) & ~1;                             // row stride: even, so every wave's row starts on a word
    int *base = s_dyn;                                           // [tiles] absolute position of the block's first pair of each tile
    uint16_t *run = reinterpret_cast<uint16_t *>(s_dyn + tp);    // [NW][tp] per wave: first its count, then its running offset
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int chunk = chunk_of_block(blockIdx.x, nchunks);
    if (chunk >= nchunks) return;
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int t = threadIdx.x; t < (NW / 2) * tp; t += NT) s_dyn[tp + t] = 0;
    for (int t = threadIdx.x; t < tiles; t += NT) base[t] = tile_start[t] + rel[(size_t)chunk * tiles + t];
    __syncthreads();
    const int c0 = chunk * S, q4 = S / NW;
    const int j0 = min(D, c0 + w * q4), j1 = min(D, c0 + (w + 1) * q4);
    uint16_t *my = run + (size_t)w * tp;
    // sweep 1: this wave's pairs per tile (16-bit counters packed two to a word: LDS atomics are 32-bit; a wave holds at most
    // S / NW < 65536 pairs, so a half never carries into its neighbour)
    walk_pairs(j0, j1, kidx, sorted, doff, rect, n, D, grid_x, win, w, lane, [&](int, int tile, uint32_t, bool active) {
        if (active) atomicAdd(reinterpret_cast<int *>(my) + (tile >> 1), (tile & 1) ? 0x10000 : 1);
    });
    __syncthreads();
    // counts -> exclusive prefix over the block's waves
    for (int t = threadIdx.x; t < tiles; t += NT) {
        int acc = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int v = run[(size_t)k * tp + t];
            run[(size_t)k * tp + t] = (uint16_t)acc;
            acc += v;
        }
    }
    __syncthreads();
    // sweep 2: rank inside the wave by ballots, in pair order
    walk_pairs(j0, j1, kidx, sorted, doff, rect, n, D, grid_x, win, w, lane, [&](int, int tile, uint32_t gid, bool active) {
        // lanes holding the same tile ("match any"): a lane differs from me in bit b where ballot(bit b) XOR (my bit b
        // replicated) is set; OR over the bits, complement
        unsigned int diff_lo = 0u, diff_hi = 0u;
#pragma unroll
        for (int b = 0; b < TILE_BITS; ++b) {
            const unsigned int rep = (unsigned int)__builtin_amdgcn_sbfe(tile, b, 1); // 0 or 0xFFFFFFFF
            const unsigned long long m = __ballot(rep != 0u);
            diff_lo |= (unsigned int)m ^ rep;
            diff_hi |= (unsigned int)(m >> 32) ^ rep;
        }
        unsigned long long peers = ~(((unsigned long long)diff_hi << 32) | diff_lo) & __ballot(active);
        if (!active) peers = 0ull;
        const int before = __popcll(peers & lt_mask);
        // every lane reads its tile's running offset (lanes of one tile read the same halfword: a broadcast), then the first of
        // them adds the group's size -- a wave's LDS operations execute in order, so no lane sees the update early
        const int old = active ? (int)my[tile] : 0;
        if (active && before == 0) my[tile] = (uint16_t)(old + __popcll(peers));
        if (active) point_list[base[tile] + old + before] = (int32_t)gid;
    });
}

} // namespace

// scratch (inside the binning workspace): hist [nchunks][tiles] u16, seg [nsegs][tiles] i32, base [nchunks][tiles] i32
BinPlan gsr_bin_plan(int64_t D, int tiles)
{
    BinPlan p;
    // chunk size: a multiple of NW * 1024 pairs (so every wave starts on a multiple of 1024, where kidx has an entry), about
    // D / 1024 -- a thousand workgroups -- and at most 57344 so that 16-bit per-chunk counts cannot overflow
    const int64_t q = (int64_t)NW * 1024;
    int64_t S = ((D / 1024 + q - 1) / q) * q;
    if (S < q) S = q;
    if (S > 57344) S = 57344;
    p.S = (int)S;
    p.nchunks = (int)((D + S - 1) / S);
    if (p.nchunks < 1) p.nchunks = 1;
    p.cps = 0;
    p.nsegs = 0;
    p.hist_bytes = gsr_align((size_t)p.nchunks * tiles * sizeof(uint16_t));
    p.seg_bytes = gsr_align((size_t)2 * tiles * sizeof(int32_t));             // tot, tile_start
    p.base_bytes = gsr_align((size_t)p.nchunks * tiles * sizeof(int32_t));    // rel
    return p;
}

bool gsr_bin_supported(int tiles) { return tiles >= 1 && tiles <= GSR_BIN_MAX_TILES; }

static hipError_t allow_lds(const void *fn, size_t bytes)
{
    return bytes > 48 * 1024 ? hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) : hipSuccess;
}

hipError_t gsr_launch_bin_count(const GeomWs &gw, int64_t N, int64_t D, int grid_x, int tiles, void *hist, hipStream_t s)
{
    const BinPlan p = gsr_bin_plan(D, tiles);
    const size_t lds = (size_t)tiles * sizeof(int);
    if (hipError_t e = allow_lds((const void *)bin_count_kernel, lds)) return e;
    hipLaunchKernelGGL(bin_count_kernel, dim3(p.nchunks), dim3(NT), lds, s, gw.kidx, gw.depth_item, gw.doff, gw.rect_sorted, (int)N, (int)D,
                       grid_x, tiles, p.S, (uint16_t *)hist);
    return hipGetLastError();
}

hipError_t gsr_launch_bin_scatter(const GeomWs &gw, int64_t N, int64_t D, int grid_x, int tiles, const void *hist, void *seg, void *base,
                                  int32_t *point_list, int32_t *ranges, hipStream_t s)
{
    const BinPlan p = gsr_bin_plan(D, tiles);
    const int n = (int)N, d = (int)D;
    const uint16_t *h = (const uint16_t *)hist;
    int32_t *sg = (int32_t *)seg, *bs = (int32_t *)base;
    int32_t *tot = sg, *tile_start = sg + tiles;
    if (p.nchunks > K2_WAVES * K2_MAXPER) return hipErrorInvalidValue; // gsr_bin_plan keeps nchunks near 1000
    hipLaunchKernelGGL(bin_prefix_kernel, dim3((tiles + 63) / 64), dim3(1024), 0, s, h, p.nchunks, tiles, bs, tot);
    hipLaunchKernelGGL(bin_tilescan_kernel, dim3(1), dim3(1024), 0, s, tot, tiles, tile_start, ranges);
    int tb = 1;
    while ((1 << tb) < tiles) ++tb;
    const size_t lds = (size_t)((tiles + 1) & ~1) * (NW * sizeof(uint16_t) + sizeof(int));
    const int nblocks = ((p.nchunks + 7) / 8) * 8; // chunk_of_block: every XCD gets the same number of (possibly empty) slots
#define SCATTER(B)                                                                                                             \
    {                                                                                                                         \
        if (hipError_t e = allow_lds((const void *)bin_scatter_kernel<B>, lds)) return e;                                     \
        hipLaunchKernelGGL(bin_scatter_kernel<B>, dim3(nblocks), dim3(NT), lds, s, gw.kidx, gw.depth_item, gw.doff, gw.rect_sorted, n, d, \
                           grid_x, tiles, p.S, p.nchunks, bs, tile_start, point_list);                                                               \
    }
    switch (tb) {
    case 1: case 2: case 3: case 4: case 5: case 6: case 7: case 8: SCATTER(8); break;
    case 9: SCATTER(9); break;
    case 10: SCATTER(10); break;
    case 11: SCATTER(11); break;
    case 12: SCATTER(12); break;
    case 13: SCATTER(13); break;
    default: SCATTER(14); break;
    }
#undef SCATTER
    return hipGetLastError();
}
