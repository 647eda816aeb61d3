// blend_fwd.hip -- per-tile front-to-back alpha blending for gfx950.
//
// What the reference's wp_render_gaussians does per pixel (forward.py:385-515), restructured for CDNA4:
// one 256-thread workgroup per 16x16 tile, each of its four waves owning one 8x8 pixel block (lane = pixel).  The tile's
// sorted list is staged through LDS in batches of 256 entries, each entry gathered ONCE per tile as a single 64-byte record
// (the reference gathers four arrays per pixel per entry).  While staging, the thread that fetched an entry tests it against
// the tile's eight 8x4 blocks (exact convex minimum of the conic over the block rectangle vs ln(255 o), conservative, behind
// an axis-aligned bounding-box pre-test; the exact test runs only for the blocks the box overlaps) and stores an 8-bit hit
// mask.  A wave then turns the masks of a batch into its own compact list of live entries (ballot + mbcnt, byte offsets in
// LDS) and walks that list with a two-deep software pipeline, so entries that cannot reach alpha >= 1/255 inside its block --
// most of a tile's list -- cost it nothing and the walk needs no scalar bit loop (the round-1 kernel spent 43 M scalar
// instructions there, more issue cycles than its vector work).  The masks are also written out (GsrBinning.block_masks) so
// the backward's per-block compaction reads one byte per entry instead of re-deriving the test from the records.
//
// Float operations are in the reference's order (no contraction in the transmittance chain; the conic is pre-scaled by exact
// powers of two) so the discrete tests (power > 0, alpha < 1/255, T < 1e-4) agree with the CPU oracle except where exp()
// itself rounds differently: exp is v_exp_f32(power * log2 e), relative error < 5e-7 for power in [-5.6, 0].  Colour is
// accumulated with fused multiply-adds of c and (alpha T): the same sum to a few ulps.  The common step is branch-free and
// mask-free: a finished (or off-image) pixel is parked at x = 1e15, where every entry's alpha underflows to 0; only a step in
// which some pixel would saturate takes the masked path.  `n_contrib` is the 1-based LIST position of the last contributing
// entry, so skipping dead entries does not change it.
//
// Round 4: a wave whose 64 pixels have all saturated leaves the workgroup instead of serving the tile's barriers until its slowest
// block is done (s_barrier waits for surviving waves only), and the batch shrinks to 64 entries per surviving wave: its wave slot
// goes to the next tile's workgroup that much earlier (103.5 against 105.4 us at C3, 229 against 234 at C5; rocprofv3 averages of
// three interleaved runs, profiles/r04_e_fwd_early_exit_kernel_averages.txt).
//
// Measured and rejected in round 2: eight waves per tile, each blending TWO entries per instruction over an 8x4 block
// (v_permlane32_swap hand-over of 1 - alpha): bit-identical results, but splats at C3 are as large as the blocks, so the finer
// blocks cull almost nothing (168 live entries per 8x4 block against 168 per 8x8 block) and the pair step costs 28 vector
// instructions against 2 x 22: 0.149 ms against 0.128 ms.
#include <algorithm>

#include "gsr_internal.h"

// GSR_TIMELINE (diagnostic build only, never the product): per-phase shader-cycle totals over all waves
#ifdef GSR_TIMELINE
constexpr int TL_MAX_WAVES = 1 << 17;
__device__ unsigned long long g_fwd_wave[TL_MAX_WAVES][8]; // one row per wave, plain stores (atomics would clog the memory pipe)
#define TL_DECL long long tl_t = __builtin_amdgcn_s_memtime(); unsigned long long tl_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TL(k) { const long long tl_n = __builtin_amdgcn_s_memtime(); tl_acc[k] += (unsigned long long)(tl_n - tl_t); tl_t = tl_n; }
#define TL_COUNT(k, v) tl_acc[k] += (v);
#define TL_FLUSH if (lane == 0) { const int tw = (tile * 4 + wv) & (TL_MAX_WAVES - 1); for (int q = 0; q < 8; ++q) g_fwd_wave[tw][q] = tl_acc[q]; }
#elif defined(GSR_CENSUS)
// GSR_CENSUS (diagnostic build, `make census`): the product kernel plus three scalar stamps per wave (HW_ID | XCC_ID,
// s_memrealtime start / end) for tools/residency.py
__device__ unsigned long long g_fwd_census[1 << 17][4];
__device__ const int *g_fwd_order = nullptr; // experiment (tools/residency.py --fwd-order): launch slot -> tile
#define TL_DECL const unsigned long long tl_r0 = __builtin_amdgcn_s_memrealtime();
#define TL(k)
#define TL_COUNT(k, v)
#define TL_FLUSH if (lane == 0) { unsigned long long *cw = g_fwd_census[(tile * 4 + wv) & ((1 << 17) - 1)]; \
        cw[0] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32); \
        cw[1] = tl_r0; cw[2] = __builtin_amdgcn_s_memrealtime(); }
#else
#define TL_DECL
#define TL(k)
#define TL_COUNT(k, v)
#define TL_FLUSH
#endif

namespace {

// What a wave leaves for the next frame's tile order (gsr_internal.h "forward tile order"): 1 = its measured life in 100 MHz ticks
// (the product since late round 4: C3 blend_fwd 98.0 -> 96.3 us in two same-box A/B rounds, the Lego trainer's 52.4 -> 49.0 us per
// iteration, profiles/r04_s_fwd_cost_is_wave_life.txt), 0 = entries walked and staged (max of walked + staged / 2 ranked the tiles)
#ifndef GSR_FWD_COST_LIFE
#define GSR_FWD_COST_LIFE 1
#endif
constexpr int BATCH = 256;
constexpr int NWAVES = 4;

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// minimise 0.5*qa*u^2 + 0.5*qc*v^2 + qb*u*v over v in [vlo, vhi];  nbr = -qb / qc
__device__ __forceinline__ float edge_min_q(float qa, float qb, float qc, float nbr, float u, float vlo, float vhi)
{
    float v = nbr * u;
    v = fminf(vhi, fmaxf(vlo, v));
    return 0.5f * (qa * u * u + qc * v * v) + qb * u * v;
}
// Can alpha reach 1/255 anywhere in pixels [x0,x0+7] x [y0,y0+3]?  Conservative (small slack in `lim`).
// alpha = o exp(power) >= 1/255  <=>  q(d) = 0.5 (a dx^2 + c dy^2) + b dx dy <= ln(255 o), d = centre - pixel.  q is a convex
// quadratic, so its minimum over the rectangle is 0 if the centre is inside, else it lies on an edge whose supporting line
// has the centre on its OUTER side (KKT: A (x* - c) = -lambda n with lambda >= 0 gives n.(c - x*) >= 0): for an axis-aligned
// rectangle that is at most one vertical and one horizontal edge, each a clamped 1-D quadratic.  The same test, with the same
// slack, is what the backward's own compaction applies when it has no masks (blend_bwd_splat.hip).
__device__ __forceinline__ bool block_may_hit(float gx, float gy, float ca, float cb, float cc, float nb_rc, float nb_ra, float lim, float x0,
                                              float y0)
{
    const float dxl = gx - (x0 + 7.0f), dxh = gx - x0, dyl = gy - (y0 + 3.0f), dyh = gy - y0;
    const bool in_x = dxl <= 0.0f && dxh >= 0.0f, in_y = dyl <= 0.0f && dyh >= 0.0f;
    const float qx = edge_min_q(ca, cb, cc, nb_rc, dxh < 0.0f ? dxh : dxl, dyl, dyh); // the vertical edge facing the centre
    const float qy = edge_min_q(cc, cb, ca, nb_ra, dyh < 0.0f ? dyh : dyl, dxl, dxh); // the horizontal edge facing the centre
    const float qmin = in_x ? (in_y ? 0.0f : qy) : (in_y ? qx : fminf(qx, qy));
    return qmin <= lim;
}

// LDS image of one staged entry: 48 bytes = {x, y, -a/2, -b | -c/2, opacity, r, g | b, 1/depth, -, -}.  The conic is stored as
// (-a/2, -b, -c/2): scaling by a power of two and negation commute with float rounding, so
//     power = (-a/2 dx) dx + (-c/2 dy) dy + (-b dx) dy
// is bit for bit the reference's -0.5 (a dx dx + c dy dy) - b dx dy (forward.py:477-479) with one multiply less.
// A 48-byte stride keeps the staging threads' 16-byte stores free of bank conflicts (12 i mod 32 over 8 lanes hits 8 distinct
// 4-bank groups) and lets the blend loop address all three pieces of a record from ONE byte offset, which is what the waves'
// compacted lists hold.
constexpr int REC_BYTES = 48;
constexpr int LIST_PAD = 8;       // sentinel offsets behind a wave's list: the walk reads up to three entries ahead without bounds tests
constexpr float PARKED_X = 1e15f; // x coordinate of a finished pixel: (-a/2 dx) dx ~ -1e30 a, so alpha underflows to exactly 0

struct StagedRec {
    float4 a, b;
    float2 c;
};

// Block filing for the backward blend (gsr_internal.h "block order"): this wave's two 8x4 blocks (rows 0-3 = lanes 0-31, rows
// 4-7 = lanes 32-63) go under (band of the tile, cost class, shard), the cost being what the backward's compaction will keep for
// the block -- the list entries whose mask names it, up to the block's last contributor -- counted from the mask bytes this
// workgroup wrote (the last staging barrier is behind every wave; same CU, so its L1 serves them).  Kept OUT of line on purpose:
// inlined -- inside the batch loop or behind it -- its few registers push the kernel past 64 VGPRs, the scheduler then relaxes the
// walk loop as well (70 VGPRs, 7 workgroups per CU instead of 8) and the kernel goes from 105 to 132 us; as a call made once per
// wave, with next to nothing live across it, the kernel keeps its 64 VGPRs / 73 SGPRs and needs no scratch.
__device__ __attribute__((noinline)) void file_blocks(const uint8_t *__restrict__ block_masks, int32_t *__restrict__ block_order, int bo_cap,
                                                      int n_tiles, int grid_x, int tile, int start, int last)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int kept = last;
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) kept = max(kept, __shfl_xor(kept, d, 64)); // per 32-lane half: the block's max n_contrib
    const int blk = (wv >> 1) * 4 + (wv & 1) + ((lane >> 5) << 1);
    const unsigned rep = 0x01010101u << blk;                                      // the block's bit in each of four mask bytes
    const int lo = start, hi = start + kept;
    int hits = 0;
    for (int p = (start & ~15) + 16 * (lane & 31); p < hi; p += 16 * 32) {        // 16-byte aligned walk over the bytes [lo, hi)
        const uint4 v = *reinterpret_cast<const uint4 *>(block_masks + p);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = p + 4 * j;                                              // bytes q .. q + 3: keep those inside [lo, hi)
            unsigned keep = 0xFFFFFFFFu;
            if (q < lo) keep &= (lo - q >= 4) ? 0u : (0xFFFFFFFFu << (8 * (lo - q)));
            if (q + 4 > hi) keep &= (hi - q <= 0) ? 0u : (0xFFFFFFFFu >> (8 * (q + 4 - hi)));
            hits += __popc(w[j] & rep & keep);
        }
    }
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) hits += __shfl_xor(hits, d, 64);
    if ((lane & 31) == 0) {
        const int band = gsr_bo_band(tile, n_tiles, grid_x);
        const int q = (band * GSR_BO_CLASSES + gsr_bo_class(hits)) * GSR_BO_SHARDS + ((tile - band * gsr_bo_tiles_per_band(n_tiles, grid_x)) & (GSR_BO_SHARDS - 1));
        const int pos = atomicAdd(&block_order[q], 1);
        if (pos < bo_cap) block_order[GSR_BO_HEADER + (size_t)q * bo_cap + pos] = tile * 8 + blk;
    }
}

__global__ __launch_bounds__(256) void blend_forward_kernel(int W, int H, int grid_x, float bg0, float bg1, float bg2,
                                                            const int32_t *__restrict__ ranges,
                                                            const int32_t *__restrict__ point_list,
                                                            const BlendRec *__restrict__ rec, float *__restrict__ image,
                                                            float *__restrict__ inv_depth, float *__restrict__ final_T,
                                                            int32_t *__restrict__ n_contrib, uint8_t *__restrict__ block_masks, int xcd_map,
                                                            int n_tiles, int32_t *__restrict__ block_order, int bo_cap,
                                                            float4 *__restrict__ clear4, long long clear_n4, int clear_wgs,
                                                            const int32_t *__restrict__ tile_order, int32_t *__restrict__ tile_cost)
{
    // Spare workgroups behind the tiles' (clear_wgs of them, when the caller handed over its backward workspace): they clear the
    // backward's accumulator records.  They are dispatched after every tile's workgroup, i.e. as the kernel starts to drain and
    // CUs fall idle, and the blend hardly uses HBM -- the 64 MB the backward used to clear in a launch of its own (12 us at C3,
    // 48 us at C5) cost next to nothing here.
    if (clear_wgs > 0 && (int)blockIdx.x >= (int)gridDim.x - clear_wgs) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (long long i = ((long long)blockIdx.x - ((long long)gridDim.x - clear_wgs)) * 256 + threadIdx.x; i < clear_n4; i += (long long)clear_wgs * 256)
            clear4[i] = z;
        return;
    }
    __shared__ __attribute__((aligned(16))) unsigned char s_rec[(BATCH + 1) * REC_BYTES]; // + one sentinel record (opacity 0: never valid)
    __shared__ uint8_t s_mask[BATCH];                      // bit k: entry may touch 8x4 block k (k & 1 = x half, k >> 1 = row band)
    __shared__ uint16_t s_list[NWAVES][BATCH + LIST_PAD];  // per wave: byte offsets (into s_rec) of its live entries, in list order

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // xcd_map: workgroup ids go to the eight XCDs round-robin; with the map XCD x renders the x-th eighth of the tiles (a band
    // of rows), so neighbouring tiles -- which share most of their Gaussians -- share an L2 (grid = 8 * ceil(tiles / 8))
    int tile = blockIdx.x;
    if (xcd_map) {
        tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
        if (tile >= n_tiles) return;
    } else if (tile_order) {
        tile = tile_order[blockIdx.x]; // heaviest class first, by what the tiles cost a frame ago (gsr_internal.h "forward tile order")
    }
#ifdef GSR_CENSUS
    if (g_fwd_order) tile = g_fwd_order[blockIdx.x];
#endif
    const int tile_x = tile % grid_x, tile_y = tile / grid_x;
    const int pix_x = tile_x * 16 + (wv & 1) * 8 + (lane & 7);
    const int pix_y = tile_y * 16 + (wv >> 1) * 8 + (lane >> 3);
    const float tx0 = (float)(tile_x * 16), ty0 = (float)(tile_y * 16);
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const int my_bits = 5 << ((wv >> 1) * 4 + (wv & 1)); // this wave's 8x8 block = the 8x4 blocks k0 and k0 + 2

#if GSR_FWD_COST_LIFE
    const unsigned long long t_born = wall_clock64();
#endif
    const int2 range = *reinterpret_cast<const int2 *>(ranges + 2 * tile);
    const int start = range.x, end = range.y;

    float T = 1.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f, cd = 0.0f;
    int last = 0;
    bool done = !(pix_x < W && pix_y < H);
    float pixf_x = done ? PARKED_X : (float)pix_x, pixf_y = (float)pix_y;
    asm volatile("" : "+v"(pixf_x), "+v"(pixf_y)); // keep the converted coordinates in registers (hipcc re-converts them per entry otherwise)
    if (tid < 3) reinterpret_cast<float4 *>(s_rec + BATCH * REC_BYTES)[tid] = make_float4(0.f, 0.f, 0.f, 0.f); // the sentinel

    TL_DECL
    // A wave whose 64 pixels have all saturated LEAVES (s_barrier waits for the surviving waves of a workgroup only --
    // tools/barrier_exit_probe.hip), instead of serving the tile's barriers until its slowest block is done: its wave slot is free
    // for the next tile's workgroup that much earlier.  The batch is then 64 entries per surviving wave.  Who stages what is agreed
    // one batch ahead: `alive_cur` (the set this batch was prefetched under) and `alive_next` (s_alive as every wave reads it
    // behind the top barrier, i.e. after the leavers of the last walk cleared their bits).  A leaver still stages its share of
    // the batch that was prefetched with it, then goes: the others' next wait for it falls under their walk of that batch.
    __shared__ int s_alive;
    if (tid == 0) s_alive = 0xF;
    [[maybe_unused]] int walked = 0;
    int alive_cur = 0xF; // walked: list entries this wave blended (its share of the tile's cost, for the next frame's order)
    const int wv_u = __builtin_amdgcn_readfirstlane(wv); // wave-uniform copy: the bookkeeping below stays in scalar registers
    auto rank_in = [&](int set) { return __popc(set & ((1 << wv_u) - 1)); };
    int nid = (start + tid < end) ? point_list[start + tid] : -1;
    float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na;
    float2 ncd = make_float2(0.f, 0.f);
    if (nid >= 0) {
        const float4 *rp = reinterpret_cast<const float4 *>(rec + nid);
        na = rp[0]; nb = rp[1];
        ncd = *reinterpret_cast<const float2 *>(rp + 2);
    }
    [[maybe_unused]] int staged = 0; // list entries staged by the time this wave leaves
    for (int base = start; base < end;) {
        __syncthreads(); // every surviving wave has walked the last batch (LDS reuse) and the leavers' bits are cleared
        TL(0)
        const int alive_next = __builtin_amdgcn_readfirstlane(s_alive);
        if (alive_next == 0) break; // the last block saturated: nothing left to blend in this tile
        const int cnt = min(64 * __popc(alive_cur), end - base);
        const int e = 64 * rank_in(alive_cur) + lane; // my entry of this batch
        if (e < cnt) {
            const float4 a = na, b = nb;
            float4 *dst = reinterpret_cast<float4 *>(s_rec + e * REC_BYTES);
            dst[0] = make_float4(a.x, a.y, -0.5f * a.z, -a.w);
            dst[1] = make_float4(-0.5f * b.x, b.y, b.z, b.w);
            *reinterpret_cast<float2 *>(dst + 2) = ncd;
            int m = 0;
            if (b.y * 255.0f >= 1.0f) {
                const float lim = __builtin_amdgcn_logf(b.y * 255.0f) * 0.6931471805599453f * 1.0001f + 1e-3f;
                const float det = a.z * b.x - a.w * a.w;
                const bool boxless = !(det > 0.0f);
                const float kdet = 2.0f * lim * fast_rcp(det);
                const float hx = __builtin_amdgcn_sqrtf(kdet * b.x) * 1.001f + 0.05f, hy = __builtin_amdgcn_sqrtf(kdet * a.z) * 1.001f + 0.05f;
                const float lx = a.x - hx - tx0, rx = a.x + hx - tx0, ly = a.y - hy - ty0, ry = a.y + hy - ty0;
                int cand;
                {
                    const int xm = (boxless || (lx <= 7.0f && rx >= 0.0f) ? 0x55 : 0) | (boxless || (lx <= 15.0f && rx >= 8.0f) ? 0xAA : 0);
                    int ym = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (boxless || (ly <= (float)(4 * r + 3) && ry >= (float)(4 * r))) ym |= 3 << (2 * r);
                    cand = xm & ym;
                }
                const float nb_rc = -a.w * fast_rcp(b.x), nb_ra = -a.w * fast_rcp(a.z);
                while (cand) {
                    const int k = __builtin_ctz(cand);
                    cand &= cand - 1;
                    if (block_may_hit(a.x, a.y, a.z, a.w, b.x, nb_rc, nb_ra, lim, tx0 + (float)((k & 1) * 8), ty0 + (float)((k >> 1) * 4))) m |= 1 << k;
                }
            }
            s_mask[e] = (uint8_t)m;
            if (block_masks) block_masks[base + e] = (uint8_t)m;
        }
        TL(2)
        const bool staying = (alive_next >> wv_u) & 1;
        if (staying) { // my share of the next batch, as the next batch's stagers will be
            const int nidx = base + cnt + 64 * rank_in(alive_next) + lane;
            nid = (nidx < min(end, base + cnt + 64 * __popc(alive_next))) ? point_list[nidx] : -1;
            if (nid >= 0) {
                const float4 *rp = reinterpret_cast<const float4 *>(rec + nid);
                na = rp[0]; nb = rp[1];
                ncd = *reinterpret_cast<const float2 *>(rp + 2);
            }
        }
        __syncthreads();
        TL(3)
        if (!staying) break; // I cleared my bit after my last walk; my share of this batch is staged: gone
        // this wave's live entries of the batch, compacted in list order (LDS operations of one wave execute in order, so the
        // list can be read back without a barrier); sentinels behind it
        int n = 0;
        for (int g = 0; g < cnt; g += 64) {
            const int mv = (g + lane < cnt) ? (int)s_mask[g + lane] : 0;
            const bool hit = (mv & my_bits) != 0;
            const unsigned long long bits = __ballot(hit);
            if (hit) s_list[wv][n + __popcll(bits & lt_mask)] = (uint16_t)((g + lane) * REC_BYTES);
            n += __popcll(bits);
        }
        if (lane < LIST_PAD) s_list[wv][n + lane] = (uint16_t)(BATCH * REC_BYTES);
        // lanes read list entries other lanes wrote: in order in hardware, but the compiler must not move the reads up
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        TL(4) // list build
        TL_COUNT(7, (unsigned long long)n)
        walked += n;

        // walk the list; two-deep software pipeline: the offset of entry k+2 and the record of entry k+1 are in flight while
        // entry k is blended
        const uint16_t *lp = &s_list[wv][0];
        auto fetch = [&](int off) {
            StagedRec r;
            const float4 *q = reinterpret_cast<const float4 *>(s_rec + off);
            r.a = q[0]; r.b = q[1];
            r.c = *reinterpret_cast<const float2 *>(q + 2);
            return r;
        };
        // Two register sets alternate (no register rotation); an odd list ends on a sentinel, which contributes nothing.
#define GSR_BLEND(R, OFF)                                                                                                      \
    {                                                                                                                         \
        const float dx = R.a.x - pixf_x, dy = R.a.y - pixf_y;                                                                 \
        const float power = (R.a.z * dx * dx + R.b.x * dy * dy) + R.a.w * dx * dy; /* see REC_BYTES: the reference's value */   \
        const float alpha = fminf(0.99f, R.b.y * fast_exp(power));                                                            \
        /* forward.py:480-484: skip when power > 0 or alpha < 1/255 -- as selects, not branches */                            \
        const float t = power > 0.0f ? 0.0f : alpha;                                                                          \
        const bool contributes = !(t < (1.0f / 255.0f));                                                                      \
        const float a_eff = contributes ? t : 0.0f;                                                                           \
        const float test_T = T * (1.0f - a_eff); /* forward.py:486 */                                                          \
        float w = a_eff * T, T_next = test_T; /* a_eff = 0 where the entry does not contribute: w = 0 and T_next = T exactly */ \
        bool counts = contributes;                                                                                            \
        if (__builtin_amdgcn_ballot_w64(test_T < 0.0001f) != 0ull) {                                                          \
            /* some pixel saturates at this entry (forward.py:487-489): the entry is not applied there and the pixel ends.    \
               Only these four values change; the updates below are one code path (as two, hipcc reconciled the register     \
               sets of the two paths with six v_mov per entry on the COMMON one) */                                          \
            const bool sat = contributes && (test_T < 0.0001f);                                                               \
            w = sat ? 0.0f : w;                                                                                               \
            T_next = sat ? T : T_next;                                                                                        \
            counts = contributes && !sat;                                                                                     \
            if (sat) pixf_x = PARKED_X; /* the pixel is finished (`done` is re-derived from this after the walk) */           \
            if (__all(pixf_x == PARKED_X)) break; /* all 64 pixels are saturated (none of them applies this entry either) */  \
        }                                                                                                                     \
        cr = __builtin_fmaf(R.b.z, w, cr); cg = __builtin_fmaf(R.b.w, w, cg);                                                 \
        cb = __builtin_fmaf(R.c.x, w, cb); cd = __builtin_fmaf(R.c.y, w, cd);                                                 \
        if (counts) last_off = OFF;                                                                                           \
        T = T_next;                                                                                                           \
    }
        int off0 = lp[0], off1 = lp[1];
        StagedRec r0 = fetch(off0);
        int last_off = -1;
        for (int k = 0; k < n; k += 2) {
            const StagedRec r1 = fetch(off1);      // entry k+1 (or a sentinel)
            const int off2 = lp[2];
            GSR_BLEND(r0, off0)
            r0 = fetch(off2);                      // entry k+2
            off0 = off2;
            const int off3 = lp[3];
            lp += 2;
            GSR_BLEND(r1, off1)
            off1 = off3;
        }
#undef GSR_BLEND
        // n_contrib = 1-based list position of the last contributing entry: offset / 48 by multiply-shift (exact below 2^16)
        if (last_off >= 0) last = base - start + (int)(((unsigned)last_off * 43691u) >> 21) + 1;
        done = pixf_x == PARKED_X;
        TL(5)
        base += cnt;
        staged = base - start;
        alive_cur = alive_next;
        if (__all(done) && base < end) {
            if (lane == 0) atomicAnd(&s_alive, ~(1 << wv_u)); // the next top barrier is behind this
        }
    }
    if (pix_x < W && pix_y < H) {
        const size_t px = (size_t)pix_y * W + pix_x;
        final_T[px] = T;
        n_contrib[px] = last;
        image[3 * px] = cr + T * bg0;
        image[3 * px + 1] = cg + T * bg1;
        image[3 * px + 2] = cb + T * bg2;
        inv_depth[px] = cd;
    }
#if GSR_FWD_COST_LIFE
    if (tile_cost && lane == 0) tile_cost[tile * 4 + wv] = (int)min((unsigned long long)0x7FFF, wall_clock64() - t_born) << 16;
#else
    if (tile_cost && lane == 0) tile_cost[tile * 4 + wv] = (min(walked, 0x7FFF) << 16) | min(staged, 0xFFFF);
#endif
    if (block_order) file_blocks(block_masks, block_order, bo_cap, n_tiles, grid_x, tile, start, last);
    TL(6)
    TL_FLUSH
}

} // namespace

#ifdef GSR_CENSUS
extern "C" int gsr_debug_fwd_order(const int *order_dev) // device array [tiles] or NULL
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_order), &order_dev, sizeof(order_dev)) == hipSuccess ? 0 : -1;
}
extern "C" int gsr_debug_fwd_census(unsigned long long *out /* [waves][4] */, int waves, int clear)
{
    if (waves > (1 << 17)) return -1;
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_fwd_census)) != hipSuccess) return -1;
        return hipMemset(p, 0, sizeof(unsigned long long) * 4 * (size_t)waves) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fwd_census), sizeof(unsigned long long) * 4 * (size_t)waves) == hipSuccess ? 0 : -1;
}
#endif
#ifdef GSR_TIMELINE
extern "C" int gsr_debug_fwd_phases(unsigned long long *out /* [waves][8] */, int waves)
{
    if (waves > TL_MAX_WAVES) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fwd_wave), sizeof(unsigned long long) * 8 * (size_t)waves) == hipSuccess ? 0 : -1;
}
#endif

int gsr_fwd_xcd_map = 0; // GSR_FWD_XCD (see the kernel)
int gsr_fwd_no_order = 0; // GSR_FWD_NO_ORDER: row-major dispatch instead of last frame's cost classes

hipError_t gsr_launch_blend_forward(const CamK &cam, const int32_t *ranges, const int32_t *point_list, const BlendRec *rec,
                                    const GsrImage &img, uint8_t *block_masks, int32_t *block_order, void *clear, size_t clear_bytes,
                                    hipStream_t s, const int32_t *tile_order, int32_t *tile_cost)
{
    const int tiles = cam.grid_x * cam.grid_y;
    if (tiles <= 0) return hipSuccess;
    // the clear is part of the call's contract (gsr.h GsrBinning.backward_ws): when this launch cannot host the spare workgroups
    // (the XCD map derives the tile from gridDim.x) it is a memset in front of the kernel instead
    if (clear && gsr_fwd_xcd_map) {
        if (hipError_t e = hipMemsetAsync(clear, 0, clear_bytes, s)) return e;
    }
    const long long clear_n4 = (clear && !gsr_fwd_xcd_map) ? (long long)(clear_bytes / 16) : 0;
    const int clear_wgs = (int)std::min<long long>(2048, (clear_n4 + 255) / 256);
    const int grid = (gsr_fwd_xcd_map ? 8 * ((tiles + 7) / 8) : tiles) + clear_wgs;
    hipLaunchKernelGGL(blend_forward_kernel, dim3(grid), dim3(256), 0, s, cam.W, cam.H, cam.grid_x, cam.bg[0], cam.bg[1], cam.bg[2],
                       ranges, point_list, rec, img.image, img.inv_depth, img.final_T, img.n_contrib, block_masks, gsr_fwd_xcd_map, tiles,
                       block_masks ? block_order : nullptr, gsr_bo_cap(tiles, cam.grid_x), reinterpret_cast<float4 *>(clear), clear_n4, clear_wgs,
                       gsr_fwd_xcd_map ? nullptr : tile_order, tile_cost);
    return hipGetLastError();
}
