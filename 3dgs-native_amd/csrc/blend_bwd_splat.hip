// blend_bwd_splat.hip -- Gaussian-parallel gradient replay for gfx950 (default backward blend).
//
// Same mathematics as the reference's wp_render_backward_kernel (backward.py:559-706), parallelised
// the other way round.  The reference (and this library's first version) gives a lane a PIXEL and walks the list, so
// every (pixel, entry) pair produces nine values that must be summed over pixels.  Here a lane owns a
// list ENTRY and the wave walks the pixels, so the nine gradients of an entry accumulate in that
// lane's registers with no cross-lane reduction at all; what crosses lanes instead is the per-pixel
// recurrence over entries, and that is two wave-wide DPP scans per pixel:
//     R_k = prod_{deeper j <= k} 1/(1 - alpha_j)      ->  T_k = T_final * R_k          (backward.py:658)
//     Q_k = sum_{deeper j <  k} alpha_j T_j (c_j . dL_dpixel)  ( = T_k (1-alpha_k) accum_rec . dL_dpixel, :667-671)
// so that dL_dalpha_k = T_k (c_k . dL_dpixel) - (Q_k + T_final bg . dL_dpixel) / (1 - alpha_k)   (:671-680).
// One single-wave workgroup per 8x4 pixel block (8 per tile, 20 000 at 800x800; 8x8 and 4x4 are also
// instantiated) replays its tile's list back to front.  Most entries of a tile's list cannot touch a given block, so the wave first
// COMPACTS the stream: 64 candidates at a time are tested against the block rectangle (exact convex
// minimum of the conic over the rectangle vs ln(255 o), conservative) and survivors are queued in an
// LDS ring in order; each full bucket of survivors (32 since round 4, 64 before; lane 0 = deepest) then runs the pixel loop, so
// lanes are mostly live.  Per-pixel carries (P, Q) live in LDS between buckets; pixels whose n_contrib
// ends before the bucket are skipped wave-uniformly.
// After a bucket each lane holds the block's complete gradient for its entry; an LDS transpose lets 16
// lanes write one 64-byte GradRec with float atomics (4 whole 64-B requests per wave instruction).
//
// Differences from the reference in float rounding only: sums over pixels/entries are re-associated
// (quirk Q15), per-entry constants are factored out of the pixel sums, FMA contraction is on (except in `power`, which is
// evaluated in the forward's operation order so the replayed alphas are the forward's), and 1/x uses v_rcp_f32 (1 ulp).
#include <algorithm>

#include "gsr_internal.h"

// This kernel re-associates float sums anyway (quirk Q15), so fused multiply-adds are allowed here.
#pragma clang fp contract(fast)

// GSR_TIMELINE (diagnostic build only, never the product): per-phase shader cycles of every wave (tools/bwd_timeline.py)
#ifdef GSR_TIMELINE
constexpr int TLB_MAX_WAVES = 1 << 17;
__device__ unsigned long long g_bwd_wave[TLB_MAX_WAVES][12];
// [5] = start (shader clock), [8] = HW_ID | XCC_ID << 32, [9] / [10] = s_memrealtime (100 MHz, chip-wide) at start / end
#define TL_DECL long long tl_t = __builtin_amdgcn_s_memtime(); const unsigned long long tl_r0 = __builtin_amdgcn_s_memrealtime(); unsigned long long tl_acc[8] = {0, 0, 0, 0, 0, (unsigned long long)tl_t, 0, 0};
#define TL(k) { const long long tl_n = __builtin_amdgcn_s_memtime(); tl_acc[k] += (unsigned long long)(tl_n - tl_t); tl_t = tl_n; }
#define TL_COUNT(k, v) tl_acc[k] += (v);
#define TL_FLUSH if (threadIdx.x == 0) { const int tw = bid & (TLB_MAX_WAVES - 1); for (int q = 0; q < 8; ++q) g_bwd_wave[tw][q] = tl_acc[q]; \
        g_bwd_wave[tw][8] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32); \
        g_bwd_wave[tw][9] = tl_r0; g_bwd_wave[tw][10] = __builtin_amdgcn_s_memrealtime(); }
#elif defined(GSR_CENSUS)
// GSR_CENSUS (diagnostic build, `make census`): the product kernel, same registers and LDS, plus three scalar stamps per wave --
// where it ran (HW_ID, XCC_ID) and its s_memrealtime start / end -- so residency per CU is measured on the real code object
// (the GSR_TIMELINE build needs 94 VGPRs: its waves are capped at 5 per SIMD, the product's 62 are not).  tools/residency.py
__device__ unsigned long long g_bwd_census[1 << 17][4];
__device__ const int *g_bwd_order = nullptr; // experiment (tools/residency.py --lpt): launch slot -> block, e.g. longest-lived first
#define TL_DECL const unsigned long long tl_r0 = __builtin_amdgcn_s_memrealtime();
#define TL(k)
#define TL_COUNT(k, v)
#define TL_FLUSH if (threadIdx.x == 0) { unsigned long long *cw = g_bwd_census[bid & ((1 << 17) - 1)]; \
        cw[0] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32); \
        cw[1] = tl_r0; cw[2] = __builtin_amdgcn_s_memrealtime(); }
#else
#define TL_DECL
#define TL(k)
#define TL_COUNT(k, v)
#define TL_FLUSH
#endif

namespace {

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get(float v, float identity)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
// inclusive scans over the 64 lanes (lane 0 first)
// One fused v_mul_f32_dpp per step: lanes whose DPP source is out of the row (or whose row is masked)
// are write-disabled and keep v, which is exactly "multiply by 1".  hipcc emits v_mov + v_mov_dpp +
// v_mul for the builtin form.  A VALU write followed by a DPP read of the same VGPR needs 2 wait
// states; the s_nop is inside the asm because hipcc does not pad around inline asm.
__device__ __forceinline__ float wave_scan_mul(float v)
{
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ float wave_scan_add(float v)
{
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}
// The same scans over each 32-lane half of the wave separately (lanes 0-31 and 32-63 hold different pixels): without the
// last, row_bcast:31 step; row_bcast:15 with row mask 0xa feeds rows 1 and 3 from rows 0 and 2.
__device__ __forceinline__ float half_scan_mul(float v)
{
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ float half_scan_add(float v)
{
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}
// ... and over each 16-lane DPP row separately (four pixels per step): the four row_shr steps alone
__device__ __forceinline__ float row_scan_mul(float v)
{
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ float row_scan_add(float v)
{
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
}

// Conservative test: can the Gaussian reach alpha >= 1/255 anywhere in the pixel rectangle
// [x0,x0+7] x [y0,y0+7]?  alpha = o*exp(power) >= 1/255  <=>  q(d) = 0.5(a dx^2 + c dy^2) + b dx dy <= ln(255 o).
// q is a convex quadratic (the conic is the inverse of a positive-definite covariance), so its minimum
// over the rectangle is 0 if the centre is inside, else it lies on one of the four edges, where it is a
// clamped 1-D quadratic.  A small slack keeps the test a strict superset of the exact per-pixel test.
__device__ __forceinline__ float edge_min_q(float qa, float qb, float qc, float u, float vlo, float vhi)
{
    // minimise 0.5*qa*u^2 + 0.5*qc*v^2 + qb*u*v over v in [vlo, vhi]
    float v = -qb * u * fast_rcp(qc);
    v = fminf(vhi, fmaxf(vlo, v));
    return 0.5f * (qa * u * u + qc * v * v) + qb * u * v;
}
__device__ __forceinline__ bool block_may_hit(float gx, float gy, float ca, float cb, float cc, float opacity, float x0, float y0, float ex, float ey)
{
    if (!(opacity * 255.0f >= 1.0f)) return false;
    const float tau = __builtin_amdgcn_logf(opacity * 255.0f) * 0.6931471805599453f; // ln via v_log_f32 (argument >= 1: no denormal path); >= 0
    // d = g - pixel, pixel in [x0, x0+ex] -> d in [gx-x0-ex, gx-x0]
    const float dxl = gx - (x0 + ex), dxh = gx - x0, dyl = gy - (y0 + ey), dyh = gy - y0;
    float qmin;
    if (dxl <= 0.0f && dxh >= 0.0f && dyl <= 0.0f && dyh >= 0.0f) qmin = 0.0f;
    else {
        qmin = edge_min_q(ca, cb, cc, dxl, dyl, dyh);
        qmin = fminf(qmin, edge_min_q(ca, cb, cc, dxh, dyl, dyh));
        qmin = fminf(qmin, edge_min_q(cc, cb, ca, dyl, dxl, dxh));
        qmin = fminf(qmin, edge_min_q(cc, cb, ca, dyh, dxl, dxh));
    }
    return qmin <= tau * 1.0001f + 1e-3f;
}

// power = -0.5 (a dx^2 + c dy^2) - b dx dy in the forward's (= the reference's, forward.py:477-479 / backward.py:641-643)
// operation order, every product rounded on its own.  For needle-like splats (100:1 and more, hundreds of pixels long) the
// three terms cancel to a few units out of 1e4..1e6, and a fused form replays alphas that differ from the forward's by
// per cent: tests/test_gpu_fuzz.py seed 63 had dL_dconic 3.5 % off for a 1.28 x 0.008 x 0.003 Gaussian.
#pragma clang fp contract(off)
// With the conic pre-scaled to (-a/2, -b, -c/2) the same roundings come out of one multiply less: scaling by a power of two
// commutes with rounding, so (-a/2 dx) dx + (-c/2 dy) dy = -0.5 (a dx dx + c dy dy) and (-b dx) dy = -(b dx dy) bit for bit
// (blend_fwd.hip stages its records the same way).  The dy term is the same for a whole pixel row: row_term() once per row.
__device__ __forceinline__ float row_term(float cc2, float dy) { return (cc2 * dy) * dy; }
__device__ __forceinline__ float power_ref_order(float ca2, float cb2, float tc, float dx, float dy)
{
    return ((ca2 * dx) * dx + tc) + (cb2 * dx) * dy;
}
#pragma clang fp contract(fast)

// Candidate chunks (of 64) requested per fill iteration on the masks path: more chunks per memory round trip shorten the fill (a
// quarter of a wave's life).  Round 2 let the ring grow with them (128 entries per chunk), and its LDS is occupancy: 1 / 2 / 4
// chunks ran the kernel in 170.0 / 177.4 / 175.1 us at C3.  Since round 3 the ring stays at 128 entries and a requested chunk is
// consumed only while another 64 survivors would still fit (otherwise it is requested again later -- at C3's 12 % hit rate that is
// rare): 1 / 2 / 3 / 4 chunks 154.7 / 152.6 / 151.8 / 152.5 us at C3, 494 / 484 / - / 485 at C2i, 420 / 428 / - / 417 at C5.
#ifndef GSR_FILL_K
#define GSR_FILL_K 4
#endif
constexpr int FILL_K = GSR_FILL_K;
// GSR_BWD_BUCKET: entries per bucket -- 32 (the product), 64 (rounds 2-3) or 16.  A bucket of at most 32 entries runs TWO pixels per
// step (each 32-lane half holds the bucket against its own pixel; scans of five DPP steps instead of six), one of at most 16
// FOUR (GSR_BWD_FOUR: 16-lane rows, four-step scans) -- the "16-entry x 4-pixel lane layout" of round 3's notes, here as the form
// the LAST bucket of a block takes.  Fewer scan steps per (entry, pixel) pair against more buckets (record gather, LDS transpose,
// flush per bucket).  Round 4, rocprofv3 kernel averages over 60 launches, three interleaved runs per build on one box
// (profiles/r04_d_bwd_bucket_kernel_averages.txt -- the kernel's run-to-run spread is +-4 %, so single A/B pairs mislead):
//   C3: 64 -> 150.5 us, 32 -> 147.0, 32 + four-pixel last bucket + skipped finished pixel pairs -> 142.4 (lowest in every run);
//   C0: 64 -> 226.0, 32 -> 219.0;  16 throughout (four pixels everywhere): C3 145.6, C0 233.6, C5 406 against 414 / 410 (tools/ab.sh).
#ifndef GSR_BWD_BUCKET
#define GSR_BWD_BUCKET 32
#endif
#ifndef GSR_BWD_FOUR
#define GSR_BWD_FOUR 1      // 1: a bucket of at most 16 entries runs four pixels per step (16-lane rows, four-step scans)
#endif
#ifndef GSR_BWD_SKIP2
#define GSR_BWD_SKIP2 1     // 1: the two-pixel form skips a step when both pixels' replays have ended
#endif
#ifndef GSR_BWD_BUCKET_WIDE
#define GSR_BWD_BUCKET_WIDE GSR_BWD_BUCKET   // the same for the 8x8 blocks (A/B)
#endif
constexpr int BUCKET = GSR_BWD_BUCKET;
constexpr int QCAP = 128;                 // ring of compacted entries (power of two, >= 63 + 64): a chunk is consumed only while it fits

// USE_MASKS: the per-block hit masks the forward wrote (GsrBinning.block_masks) replace the compaction's own test.  They are
// per 8x4 block; an 8x8 block ORs the bits of its two halves, a 4x4 block keeps its own (finer) test.
template <int BW, int BH, bool USE_MASKS>
__global__ __launch_bounds__(64) void blend_backward_splat_kernel(int W, int H, int grid_x, float bg0, float bg1, float bg2,
                                                                  const int32_t *__restrict__ ranges,
                                                                  const int32_t *__restrict__ point_list,
                                                                  const BlendRec *__restrict__ rec,
                                                                  const float *__restrict__ final_T,
                                                                  const int32_t *__restrict__ n_contrib,
                                                                  const float *__restrict__ dL_dpixels,
                                                                  const uint8_t *__restrict__ block_masks, GradRec *__restrict__ acc, int dbg, int xcd_map, int n_blocks,
                                                                  const int32_t *__restrict__ block_order, int bo_cap)
{
    constexpr int NPIX = BW * BH;            // pixels of the block this wave owns
    constexpr int PER_TILE = 256 / NPIX;     // blocks per 16x16 tile
    constexpr int NBX = 16 / BW;
    // px, py, and the pixel's two carries between buckets: T = T_final * product of deeper 1/(1-alpha) (the transmittance in
    // front of the next bucket's deepest entry) and Q = T_final (bg . dpix) + sum of deeper alpha*T*(c . dpix)
    __shared__ float4 s_pq[NPIX];
    __shared__ float4 s_pb[NPIX];   // dpix r,g,b, kept (as int bits)
    __shared__ int2 s_ring[QCAP];   // compacted survivors: (Gaussian id, list index); records are re-gathered (L2 hits)
    __shared__ float s_g[64][9];    // per-entry gradients for the transposed flush (odd stride: no bank conflicts)
    __shared__ int s_id[64];

    const int lane = threadIdx.x;
    // Workgroup ids are handed to the eight XCDs round-robin, so consecutive ids -- the PER_TILE blocks of one tile, which read
    // the same list, masks and records -- land on eight different L2s.  xcd_map != 0: XCD x takes the x-th eighth of the blocks
    // instead, so that a tile's blocks run on ONE XCD, next to each other in its queue (grid = 8 * ceil(blocks / 8)).
    // xcd_map 2: XCD x takes the tiles t with t % 8 == x (interleaved: every XCD gets the same mix of deep and shallow tiles,
    // where the bands of map 1 give the XCDs holding the image centre more work than the others).
    int bid = blockIdx.x;
    const bool ordered = block_order && block_order[GSR_BO_FLAG] != 0; // the forward filed the blocks (it skips large images)
    if (ordered) {
    } else if (xcd_map == 1 || block_order) {
        const int per = (n_blocks + 7) >> 3, k = blockIdx.x >> 3; // (the grid of the ordered path may be larger than this needs)
        if (k >= per) return;
        bid = (blockIdx.x & 7) * per + k;
        if (bid >= n_blocks) return;
    } else if (xcd_map == 2) {
        const int k = blockIdx.x >> 3; // position in this XCD's queue
        bid = ((k / PER_TILE) * 8 + (blockIdx.x & 7)) * PER_TILE + (k % PER_TILE);
        if (bid >= n_blocks) return;
    }
    if (ordered) {
        // The forward filed every block under (band of its tile, cost class) -- gsr_internal.h "block order".  Launch slot
        // blockIdx.x runs on XCD blockIdx.x % 8 as the (blockIdx.x / 8)-th workgroup of that XCD's queue: it takes the band's
        // k-th block counting from the heaviest class down.  Lane c looks at class CLASSES - 1 - c.
        const int band = blockIdx.x & 7, k = blockIdx.x >> 3;
        // the band's 512 queues in walking order (class descending, shard ascending): lane l owns positions 8 l .. 8 l + 7
        int c[8];
        {
            const int cls = GSR_BO_CLASSES - 1 - (lane >> 1), sh0 = (lane & 1) * 8;
            const int4 *cp = reinterpret_cast<const int4 *>(block_order + (band * GSR_BO_CLASSES + cls) * GSR_BO_SHARDS + sh0);
            const int4 a = cp[0], b = cp[1];
            c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
        }
        int tot = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) tot += c[j];
        int incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        const unsigned long long owner = __ballot(incl > k);
        if (owner == 0ull) return; // beyond the band's blocks (the grid is rounded up)
        const int src = __builtin_ctzll(owner);
        int basej = incl - tot, jj = 0; // every lane searches its own eight; only lane src's answer is used
#pragma unroll
        for (int j = 0; j < 7; ++j)
            if (basej + c[j] <= k && jj == j) { basej += c[j]; jj = j + 1; }
        const int pos = __shfl(8 * lane + jj, src, 64), idx = k - __shfl(basej, src, 64);
        const int q = (band * GSR_BO_CLASSES + (GSR_BO_CLASSES - 1 - (pos >> 4))) * GSR_BO_SHARDS + (pos & 15);
        bid = block_order[GSR_BO_HEADER + (size_t)q * bo_cap + idx];
        bid = __builtin_amdgcn_readfirstlane(bid);
    }
#ifdef GSR_CENSUS
    if (g_bwd_order) bid = g_bwd_order[blockIdx.x];
#endif
    const int tile = bid / PER_TILE, sub = bid % PER_TILE;
    const int tile_x = tile % grid_x, tile_y = tile / grid_x;
    const int2 range = *reinterpret_cast<const int2 *>(ranges + 2 * tile);
    const int start = range.x, end = range.y;
    if (end <= start) return;
    TL_DECL

    // lane q prepares the constants of pixel q of the block
    const int bx0 = tile_x * 16 + (sub % NBX) * BW, by0 = tile_y * 16 + (sub / NBX) * BH;
    const int my_x = bx0 + (lane % BW), my_y = by0 + (lane / BW);
    int kept = start;
    if (lane < NPIX) {
        float Tfin = 0.0f, d0 = 0.0f, d1 = 0.0f, d2 = 0.0f;
        if (my_x < W && my_y < H) {
            const size_t px = (size_t)my_y * W + my_x;
            Tfin = final_T[px];
            kept = min(end, start + n_contrib[px]);
            d0 = dL_dpixels[3 * px]; d1 = dL_dpixels[3 * px + 1]; d2 = dL_dpixels[3 * px + 2];
        }
        float bgdot = bg0 * d0;
        bgdot += bg1 * d1;
        bgdot += bg2 * d2;
        s_pq[lane] = make_float4((float)my_x, (float)my_y, Tfin, Tfin * bgdot);
        s_pb[lane] = make_float4(d0, d1, d2, __int_as_float(kept));
    }
    const int hi_all = wave_max_i(kept);
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
    const float fx0 = (float)bx0, fy0 = (float)by0;
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    // mask bit k <-> 8x4 block k of the tile: x half k & 1, row band k >> 1
    const int mask_shift = BH == 4 ? sub : ((sub >> 1) * 4 + (sub & 1)), mask_bits = BH == 4 ? 1 : 5;
    __syncthreads();

    int cursor = hi_all; // next list index (exclusive) to pull candidates from, moving towards `start`
    int head = 0, qn = 0; // ring state (wave-uniform)
    constexpr int bucket = NPIX == 64 ? GSR_BWD_BUCKET_WIDE : BUCKET; // entries per bucket (see GSR_BWD_BUCKET above)
    TL(0) // prologue
    for (;;) {
        // ---- fill: pull candidates (deepest first) until a full bucket is queued or the list is exhausted ----
        while (qn < bucket && cursor > start) {
            if (USE_MASKS) {
                // The forward already tested every staged entry against the tile's eight 8x4 blocks (blend_fwd.hip): one byte
                // per entry, read coalesced, instead of two 16-byte gathers and the convex test per candidate.  FILL_K chunks of
                // 64 candidates are requested together, one memory round trip per iteration.
                int ids[FILL_K], mvs[FILL_K];
#pragma unroll
                for (int j = 0; j < FILL_K; ++j) {
                    const int idx = cursor - 1 - (j * 64 + lane);
                    ids[j] = 0; mvs[j] = 0;
                    if (idx >= start) {
                        ids[j] = point_list[idx];
                        mvs[j] = (int)block_masks[idx];
                    }
                }
                int used = 0;
#pragma unroll
                for (int j = 0; j < FILL_K; ++j) {
                    if (j > 0 && qn > QCAP - 64) break; // (wave-uniform) the ring may not take another 64: the chunk is requested again later
                    const int idx = cursor - 1 - (j * 64 + lane);
                    const bool hit = idx >= start && ((mvs[j] >> mask_shift) & mask_bits) != 0;
                    const unsigned long long m = __ballot(hit);
                    if (hit) s_ring[(head + qn + __popcll(m & lt_mask)) & (QCAP - 1)] = make_int2(ids[j], idx);
                    qn += __popcll(m);
                    used = j + 1;
                }
                cursor = max(start, cursor - 64 * used);
                continue;
            }
            const int lo = max(start, cursor - 64);
            const int idx = cursor - 1 - lane;
            bool hit = false;
            int id = 0;
            if (idx >= lo) {
                id = point_list[idx];
                const float4 *rp = reinterpret_cast<const float4 *>(rec + id);
                float4 a = rp[0], b = rp[1];
                // both 16-byte loads complete here: left alone, hipcc sinks the first one below the opacity test of
                // block_may_hit and the fill loop pays three dependent memory round trips per iteration instead of two
                asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y));
                hit = block_may_hit(a.x, a.y, a.z, a.w, b.x, b.y, fx0, fy0, (float)(BW - 1), (float)(BH - 1));
            }
            const unsigned long long m = __ballot(hit);
            if (hit) s_ring[(head + qn + __popcll(m & lt_mask)) & (QCAP - 1)] = make_int2(id, idx);
            qn += __popcll(m);
            cursor = lo;
        }
        if (qn == 0) break;
        __syncthreads();
        TL(1) // fill

        // ---- one bucket: lane k takes the k-th queued entry (still deepest first) ----
        const int n = min(bucket, qn);
        // A bucket of at most 32 entries (every block's last one, half of the time) runs TWO pixels per step: lanes 0-31 hold
        // the entries against pixel 2s, lanes 32-63 the same entries against pixel 2s + 1, and the scans stay inside each half.
        const bool four = n <= 16 && BW % 4 == 0 && GSR_BWD_FOUR; // ... and one of at most 16 FOUR pixels per step, a 16-lane row each
        const bool two = !four && n <= 32 && (NPIX % 2 == 0);
        const int elane = four ? (lane & 15) : two ? (lane & 31) : lane; // which entry of the bucket this lane holds
        const bool valid = elane < n;
        const int slot = (head + elane) & (QCAP - 1);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        float colb = 0.0f;
        int idx = 0x7FFFFFFF, id = 0;
        if (valid) {
            const int2 e = s_ring[slot];
            id = e.x;
            idx = e.y;
            const float4 *rp = reinterpret_cast<const float4 *>(rec + id);
            a = rp[0]; // xy.x xy.y con.a con.b
            b = rp[1]; // con.c opacity r g
            colb = rp[2].x;
        }
#ifdef GSR_TIMELINE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TL(2) // record gather
        TL_COUNT(6, 1ull)
#endif
        const int idx_min = __shfl(idx, n - 1, 64); // shallowest entry of the bucket
        head = (head + n) & (QCAP - 1);
        qn -= n;

        // Per-pixel work accumulates only the moments every gradient is linear in; the per-entry constants
        // (conic, opacity, 0.5*W, 0.5*H) are applied once per bucket below:
        //   h = dL/dG * G            S1 = sum h dx      S2 = sum h dy
        //   Sxx = sum h dx^2         Sxy = sum h dx dy  Syy = sum h dy^2      Sop = sum G dL/dalpha
        float g_c0 = 0.f, g_c1 = 0.f, g_c2 = 0.f, S1 = 0.f, S2 = 0.f, Sxx = 0.f, Sxy = 0.f, Syy = 0.f, Sop = 0.f;
        bool touched = false;
        const float ca2 = -0.5f * a.z, cb2 = -a.w, cc2 = -0.5f * b.x; // exact

#define GSR_PIXEL_STEP(Q, SCAN_MUL, SCAN_ADD, CARRY_LANE)                                                                      \
    {                                                                                                                         \
        const float4 pq = s_pq[Q];                                                                                            \
        const float d_x = a.x - pq.x;                                                                                         \
        const float power = power_ref_order(ca2, cb2, tc, d_x, d_y);                                                          \
        const float G = fast_exp(power);                                                                                      \
        const float alpha = fminf(0.99f, b.y * G);                                                                            \
        const bool live = (idx < pkept) && !(power > 0.0f) && !(alpha < (1.0f / 255.0f));                                      \
        const float inv = fast_rcp(1.0f - alpha); /* 1/(1-alpha): scanned as a product, and reused in dL/dalpha */            \
        const float m = live ? inv : 1.0f;                                                                                    \
        const float Pi = SCAN_MUL(m);                                                                                         \
        const float T = pq.z * Pi; /* transmittance in front of this entry = T_final / prod(1-alpha) over it and all deeper */ \
        const float cd = b.z * pb.x + b.w * pb.y + colb * pb.z;                                                               \
        const float w = alpha * T;                                                                                            \
        const float qv = live ? w * cd : 0.0f;                                                                                \
        const float Qi = SCAN_ADD(qv);                                                                                        \
        if (CARRY_LANE) *reinterpret_cast<float2 *>(&s_pq[Q].z) = make_float2(T, pq.w + Qi);                                  \
        if (live) {                                                                                                           \
            touched = true;                                                                                                   \
            const float Qe = pq.w + (Qi - qv); /* background term + deeper entries only */                                    \
            const float dL_dalpha = T * cd - Qe * inv;                                                                        \
            g_c0 += w * pb.x; g_c1 += w * pb.y; g_c2 += w * pb.z;                                                             \
            const float gd = G * dL_dalpha;                                                                                   \
            Sop += gd;                                                                                                        \
            const float h = b.y * gd; /* dL/dG * G = o * dL/dalpha * G */                                                     \
            const float hx = h * d_x, hy = h * d_y;                                                                           \
            S1 += hx; S2 += hy;                                                                                               \
            Sxx += hx * d_x; Sxy += hx * d_y; Syy += hy * d_y;                                                                \
        }                                                                                                                     \
    }
        const int rows_run = GSR_ABL(dbg, 2) ? 1 : BH;
        if (four) {
            const int sub = lane >> 4;
            for (int r = 0; r < rows_run; ++r) {
                const float d_y = a.y - (fy0 + (float)r), tc = row_term(cc2, d_y);
                for (int q4 = r * BW; q4 < (GSR_ABL(dbg, 2) ? 4 : (r + 1) * BW); q4 += 4) {
                    const int q = q4 + sub;
                    const float4 pb = s_pb[q];
                    const int pkept = __float_as_int(pb.w);
                    if (__ballot(pkept > idx_min) == 0ull) continue; // all four pixels' replays end before every entry of the bucket
                    TL_COUNT(7, 1ull)
                    GSR_PIXEL_STEP(q, row_scan_mul, row_scan_add, (lane & 15) == 15)
                }
            }
            // an entry's sums are split over its four lanes (one per row): add them up (rows 1-3 then hold copies and stay out of the flush)
#define GSR_SUM4(v) v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            GSR_SUM4(g_c0) GSR_SUM4(g_c1) GSR_SUM4(g_c2) GSR_SUM4(S1) GSR_SUM4(S2) GSR_SUM4(Sxx) GSR_SUM4(Sxy) GSR_SUM4(Syy) GSR_SUM4(Sop)
#undef GSR_SUM4
            int t4 = (int)touched;
            t4 |= __shfl_xor(t4, 16, 64);
            t4 |= __shfl_xor(t4, 32, 64);
            touched = t4 != 0 && lane < 16;
        } else if (!two) {
            for (int r = 0; r < rows_run; ++r) {
                const float d_y = a.y - (fy0 + (float)r), tc = row_term(cc2, d_y);
                for (int q = r * BW; q < (GSR_ABL(dbg, 2) ? 1 : (r + 1) * BW); ++q) {
                    const float4 pb = s_pb[q];
                    const int pkept = __float_as_int(pb.w);
                    if (pkept <= idx_min) continue; // wave-uniform: this pixel's replay ends before every entry of the bucket
                    TL_COUNT(7, 1ull)
                    GSR_PIXEL_STEP(q, wave_scan_mul, wave_scan_add, lane == 63)
                }
            }
        } else {
            const int sub = lane >> 5;
            for (int r = 0; r < rows_run; ++r) {
                const float d_y = a.y - (fy0 + (float)r), tc = row_term(cc2, d_y);
                for (int q2 = r * BW; q2 < (GSR_ABL(dbg, 2) ? 2 : (r + 1) * BW); q2 += 2) {
                    const int q = q2 + sub;
                    const float4 pb = s_pb[q];
                    const int pkept = __float_as_int(pb.w);
#if GSR_BWD_SKIP2
                    if (__ballot(pkept > idx_min) == 0ull) continue; // both pixels' replays end before every entry of the bucket
#endif
                    TL_COUNT(7, 1ull)
                    GSR_PIXEL_STEP(q, half_scan_mul, half_scan_add, (lane & 31) == 31)
                }
            }
            // an entry's sums are split over its two lanes: add the halves (lanes 32-63 then hold copies and stay out of the flush)
            g_c0 += __shfl_xor(g_c0, 32, 64); g_c1 += __shfl_xor(g_c1, 32, 64); g_c2 += __shfl_xor(g_c2, 32, 64);
            S1 += __shfl_xor(S1, 32, 64); S2 += __shfl_xor(S2, 32, 64);
            Sxx += __shfl_xor(Sxx, 32, 64); Sxy += __shfl_xor(Sxy, 32, 64); Syy += __shfl_xor(Syy, 32, 64);
            Sop += __shfl_xor(Sop, 32, 64);
            const int partner_touched = __shfl_xor((int)touched, 32, 64); // unconditionally: a cross-lane read must not sit behind ||
            touched = (touched || partner_touched != 0) && lane < 32;
        }
#undef GSR_PIXEL_STEP
        TL(3) // pixel loop
        // dL/dmean2D = dL/dG * dG/ddel * 0.5*(W,H), dG/ddelx = -G (a dx + b dy); dL/dconic = -0.5 h (dx^2, dx dy, dy^2)
        const float g_mx = (2.0f * ca2 * S1 + cb2 * S2) * ddelx_dx; // = -(a S1 + b S2) 0.5 W
        const float g_my = (2.0f * cc2 * S2 + cb2 * S1) * ddely_dy; // = -(c S2 + b S1) 0.5 H
        const float g_ca = -0.5f * Sxx, g_cb = -0.5f * Sxy, g_cc = -0.5f * Syy, g_op = Sop;

        // transpose through LDS: 16 lanes per entry -> one 64-byte accumulator record each
        s_id[lane] = touched ? id : -1;
        s_g[lane][0] = g_c0; s_g[lane][1] = g_c1; s_g[lane][2] = g_c2; s_g[lane][3] = g_mx; s_g[lane][4] = g_my;
        s_g[lane][5] = g_ca; s_g[lane][6] = g_cb; s_g[lane][7] = g_cc; s_g[lane][8] = g_op;
        __syncthreads();
        const int c = lane & 15, fslot = gsr_gradrec_slot(c); // the record's layout leaves the API arrays' zero columns free
        const int flush_rows = (n + 3) >> 2; // four entries per wave instruction; entries beyond n have nothing
#pragma unroll 4
        for (int r = 0; r < flush_rows; ++r) {
            const int e = r * 4 + (lane >> 4);
            const int eid = s_id[e];
            if (c < 9 && eid >= 0 && !GSR_ABL(dbg, 1)) unsafeAtomicAdd(&acc[eid].f[fslot], s_g[e][c]);
        }
        __syncthreads();
        TL(4) // flush
    }
    TL_FLUSH
}

// Rebuild blend records from the forward's per-Gaussian outputs (backward() receives them as
// arguments: means2D, conic_opacity, rgb -- reference backward.py:975-980).
__global__ __launch_bounds__(256) void pack_records_kernel(const float *__restrict__ xy, const float *__restrict__ conic_opacity,
                                                           const float *__restrict__ rgb, const float *__restrict__ depths,
                                                           BlendRec *__restrict__ rec, int64_t N)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float2 p = *reinterpret_cast<const float2 *>(xy + 2 * i);
    const float4 co = *reinterpret_cast<const float4 *>(conic_opacity + 4 * i);
    const float r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
    const float d = depths ? depths[i] : 0.0f;
    float4 *rp = reinterpret_cast<float4 *>(rec + i);
    rp[0] = make_float4(p.x, p.y, co.x, co.y);
    rp[1] = make_float4(co.z, co.w, r, g);
    rp[2] = make_float4(b, d != 0.0f ? 1.0f / d : 0.0f, 0.0f, 0.0f);
}

} // namespace

hipError_t gsr_launch_pack_records(const GsrGeom &g, BlendRec *rec, int64_t N, hipStream_t s)
{
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_records_kernel, dim3((unsigned)gsr_div_up(N, 256)), dim3(256), 0, s, g.xy, g.conic_opacity, g.rgb,
                       g.depths, rec, N);
    return hipGetLastError();
}

#ifdef GSR_CENSUS
extern "C" int gsr_debug_bwd_order(const int *order_dev) // device array [blocks] or NULL
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_order), &order_dev, sizeof(order_dev)) == hipSuccess ? 0 : -1;
}
extern "C" int gsr_debug_bwd_census(unsigned long long *out /* [waves][4] */, int waves, int clear)
{
    if (waves > (1 << 17)) return -1;
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_bwd_census)) != hipSuccess) return -1;
        return hipMemset(p, 0, sizeof(unsigned long long) * 4 * (size_t)waves) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_census), sizeof(unsigned long long) * 4 * (size_t)waves) == hipSuccess ? 0 : -1;
}
#endif
#ifdef GSR_TIMELINE
extern "C" int gsr_debug_bwd_phases(unsigned long long *out /* [waves][12] */, int waves)
{
    if (waves > TLB_MAX_WAVES) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_wave), sizeof(unsigned long long) * 12 * (size_t)waves) == hipSuccess ? 0 : -1;
}
#endif

int gsr_bwd_block = 0;    // GSR_BWD_BLOCK: 0 = by the frame's tile pairs per Gaussian (see the launcher)
int gsr_bwd_no_order = 0;  // GSR_BWD_NO_ORDER: ignore the forward's block order (A/B)
int gsr_bwd_xcd_map = 1;   // GSR_BWD_XCD (see the kernel): on by default, 171 -> 165 us at C3
int gsr_debug_flags = 0; // see gsr_internal.h

hipError_t gsr_launch_blend_backward_splat(const CamK &cam, const int32_t *ranges, const int32_t *point_list, const BlendRec *rec,
                                           const GsrImage &img, const float *dL_dpixels, const uint8_t *block_masks,
                                           const int32_t *block_order, GradRec *acc, int64_t N, int64_t D, hipStream_t s)
{
    const int tiles = cam.grid_x * cam.grid_y;
    if (tiles <= 0) return hipSuccess;
#define LAUNCH(BW, BH, M)                                                                                                     \
    do {                                                                                                                      \
        const int nblk = tiles * (256 / ((BW) * (BH)));                                                                       \
        const int ppt = 256 / ((BW) * (BH)); /* map 2 walks whole groups of 8 tiles: grid = 8 * ppt * ceil(tiles / 8) */        \
        const int32_t *bo = ((M) && (BW) == 8 && (BH) == 4 && !gsr_bwd_no_order) ? block_order : nullptr; /* 8x4 blocks only */    \
        const int bo_cap = gsr_bo_cap(tiles, cam.grid_x);                                                                                 \
        const int grid = bo ? 8 * 8 * gsr_bo_tiles_per_band(tiles, cam.grid_x) : gsr_bwd_xcd_map == 2 ? 8 * ppt * ((tiles + 7) / 8) : gsr_bwd_xcd_map ? 8 * ((nblk + 7) / 8) : nblk; \
        hipLaunchKernelGGL((blend_backward_splat_kernel<BW, BH, M>), dim3(grid), dim3(64), 0, s, cam.W, cam.H, cam.grid_x,    \
                           cam.bg[0], cam.bg[1], cam.bg[2], ranges, point_list, rec, img.final_T, img.n_contrib, dL_dpixels,  \
                           block_masks, acc, gsr_debug_flags, gsr_bwd_xcd_map, nblk, bo, bo_cap);                             \
    } while (0)
    // Pixels per wave.  The block masks discard an entry for a whole block, so small splats (few tile pairs per Gaussian) want the
    // finer 8x4 blocks; splats that cover their tiles anyway want 8x8, which stages every entry in half as many waves.  Measured
    // over D / N from 2.7 to 262 (tools/bwd_block_sweep.py, profiles/r04_q_bwd_block_size_sweep.txt): at 800 x 800 the curves
    // cross at D / N of 18-22 (8x4 is 10-25 % faster below 13, 8x8 is 14-24 % faster above 40); images of more than
    // GSR_BO_MAX_TILES tiles, whose 8x4 blocks the forward does not file by cost, cross at 4-5 already (1080p: 8x8 is 5-24 %
    // faster from 5.4 up).  GSR_BWD_BLOCK = 32, 64 or 16 (4x4) forces one size.
    const int block_px = gsr_bwd_block_px(N, D, tiles);
    switch (block_px) {
    case 16: LAUNCH(4, 4, false); break;
    case 64: if (block_masks) LAUNCH(8, 8, true); else LAUNCH(8, 8, false); break;
    default: if (block_masks) LAUNCH(8, 4, true); else LAUNCH(8, 4, false); break;
    }
#undef LAUNCH
    return hipGetLastError();
}
