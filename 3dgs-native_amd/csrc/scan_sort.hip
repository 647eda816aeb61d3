// scan_sort.hip -- binning stage for gfx950: scans, stable radix passes, key expansion, tile ranges.
//
// The reference builds one (tile << 32 | depth bits) int64 key per (Gaussian, tile) pair in id order
// and sorts all of them with an 8-pass 64-bit radix sort (forward.py:518-558, :791-824).  The result
// is the list ordered by (tile, depth bits, id) -- ties broken by id because the sort is stable and
// pairs are emitted in id order (quirk Q13).  We produce the SAME list with much less HBM traffic:
//   1. sort the N Gaussians once by depth bits (stable, from id order)      -> order (depth, id)
//   2. expand them to (tile << id_shift | id) items in that order            -> order (depth, id) per tile
//      (32-bit items when tile bits + id bits <= 32, else 64-bit with id_shift = 32)
//   3. stable-partition the D items by tile id only (ceil(log2(tiles)/8) passes instead of 8, digit bits split
//      evenly: 12 tile bits -> 6 + 6)
// All sorting is one kernel family: a stable LSD radix pass over 32- or 64-bit items on a 4..8-bit digit.
// The last pass of the depth sort also carries each Gaussian's tile rectangle and tile count to its sorted position.
// Wave64 ballots give each item its rank among equal digits (no per-item atomics), an LDS reorder
// makes the scatter write contiguous runs.  No inter-workgroup spin-waits anywhere: every dependency
// is a kernel boundary.  (A single-kernel-per-pass variant with ticketed chunks and decoupled look-back
// over 8-byte sc1 status words was built and measured in round 1: bit-identical output but 1.4-2.1x
// SLOWER on MI355X -- depth sort 0.196 vs 0.093 ms, tile sort 0.187 vs 0.131 ms at C3 -- because every
// look-back hop is a ~1-3 us memory-side round trip and all chunks run in lockstep; see DESIGN.md.)
#include <algorithm>

#include "gsr_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------
// wave / block scan helpers (wave64)
// ---------------------------------------------------------------------------------------------
// Inclusive scan over the 64 lanes in six fused v_add_u32_dpp steps (row_shr 1,2,4,8, row_bcast 15,31).
// Lanes whose DPP source is outside the row / whose row is masked are write-disabled and keep their
// value, which is "add 0".  (ds_bpermute-based __shfl_up costs an LDS round trip per step.)
__device__ __forceinline__ int wave_incl_scan(int v)
{
    asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}

// Inclusive scan of one int per thread across a 256-thread block; *total gets the block sum.
__device__ __forceinline__ int block_incl_scan_256(int v, int *lds4 /* >= 4 ints */, int *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int s = wave_incl_scan(v);
    if (lane == 63) lds4[w] = s;
    __syncthreads();
    int add = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int c = lds4[k];
        if (k < w) add += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return s + add;
}

// the same across a block of NW waves (every thread calls it; lds >= NW ints)
template <int NW> __device__ __forceinline__ int block_incl_scan_nw(int v, int *lds, int *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int s = wave_incl_scan(v);
    if (lane == 63) lds[w] = s;
    __syncthreads();
    int add = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        int c = lds[k];
        if (k < w) add += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return s + add;
}

// ---------------------------------------------------------------------------------------------
// device-wide scan in two kernels: per-wave sums -> per-wave rescan (each wave first adds up the sums before it).
// The unit of work is a WAVE owning 1024 consecutive items, walked in 16 rounds of 64 so every load and
// store is a fully coalesced 256-byte access and no block barrier is needed.
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_WAVE_ITEMS = 1024;
constexpr int SCAN_ROUNDS = SCAN_WAVE_ITEMS / 64;

template <int MODE>
__global__ __launch_bounds__(256) void scan_reduce_kernel(const int32_t *__restrict__ in, int32_t *__restrict__ wave_sums, int64_t n)
{
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t base = wid * SCAN_WAVE_ITEMS;
    if (base >= n) return;
    int s = 0;
#pragma unroll
    for (int r = 0; r < SCAN_ROUNDS; ++r) {
        const int64_t k = base + r * 64 + lane;
        s += (k < n) ? in[k] : 0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) wave_sums[wid] = s;
}

// SUMS_PER_UNIT: `wave_sums` holds that many partial sums per 1024-item unit (1 from scan_reduce_kernel; 4 when the kernel that
// produced `in` wrote one sum per 256 items itself -- preprocess does, which saves the reduce launch of the id-order scan).
// What the depth sort needs to know about this frame's depths (device memory: the host never sees it).  The visible depth bits
// lie in [min_bits, min_bits + range - 1]; the sort key of a Gaussian is umin(bits - min_bits, range) -- the same order as the
// bits themselves (positive floats order like their bit patterns), culled Gaussians (bits 0xFFFFFFFF) all at `range`, behind
// every visible one, in id order -- and only the low `8 * npass` bits of it can differ.  A typical scene spans less than two
// octaves of depth around the camera distance: range < 2^24, three 8-bit passes instead of four.
struct DepthCtl { // what a depth kernel works with, derived from DepthCtlRaw when it starts (depth_ctl_load)
    uint32_t min_bits, range;
    int32_t npass, first; // passes 0 .. 3 are launched; pass p sorts digit p - first, passes below `first` = 4 - npass exit at once
    int32_t n_vis;        // visible Gaussians: the first active pass drops the culled ones, the later passes move n_vis items
};
// In memory (the uint4 behind the last block's extremes, cleared by preprocess): the frame's extremes and visible count, combined
// by atomics from the few workgroups of the scan's launch that reduce the per-block values.  The minimum is kept complemented so
// that zero is the identity of every field.
struct DepthCtlRaw {
    uint32_t inv_min, max_bits;
    int32_t n_vis;
    uint32_t pad;
};
__device__ __forceinline__ DepthCtl depth_ctl_load(const DepthCtlRaw *__restrict__ raw, int force_npass)
{
    const uint4 r = *reinterpret_cast<const uint4 *>(raw);
    const DepthPlan p = gsr_depth_plan(~r.x, r.y, force_npass);
    DepthCtl c;
    c.min_bits = p.min_bits; c.range = p.range; c.npass = p.npass; c.first = p.first; c.n_vis = (int32_t)r.z;
    return c;
}

// Run by a few workgroups of the id-order scan's launch (which follows preprocess in the stream and precedes the depth passes):
// preprocess left the smallest and largest visible depth bits of every 256-Gaussian block (0xFFFFFFFF / 0 for a block without a
// visible one) and its visible count.  Workgroup k of K reduces its share of the blocks (at most 2048 when K < 16: one round of
// loads per thread -- as ONE workgroup this was ten dependent rounds at 5 M Gaussians, 16 us that the whole launch waited for)
// and adds it to DepthCtlRaw; its extremes also go to the pinned host words 2 + 2k, 3 + 2k (the host's launch guess for the next
// frame: api.hip).
__device__ __forceinline__ void depth_ctl_from_blocks(const uint32_t *__restrict__ blk_minmax, int nblk, DepthCtlRaw *__restrict__ raw, int k, int K,
                                                      int32_t *host_words)
{
    __shared__ uint32_t s_lo[4], s_hi[4];
    __shared__ int s_nv[4];
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    int nv = 0;
    const uint4 *p4 = reinterpret_cast<const uint4 *>(blk_minmax); // per block {min, max, visible count, -}
    const int per = (nblk + K - 1) / K, b_begin = k * per, b_end = min(nblk, b_begin + per);
    for (int b0 = b_begin; b0 < b_end; b0 += 256 * 8) {
        uint4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int b = b0 + q * 256 + (int)threadIdx.x;
            v[q] = b < b_end ? p4[b] : make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            lo = min(lo, v[q].x);
            hi = max(hi, v[q].y);
            nv += (int)v[q].z;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, d, 64));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, d, 64));
        nv += __shfl_xor(nv, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; s_nv[threadIdx.x >> 6] = nv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3]));
        hi = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
        nv = s_nv[0] + s_nv[1] + s_nv[2] + s_nv[3];
        if (~lo) atomicMax(&raw->inv_min, ~lo);
        if (hi) atomicMax(&raw->max_bits, hi);
        if (nv) atomicAdd(&raw->n_vis, nv);
        if (host_words) { host_words[2 + 2 * k] = (int32_t)lo; host_words[3 + 2 * k] = (int32_t)hi; }
    }
}

template <int MODE, int SUMS_PER_UNIT>
__device__ __forceinline__ void scan_final_block(const int32_t *__restrict__ in, const int32_t *__restrict__ wave_sums, int32_t *__restrict__ out,
                                                 int64_t n, int32_t *total_out, int block)
{
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)block * 4 + (threadIdx.x >> 6);
    const int64_t base = wid * SCAN_WAVE_ITEMS;
    if (base >= n) return;
    // all 16 rounds are loaded before anything else: left to itself the compiler emits load -> wait -> scan -> store per
    // round (the stores' branches fence the loads), a chain of 16 memory round trips per wave
    int v[SCAN_ROUNDS];
#pragma unroll
    for (int r = 0; r < SCAN_ROUNDS; ++r) {
        const int64_t k = base + r * 64 + lane;
        v[r] = (k < n) ? in[k] : 0;
    }
    // this wave's offset = sum of the sums of all earlier waves: a few KB read per wave beats a third kernel launch
    // (16-byte loads, four in flight: as a scalar loop this was up to 16 dependent round trips for the last waves; the array is
    // padded by 4 ints, elements at or past the limit are masked)
    int carry = 0;
    {
        const int limit = (int)wid * SUMS_PER_UNIT;
        const int4 *s4 = reinterpret_cast<const int4 *>(wave_sums);
#pragma unroll 4
        for (int j = lane * 4; j < limit; j += 256) {
            const int4 q = s4[j >> 2];
            carry += q.x + (j + 1 < limit ? q.y : 0) + (j + 2 < limit ? q.z : 0) + (j + 3 < limit ? q.w : 0);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) carry += __shfl_xor(carry, d, 64);
    int32_t total = 0;
#pragma unroll
    for (int r = 0; r < SCAN_ROUNDS; ++r) {
        const int inc = wave_incl_scan(v[r]);
        const int64_t k = base + r * 64 + lane;
        if (k == n - 1) total = carry + inc;
        v[r] = carry + (MODE == 0 ? inc : inc - v[r]); // mode 0 inclusive, mode 2 exclusive
        carry += __shfl(inc, 63, 64);
    }
#pragma unroll
    for (int r = 0; r < SCAN_ROUNDS; ++r) {
        const int64_t k = base + r * 64 + lane;
        if (k < n) out[k] = v[r];
    }
    // the grand total goes straight to a (pinned, device-visible) host word: no separate copy kernel for D
    if (total_out && base + SCAN_WAVE_ITEMS >= n && ((n - 1 - base) & 63) == lane) *total_out = total;
}

template <int MODE, int SUMS_PER_UNIT>
__global__ __launch_bounds__(256) void scan_final_kernel(const int32_t *__restrict__ in, const int32_t *__restrict__ wave_sums,
                                                         int32_t *__restrict__ out, int64_t n, int32_t *total_out,
                                                         const uint32_t *__restrict__ blk_minmax, int nblk, DepthCtlRaw *__restrict__ ctl)
{
    if (ctl && blockIdx.x == gridDim.x - 1) { // one extra workgroup, launched for this alone: off the scan's critical path
        depth_ctl_from_blocks(blk_minmax, nblk, ctl, 0, 1, total_out);
        return;
    }
    scan_final_block<MODE, SUMS_PER_UNIT>(in, wave_sums, out, n, total_out, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// stable radix pass (4..8-bit digit) on 64-bit items
// ---------------------------------------------------------------------------------------------
// Chunk = 256 * ITEMS items per block (each wave owns 64*ITEMS consecutive items).  ITEMS = 16 for the big
// D-item passes (fewer, fatter blocks); ITEMS = 4 when n is small, so the launch still fills the chip.

// Per pass two kernels (round 1 had a third, a row scan over the block histograms, between them: six 5-us launches per
// frame that each scanned under a megabyte).  The histogram kernel writes its block's counts block-major, hist[block][digit],
// and also ADDS them into a small accumulator array (sb adds per address: one row of totals over ALL blocks measured 10 us
// per pass of same-address serialisation at nb = 1024):
//     acc[256 + (block / sb) * 256 + d]   total of digit d over the super-block of `sb` consecutive blocks
// from which the scatter kernel of the same pass rebuilds, per digit, the digit's total (all nb / sb super-blocks) and the
// number of items before its block (the super-blocks before its own, plus the blocks before it inside its own, < sb rows of
// hist) -- about 2 sqrt(nb) coalesced 16-byte loads spread over the block instead of a launch.  acc[0..255] receives the
// totals from block 0 of a FINAL scatter (ranges_fixup_kernel reads them).  The accumulators of a pass are zeroed by the kernel that precedes its
// histogram kernel in the stream (preprocess / expand for the first pass, the previous pass's scatter after that).
// DEPTH (the four depth passes over the 64-bit depth|id items): the digit comes from the reduced key umin(bits - min, range)
// (DepthCtl above), which pass this is decides the digit and which of the two ping-pong buffers is the input, and a pass the
// frame does not need returns at once -- all read from device memory, so the host launches the same four passes every frame.
struct DepthPass {
    const DepthCtlRaw *ctl;
    int force_npass;     // GSR_DEBUG bit 8 (tests): four passes whatever the range
    int pass;            // 0 .. 3
    uint64_t *buf[2];    // ping-pong buffers; the first ACTIVE pass reads buf[0]
    int32_t *acc_first;  // the accumulators of the first ACTIVE pass, whichever pass that is: its histogram is made ahead of the plan,
                         // beside the id-order scan (scan_ctl_hist_kernel), into an array of its own
    int launched_first;  // the host launched the passes launched_first .. 3 only (its guess from the previous frame): if this
                         // frame needs an earlier one (ctl->first < launched_first), every launched pass leaves the data alone
                         // and the host, told by the readback, launches all four
    int pack_ok;         // the tile grid fits 6 bits per coordinate and the ids 24: see "packed depth items" below
};
// Packed depth items (late round 4).  The last depth pass hands the rest of the pipeline each Gaussian's id, tile rectangle and
// tile count in depth order; fetching the rectangle BY ID at that point is a random 8-byte gather (121 MB moved for 8 needed at
// C3, 20 us against 10 for a plain pass).  The first ACTIVE pass reads its items in id order, where the rectangle is a coalesced
// read -- and once that pass has consumed the key's low byte, what is left of a two- or three-pass key (8 or 16 bits) shares a
// 64-bit item with the rectangle at 6 bits per coordinate and a 24-bit id:
//     [63:48] key >> 8   [47:24] x0 | y0 << 6 | x1 << 12 | y1 << 18   [23:0] id
// So when the frame's plan has two or three passes (decided on the device, like the plan) and the sizes fit (pack_ok, decided by the
// host), the first active pass writes such items, the later passes take their digits from bits 48.. and the last one unpacks instead
// of gathering.  Same order, same outputs; a four-pass frame (16 + 24 + 24 + 8 > 64) and a one-pass frame (its only pass reads in
// id order anyway) keep the plain items.
__device__ __forceinline__ bool depth_items_packed(const DepthPass &dp, const DepthCtl &c) { return dp.pack_ok && c.npass >= 2 && c.npass <= 3; }
__device__ __forceinline__ unsigned long long depth_item_pack(uint32_t key, unsigned long long rect_raw, uint32_t id)
{
    const uint32_t r24 = (uint32_t)(rect_raw & 63ull) | ((uint32_t)((rect_raw >> 16) & 63ull) << 6) | ((uint32_t)((rect_raw >> 32) & 63ull) << 12) |
                         ((uint32_t)((rect_raw >> 48) & 63ull) << 18);
    return ((unsigned long long)((key >> 8) & 0xFFFFu) << 48) | ((unsigned long long)r24 << 24) | (unsigned long long)(id & 0xFFFFFFu);
}
__device__ __forceinline__ unsigned long long depth_item_rect(unsigned long long item) // raw TileRect bits (4 x uint16)
{
    const uint32_t r24 = (uint32_t)(item >> 24) & 0xFFFFFFu;
    return (unsigned long long)(r24 & 63u) | ((unsigned long long)((r24 >> 6) & 63u) << 16) | ((unsigned long long)((r24 >> 12) & 63u) << 32) |
           ((unsigned long long)((r24 >> 18) & 63u) << 48);
}
template <bool DEPTH, int BITS, typename ItemT>
__device__ __forceinline__ int radix_digit(ItemT item, int shift, uint32_t kmin, uint32_t krange)
{
    if constexpr (DEPTH) return (int)((min((uint32_t)(item >> 32) - kmin, krange) >> shift) & 255u);
    else return (int)((item >> shift) & ((1 << BITS) - 1));
}

template <int RADIX_ITEMS, int BITS, typename ItemT, bool DEPTH>
__device__ __forceinline__ void radix_hist_block(const ItemT *__restrict__ in, int32_t *__restrict__ hist, int32_t *__restrict__ acc, int64_t n, int shift,
                                                 int sb, int block, uint32_t kmin, uint32_t krange, bool drop_culled)
{
    constexpr int CHUNK = 256 * RADIX_ITEMS;
    constexpr int RADIX = 1 << BITS;
    __shared__ int h[RADIX];
    if (threadIdx.x < RADIX) h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)block * CHUNK;
    // loads first, LDS atomics after: otherwise every round waits for its own load (RADIX_ITEMS serial round trips)
    // (16-byte loads: a histogram does not care which thread counts which item; `in` and every chunk start are 16-byte aligned)
    constexpr int VEC = 16 / (int)sizeof(ItemT);
    static_assert(RADIX_ITEMS % VEC == 0, "whole 16-byte loads per thread");
    struct alignas(16) Vec { ItemT v[VEC]; };
    Vec item[RADIX_ITEMS / VEC];
#pragma unroll
    for (int r = 0; r < RADIX_ITEMS / VEC; ++r) {
        const int64_t k = base + (int64_t)(r * 256 + (int)threadIdx.x) * VEC;
        if (k + VEC <= n) item[r] = *reinterpret_cast<const Vec *>(in + k);
        else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) item[r].v[j] = k + j < n ? in[k + j] : (ItemT)0;
        }
    }
#pragma unroll
    for (int r = 0; r < RADIX_ITEMS / VEC; ++r) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int64_t k = base + (int64_t)(r * 256 + (int)threadIdx.x) * VEC + j;
            bool take = k < n;
            if constexpr (DEPTH) take = take && !(drop_culled && (uint32_t)(item[r].v[j] >> 32) == 0xFFFFFFFFu);
            if (take) atomicAdd(&h[radix_digit<DEPTH, BITS>(item[r].v[j], shift, kmin, krange)], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < RADIX) {
        const int c = h[threadIdx.x];
        hist[(size_t)block * RADIX + threadIdx.x] = c;
        if (c) atomicAdd(&acc[256 + (block / sb) * 256 + threadIdx.x], c);
    }
}

template <int RADIX_ITEMS, int BITS, typename ItemT, bool DEPTH = false>
__global__ __launch_bounds__(256) void radix_hist_kernel(const ItemT *__restrict__ in, int32_t *__restrict__ hist, int32_t *__restrict__ acc,
                                                         int64_t n, int shift, int sb, DepthPass dp)
{
    constexpr int CHUNK = 256 * RADIX_ITEMS;
    uint32_t kmin = 0u, krange = 0u;
    if constexpr (DEPTH) {
        const DepthCtl c = depth_ctl_load(dp.ctl, dp.force_npass);
        if (dp.pass < c.first || c.first < dp.launched_first) return; // this frame's keys need fewer passes / more than were launched
        const int rel = dp.pass - c.first;
        if (rel == 0) return; // the first active pass's histogram exists already (scan_ctl_hist_kernel)
        in = reinterpret_cast<const ItemT *>(dp.buf[rel & 1]);
        shift = 8 * rel;
        kmin = c.min_bits; krange = c.range;
        if (depth_items_packed(dp, c)) { shift = 16 + 8 * (rel - 1); kmin = 0u; krange = 0xFFFFFFFFu; } // the digit sits in bits 48 + 8 (rel - 1) ..
        n = c.n_vis; // the first active pass dropped the culled ones: the later passes see the n_vis survivors only
        if ((int64_t)blockIdx.x * CHUNK >= n) return;
    }
    radix_hist_block<RADIX_ITEMS, BITS, ItemT, DEPTH>(in, hist, acc, n, shift, sb, (int)blockIdx.x, kmin, krange, false);
}

// The id-order scan behind preprocess (scan_final_kernel<0, 4>), the depth sort's pass plan (its control workgroup) and the
// histogram of the depth sort's first ACTIVE pass in ONE launch.  That histogram does not depend on the plan: the first active
// pass always reads the items preprocess wrote, counts the visible ones, and its digit is the lowest byte of the raw depth
// bits (DepthCtl.min_bits has a zero low byte).  The scan's workgroups come first in the grid (D reaches the host early), the
// histogram's fill the chip behind them: 5 us at C3, 25 us at C5 no longer stand between preprocess and the sort.
template <int HIST_ITEMS>
__global__ __launch_bounds__(256) void scan_ctl_hist_kernel(const int32_t *__restrict__ in, const int32_t *__restrict__ wave_sums, int32_t *__restrict__ out,
                                                            int64_t n, int32_t *total_out, const uint32_t *__restrict__ blk_minmax, int nblk,
                                                            DepthCtlRaw *__restrict__ ctl, int n_ctl, int nb_scan,
                                                            const uint64_t *__restrict__ items, int32_t *__restrict__ hist, int32_t *__restrict__ acc_first, int sb)
{
    if ((int)blockIdx.x < nb_scan) {
        scan_final_block<0, 4>(in, wave_sums, out, n, total_out, (int)blockIdx.x);
    } else if ((int)blockIdx.x < nb_scan + n_ctl) {
        depth_ctl_from_blocks(blk_minmax, nblk, ctl, (int)blockIdx.x - nb_scan, n_ctl, total_out);
    } else {
        radix_hist_block<HIST_ITEMS, 8, uint64_t, true>(items, hist, acc_first, n, 0, sb, (int)blockIdx.x - nb_scan - n_ctl, 0u, 0xFFFFFFFFu, true);
    }
}

// Passes over many blocks (nb > GSR_RADIX_PREFIX_NB; C5's tile partition has 7 866) put one small launch between the two
// kernels after all: with only the two-level sums a scatter block read ~1.5 sqrt(nb) rows (68 KB per block, 0.5 GB per pass at
// C5: tile partition 395 -> 435 us).  One workgroup turns the super-block rows into exclusive prefixes in place and writes the
// digit totals to acc[0..255]; super-blocks are then a fixed 32 blocks, so a scatter block reads 2 + (< 32) rows.
__global__ __launch_bounds__(1024) void radix_superscan_kernel(int32_t *__restrict__ acc, int nsuper, DepthPass dp)
{
    if (dp.ctl) { // a depth pass: skipped passes have nothing to scan, the first active one keeps its sums in an array of its own
        const DepthCtl c = depth_ctl_load(dp.ctl, dp.force_npass);
        if (dp.pass < c.first || c.first < dp.launched_first) return;
        if (dp.pass == c.first) acc = dp.acc_first;
    }
    // Four workgroups, 64 digits each (16 threads x 4 digits per row); 64 groups of threads share the rows, a contiguous run each:
    // at C5 (246 rows) a group has 4 rows -- one round of loads, kept in registers for the store pass -- where one workgroup of 16
    // groups walked 16 rows twice (11 us per launch, two launches per frame; now 2 x ~5).
    __shared__ int4 s_part[64][16];
    const int ql = threadIdx.x & 15, g = threadIdx.x >> 4, q = (int)blockIdx.x * 16 + ql;
    const int per = (nsuper + 63) / 64, t0 = g * per, t1 = min(nsuper, t0 + per);
    int4 *rows = reinterpret_cast<int4 *>(acc + 256) + q;
    int4 sum = make_int4(0, 0, 0, 0);
    int4 v[8];
    const bool held = per <= 8; // the group's rows stay in registers between the two passes
    if (held) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = t0 + k < t1 ? rows[(size_t)(t0 + k) * 64] : make_int4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { sum.x += v[k].x; sum.y += v[k].y; sum.z += v[k].z; sum.w += v[k].w; }
    } else {
#pragma unroll 8
        for (int t = t0; t < t1; ++t) {
            const int4 w = rows[(size_t)t * 64];
            sum.x += w.x; sum.y += w.y; sum.z += w.z; sum.w += w.w;
        }
    }
    s_part[g][ql] = sum;
    __syncthreads();
    int4 run = make_int4(0, 0, 0, 0), all = run;
#pragma unroll 16
    for (int k = 0; k < 64; ++k) {
        const int4 w = s_part[k][ql];
        if (k < g) { run.x += w.x; run.y += w.y; run.z += w.z; run.w += w.w; }
        all.x += w.x; all.y += w.y; all.z += w.z; all.w += w.w;
    }
    if (g == 0) reinterpret_cast<int4 *>(acc)[q] = all;
    if (held) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (t0 + k < t1) rows[(size_t)(t0 + k) * 64] = run;
            run.x += v[k].x; run.y += v[k].y; run.z += v[k].z; run.w += v[k].w;
        }
    } else {
        for (int tb = t0; tb < t1; tb += 8) { // eight rows at a time: loads together, then the stores
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = tb + k < t1 ? rows[(size_t)(tb + k) * 64] : make_int4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (tb + k < t1) rows[(size_t)(tb + k) * 64] = run;
                run.x += v[k].x; run.y += v[k].y; run.z += v[k].z; run.w += v[k].w;
            }
        }
    }
}

// CARRY (last depth pass only): the item's tile rectangle is fetched by id and written, with its tile count, to the
// item's final position -- so the depth-order scan and the expansion stream rect_sorted / cnt_sorted instead of each
// gathering through the sorted ids (two random 64-byte-sector reads per Gaussian become one).
struct ScatterCarry {
    const TileRect *rect;   // [n] by id
    TileRect *rect_sorted;  // [n] in output order
    int32_t *cnt_sorted;    // [n] (x1-x0)*(y1-y0) in output order
    uint32_t *id_sorted;    // [n] Gaussian ids in output order (instead of the 8-byte items: the expansion needs only these)
    int64_t n_total;        // all Gaussians, culled included (cnt_sorted behind the visible ones is zero-filled)
};

// FINAL (last pass of the tile partition only): the pass's output IS the sorted list, so instead of the items it writes what
// the reference's last two steps produce from them (forward.py:806-824 un-padding copies, :561-586 wp_identify_tile_ranges):
// the Gaussian ids go straight to point_list, and tile boundaries are written to `ranges` wherever both neighbours are in
// sight -- inside a block's run of one digit, which is contiguous in the output.  The first and last tile of every run go
// to an edge table [RADIX][nb] (a run's predecessor in the output is the previous non-empty run of the same digit, written
// by another workgroup), which ranges_fixup_kernel resolves.  This replaces a kernel that re-read all D items.
struct ScatterFinal {
    int32_t *point_list;  // [n]
    int32_t *ranges;      // [tiles * 2], pre-zeroed
    int32_t *edge_first;  // [RADIX * nb] tile of the run's first item, -1 for an empty run
    int32_t *edge_last;   // [RADIX * nb] tile of the run's last item
    int32_t *edge_pos;    // [RADIX * nb] output position of the run's first item
    int32_t *totals;      // [RADIX] items per digit, for ranges_fixup_kernel (written by block 0; a non-final pass writes them too when asked)
    int id_shift;
    // Narrowed items (two-pass partitions whose tile + id bits exceed 32, e.g. 1080p with 5 M Gaussians: 13 + 23).  The first pass
    // reads the 64-bit items (tile << 32 | id) and WRITES 32-bit ones, (tile >> bits0) << narrow_id_bits | id: the digit it has
    // just sorted by is implied by the item's position in its output (that output is grouped by it), so the second pass -- its
    // histogram and its scatter, the FINAL one -- moves 4 bytes per item instead of 8 and recovers the digit from the first
    // pass's digit totals (low_totals, low_bits): a search of <= 256 segment starts per item.
    int narrow_id_bits;          // non-final pass: > 0 = write 32-bit items with the id in this many low bits
    const int32_t *low_totals;   // FINAL: the first pass's items per digit ...
    int low_bits;                // ... and its digit width (0: the items hold the whole tile id)
};

// PACKCAP (DEPTH, not CARRY): the host allows packed depth items (DepthPass.pack_ok), so this pass may be the one that packs them and
// needs the rectangles' LDS image -- 32 KB at 4096-item chunks, which a frame that cannot pack (C5's 120 x 68 grid) must not pay for.
template <int RADIX_ITEMS, int BITS, typename ItemT, bool CARRY = false, bool FINAL = false, bool DEPTH = false, int THREADS = 256, bool LOWREC = false,
          bool PACKCAP = false>
__global__ __launch_bounds__(THREADS) void radix_scatter_kernel(const ItemT *__restrict__ in, ItemT *__restrict__ out,
                                                            const int32_t *__restrict__ hist, const int32_t *__restrict__ acc,
                                                            int64_t n, int shift, int nb, int sb, bool prefixed, int32_t *__restrict__ zero_acc,
                                                            int zero_n, ScatterCarry carry, ScatterFinal fin, DepthPass dp)
{
    constexpr int CHUNK = THREADS * RADIX_ITEMS;
    constexpr int NW = THREADS / 64; // waves per workgroup (GSR_RADIX_WIDE_WG: 8 waves of 8 items instead of 4 of 16 for the same chunk)
    constexpr int RADIX = 1 << BITS;
    __shared__ ItemT s_items[CHUNK];    // items reordered by digit
    __shared__ unsigned long long s_rect[(CARRY || PACKCAP) ? CHUNK : 1]; // the items' rectangles (raw bits), reordered with them: CARRY, and the depth pass that packs them into the items
    __shared__ int s_wcnt[NW][RADIX];              // per-wave digit counts -> per-wave start offsets
    __shared__ int s_before[RADIX];                // items of each digit in earlier blocks
    __shared__ int s_total[RADIX];                 // items of each digit in all blocks
    // thread d reads [d] of the two arrays above into registers before it writes [d] of these two: they share the space
    // (2 KB that decide between 3 and 4 workgroups per CU for 64-bit items with 8-bit digits)
    int *const s_dstart = s_total;                 // first LDS slot of each digit
    int *const s_gbase = s_before;                 // global position of the block's first item of each digit
    __shared__ int s_dcnt[FINAL ? RADIX : 1];      // FINAL: items of each digit in this block
    __shared__ uint8_t s_low[LOWREC ? CHUNK : 1];  // LOWREC (FINAL with narrowed items): each item's first-pass digit, reordered with it
    __shared__ int s_seg[LOWREC ? 257 : 1];        // ... and where each first-pass digit's segment of the input starts
    __shared__ int s_tmp[NW];
    __shared__ int s_valid_n;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t block_base = (int64_t)blockIdx.x * CHUNK;
    const int64_t wave_base = block_base + (int64_t)w * 64 * RADIX_ITEMS;
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));

    if (tid < RADIX) {
#pragma unroll
        for (int k = 0; k < NW; ++k) s_wcnt[k][tid] = 0;
        s_before[tid] = 0;
        s_total[tid] = 0;
    }
    // the next pass's accumulators (see radix_hist_kernel) are cleared here: nothing reads them before that pass's histogram
    // (also by a depth pass that then finds it has nothing to sort)
    for (int z = blockIdx.x * THREADS + tid; z < zero_n; z += gridDim.x * THREADS) zero_acc[z] = 0;
    uint32_t kmin = 0u, krange = 0u;
    bool drop_culled = false;
    bool packed = false, pack_now = false, unpack = false; // DEPTH: "packed depth items" above (all wave-uniform)
    int64_t n_load = n; // items readable in `in` (the index clamp of the loads)
    if constexpr (DEPTH) {
        const DepthCtl c = depth_ctl_load(dp.ctl, dp.force_npass);
        if (dp.pass < c.first || c.first < dp.launched_first) return;
        const int rel = dp.pass - c.first;
        if (rel == 0) acc = dp.acc_first; // (see scan_ctl_hist_kernel)
        in = reinterpret_cast<const ItemT *>(dp.buf[rel & 1]);
        out = reinterpret_cast<ItemT *>(dp.buf[(rel + 1) & 1]);
        shift = 8 * rel;
        kmin = c.min_bits; krange = c.range;
        packed = depth_items_packed(dp, c);
        pack_now = PACKCAP && packed && rel == 0; // this pass reads plain items and writes packed ones (pack_ok implies a PACKCAP launch)
        if (packed && rel > 0) { shift = 16 + 8 * (rel - 1); kmin = 0u; krange = 0xFFFFFFFFu; unpack = true; } // packed items in
        // the first active pass reads all n items and drops the culled ones (they have no tiles: nothing downstream wants them);
        // the later passes move the n_vis survivors.  Everything this pass writes lands in [0, n_vis).
        drop_culled = rel == 0;
        if (rel > 0) n = n_load = c.n_vis;
        if constexpr (CARRY) {
            // the depth-order scan runs over all N counts: those behind the survivors are zero
            const int64_t z0 = max((int64_t)c.n_vis, block_base), z1 = min(block_base + (int64_t)CHUNK, carry.n_total);
            for (int64_t z = z0 + tid; z < z1; z += THREADS) carry.cnt_sorted[z] = 0;
        }
        if (block_base >= n) return;
    }
    __syncthreads();
    if constexpr (LOWREC) { // exclusive prefix of the first pass's digit totals = the segments of this pass's input
        int tot;
        const int v = (tid < (1 << fin.low_bits)) ? fin.low_totals[tid] : 0;
        const int inc = block_incl_scan_nw<NW>(tid < 256 ? v : 0, s_tmp, &tot);
        if (tid < 256) s_seg[tid + 1] = inc;
        if (tid == 0) s_seg[0] = 0;
        __syncthreads();
    }

    // items of each digit in earlier blocks = whole super-blocks (accumulators) + the earlier blocks of the own super-block
    // (block histograms): about 2 sqrt(nb) terms per digit, spread over the block's 256 / RADIX threads per digit, coalesced
    // along the digit.  Done BEFORE the items are loaded: in flight under the ranking (or beside the item loads) the partial sums
    // and the unrolled loads' targets cost 20-30 registers (4 -> 3 waves per SIMD for 64-bit items: C5's tile partition 395 ->
    // 435 us); the one exposed round trip costs less
    int4 before_part = make_int4(0, 0, 0, 0), total_part = make_int4(0, 0, 0, 0);
    constexpr int TPT = RADIX / 4;  // threads per term: each adds four consecutive digits (one 16-byte load)
    {
        constexpr int GROUPS = THREADS / TPT;
        const int d4 = (tid % TPT) * 4, g = tid / TPT, my_sb = blockIdx.x / sb, nsuper = (nb + sb - 1) / sb;
        // (two plain strided loops, so that each unrolled body issues its loads together)
        const int32_t *rows = hist + (size_t)my_sb * sb * RADIX + d4; // the earlier blocks of the own super-block
        const int within = (int)blockIdx.x - my_sb * sb;
#pragma unroll 4
        for (int t = g; t < within; t += GROUPS) {
            const int4 v = *reinterpret_cast<const int4 *>(rows + (size_t)t * RADIX);
            before_part.x += v.x; before_part.y += v.y; before_part.z += v.z; before_part.w += v.w;
        }
        if (prefixed) { // many blocks: radix_superscan_kernel left the totals and every super-block's exclusive prefix
            if (g == 0) {
                total_part = *reinterpret_cast<const int4 *>(acc + d4);
                const int4 v = *reinterpret_cast<const int4 *>(acc + 256 + (size_t)my_sb * 256 + d4);
                before_part.x += v.x; before_part.y += v.y; before_part.z += v.z; before_part.w += v.w;
            }
        } else {
            const int32_t *sup = acc + 256 + d4;
#pragma unroll 4
            for (int t = g; t < nsuper; t += GROUPS) {
                const int4 v = *reinterpret_cast<const int4 *>(sup + (size_t)t * 256);
                total_part.x += v.x; total_part.y += v.y; total_part.z += v.z; total_part.w += v.w;
                if (t < my_sb) { before_part.x += v.x; before_part.y += v.y; before_part.z += v.z; before_part.w += v.w; }
            }
        }
    }
    {
        const int d4 = (tid % TPT) * 4;
        if (before_part.x) atomicAdd(&s_before[d4 + 0], before_part.x);
        if (before_part.y) atomicAdd(&s_before[d4 + 1], before_part.y);
        if (before_part.z) atomicAdd(&s_before[d4 + 2], before_part.z);
        if (before_part.w) atomicAdd(&s_before[d4 + 3], before_part.w);
        if (total_part.x) atomicAdd(&s_total[d4 + 0], total_part.x);
        if (total_part.y) atomicAdd(&s_total[d4 + 1], total_part.y);
        if (total_part.z) atomicAdd(&s_total[d4 + 2], total_part.z);
        if (total_part.w) atomicAdd(&s_total[d4 + 3], total_part.w);
    }

    // pass 1: rank every item among equal digits of its wave, in index order
    ItemT item[RADIX_ITEMS];
    unsigned long long rc[(CARRY || PACKCAP) ? RADIX_ITEMS : 1]; // raw TileRect bits
    int rank[RADIX_ITEMS]; // rank within (wave, digit)
    uint8_t low[LOWREC ? RADIX_ITEMS : 1]; // LOWREC: the item's first-pass digit
    bool valid_bits[RADIX_ITEMS];
    // every load is issued before the ranking starts (the ranking's branches would otherwise pin each load to its own
    // round: RADIX_ITEMS serial memory round trips per wave)
#pragma unroll
    for (int r = 0; r < RADIX_ITEMS; ++r) {
        const int64_t k = wave_base + r * 64 + lane;
        const ItemT v = in[k < n_load ? k : n_load - 1];
        item[r] = k < n_load ? v : (ItemT)~(ItemT)0;
    }
    if constexpr (CARRY) { // the random rectangle fetches are in flight during the ranking
        // raw 8-byte loads at a clamped index, no branch: as `valid ? rect[id] : {}` each fetch got its own exec-masked
        // block with a full wait behind it, i.e. RADIX_ITEMS serial random round trips
        if (!unpack) {
#pragma unroll
            for (int r = 0; r < RADIX_ITEMS; ++r) {
                const int64_t k = wave_base + r * 64 + lane;
                const uint32_t id = k < n_load ? (uint32_t)item[r] : 0u;
                rc[r] = reinterpret_cast<const unsigned long long *>(carry.rect)[id];
            }
        } else { // packed items carry their rectangle
#pragma unroll
            for (int r = 0; r < RADIX_ITEMS; ++r) rc[r] = depth_item_rect((unsigned long long)item[r]);
        }
    } else if constexpr (PACKCAP) {
        if (pack_now) { // the items are still in id order here (item k is Gaussian k): the rectangles are a coalesced read
#pragma unroll
            for (int r = 0; r < RADIX_ITEMS; ++r) {
                const int64_t k = wave_base + r * 64 + lane;
                rc[r] = reinterpret_cast<const unsigned long long *>(carry.rect)[k < n_load ? k : n_load - 1];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RADIX_ITEMS; ++r) {
        const int64_t k = wave_base + r * 64 + lane;
        bool valid = k < n;
        if constexpr (DEPTH) valid = valid && !(drop_culled && (uint32_t)(item[r] >> 32) == 0xFFFFFFFFu);
        valid_bits[r] = valid;
        if constexpr (LOWREC) {
            int p = 0; // the last segment that starts at or before input position k (empty segments share their successor's start)
#pragma unroll
            for (int step = 128; step >= 1; step >>= 1)
                if (s_seg[p + step] <= (int)min(k, n - 1)) p += step;
            low[r] = (uint8_t)p;
        }
        const int d = radix_digit<DEPTH, BITS>(item[r], shift, kmin, krange);
        // lanes holding the same digit ("match any"): a lane differs from me in bit b where ballot(bit b) XOR (my bit b
        // replicated) is set; OR over the bits, complement.  Written on 32-bit halves with the replicated bit as one signed
        // bit-field extract, so a digit bit costs 1 extract + 1 compare + 2 xor + 2 or instead of the 9 operations hipcc
        // makes of `peers &= bit ? m : ~m`.
        unsigned int diff_lo = 0u, diff_hi = 0u;
#pragma unroll
        for (int b = 0; b < BITS; ++b) {
            const unsigned int rep = (unsigned int)__builtin_amdgcn_sbfe(d, b, 1); // 0 or 0xFFFFFFFF
            const unsigned long long m = __ballot(rep != 0u);
            diff_lo |= (unsigned int)m ^ rep;
            diff_hi |= (unsigned int)(m >> 32) ^ rep;
        }
        unsigned long long peers = ~(((unsigned long long)diff_hi << 32) | diff_lo) & __ballot(valid);
        if (!valid) peers = 0ull; // invalid lanes rank nothing
        const int before = __popcll(peers & lt_mask);
        // every lane reads its digit's running count (lanes of one digit read the same word: an LDS broadcast), then the
        // first of them adds the group's size -- a wave's LDS operations execute in order, so no lane sees the update early
        const int old = valid ? s_wcnt[w][d] : 0;
        if (valid && before == 0) s_wcnt[w][d] = old + __popcll(peers);
        rank[r] = old + before;
    }
    __syncthreads();

    // per digit: prefix over the 4 waves, block total, then exclusive scan over digits
    {
        const int d = tid;
        const bool own = d < RADIX; // thread d owns digit d
        int run = 0;
        if (own) {
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const int c = s_wcnt[k][d];
                s_wcnt[k][d] = run;
                run += c;
            }
        }
        int tot;
        const int inc = block_incl_scan_nw<NW>(run, s_tmp, &tot);
        if (tid == 0) s_valid_n = tot; // items of this block that take part (all of its chunk, unless culled ones were dropped)
        // digit base over the whole array = sum of totals of smaller digits; then the items of this digit in earlier blocks:
        // whole super-blocks from the accumulators, the rest of the own super-block from the block histograms
        const int td = own ? s_total[d] : 0;
        const int before = own ? s_before[d] : 0;
        const int tinc = block_incl_scan_nw<NW>(td, s_tmp, &tot);
        if (own) {
            s_dstart[d] = inc - run;
            s_gbase[d] = tinc - td + before;
            if constexpr (!FINAL && sizeof(ItemT) == 8 && !DEPTH) {
                if (fin.totals && blockIdx.x == 0) fin.totals[d] = td; // (a narrowing pass: the final pass finds each item's digit of THIS pass from these)
            }
            if constexpr (FINAL) {
                if (blockIdx.x == 0) fin.totals[d] = td;
                s_dcnt[d] = run;
                if (run == 0) fin.edge_first[(size_t)d * nb + blockIdx.x] = -1;
            }
        }
    }
    __syncthreads();

    // pass 2: reorder in LDS by digit (stable)
#pragma unroll
    for (int r = 0; r < RADIX_ITEMS; ++r) {
        if (valid_bits[r]) {
            const int d = radix_digit<DEPTH, BITS>(item[r], shift, kmin, krange);
            const int slot = s_dstart[d] + s_wcnt[w][d] + rank[r];
            s_items[slot] = item[r];
            if constexpr (LOWREC) s_low[slot] = low[r];
            if constexpr (CARRY) s_rect[slot] = rc[r];
            else if constexpr (PACKCAP) { if (pack_now) s_rect[slot] = rc[r]; }
        }
    }
    __syncthreads();

    // pass 3: contiguous runs out to global memory
    const int valid_n = s_valid_n;
#pragma unroll 4
    for (int r = 0; r < RADIX_ITEMS; ++r) {
        const int slot = r * THREADS + tid;
        if (slot < valid_n) {
            const ItemT it = s_items[slot];
            const int d = radix_digit<DEPTH, BITS>(it, shift, kmin, krange);
            const int64_t pos = (int64_t)s_gbase[d] + (slot - s_dstart[d]);
            if constexpr (FINAL) {
                uint32_t tile = (uint32_t)(it >> fin.id_shift);
                if constexpr (LOWREC) tile = (tile << fin.low_bits) | (uint32_t)s_low[slot];
                fin.point_list[pos] = (int32_t)(uint32_t)(it & (((ItemT)1 << fin.id_shift) - 1));
                const int rel = slot - s_dstart[d];
                const size_t e = (size_t)d * nb + blockIdx.x;
                if (rel > 0) {
                    uint32_t prev = (uint32_t)(s_items[slot - 1] >> fin.id_shift);
                    if constexpr (LOWREC) prev = (prev << fin.low_bits) | (uint32_t)s_low[slot - 1];
                    if (prev != tile) {
                        fin.ranges[2 * prev + 1] = (int32_t)pos;
                        fin.ranges[2 * tile] = (int32_t)pos;
                    }
                } else {
                    fin.edge_first[e] = (int32_t)tile;
                    fin.edge_pos[e] = (int32_t)pos;
                }
                if (rel == s_dcnt[d] - 1) fin.edge_last[e] = (int32_t)tile;
            } else if constexpr (CARRY) {
                carry.id_sorted[pos] = unpack ? ((uint32_t)it & 0xFFFFFFu) : (uint32_t)it; // the last depth pass: ids, rectangles and counts leave, not the items
            } else {
                bool narrowed = false;
                if constexpr (PACKCAP) {
                    if (pack_now) { // key = the reduced depth key this pass's digit came from; its low byte is spent
                        const uint32_t key = min((uint32_t)(it >> 32) - kmin, krange);
                        out[pos] = (ItemT)depth_item_pack(key, s_rect[slot], (uint32_t)it);
                        narrowed = true;
                    }
                }
                if constexpr (sizeof(ItemT) == 8 && !DEPTH) { // (uniform) the digit just sorted by leaves the item: see ScatterFinal
                    if (fin.narrow_id_bits) {
                        reinterpret_cast<uint32_t *>(out)[pos] = (uint32_t)(((it >> (shift + BITS)) << fin.narrow_id_bits) | (it & (((ItemT)1 << fin.narrow_id_bits) - 1)));
                        narrowed = true;
                    }
                }
                if (!narrowed) out[pos] = it;
            }
            if constexpr (CARRY) {
                const unsigned long long q = s_rect[slot]; // TileRect {x0, y0, x1, y1}, 16 bits each, little endian
                reinterpret_cast<unsigned long long *>(carry.rect_sorted)[pos] = q;
                const int x0 = (int)(q & 0xFFFF), y0 = (int)((q >> 16) & 0xFFFF), x1 = (int)((q >> 32) & 0xFFFF), y1 = (int)(q >> 48);
                carry.cnt_sorted[pos] = (x1 - x0) * (y1 - y0);
            }
        }
    }
}

// One 1024-thread workgroup per digit: joins consecutive non-empty runs of the digit (ScatterFinal's edge table).  Between
// two digits the tile always changes (the digit is the top bits of the tile id), so a digit's first run always opens a
// range at the digit's first position and its last run closes one at the digit's end; inside the digit a range boundary
// sits wherever a run's first tile differs from the previous non-empty run's last tile.
__global__ __launch_bounds__(1024) void ranges_fixup_kernel(const int32_t *__restrict__ edge_first, const int32_t *__restrict__ edge_last,
                                                            const int32_t *__restrict__ edge_pos, const int32_t *__restrict__ totals, int nb,
                                                            int radix, int32_t *__restrict__ ranges)
{
    __shared__ int s_last[16]; // per wave: last tile of its last non-empty run, -1 if it has none
    const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int total = totals[d];
    if (total == 0) return;
    int digit_start = 0;
    for (int k = lane; k < d; k += 64) digit_start += totals[k];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) digit_start += __shfl_xor(digit_start, o, 64);
    const size_t row = (size_t)d * nb;
    // each wave owns a contiguous slice of the digit's nb runs
    const int per = (nb + 15) / 16;
    const int b0 = wv * per, b1 = min(nb, b0 + per);
    // pass 1: this wave's last non-empty run
    int my_last = -1;
    for (int base = b0; base < b1; base += 64) {
        const int b = base + lane;
        const bool ne = b < b1 && edge_first[row + b] >= 0;
        const unsigned long long m = __ballot(ne);
        if (m) {
            const int hi = 63 - __builtin_clzll(m);
            my_last = __shfl(ne ? edge_last[row + b] : -1, hi, 64);
        }
    }
    if (lane == 0) s_last[wv] = my_last;
    __syncthreads();
    int carry = -1; // last tile of the nearest non-empty run before this wave's slice
    for (int k = wv - 1; k >= 0; --k)
        if (s_last[k] >= 0) { carry = s_last[k]; break; }
    // pass 2: boundaries
    for (int base = b0; base < b1; base += 64) {
        const int b = base + lane;
        const bool ne = b < b1 && edge_first[row + b] >= 0;
        const int first = ne ? edge_first[row + b] : -1, last = ne ? edge_last[row + b] : -1, pos = ne ? edge_pos[row + b] : 0;
        const unsigned long long m = __ballot(ne);
        const unsigned long long below = m & (lane == 0 ? 0ull : (~0ull >> (64 - lane)));
        const int src = below ? 63 - __builtin_clzll(below) : 0;
        const int prev_in_chunk = __shfl(last, src, 64);
        const int prev = below ? prev_in_chunk : carry;
        if (ne) {
            if (prev < 0) ranges[2 * first] = pos;                 // the digit's first run: pos == digit_start
            else if (prev != first) { ranges[2 * prev + 1] = pos; ranges[2 * first] = pos; }
        }
        if (m) carry = __shfl(last, 63 - __builtin_clzll(m), 64);
    }
    // the digit's last run closes its range at the digit's end
    bool later = false;
    for (int k = wv + 1; k < 16; ++k) later = later || s_last[k] >= 0;
    if (!later && my_last >= 0 && lane == 0) ranges[2 * my_last + 1] = digit_start + total;
}

// ---------------------------------------------------------------------------------------------
// expansion: depth-ordered Gaussians -> (tile << 32 | id) items, row-major tile walk
// (same walk as reference forward.py:546-548: y outer, x inner)
// ---------------------------------------------------------------------------------------------
// Load-balanced: a wave owns 64 consecutive depth-sorted Gaussians, whose output range
// [doff[k0], doff[k0+64]) is contiguous.  Lanes walk that range 64 items at a time (fully coalesced
// 8-byte stores) and find each item's owner by a 6-step binary search over the wave's 64 offsets in
// LDS, so a Gaussian covering thousands of tiles costs no more per item than one covering four.
// G = Gaussians per wave: 64 when a Gaussian expands to a handful of tiles (C3: 7.75), 8 when it expands to dozens or hundreds
// (the reference trainer's initial point set: 5 000 Gaussians of scale 0.1 cover 130 tiles each -- at 64 per wave that was 78 waves
// for 649 k items on a 256-CU chip, 48 us; api.hip picks G from D / N).
template <typename ItemT, int G>
__global__ __launch_bounds__(256) void expand_kernel(const uint32_t *__restrict__ id_sorted, const int32_t *__restrict__ doff,
                                                     const TileRect *__restrict__ rect, ItemT *__restrict__ tile_items, int64_t n,
                                                     int grid_x, int64_t D, int id_shift, int32_t *__restrict__ ranges, int ranges_n,
                                                     int32_t *__restrict__ zero_acc, int zero_n, int32_t *__restrict__ zero_b, int zero_b_n, int bo_flag)
{
    // the accumulators of the first partition pass (radix_hist_kernel) are cleared here, like the ranges below
    for (int64_t z = (int64_t)blockIdx.x * 256 + threadIdx.x; z < zero_n; z += (int64_t)gridDim.x * 256) zero_acc[z] = 0;
    // ... and the counters of the block-order queues the forward blend fills (GsrBinning.block_order)
    // (and the header's `filed` flag: will the forward blend file them?)
    for (int64_t z = (int64_t)blockIdx.x * 256 + threadIdx.x; z < zero_b_n; z += (int64_t)gridDim.x * 256) zero_b[z] = (z == GSR_BO_FLAG) ? bo_flag : 0;
    // also clears the tile ranges (filled later by ranges_kernel; untouched tiles must read (0,0)): saves a memset launch
    for (int64_t z = (int64_t)blockIdx.x * 256 + threadIdx.x; z < ranges_n; z += (int64_t)gridDim.x * 256) ranges[z] = 0;
    __shared__ int s_off[4][G];
    __shared__ TileRect s_rect[4][G];
    __shared__ uint32_t s_gid[4][G];
    __shared__ float s_inv[4][G]; // 1 / (rectangle width in tiles)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t k0 = ((int64_t)blockIdx.x * 4 + w) * G;
    const int64_t k = k0 + lane;
    int off = (int)D;
    TileRect rc = {0, 0, 0, 0};
    uint32_t id = 0;
    // (requested with the wave's other loads: read where it is used, after the barrier, it was a second serial round trip)
    int next_off = (k0 + G < n) ? doff[k0 + G] : (int)D;
    asm volatile("" : "+v"(next_off));
    if (lane < G && k < n) {
        id = id_sorted[k];
        off = doff[k];
        rc = rect[k]; // rectangles arrive in depth order (carried by the last sort pass); culled Gaussians have empty ones
    }
    if (lane < G) {
        s_off[w][lane] = off;
        s_rect[w][lane] = rc;
        s_gid[w][lane] = id;
        s_inv[w][lane] = rc.x1 > rc.x0 ? 1.0f / (float)((int)rc.x1 - (int)rc.x0) : 0.0f;
    }
    __syncthreads();
    if (k0 >= n) return;
    const int begin = s_off[w][0];
    // never write past the caller's D, even if it under-reports the count gsr_forward_count returned
    const int end = min((int)D, next_off);
    for (int j = begin + lane; j < end; j += 64) {
        int lo = 0; // last k with off[k] <= j (zero-count Gaussians share their successor's offset)
#pragma unroll
        for (int step = G / 2; step >= 1; step >>= 1)
            if (s_off[w][lo + step] <= j) lo += step;
        const TileRect r = s_rect[w][lo];
        const int t = j - s_off[w][lo];
        const int wd = (int)r.x1 - (int)r.x0;
        // row-major walk: y = t / wd, x = t % wd (reference forward.py:546-548).  t < 2^24 (a rectangle has at most
        // 4096 x 4096 tiles), so the quotient comes from one float multiply by the Gaussian's 1/wd and a +-1 correction
        // instead of the ~25-instruction integer division
        int y = (int)((float)t * s_inv[w][lo]);
        int x = t - y * wd;
        if (x < 0) { --y; x += wd; }
        else if (x >= wd) { ++y; x -= wd; }
        tile_items[j] = (ItemT)(((ItemT)(uint32_t)(((int)r.y0 + y) * grid_x + (int)r.x0 + x) << id_shift) | (ItemT)s_gid[w][lo]);
    }
}

// ---------------------------------------------------------------------------------------------
// expansion, by OUTPUT block (the product path; the kernel above is the alternative path of GSR_DEBUG bit 9)
// ---------------------------------------------------------------------------------------------
// The depth-order offsets are an exclusive scan of the sorted tile counts.  As a device-wide scan that was two launches
// (12 us at C3 for 4 MB) whose output, 4 bytes per Gaussian, the expansion read straight back.  Now one small streaming kernel
// leaves two levels of partial sums and the expansion does the rest itself: per workgroup of this kernel (4096 depth-sorted
// Gaussians) the total (sum4096), and per 256 Gaussians the exclusive prefix INSIDE its 4096 (pre256).  (A single-kernel
// version in which the last workgroup to finish turned the sums into global prefixes was tried first: sums written by one
// XCD and read by another need device-scope release/acquire, i.e. whole-L2 write-backs -- 72 us -- or atomics around the
// caches -- 13 us at C3, 39 us at C5; this kernel is 5 us.)
// It also clears what the expansion's workgroups can no longer clear for the kernel behind them, because they now ADD to
// it: the first partition pass's accumulators -- and the tile ranges and the block-order header, as the expansion did.
__global__ __launch_bounds__(1024) void depth_block_offsets_kernel(const int32_t *__restrict__ cnt_sorted, int64_t n, int32_t *__restrict__ pre256,
                                                                   int nsum, int32_t *__restrict__ sum4096, int32_t *__restrict__ ranges, int ranges_n,
                                                                   int32_t *__restrict__ zero_acc, int zero_n, int32_t *__restrict__ zero_b, int zero_b_n,
                                                                   int bo_flag)
{
    for (int64_t z = (int64_t)blockIdx.x * 1024 + threadIdx.x; z < zero_n; z += (int64_t)gridDim.x * 1024) zero_acc[z] = 0;
    for (int64_t z = (int64_t)blockIdx.x * 1024 + threadIdx.x; z < zero_b_n; z += (int64_t)gridDim.x * 1024) zero_b[z] = (z == GSR_BO_FLAG) ? bo_flag : 0;
    for (int64_t z = (int64_t)blockIdx.x * 1024 + threadIdx.x; z < ranges_n; z += (int64_t)gridDim.x * 1024) ranges[z] = 0;
    __shared__ int s_wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int blk = (int)blockIdx.x * 16 + w;
    int v = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t k = (int64_t)blk * 256 + r * 64 + lane;
        if (k < n) v += cnt_sorted[k];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if (lane == 0) s_wsum[w] = v;
    __syncthreads();
    if (tid < 16) {
        int before = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int t = s_wsum[k];
            if (k < tid) before += t;
            tot += t;
        }
        if ((int)blockIdx.x * 16 + tid < nsum) pre256[blockIdx.x * 16 + tid] = before;
        if (tid == 0) sum4096[blockIdx.x] = tot;
    }
}

// One workgroup per radix block of the OUTPUT: workgroup b writes the items [b * chunk, (b + 1) * chunk) -- whatever Gaussians
// they belong to -- so (1) the work per workgroup is the same whether a Gaussian covers four tiles or four thousand (the
// wave-per-64-Gaussians kernel above needed a second shape for the trainer's initial point set), and (2) the workgroup has seen
// exactly one block of the first partition pass and leaves that pass's histogram row and super-block sums behind: the pass's
// histogram kernel, which re-read all D items, is not launched.  It finds the 256-Gaussian block holding its first item by a
// two-level search of pre256 (two dependent loads), then walks blocks of 256 Gaussians: rectangle -> count -> offsets by a
// block scan on top of pre256[B], and every thread places items by an 8-step search of the 256 offsets in LDS.
template <typename ItemT>
__global__ __launch_bounds__(256) void expand_blocks_kernel(const uint32_t *__restrict__ id_sorted, const int32_t *__restrict__ cnt_sorted,
                                                            const int32_t *__restrict__ pre256, const int32_t *__restrict__ sum4096, int nsum,
                                                            const TileRect *__restrict__ rect, ItemT *__restrict__ tile_items, int64_t n, int grid_x,
                                                            int64_t D, int id_shift, int chunk, int digit_mask, int32_t *__restrict__ hist,
                                                            int32_t *__restrict__ acc, int sb)
{
    __shared__ int s_off[257]; // [256] = the end of the block's items
    __shared__ TileRect s_rect[256];
    __shared__ uint32_t s_gid[256];
    __shared__ float s_inv[256]; // 1 / (rectangle width in tiles)
    __shared__ int s_h[256];
    __shared__ int s_tmp[4];
    const int tid = threadIdx.x;
    const int lo = (int)((int64_t)blockIdx.x * chunk), hi = (int)min(D, (int64_t)lo + chunk); // never past the caller's D
    s_h[tid] = 0;
    // The 256-Gaussian block B0 holding the owner of item lo (the last B whose offset is <= lo: empty Gaussians share their
    // successor's offset) and that block's offset, from the two levels depth_block_offsets_kernel left: scan the 4096-totals
    // (every workgroup does; there are N / 4096 of them), pick the 4096 S, then pick among its sixteen 256-prefixes.
    __shared__ int s_pick[2];
    int B0, base0;
    {
        const int nsuper = (nsum + 15) >> 4, per = (nsuper + 255) / 256;
        const int q0 = min(nsuper, tid * per), q1 = min(nsuper, q0 + per);
        int strip = 0;
        for (int q = q0; q < q1; ++q) strip += sum4096[q];
        int tot;
        const int excl = block_incl_scan_256(strip, s_tmp, &tot) - strip; // offset of 4096-block q0
        int cnt = 0, run = excl, at = excl; // how many of my 4096-blocks start at or before lo, and the offset of the last of them
        for (int q = q0; q < q1; ++q) {
            if (run <= lo) { ++cnt; at = run; }
            run += sum4096[q];
        }
        int csum;
        const int cincl = block_incl_scan_256(cnt, s_tmp, &csum);
        // the 4096-blocks starting at or before lo are a prefix of all of them: the last one, S = csum - 1, is in the strip of
        // the one thread whose inclusive count reaches csum with cnt > 0
        if (cnt > 0 && cincl == csum) { s_pick[0] = csum - 1; s_pick[1] = at; }
        __syncthreads();
        const int S = s_pick[0], sbase = s_pick[1];
        const int b = S * 16 + tid;
        const int off = (tid < 16 && b < nsum) ? sbase + pre256[b] : 0;
        const int t2 = __syncthreads_count(tid < 16 && b < nsum && off <= lo); // >= 1: the 4096's first block starts at sbase <= lo
        if (tid == t2 - 1) s_pick[1] = off;
        B0 = S * 16 + t2 - 1;
        __syncthreads();
        base0 = s_pick[1];
    }
    int base = base0;
    // (software pipeline: the next block's rectangle and id are requested before this block's items are placed)
    // (the count comes from cnt_sorted, the array the two levels of sums were made from -- it is zero behind the visible
    // Gaussians, where rect_sorted and id_sorted hold whatever an earlier frame left: the sort drops the culled ones)
    TileRect rc = {0, 0, 0, 0};
    uint32_t id = 0;
    int cnt = 0;
    {
        const int64_t k = (int64_t)B0 * 256 + tid;
        if (k < n) { rc = rect[k]; id = id_sorted[k]; cnt = cnt_sorted[k]; } // in depth order (carried by the last sort pass)
    }
    for (int B = B0; B < nsum; ++B) {
        const int wd = cnt > 0 ? (int)rc.x1 - (int)rc.x0 : 0;
        int tot;
        const int inc = block_incl_scan_256(cnt, s_tmp, &tot); // (its barriers also fence the previous block's readers of the arrays below)
        s_off[tid] = base + inc - cnt;
        s_rect[tid] = rc;
        s_gid[tid] = id;
        s_inv[tid] = wd > 0 ? 1.0f / (float)wd : 0.0f;
        const int end = base + tot;
        if (tid == 255) s_off[256] = end;
        int next_cnt = 0;
        if (end < hi && B + 1 < nsum) {
            const int64_t k = (int64_t)(B + 1) * 256 + tid;
            rc = TileRect{0, 0, 0, 0};
            id = 0;
            next_cnt = 0;
            if (k < n) { rc = rect[k]; id = id_sorted[k]; next_cnt = cnt_sorted[k]; }
        }
        __syncthreads();
        const int j0 = max(lo, base), j1 = min(hi, end);
        // 32 bytes of consecutive items per thread (eight 4-byte or four 8-byte items; a wave writes 2 KB of consecutive items as
        // 16-byte pieces -- 64 bytes per thread measured 25 % slower at C5): one 8-step search of the offsets for the thread's first
        // item, then the walk is incremental -- next column, next row, next Gaussian -- instead of a search and a quotient per item
        constexpr int PT = 32 / (int)sizeof(ItemT);
        for (int jb = (j0 & ~(PT - 1)) + tid * PT; jb < j1; jb += 256 * PT) {
            const int jf = max(jb, j0); // the thread's first item inside [j0, j1)
            int p = 0; // last entry with off <= jf
#pragma unroll
            for (int step = 128; step >= 1; step >>= 1)
                if (s_off[p + step] <= jf) p += step;
            TileRect r = s_rect[p];
            uint32_t gid = s_gid[p];
            int next_off = s_off[p + 1];
            int rw = (int)r.x1 - (int)r.x0;
            // row-major walk: y = t / rw, x = t % rw (reference forward.py:546-548); t < 2^24, so the quotient comes from one float
            // multiply by the Gaussian's 1 / rw and a +-1 correction
            const int t = jf - s_off[p];
            int y = (int)((float)t * s_inv[p]);
            int x = t - y * rw;
            if (x < 0) { --y; x += rw; }
            else if (x >= rw) { ++y; x -= rw; }
            ItemT outv[PT];
#pragma unroll
            for (int u = 0; u < PT; ++u) {
                const int j = jb + u;
                outv[u] = 0;
                if (j >= jf && j < j1) {
                    if (j >= next_off) { // the next Gaussian with items (empty ones share their successor's offset)
                        do { ++p; next_off = s_off[p + 1]; } while (j >= next_off);
                        r = s_rect[p];
                        gid = s_gid[p];
                        rw = (int)r.x1 - (int)r.x0;
                        x = 0; y = 0;
                    }
                    const uint32_t tile = (uint32_t)(((int)r.y0 + y) * grid_x + (int)r.x0 + x);
                    outv[u] = (ItemT)(((ItemT)tile << id_shift) | (ItemT)gid);
                    atomicAdd(&s_h[tile & (uint32_t)digit_mask], 1);
                    if (++x == rw) { x = 0; ++y; }
                }
            }
            if (jb >= j0 && jb + PT <= j1) { // whole and 32-byte aligned: 16-byte stores
                constexpr int PER = 16 / (int)sizeof(ItemT);
                uint4 *dst = reinterpret_cast<uint4 *>(tile_items + jb);
#pragma unroll
                for (int q = 0; q < PT / PER; ++q) {
                    uint4 v;
                    if constexpr (sizeof(ItemT) == 4) v = make_uint4((uint32_t)outv[4 * q], (uint32_t)outv[4 * q + 1], (uint32_t)outv[4 * q + 2], (uint32_t)outv[4 * q + 3]);
                    else v = make_uint4((uint32_t)outv[2 * q], (uint32_t)((uint64_t)outv[2 * q] >> 32), (uint32_t)outv[2 * q + 1], (uint32_t)((uint64_t)outv[2 * q + 1] >> 32));
                    dst[q] = v;
                }
            } else {
#pragma unroll
                for (int u = 0; u < PT; ++u)
                    if (jb + u >= j0 && jb + u < j1) tile_items[jb + u] = outv[u];
            }
        }
        if (end >= hi) break;
        base = end; // the next block starts where this one ends
        cnt = next_cnt;
    }
    __syncthreads();
    // what radix_hist_kernel would have left for this block of the first partition pass
    if (tid <= digit_mask) {
        const int c = s_h[tid];
        hist[(size_t)blockIdx.x * (digit_mask + 1) + tid] = c;
        if (c) atomicAdd(&acc[256 + ((int)blockIdx.x / sb) * 256 + tid], c);
    }
}

// ---------------------------------------------------------------------------------------------
// small scenes: the whole depth stage in ONE workgroup
// ---------------------------------------------------------------------------------------------
// Up to GSR_SMALL_SORT_N Gaussians (the reference trainer starts with 5 000): ten launches of 5 us each -- four histogram /
// scatter pairs and the two-kernel depth-order scan -- to order a few thousand items is all latency.  One 1024-thread workgroup
// does the same stable LSD radix passes with both item buffers in LDS (8-bit digits of the reduced key, only the passes the
// depth range needs: the extremes are reduced in the kernel itself), then carries rectangle and count to the sorted position and
// scans the counts into the depth-order offsets.  (A bitonic network over the items -- they are distinct, so their ascending
// order as 64-bit integers IS the stable order -- was tried first: 91 barrier-separated LDS sweeps on one CU, 85 us.)
constexpr int SMALL_SORT_MAX = GSR_SMALL_SORT_N;
constexpr int SMALL_ROUNDS = SMALL_SORT_MAX / 1024; // items per thread
__global__ __launch_bounds__(1024) void depth_sort_small_kernel(const uint64_t *__restrict__ items, int n, const TileRect *__restrict__ rect,
                                                                uint32_t *__restrict__ id_sorted, TileRect *__restrict__ rect_sorted,
                                                                int32_t *__restrict__ cnt_sorted, int32_t *__restrict__ doff)
{
    __shared__ uint64_t s_buf[2][SMALL_SORT_MAX];
    __shared__ int s_wcnt[16][256]; // per-wave digit counts -> per-wave start offsets within the digit
    __shared__ int s_dstart[256];   // first slot of each digit
    __shared__ uint32_t s_lo[16], s_hi[16];
    __shared__ int s_wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    // load + the visible depth range (culled items carry 0xFFFFFFFF: they end up last, in id order)
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (int i = tid; i < SMALL_SORT_MAX; i += 1024) {
        const uint64_t it = i < n ? items[i] : ~0ull;
        s_buf[0][i] = it;
        const uint32_t bits = (uint32_t)(it >> 32);
        if (bits != 0xFFFFFFFFu) { lo = min(lo, bits); hi = max(hi, bits); }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, d, 64));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, d, 64));
    }
    if (lane == 0) { s_lo[w] = lo; s_hi[w] = hi; }
    __syncthreads();
    for (int k = 0; k < 16; ++k) { lo = min(lo, s_lo[k]); hi = max(hi, s_hi[k]); }
    const uint32_t kmin = lo > hi ? 0xFFFFFFFFu : lo, krange = lo > hi ? 0u : hi - lo + 1u;
    const int npass = max(1, (32 - __builtin_clz(krange | 1u) + 7) / 8);
    const int per_wave = SMALL_SORT_MAX / 16; // each wave owns 512 consecutive slots, walked in rounds of 64 (index order)
    int cur = 0;
    for (int pass = 0; pass < npass; ++pass) {
        const int shift = 8 * pass;
        for (int d = tid; d < 16 * 256; d += 1024) (&s_wcnt[0][0])[d] = 0;
        __syncthreads();
        int rank[SMALL_ROUNDS], dig[SMALL_ROUNDS];
#pragma unroll
        for (int r = 0; r < SMALL_ROUNDS; ++r) {
            const int i = w * per_wave + r * 64 + lane;
            const bool valid = i < n;
            const int d = radix_digit<true, 8>(s_buf[cur][i], shift, kmin, krange);
            unsigned int diff_lo = 0u, diff_hi = 0u;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const unsigned int rep = (unsigned int)__builtin_amdgcn_sbfe(d, b, 1);
                const unsigned long long mm = __ballot(rep != 0u);
                diff_lo |= (unsigned int)mm ^ rep;
                diff_hi |= (unsigned int)(mm >> 32) ^ rep;
            }
            unsigned long long peers = ~(((unsigned long long)diff_hi << 32) | diff_lo) & __ballot(valid);
            if (!valid) peers = 0ull;
            const int before = __popcll(peers & lt_mask);
            const int old = valid ? s_wcnt[w][d] : 0;
            if (valid && before == 0) s_wcnt[w][d] = old + __popcll(peers);
            rank[r] = old + before;
            dig[r] = d;
        }
        __syncthreads();
        // per digit: prefix over the 16 waves, then an exclusive scan over the digits (threads 0 .. 255 = 4 waves)
        int run = 0;
        if (tid < 256) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int c = s_wcnt[k][tid];
                s_wcnt[k][tid] = run;
                run += c;
            }
        }
        const int incl = wave_incl_scan(run);
        if (tid < 256 && lane == 63) s_wsum[w] = incl;
        __syncthreads();
        if (tid < 256) {
            int base = incl - run;
            for (int k = 0; k < w; ++k) base += s_wsum[k];
            s_dstart[tid] = base;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SMALL_ROUNDS; ++r) {
            const int i = w * per_wave + r * 64 + lane;
            if (i < n) s_buf[cur ^ 1][s_dstart[dig[r]] + s_wcnt[w][dig[r]] + rank[r]] = s_buf[cur][i];
        }
        __syncthreads();
        cur ^= 1;
    }
    // carry + exclusive scan of the tile counts in sorted order: thread t owns the positions [t * per, t * per + per)
    const int per = (n + 1023) >> 10;
    int cnt[SMALL_ROUNDS], sum = 0;
#pragma unroll
    for (int q = 0; q < SMALL_ROUNDS; ++q) {
        const int pos = tid * per + q;
        cnt[q] = 0;
        if (q < per && pos < n) {
            const uint32_t id = (uint32_t)s_buf[cur][pos];
            const unsigned long long r = reinterpret_cast<const unsigned long long *>(rect)[id];
            const int x0 = (int)(r & 0xFFFF), y0 = (int)((r >> 16) & 0xFFFF), x1 = (int)((r >> 32) & 0xFFFF), y1 = (int)(r >> 48);
            cnt[q] = (x1 - x0) * (y1 - y0);
            id_sorted[pos] = id;
            reinterpret_cast<unsigned long long *>(rect_sorted)[pos] = r;
            cnt_sorted[pos] = cnt[q];
        }
        sum += cnt[q];
    }
    const int incl = wave_incl_scan(sum);
    if (lane == 63) s_wsum[w] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int k = 0; k < w; ++k) base += s_wsum[k];
#pragma unroll
    for (int q = 0; q < SMALL_ROUNDS; ++q) {
        const int pos = tid * per + q;
        if (q < per && pos < n) doff[pos] = base;
        base += cnt[q];
    }
}

} // namespace

hipError_t gsr_launch_scan(const int32_t *in, const uint64_t *items, int32_t *out, int32_t *block_tmp, int64_t n, int mode,
                           int32_t *total_out, bool sums_per_256_ready, hipStream_t s, const uint32_t *blk_minmax, void *depth_ctl)
{
    if (n <= 0) return hipSuccess;
    const int nw = (int)gsr_div_up(n, GSR_SCAN_WAVE_ITEMS); // wave-sized units; block_tmp holds one sum per unit
    const int nb = (nw + 3) / 4;
    (void)items;
    DepthCtlRaw *ctl = (DepthCtlRaw *)depth_ctl; // only the id-order scan behind preprocess is asked to fill it (one workgroup here)
    const int nblk = (int)gsr_div_up(n, 256);
    if (mode == 0 && sums_per_256_ready) { // block_tmp already holds one sum per 256 items (preprocess.hip)
        hipLaunchKernelGGL((scan_final_kernel<0, 4>), dim3(nb + (ctl ? 1 : 0)), dim3(256), 0, s, in, block_tmp, out, n, total_out, blk_minmax, nblk, ctl);
    } else if (mode == 0) {
        hipLaunchKernelGGL(scan_reduce_kernel<0>, dim3(nb), dim3(256), 0, s, in, block_tmp, n);
        hipLaunchKernelGGL((scan_final_kernel<0, 1>), dim3(nb + (ctl ? 1 : 0)), dim3(256), 0, s, in, block_tmp, out, n, total_out, blk_minmax, nblk, ctl);
    } else if (mode == 2) {
        hipLaunchKernelGGL(scan_reduce_kernel<2>, dim3(nb), dim3(256), 0, s, in, block_tmp, n);
        hipLaunchKernelGGL((scan_final_kernel<2, 1>), dim3(nb), dim3(256), 0, s, in, block_tmp, out, n, total_out, (const uint32_t *)nullptr, 0,
                           (DepthCtlRaw *)nullptr);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// chunk size and super-block size of a pass over n items (gsr_internal.h: gsr_radix_blocks, gsr_radix_sb)
struct PassGeom {
    int tier;      // 0: GSR_RADIX_TINY_CHUNK, 1: GSR_RADIX_SMALL_CHUNK, 2: GSR_RADIX_CHUNK (gsr_internal.h)
    bool prefixed;
    int nb, sb, chunk;
};
static PassGeom pass_geom(int64_t n)
{
    PassGeom g;
    g.tier = (gsr_debug_flags & 64) ? 2 : n <= GSR_RADIX_TINY_N ? 0 : n <= GSR_RADIX_SMALL_N ? 1 : 2; // GSR_DEBUG bit 6: the large-n path at any n (tests)
    g.chunk = g.tier == 0 ? GSR_RADIX_TINY_CHUNK : g.tier == 1 ? GSR_RADIX_SMALL_CHUNK : GSR_RADIX_CHUNK;
    g.nb = (int)gsr_div_up(n, g.chunk);
    g.sb = gsr_radix_sb(g.nb);
    g.prefixed = g.nb > GSR_RADIX_PREFIX_NB || (gsr_debug_flags & 128); // GSR_DEBUG bit 7: at any block count (tests)
    return g;
}

template <int BITS, typename ItemT, bool CARRY, bool FINAL, bool DEPTH = false, bool LOWREC = false, bool PACKCAP = false>
static void radix_pass_launch(const ItemT *in, ItemT *out, int32_t *hist, int32_t *acc, int64_t n, int shift, int32_t *zero_acc, int zero_n,
                              const ScatterCarry &carry, const ScatterFinal &fin, hipStream_t s, const DepthPass &dp = DepthPass{}, bool hist_ready = false)
{
    const PassGeom g = pass_geom(n);
    // (a skipped depth pass leaves its accumulator rows zero: the super-block scan of a many-block pass then scans zeros)
    // hist_ready: the kernel that produced `in` left this pass's block histograms and super-block sums (expand_blocks_kernel)
    if (hist_ready) {
    } else if (g.tier == 0) {
        hipLaunchKernelGGL((radix_hist_kernel<GSR_RADIX_TINY_CHUNK / 256, BITS, ItemT, DEPTH>), dim3(g.nb), dim3(256), 0, s, in, hist, acc, n, shift, g.sb, dp);
    } else if (g.tier == 1) {
        hipLaunchKernelGGL((radix_hist_kernel<GSR_RADIX_SMALL_CHUNK / 256, BITS, ItemT, DEPTH>), dim3(g.nb), dim3(256), 0, s, in, hist, acc, n, shift, g.sb, dp);
    } else {
        hipLaunchKernelGGL((radix_hist_kernel<GSR_RADIX_CHUNK / 256, BITS, ItemT, DEPTH>), dim3(g.nb), dim3(256), 0, s, in, hist, acc, n, shift, g.sb, dp);
    }
    if (g.prefixed) hipLaunchKernelGGL(radix_superscan_kernel, dim3(4), dim3(1024), 0, s, acc, (g.nb + g.sb - 1) / g.sb, dp);
    if (g.tier == 0) {
        hipLaunchKernelGGL((radix_scatter_kernel<GSR_RADIX_TINY_CHUNK / 256, BITS, ItemT, CARRY, FINAL, DEPTH, 256, LOWREC, PACKCAP>), dim3(g.nb), dim3(256), 0, s, in, out, hist, acc, n, shift,
                           g.nb, g.sb, g.prefixed, zero_acc, zero_n, carry, fin, dp);
    } else if (g.tier == 1) {
        hipLaunchKernelGGL((radix_scatter_kernel<GSR_RADIX_SMALL_CHUNK / GSR_RADIX_SMALL_WG, BITS, ItemT, CARRY, FINAL, DEPTH, GSR_RADIX_SMALL_WG, LOWREC, PACKCAP>), dim3(g.nb), dim3(GSR_RADIX_SMALL_WG),
                           0, s, in, out, hist, acc, n, shift, g.nb, g.sb, g.prefixed, zero_acc, zero_n, carry, fin, dp);
    } else {
        hipLaunchKernelGGL((radix_scatter_kernel<GSR_RADIX_CHUNK / GSR_RADIX_WG, BITS, ItemT, CARRY, FINAL, DEPTH, GSR_RADIX_WG, LOWREC, PACKCAP>), dim3(g.nb), dim3(GSR_RADIX_WG), 0, s, in, out,
                           hist, acc, n, shift, g.nb, g.sb, g.prefixed, zero_acc, zero_n, carry, fin, dp);
    }
    if constexpr (FINAL)
        hipLaunchKernelGGL(ranges_fixup_kernel, dim3(1 << BITS), dim3(1024), 0, s, fin.edge_first, fin.edge_last, fin.edge_pos, acc, g.nb, 1 << BITS,
                           fin.ranges);
}

template <typename ItemT, bool FINAL>
static hipError_t radix_pass_any(const ItemT *in, ItemT *out, int32_t *hist, int32_t *acc, int64_t n, int shift, int bits, int32_t *zero_acc,
                                 int zero_n, const ScatterFinal &fin, hipStream_t s, bool hist_ready)
{
    const ScatterCarry nc{};
    const DepthPass nd{};
    if constexpr (FINAL && sizeof(ItemT) == 4) {
        if (fin.low_bits) { // narrowed items: the kernel that also recovers the first pass's digit (ScatterFinal)
            switch (bits) {
            case 4: radix_pass_launch<4, ItemT, false, true, false, true>(in, out, hist, acc, n, shift, zero_acc, zero_n, nc, fin, s, nd, hist_ready); break;
            case 5: radix_pass_launch<5, ItemT, false, true, false, true>(in, out, hist, acc, n, shift, zero_acc, zero_n, nc, fin, s, nd, hist_ready); break;
            case 6: radix_pass_launch<6, ItemT, false, true, false, true>(in, out, hist, acc, n, shift, zero_acc, zero_n, nc, fin, s, nd, hist_ready); break;
            case 7: radix_pass_launch<7, ItemT, false, true, false, true>(in, out, hist, acc, n, shift, zero_acc, zero_n, nc, fin, s, nd, hist_ready); break;
            case 8: radix_pass_launch<8, ItemT, false, true, false, true>(in, out, hist, acc, n, shift, zero_acc, zero_n, nc, fin, s, nd, hist_ready); break;
            default: return hipErrorInvalidValue;
            }
            return hipGetLastError();
        }
    }
    switch (bits) {
    case 4: radix_pass_launch<4, ItemT, false, FINAL>(in, out, hist, acc, n, shift, zero_acc, zero_n, nc, fin, s, nd, hist_ready); break;
    case 5: radix_pass_launch<5, ItemT, false, FINAL>(in, out, hist, acc, n, shift, zero_acc, zero_n, nc, fin, s, nd, hist_ready); break;
    case 6: radix_pass_launch<6, ItemT, false, FINAL>(in, out, hist, acc, n, shift, zero_acc, zero_n, nc, fin, s, nd, hist_ready); break;
    case 7: radix_pass_launch<7, ItemT, false, FINAL>(in, out, hist, acc, n, shift, zero_acc, zero_n, nc, fin, s, nd, hist_ready); break;
    case 8: radix_pass_launch<8, ItemT, false, FINAL>(in, out, hist, acc, n, shift, zero_acc, zero_n, nc, fin, s, nd, hist_ready); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// One stable pass on the `bits`-wide digit at `shift` (bits in 4..8).  Ranking costs one ballot per digit bit,
// so passes use the narrowest digits that cover the key: 6+6 bits for the 12-bit tile ids of an 800x800 image.
// item_bytes: 8 (uint64 items) or 4 (uint32 items: tile id and Gaussian id share one word when they fit).
// `acc`: this pass's accumulators (gsr_radix_acc_ints(n) ints, zero when the pass's first kernel runs); `zero_acc`: the
// accumulators of the NEXT pass over the same n, cleared by this pass's scatter (or NULL).
hipError_t gsr_launch_radix_pass(const void *in, void *out, int32_t *hist, int32_t *acc, int64_t n, int shift, int bits, int item_bytes,
                                 int32_t *zero_acc, hipStream_t s, bool hist_ready, int narrow_id_bits, int32_t *totals_out)
{
    if (n <= 0) return hipSuccess;
    const int zero_n = zero_acc ? (int)gsr_radix_acc_ints(n) : 0;
    ScatterFinal opt{};
    opt.narrow_id_bits = item_bytes == 8 ? narrow_id_bits : 0; // 64-bit items in, 32-bit items out (see ScatterFinal)
    opt.totals = opt.narrow_id_bits ? totals_out : nullptr;
    if (item_bytes == 4)
        return radix_pass_any<uint32_t, false>((const uint32_t *)in, (uint32_t *)out, hist, acc, n, shift, bits, zero_acc, zero_n, opt, s, hist_ready);
    return radix_pass_any<uint64_t, false>((const uint64_t *)in, (uint64_t *)out, hist, acc, n, shift, bits, zero_acc, zero_n, opt, s, hist_ready);
}

int gsr_no_narrowing = 0; // GSR_NO_NARROWING (gsr_internal.h)
int gsr_no_depth_pack = 0; // GSR_NO_DEPTH_PACK (gsr_internal.h)

// The LAST pass of the tile partition: histogram, then a scatter that writes point_list and the in-sight range boundaries
// directly (ScatterFinal), and the edge fix-up.  `edge` holds 3 * (1 << bits) * nb int32 (gsr_radix_blocks(n) = nb).
hipError_t gsr_launch_radix_final_pass(const void *in, int32_t *hist, int32_t *acc, int64_t n, int shift, int bits, int item_bytes,
                                       int id_shift, int32_t *point_list, int32_t *ranges, int32_t *edge, hipStream_t s, bool hist_ready,
                                       const int32_t *low_totals, int low_bits)
{
    if (n <= 0) return hipSuccess;
    const size_t per = ((size_t)1 << bits) * (size_t)gsr_radix_blocks(n);
    const ScatterFinal fin{point_list, ranges, edge, edge + per, edge + 2 * per, acc, id_shift, 0, low_totals, low_totals ? low_bits : 0};
    if (item_bytes == 4) return radix_pass_any<uint32_t, true>((const uint32_t *)in, (uint32_t *)nullptr, hist, acc, n, shift, bits, nullptr, 0, fin, s, hist_ready);
    return radix_pass_any<uint64_t, true>((const uint64_t *)in, (uint64_t *)nullptr, hist, acc, n, shift, bits, nullptr, 0, fin, s, hist_ready);
}

// The id-order scan of tiles_touched into point_offsets (D to the pinned host word), the depth sort's pass plan and the first
// active depth pass's histogram, one launch (scan_ctl_hist_kernel).  Not for the small-scene path (gsr_small_depth_path).
hipError_t gsr_launch_scan_ctl_hist(const int32_t *tiles_touched, int32_t *point_offsets, const GeomWs &ws, int64_t n, int32_t *total_out, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const int nb_scan = ((int)gsr_div_up(n, GSR_SCAN_WAVE_ITEMS) + 3) / 4, nblk = (int)gsr_div_up(n, 256), n_ctl = gsr_depth_ctl_wgs(n);
    const PassGeom g = pass_geom(n);
    DepthCtlRaw *ctl = (DepthCtlRaw *)ws.depth_ctl;
#define GSR_SCH(CH)                                                                                                                                    \
    hipLaunchKernelGGL((scan_ctl_hist_kernel<(CH) / 256>), dim3(nb_scan + n_ctl + g.nb), dim3(256), 0, s, tiles_touched, ws.scan_tmp, point_offsets, n, \
                       total_out, ws.blk_minmax, nblk, ctl, n_ctl, nb_scan, ws.depth_item, ws.hist, ws.acc_first, g.sb)
    if (g.tier == 0) GSR_SCH(GSR_RADIX_TINY_CHUNK);
    else if (g.tier == 1) GSR_SCH(GSR_RADIX_SMALL_CHUNK);
    else GSR_SCH(GSR_RADIX_CHUNK);
#undef GSR_SCH
    return hipGetLastError();
}

// The depth sort: Gaussians by depth bits, stable from id order.  Four 8-bit passes over the 64-bit (depth bits << 32 | id) items
// are launched; how many of them this frame's depth range needs is decided on the device (DepthCtl, filled by the id-order scan),
// the others return at once.  The last pass writes, instead of the sorted items, what the rest of the pipeline reads: the ids,
// and each Gaussian's tile rectangle and tile count carried to its sorted position.
bool gsr_small_depth_path(int64_t n) { return n <= GSR_SMALL_SORT_N && !(gsr_debug_flags & 1024); } // GSR_DEBUG bit 10: never (tests)

hipError_t gsr_launch_depth_sort(const GeomWs &ws, int64_t n, hipStream_t s, int launch_passes, int pack_ok)
{
    if (n <= 0) return hipSuccess;
    launch_passes = std::min(4, std::max(1, launch_passes));
    if (gsr_small_depth_path(n)) { // sorts, carries AND scans: the caller skips the depth-order scan
        hipLaunchKernelGGL(depth_sort_small_kernel, dim3(1), dim3(1024), 0, s, ws.depth_item, (int)n, ws.rect, ws.id_sorted, ws.rect_sorted,
                           ws.cnt_sorted, ws.doff);
        return hipGetLastError();
    }
    const int zero_n = (int)gsr_radix_acc_ints(n);
    const ScatterCarry carry{ws.rect, ws.rect_sorted, ws.cnt_sorted, ws.id_sorted, n};
    for (int pass = 4 - launch_passes; pass < 4; ++pass) {
        const DepthPass dp{(const DepthCtlRaw *)ws.depth_ctl, (gsr_debug_flags & 256) ? 4 : 0, pass, {ws.depth_item, ws.sort_tmp}, ws.acc_first, 4 - launch_passes, pack_ok};
        // pass p accumulates into acc[p & 1] (both cleared by preprocess) and clears the other one for pass p + 1 -- except the
        // first ACTIVE pass, whose histogram and sums were made beside the id-order scan (gsr_launch_scan_ctl_hist) in acc_first.
        // The first LAUNCHED pass is either skipped by the plan or the first active one: its histogram kernel is not launched.
        const bool no_hist = pass == 4 - launch_passes;
        if (pass < 3 && pack_ok)
            radix_pass_launch<8, uint64_t, false, false, true, false, true>(ws.depth_item, ws.sort_tmp, ws.hist, ws.acc[pass & 1], n, 0, ws.acc[(pass + 1) & 1], zero_n,
                                                                            carry /* (its rect array: the pass that packs the items reads it) */, ScatterFinal{}, s, dp, no_hist);
        else if (pass < 3)
            radix_pass_launch<8, uint64_t, false, false, true>(ws.depth_item, ws.sort_tmp, ws.hist, ws.acc[pass & 1], n, 0, ws.acc[(pass + 1) & 1], zero_n,
                                                               ScatterCarry{}, ScatterFinal{}, s, dp, no_hist);
        else
            radix_pass_launch<8, uint64_t, true, false, true>(ws.depth_item, ws.sort_tmp, ws.hist, ws.acc[pass & 1], n, 0, nullptr, 0, carry, ScatterFinal{}, s, dp,
                                                              no_hist);
    }
    return hipGetLastError();
}

hipError_t gsr_launch_expand(const uint32_t *id_sorted, const int32_t *doff, const TileRect *rect, void *tile_items, int64_t n,
                             int grid_x, int64_t D, int id_shift, int item_bytes, int32_t *ranges, int ranges_n, int32_t *zero_acc, int zero_n,
                             int32_t *zero_b, int zero_b_n, int bo_flag, hipStream_t s)
{
    if (n <= 0 || D <= 0) return hipSuccess;
    const bool few = D / n >= 32;
#define EXPAND(T, G)                                                                                                          \
    hipLaunchKernelGGL((expand_kernel<T, G>), dim3((unsigned)gsr_div_up(n, 4 * (G))), dim3(256), 0, s, id_sorted, doff, rect, (T *)tile_items, n, grid_x, D, \
                       id_shift, ranges, ranges_n, zero_acc, zero_n, zero_b, zero_b_n, bo_flag)
    if (item_bytes == 4) { if (few) EXPAND(uint32_t, 8); else EXPAND(uint32_t, 64); }
    else { if (few) EXPAND(uint64_t, 8); else EXPAND(uint64_t, 64); }
#undef EXPAND
    return hipGetLastError();
}

// The product path of the expansion: gsr_launch_depth_block_offsets, then gsr_launch_expand_blocks (see the kernels).
hipError_t gsr_launch_depth_block_offsets(const GeomWs &ws, int64_t n, int32_t *ranges, int ranges_n, int32_t *zero_acc, int zero_n, int32_t *zero_b,
                                          int zero_b_n, int bo_flag, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const int nsum = (int)gsr_div_up(n, 256);
    hipLaunchKernelGGL(depth_block_offsets_kernel, dim3((unsigned)gsr_div_up(nsum, 16)), dim3(1024), 0, s, ws.cnt_sorted, n, ws.scan_tmp, nsum, ws.sum4096, ranges,
                       ranges_n, zero_acc, zero_n, zero_b, zero_b_n, bo_flag);
    return hipGetLastError();
}
// `bits0`: the digit width of the first partition pass over the D items (its digit = the low bits0 bits of the tile id); `hist`,
// `acc`: that pass's block histograms and (zeroed) accumulators, which this kernel fills -- launch the pass with hist_ready.
hipError_t gsr_launch_expand_blocks(const GeomWs &ws, void *tile_items, int64_t n, int grid_x, int64_t D, int id_shift, int item_bytes, int bits0,
                                    int32_t *hist, int32_t *acc, hipStream_t s)
{
    if (n <= 0 || D <= 0) return hipSuccess;
    const PassGeom g = pass_geom(D);
    const int chunk = g.chunk, nsum = (int)gsr_div_up(n, 256);
    if (item_bytes == 4)
        hipLaunchKernelGGL(expand_blocks_kernel<uint32_t>, dim3(g.nb), dim3(256), 0, s, ws.id_sorted, ws.cnt_sorted, ws.scan_tmp, ws.sum4096, nsum, ws.rect_sorted, (uint32_t *)tile_items, n,
                           grid_x, D, id_shift, chunk, (1 << bits0) - 1, hist, acc, g.sb);
    else
        hipLaunchKernelGGL(expand_blocks_kernel<uint64_t>, dim3(g.nb), dim3(256), 0, s, ws.id_sorted, ws.cnt_sorted, ws.scan_tmp, ws.sum4096, nsum, ws.rect_sorted, (uint64_t *)tile_items, n,
                           grid_x, D, id_shift, chunk, (1 << bits0) - 1, hist, acc, g.sb);
    return hipGetLastError();
}
