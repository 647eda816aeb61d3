// gsr_internal.h -- shared declarations for the gfx950 rasterizer kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gsr.h"

#define GSR_WAVE 64

// Camera constants passed by value to kernels (lands in SGPRs / kernarg).
struct CamK {
    float view[16];
    float proj[16];
    float campos[3];
    float bg[3];
    float tan_fovx, tan_fovy;
    float focal_x, focal_y;
    int W, H, grid_x, grid_y;
};

// Per-Gaussian blend record, 64 B = one cache sector per gather (geom workspace).
//   f[0..1] xy, f[2..4] conic a,b,c, f[5] opacity, f[6..8] rgb, f[9] 1/depth, rest pad.
struct __attribute__((aligned(16))) BlendRec {
    float f[16];
};

// Packed tile rectangle of one Gaussian (min_x, min_y, max_x, max_y), all < 65536.
struct __attribute__((aligned(8))) TileRect {
    uint16_t x0, y0, x1, y1;
};

static inline bool gsr_aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; } // null counts as aligned
static inline int64_t gsr_div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t gsr_align(size_t x) { return (x + 255) & ~(size_t)255; }

#define GSR_FO_MAX_TILES 4096  // forward tile order (see below): images of at most this many tiles
#ifndef GSR_FO_CLASSES
#define GSR_FO_CLASSES 64
#endif
// ---- layout of the geom workspace (persists from gsr_forward_count to gsr_forward_render) ----
struct GeomWs {
    BlendRec *rec;        // [N]
    TileRect *rect;       // [N]
    uint64_t *depth_item; // [N] (depth bits << 32 | id), 0xFFFFFFFF depth for culled: the depth sort's input
    uint64_t *sort_tmp;   // [N] ping-pong partner of depth_item
    uint32_t *id_sorted;  // [N] Gaussian ids in depth order (written by the last depth-sort pass)
    uint32_t *blk_minmax; // [4 * ceil(N / 256)] per preprocess block: smallest / largest visible depth bits, visible count, -
    void *depth_ctl;      // DepthCtlRaw (scan_sort.hip), the uint4 behind the last block's extremes: this frame's visible depth range and count
    TileRect *rect_sorted; // [N] tile rectangles in depth order (written by the last depth-sort pass)
    int32_t *cnt_sorted;  // [N] tile counts in depth order (same pass)
    int32_t *doff;        // [N] exclusive tile-pair offsets in depth order
    int32_t *scan_tmp;    // [N / 256 + 4] partial sums for the scans (preprocess writes one per 256 Gaussians)
    int32_t *hist;        // [nb(N)][256] radix block histograms
    int32_t *acc[2];      // [gsr_radix_acc_ints(N)] each: digit + super-block totals of a pass; consecutive passes alternate
    int32_t *acc_first;   // [gsr_radix_acc_ints(N)] behind acc[1]: the first ACTIVE depth pass's accumulators (filled beside the id-order scan)
    int32_t *sum4096;     // [N / 4096 + 1] tile pairs per 4096 depth-sorted Gaussians (depth_block_offsets_kernel; scan_tmp holds the 256-level)
    int32_t *fwd_cost;    // [4 * GSR_FO_MAX_TILES] what the forward blend's waves cost LAST frame ("forward tile order" below);
    int32_t *fwd_order;   // [GSR_FO_MAX_TILES]     this frame's dispatch order made from it (last in the workspace)
    size_t bytes;
};
GeomWs gsr_carve_geom(void *base, int64_t N);

// ---- launchers (host functions; each enqueues on `s` and returns hipGetLastError()) ----
hipError_t gsr_launch_preprocess(const GsrScene &sc, const CamK &cam, const GsrGeom &g, const GeomWs &ws, hipStream_t s,
                                 bool make_fwd_order = false /* a spare workgroup turns ws.fwd_cost into ws.fwd_order */);

// Device-wide scan of int32.  mode 0: out[i] = inclusive scan of in[i].
// mode 2: out[i] = exclusive scan of in[i] (total_out still receives the grand total).
// The depth sort's pass plan from the frame's visible depth extremes (bit patterns of positive floats).  Shared by the device
// (every depth kernel derives it from DepthCtlRaw) and the host (its launch guess for the next frame, from the pinned words).
struct DepthPlan {
    uint32_t min_bits, range; // key = umin(bits - min_bits, range); min_bits has a zero low byte (see scan_ctl_hist_kernel)
    int npass, first;         // passes first .. 3 are active, first = 4 - npass
};
static inline __host__ __device__ DepthPlan gsr_depth_plan(uint32_t lo, uint32_t hi, int force_npass)
{
    DepthPlan p;
    if (lo > hi) { p.min_bits = 0xFFFFFFFFu; p.range = 0u; } // nothing visible: every key is 0, one pass (it carries the rectangles)
    else { p.min_bits = lo & ~255u; p.range = hi - p.min_bits + 1u; }
    int nbits = 0;
    for (uint32_t r = p.range | 1u; r; r >>= 1) ++nbits; // keys are 0 .. range
    p.npass = force_npass > 0 ? force_npass : (nbits + 7) / 8 < 1 ? 1 : (nbits + 7) / 8; // forced (tests: GSR_DEBUG bit 8 = always four): same order
    p.first = 4 - p.npass;
    return p;
}
#define GSR_DEPTH_CTL_WGS_MAX 16
// how many workgroups of the scan's launch reduce the per-block extremes (2048 blocks each at most 16: one round of loads per thread)
static inline int gsr_depth_ctl_wgs(int64_t N) { const int64_t nblk = (N + 255) / 256; const int64_t k = (nblk + 2047) / 2048; return (int)(k < 1 ? 1 : k > GSR_DEPTH_CTL_WGS_MAX ? GSR_DEPTH_CTL_WGS_MAX : k); }
#define GSR_SCAN_WAVE_ITEMS 1024   // items per wave-sized scan unit; scratch = one int32 per unit
hipError_t gsr_launch_scan(const int32_t *in, const uint64_t *items, int32_t *out, int32_t *block_tmp,
                           int64_t n, int mode, int32_t *total_out /* optional: receives the grand total */,
                           bool sums_per_256_ready /* mode 0: block_tmp already holds a sum per 256 items */, hipStream_t s,
                           const uint32_t *blk_minmax = nullptr, void *depth_ctl = nullptr /* mode 0: also derive the depth sort's DepthCtl */);
hipError_t gsr_launch_depth_sort(const GeomWs &ws, int64_t n, hipStream_t s, int launch_passes = 4 /* the last `launch_passes` of the four */,
                                 int pack_ok = 0 /* the sizes allow packed depth items (scan_sort.hip) */);
extern int gsr_no_depth_pack; // GSR_NO_DEPTH_PACK: the last depth pass always gathers the rectangles by id (A/B, tests)
hipError_t gsr_launch_scan_ctl_hist(const int32_t *tiles_touched, int32_t *point_offsets, const GeomWs &ws, int64_t n, int32_t *total_out, hipStream_t s);
#define GSR_SMALL_SORT_N 8192          // up to this many Gaussians one workgroup sorts, carries and scans (scan_sort.hip)
bool gsr_small_depth_path(int64_t n); // true: gsr_launch_depth_sort also writes the depth-order offsets (no separate scan)

// One stable LSD radix pass by the `bits`-wide (4..8) digit at `shift`; items are uint64 (item_bytes 8) or uint32 (4).
#ifndef GSR_RADIX_CHUNK
#define GSR_RADIX_CHUNK 4096
#endif
// Threads per scatter workgroup on the large-chunk path.  512 (late round 4): the same 4096-item chunk as 8 waves of 8 items per thread
// instead of 4 waves of 16 -- 63 VGPRs instead of 94, 32 waves per CU instead of 20: tile partition 72.2 -> 70.0 us at C3, 331 -> 320 us at
// C5 (profiles/r04_w_scatter_workgroup_shapes.txt; 1024 threads, and 6144- / 8192-item chunks with 512, are slower or move the time into the expansion)
#ifndef GSR_RADIX_WG
#define GSR_RADIX_WG 512
#endif
// Three sizes of pass (scan_sort.hip pass_geom): up to GSR_RADIX_TINY_N items 1024-item chunks in 256-thread workgroups (many small
// blocks keep the chip busy: at 100 k items the 2048-item form costs the depth sort 2 us); up to GSR_RADIX_SMALL_N 2048-item chunks in
// 512-thread workgroups, four items per thread as before but half the blocks, hence half the histogram rows every block sums (late
// round 4: depth sort 63.1 -> 56.8 us at C3, profiles/r04_w_scatter_workgroup_shapes.txt); above that GSR_RADIX_CHUNK.
#define GSR_RADIX_TINY_CHUNK 1024
#define GSR_RADIX_TINY_N (1 << 19)
#ifndef GSR_RADIX_SMALL_WG
#define GSR_RADIX_SMALL_WG 512
#endif
#ifndef GSR_RADIX_SMALL_CHUNK
#define GSR_RADIX_SMALL_CHUNK 2048
#endif
#ifndef GSR_RADIX_SMALL_N
#define GSR_RADIX_SMALL_N (4 << 20)
#endif
static inline int gsr_radix_chunk(int64_t n) { return n <= GSR_RADIX_TINY_N ? GSR_RADIX_TINY_CHUNK : n <= GSR_RADIX_SMALL_N ? GSR_RADIX_SMALL_CHUNK : GSR_RADIX_CHUNK; }
static inline int64_t gsr_radix_blocks(int64_t n) { const int c = gsr_radix_chunk(n); return (n + c - 1) / c; }
// A pass's blocks are grouped into super-blocks of about sqrt(nb) blocks; its accumulators are 256 digit totals followed by
// 256 per super-block (scan_sort.hip, radix_hist_kernel).
#define GSR_RADIX_PREFIX_NB 2048 // passes over more blocks than this scan their super-block rows first (radix_superscan_kernel)
static inline int gsr_radix_sb(int nb)
{
    if (nb > GSR_RADIX_PREFIX_NB) return 32;
    int sb = 16;
    while ((int64_t)sb * sb < nb) ++sb;
    return sb;
}
static inline size_t gsr_radix_acc_ints(int64_t n)
{
    const int nb = (int)gsr_radix_blocks(n), sb = gsr_radix_sb(nb); // GSR_DEBUG bit 6 cuts n into fewer blocks: fewer rows
    return 256 * (size_t)(3 + nb / sb);
}
hipError_t gsr_launch_radix_pass(const void *in, void *out, int32_t *hist /*[nb][radix]*/, int32_t *acc /* gsr_radix_acc_ints(n), zero */,
                                 int64_t n, int shift, int bits, int item_bytes, int32_t *zero_acc /* next pass's, or NULL */, hipStream_t s,
                                 bool hist_ready = false /* hist and acc were filled by gsr_launch_expand_blocks */,
                                 int narrow_id_bits = 0 /* 64-bit items only: > 0 = write 32-bit items (tile >> bits) << narrow_id_bits | id ... */,
                                 int32_t *totals_out = nullptr /* ... and leave the digit totals here for the final pass (scan_sort.hip ScatterFinal) */);

// Last pass of the tile partition: writes point_list and ranges instead of the sorted items (scan_sort.hip, ScatterFinal).
// `edge`: 3 * 256 * (gsr_radix_blocks(n) + 1) int32 of scratch.
hipError_t gsr_launch_radix_final_pass(const void *in, int32_t *hist, int32_t *acc, int64_t n, int shift, int bits, int item_bytes,
                                       int id_shift, int32_t *point_list, int32_t *ranges /* pre-zeroed */, int32_t *edge, hipStream_t s,
                                       bool hist_ready = false, const int32_t *low_totals = nullptr /* narrowed items: the first pass's digit totals */,
                                       int low_bits = 0 /* ... and digit width */);

// Tile items are (tile << id_shift | gaussian id): uint64 with id_shift = 32, or uint32 when tile bits + id bits <= 32.
hipError_t gsr_launch_expand(const uint32_t *id_sorted, const int32_t *doff, const TileRect *rect, void *tile_items,
                             int64_t n, int grid_x, int64_t D, int id_shift, int item_bytes, int32_t *ranges, int ranges_n,
                             int32_t *zero_acc, int zero_n /* accumulators of the first partition pass, cleared here */,
                             int32_t *zero_b, int zero_b_n /* the block-order header, or NULL */, int bo_flag /* its `filed` flag */, hipStream_t s);
// The product path of the expansion (scan_sort.hip): one prefix per 256 depth-sorted Gaussians (into ws.scan_tmp; also clears the
// ranges, the first partition pass's accumulators and the block-order header), then one workgroup per radix block of the output,
// which also leaves the first partition pass's histograms (launch that pass with hist_ready).
// ---- forward tile order (round 4) ----
// The forward blend's 2 500 workgroups at 800 x 800 run on 2 048 workgroup slots: the 452 that start when the first slots free up
// decide when the kernel ends (they start at 0.4 of its span and live half of it).  An oracle experiment (tools/residency.py
// --fwd-order: dispatch the tiles heaviest first by their MEASURED lives) takes the kernel from 113 to 94 us; by list length, which
// is known before the blend, from 113 to 108 (rank correlation 0.05-0.16 with the life: saturation cuts every deep list at about the
// same depth).  What does predict a tile's work is the tile's work a frame ago: a trainer comes back to its views, a viewer moves
// its camera smoothly.  So every wave of the forward blend leaves behind its cost -- its measured life in 100 MHz ticks since late
// round 4, the entries it walked and staged before (blend_fwd.hip GSR_FWD_COST_LIFE) -- (fwd_cost, 4 ints per tile, in
// the caller's geom workspace, which persists between frames as long as the caller keeps it), and the next forward on that
// workspace dispatches the tiles by cost class, heaviest first (fwd_order, made by a spare workgroup of preprocess_kernel
// every frame from whatever fwd_cost holds -- garbage in a fresh workspace gives some permutation, never a wrong one).  Execution
// order only: every tile computes what it always did.  Not for images of more than GSR_FO_MAX_TILES tiles (many rounds, no tail).
extern int gsr_fwd_no_order; // GSR_FWD_NO_ORDER: plain row-major dispatch (A/B)
extern int gsr_no_narrowing; // GSR_NO_NARROWING: the second tile pass keeps 64-bit items where the first could narrow them (A/B, tests)
hipError_t gsr_launch_depth_block_offsets(const GeomWs &ws, int64_t n, int32_t *ranges, int ranges_n, int32_t *zero_acc, int zero_n, int32_t *zero_b,
                                          int zero_b_n, int bo_flag, hipStream_t s);
hipError_t gsr_launch_expand_blocks(const GeomWs &ws, void *tile_items, int64_t n, int grid_x, int64_t D, int id_shift, int item_bytes, int bits0,
                                    int32_t *hist, int32_t *acc, hipStream_t s);
// ---- block order (GsrBinning.block_order): the backward blend's 8x4-pixel blocks, heaviest first ---------------------------
// The backward's waves live 40-90 us of a 165-us kernel, so what starts last decides when the kernel ends.  How many list
// entries the backward's compaction will keep for a block (mask hits up to the block's last contributor) is the one cheap
// quantity that ranks the blocks' cost (rank correlation 0.5 with the measured wave life; the tile's list length: 0.2; an
// oracle order by measured life: 147 us, by this count in 8 classes: 148-150 us, tools/residency.py --lpt).  The forward
// blend's waves count it for their two blocks when they are done (blend_fwd.hip file_blocks: a call, not inline code -- see
// there) and file every block under (XCD band of its tile, cost class, shard); the backward blend then walks each band's
// classes from the heaviest down.  The backward only READS the queues: any number of gsr_backward calls may follow a forward.
// (Also built and measured: the same filing beside the clearing of the accumulators, in one kernel at the start of
// gsr_backward -- the filing workgroups' chains of dependent loads crawl under that kernel's 64 MB of stores: 13 -> 36 us at C3,
// 48 -> 140 us at C5, more than the blend gains.)
// Layout, int32: [GSR_BO_BANDS][GSR_BO_CLASSES][GSR_BO_SHARDS] counters, a `filed` flag (+ 3 pad) -- written by the forward's expand_kernel --
// then [GSR_BO_BANDS][GSR_BO_CLASSES][GSR_BO_SHARDS][cap] block ids (tile * 8 + block), cap = 8 * ceil(tiles per band / shards).
// A (band, class) queue is split over 16 shards by tile (tile % 16) because the filing takes its slot with a RETURNED atomic:
// 20 000 of them onto 256 addresses serialise at the memory side; onto 4 096 they do not.
#define GSR_BO_BANDS 8
#define GSR_BO_CLASSES 32
#define GSR_BO_SHARDS 16
#define GSR_BO_QUEUES (GSR_BO_BANDS * GSR_BO_CLASSES * GSR_BO_SHARDS)
#define GSR_BO_FLAG GSR_BO_QUEUES      // 1: the forward filed the blocks (it does not for images of more than GSR_BO_MAX_TILES tiles)
#define GSR_BO_HEADER (GSR_BO_QUEUES + 4)
// Heaviest-first pays while the backward blend runs only a few rounds of waves (800x800: 20 000 blocks on 8 192 wave slots, 2.4
// rounds, 164 -> 154 us for 5 us more in the forward); at 1920x1080 (65 280 blocks, 8 rounds) the blend gains 2 % and the filing
// costs the forward more than that, so larger images keep the plain band order.
#define GSR_BO_MAX_TILES 4096
// Which band (= XCD) a tile belongs to.  GSR_BO_ROWS = 0: eight bands of consecutive tiles (an eighth of the image each: the XCDs
// holding the image centre get more work than the others); GSR_BO_ROWS = G > 0: groups of G tile rows dealt to the bands in turn.
#ifndef GSR_BO_ROWS
#define GSR_BO_ROWS 0
#endif
__host__ __device__ static inline int gsr_bo_tiles_per_band(int tiles, int grid_x)
{
#if GSR_BO_ROWS > 0
    const int grid_y = (tiles + grid_x - 1) / grid_x;
    return ((grid_y + GSR_BO_BANDS * GSR_BO_ROWS - 1) / (GSR_BO_BANDS * GSR_BO_ROWS)) * GSR_BO_ROWS * grid_x; // an upper bound
#else
    (void)grid_x;
    return (tiles + GSR_BO_BANDS - 1) / GSR_BO_BANDS;
#endif
}
__host__ __device__ static inline int gsr_bo_band(int tile, int tiles, int grid_x)
{
#if GSR_BO_ROWS > 0
    (void)tiles;
    return ((tile / grid_x) / GSR_BO_ROWS) % GSR_BO_BANDS;
#else
    return tile / gsr_bo_tiles_per_band(tiles, grid_x);
#endif
}
static inline int gsr_bo_cap(int tiles, int grid_x) { return 8 * ((gsr_bo_tiles_per_band(tiles, grid_x) + GSR_BO_SHARDS - 1) / GSR_BO_SHARDS); }
static inline size_t gsr_bo_ints(int tiles, int grid_x) { return GSR_BO_HEADER + (size_t)GSR_BO_QUEUES * (size_t)gsr_bo_cap(tiles, grid_x); }
// class of a block that keeps `hits` entries: four classes per octave from 8 entries up (class 0: fewer than 8)
__host__ __device__ static inline int gsr_bo_class(int hits)
{
    if (hits < 8) return 0;
    const int lg = 31 - __builtin_clz((unsigned)hits);          // floor(log2 hits) >= 3
    const int c = 1 + 4 * (lg - 3) + ((hits >> (lg - 2)) & 3); // two mantissa bits
    return c < GSR_BO_CLASSES ? c : GSR_BO_CLASSES - 1;
}
hipError_t gsr_launch_blend_forward(const CamK &cam, const int32_t *ranges, const int32_t *point_list,
                                    const BlendRec *rec, const GsrImage &img, uint8_t *block_masks /* optional out */,
                                    int32_t *block_order /* optional out, with block_masks */,
                                    void *clear, size_t clear_bytes /* optional: memory its spare workgroups zero (16-byte units) */, hipStream_t s,
                                    const int32_t *tile_order = nullptr /* optional: launch slot -> tile */, int32_t *tile_cost = nullptr /* optional out: 4 per tile */);

// backward
struct __attribute__((aligned(16))) GradRec { // 64 B accumulator per Gaussian (atomics target)
    float f[16];                              // 0..2 dcolor | 3..4 dmean2D, 5 = 0 | 6..7 dconic a b, 8 = 0, 9 dconic c | 10 dopacity | 0 ...
};
// The nine accumulated values sit where the reference's API arrays have them: columns 0-2, 3-5 and 6-9 of the record ARE
// dL_dcolor (N,3), dL_dmean2D (N,3: z = 0) and dL_dconic (N,4: a, b, 0, c), so a caller that owns the backward workspace per call
// reads them as strided views (include/gsr.h, GsrGrads) and geom_backward_kernel need not write 40 bytes per Gaussian of copies.
// k-th accumulated value (colour r g b, mean2D x y, conic a b c, opacity) -> float index in the record
__host__ __device__ static inline int gsr_gradrec_slot(int k) { return k + (k >= 5) + (k >= 7); }
hipError_t gsr_launch_pack_records(const GsrGeom &g, BlendRec *rec, int64_t N, hipStream_t s);
hipError_t gsr_launch_blend_backward_splat(const CamK &cam, const int32_t *ranges, const int32_t *point_list,
                                           const BlendRec *rec, const GsrImage &img, const float *dL_dpixels,
                                           const uint8_t *block_masks /* optional: the forward's */,
                                           const int32_t *block_order /* optional: the forward's, with its masks */, GradRec *acc,
                                           int64_t N, int64_t D /* choose the block size */, hipStream_t s);
hipError_t gsr_launch_geom_backward(const GsrScene &sc, const CamK &cam, const GsrGeom &g, const GradRec *acc,
                                    const GsrGrads &gr, hipStream_t s);

// tuning knobs (read once from the environment by api.hip; defaults are the measured best)
hipError_t gsr_launch_view_payload(const GsrScene &sc, const CamK &cam, const GsrGeom &g, const GradRec *acc, float *payload, hipStream_t s);
// GSR_DEBUG (environment, read once): bit 5 forces 64-bit tile items, bit 6 the large-n radix chunks, bit 7 the scanned
// super-block rows of many-block radix passes (radix_superscan_kernel), bit 8 all four depth-sort passes whatever the depth
// range, bit 9 the expansion by Gaussian (a wave per 64 or 8 of them, from a device-wide scan of the depth-order offsets, followed
// by the first partition pass's own histogram kernel) instead of the expansion by output block; bit 10 switches the
// one-workgroup depth stage of small scenes OFF (so small test scenes also run the multi-kernel chain) -- same results by
// other code paths (tests/test_gpu_alt_paths.py).  Bits 0-3 are timing ablations that give WRONG results (skip the atomics,
// one pixel per bucket, no SH fetch, no stores); they exist only in the separate ablation build (`make ablate` ->
// libgsr_hip_ablate.so, -DGSR_ABLATE, used by tools/stage_bench.sh) and are compiled out of libgsr_hip.so.
#ifdef GSR_ABLATE
#define GSR_DEBUG_ALLOWED (1 | 2 | 4 | 8 | 32 | 64 | 128 | 256 | 512 | 1024)
#define GSR_ABL(flags, bit) (((flags) & (bit)) != 0)
#else
#define GSR_DEBUG_ALLOWED (32 | 64 | 128 | 256 | 512 | 1024)
#define GSR_ABL(flags, bit) false
#endif
extern int gsr_debug_flags;
extern int gsr_fwd_xcd_map;        // GSR_FWD_XCD: neighbouring tiles of the forward blend on one XCD (blend_fwd.hip)
extern int gsr_bwd_no_order;
extern int gsr_bwd_xcd_map;        // GSR_BWD_XCD: a tile's blocks of the backward blend on one XCD (blend_bwd_splat.hip)
extern int gsr_bwd_block;          // GSR_BWD_BLOCK: pixels per wave in the Gaussian-parallel backward (64, 32, 16); 0 = per frame
#define GSR_BWD_WIDE_PAIRS 20         // D / N from which the backward blend takes 8x8 blocks instead of 8x4 ...
#define GSR_BWD_WIDE_PAIRS_UNFILED 5  // ... and for images of more than GSR_BO_MAX_TILES tiles (no block order for 8x4 there)
// pixels per backward-blend wave for a frame of N Gaussians, D tile pairs, `tiles` tiles (blend_bwd_splat.hip, the launcher): the
// forward asks too -- it files the 8x4 blocks by cost only for a frame whose backward will run 8x4 blocks
static inline int gsr_bwd_block_px(int64_t N, int64_t D, int tiles)
{
    if (gsr_bwd_block > 0) return gsr_bwd_block;
    return D >= (int64_t)(tiles > GSR_BO_MAX_TILES ? GSR_BWD_WIDE_PAIRS_UNFILED : GSR_BWD_WIDE_PAIRS) * N ? 64 : 32;
}
