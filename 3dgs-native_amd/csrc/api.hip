// api.hip -- the extern "C" entry points of libgsr_hip.so (include/gsr.h): argument checks, workspace
// carving and the launch sequence of each stage.  No device memory is allocated here.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "gsr_internal.h"

namespace {

#define HIP_TRY(expr)                                                                                                         \
    do {                                                                                                                      \
        hipError_t e_ = (expr);                                                                                               \
        if (e_ != hipSuccess) {                                                                                               \
            fprintf(stderr, "libgsr_hip: %s failed at %s:%d: %s\n", #expr, __FILE__, __LINE__, hipGetErrorString(e_));       \
            return GSR_E_HIP;                                                                                                 \
        }                                                                                                                     \
    } while (0)

struct Carver {
    char *p;
    size_t off = 0;
    explicit Carver(void *base) : p((char *)base) {}
    template <class T> T *take(size_t count)
    {
        T *r = (T *)(p ? p + off : nullptr);
        off += gsr_align(count * sizeof(T));
        return r;
    }
};

int tile_bits(int tiles)
{
    int b = 1;
    while ((1LL << b) < tiles) ++b;
    return b;
}

struct BinWs {
    int32_t *hist;       // [nb][256]
    int32_t *acc[2];     // [gsr_radix_acc_ints(D)] each: see GeomWs::acc
    int32_t *edge;       // [3 * 256 * nb] first tile / last tile / position of every (digit, block) run of the last pass
    uint64_t *tile_a;    // [D]
    uint64_t *tile_b;    // [D]
    size_t bytes;
};
BinWs carve_bin(void *base, int64_t N, int64_t D)
{
    Carver c(base);
    BinWs w;
    (void)N;
    w.hist = c.take<int32_t>(256 * ((size_t)gsr_radix_blocks(D) + 1));
    w.acc[0] = c.take<int32_t>(gsr_radix_acc_ints(D));
    w.acc[1] = c.take<int32_t>(gsr_radix_acc_ints(D));
    w.edge = c.take<int32_t>(3 * 256 * ((size_t)gsr_radix_blocks(D) + 1));
    w.tile_a = c.take<uint64_t>((size_t)D);
    w.tile_b = c.take<uint64_t>((size_t)D);
    w.bytes = c.off + 256;
    return w;
}

struct BwdWs {
    BlendRec *rec; // [N]
    GradRec *acc;  // [N]
    size_t bytes;
};
BwdWs carve_bwd(void *base, int64_t N)
{
    Carver c(base);
    BwdWs w;
    w.rec = c.take<BlendRec>((size_t)N);
    w.acc = c.take<GradRec>((size_t)N);
    w.bytes = c.off + 256;
    return w;
}

int check_scene_cam(const GsrScene *sc, const GsrCamera *cam)
{
    if (!sc || !cam) return GSR_E_NULL;
    if (sc->N < 0 || sc->N > 0x7FFFFFFFLL || cam->W <= 0 || cam->H <= 0 || sc->sh_degree < 0 || sc->sh_degree > 3) return GSR_E_DIMS;
    if ((cam->W + 15) / 16 > 65535 || (cam->H + 15) / 16 > 65535) return GSR_E_DIMS;
    if (sc->N > 0 && (!sc->means || !sc->scales || !sc->rotations || !sc->opacity || !sc->sh)) return GSR_E_NULL;
    if (!gsr_aligned16(sc->means) || !gsr_aligned16(sc->scales) || !gsr_aligned16(sc->rotations) || !gsr_aligned16(sc->opacity) ||
        !gsr_aligned16(sc->sh))
        return GSR_E_ALIGN;
    return GSR_OK;
}

CamK make_cam(const GsrCamera *c)
{
    CamK k;
    memcpy(k.view, c->view, sizeof(k.view));
    memcpy(k.proj, c->proj, sizeof(k.proj));
    memcpy(k.campos, c->campos, sizeof(k.campos));
    memcpy(k.bg, c->bg, sizeof(k.bg));
    k.tan_fovx = c->tan_fovx;
    k.tan_fovy = c->tan_fovy;
    k.focal_x = c->focal_x;
    k.focal_y = c->focal_y;
    k.W = c->W;
    k.H = c->H;
    k.grid_x = (c->W + GSR_TILE - 1) / GSR_TILE;
    k.grid_y = (c->H + GSR_TILE - 1) / GSR_TILE;
    return k;
}

bool geom_aligned(const GsrGeom *g)
{
    return gsr_aligned16(g->radii) && gsr_aligned16(g->tiles_touched) && gsr_aligned16(g->point_offsets) && gsr_aligned16(g->xy) &&
           gsr_aligned16(g->depths) && gsr_aligned16(g->cov3D) && gsr_aligned16(g->rgb) && gsr_aligned16(g->conic_opacity) &&
           gsr_aligned16(g->clamped_state) && gsr_aligned16(g->blend_records) && gsr_aligned16(g->sh_dir_grad);
}
bool grads_aligned(const GsrGrads *g)
{
    return gsr_aligned16(g->dL_dmean3D) && gsr_aligned16(g->dL_dscale) && gsr_aligned16(g->dL_drot) && gsr_aligned16(g->dL_dopacity) &&
           gsr_aligned16(g->dL_dshs) && gsr_aligned16(g->dL_dcolor) && gsr_aligned16(g->dL_dmean2D) && gsr_aligned16(g->dL_dconic) &&
           gsr_aligned16(g->dL_drgb);
}

bool geom_ok(const GsrGeom *g)
{
    // xy / conic_opacity / rgb may be absent when the caller takes them as columns of its own record buffer (GsrGeom.blend_records)
    const bool arrays = g && ((g->xy && g->rgb && g->conic_opacity) || g->blend_records);
    return arrays && g->radii && g->tiles_touched && g->point_offsets && g->depths && g->cov3D &&
           g->clamped_state;
}

// ---- stage timing (profiling aid) ----
// Process-wide by design (one benchmark drives it); `on` is atomic so the hot path pays one relaxed load when it is off,
// and every read-modify-write of the counters happens under g_timer_mu, so host threads driving different streams may
// all run with timing enabled (their samples interleave in the one record).
struct StageTimer {
    std::atomic<bool> on{false};
    int max_steps = 0, fwd_step = 0, bwd_step = 0; // recorded (sampled) steps so far
    int every = 1, fwd_calls = 0, bwd_calls = 0;    // record one call in `every`; event records cost ~3 us each
    hipEvent_t *ev = nullptr; // [max_steps][GSR_NSTAGES + 3]
    static constexpr int PER = GSR_NSTAGES + 3;
    hipEvent_t &at(int step, int k) { return ev[(size_t)step * PER + k]; }
} g_timer;
std::mutex g_timer_mu;
// event slots: 0..9 forward boundaries (before stage 0 .. after stage 8, with slot 3 = after sync),
// 10..13 backward boundaries
inline void mark(int step, int slot, hipStream_t s)
{
    if (step >= 0 && step < g_timer.max_steps) (void)hipEventRecord(g_timer.at(step, slot), s);
}
// which record (if any) a call samples into: -1 = not sampled.  The forward's two entry points share one record, as do
// the two halves of a split backward; the step is carried between them per host thread.
thread_local int t_fwd_record = -1, t_bwd_record = -1;
int timer_open(bool forward)
{
    if (!g_timer.on.load(std::memory_order_relaxed)) return -1;
    std::lock_guard<std::mutex> lk(g_timer_mu);
    if (!g_timer.on.load(std::memory_order_relaxed)) return -1;
    int &calls = forward ? g_timer.fwd_calls : g_timer.bwd_calls;
    int &step = forward ? g_timer.fwd_step : g_timer.bwd_step;
    if (g_timer.every == 0) return -1; // paused: the events exist, nothing is recorded
    if ((calls++ % g_timer.every) != 0 || step >= g_timer.max_steps) return -1;
    return step++;
}

// ---- readback slots for D ----
// 4 bytes of pinned host memory + an event per call IN FLIGHT: gsr_forward_count leases a slot of its device from this
// pool and returns it before it returns, so two host threads counting on the same device never share a word, and a
// thread that exits leaks nothing (slots are created on demand and kept for the life of the process).  With the pinned
// word the host can wait for D alone while the GPU already runs the depth sort, which does not depend on D.
struct Readback {
    int dev = -1;
    bool busy = false;
    int32_t *pinned = nullptr;
    hipEvent_t ev = nullptr;
};
std::mutex g_rb_mu;
std::vector<Readback *> g_rb_pool;
Readback *readback_acquire()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_rb_mu);
    for (Readback *r : g_rb_pool)
        if (r->dev == dev && !r->busy) {
            r->busy = true;
            return r;
        }
    Readback *r = new Readback;
    r->dev = dev;
    if (hipHostMalloc((void **)&r->pinned, 256, hipHostMallocMapped) != hipSuccess ||       // device-writable
        hipEventCreateWithFlags(&r->ev, hipEventDisableTiming) != hipSuccess) {
        if (r->pinned) (void)hipHostFree(r->pinned);
        delete r;
        return nullptr;
    }
    r->busy = true;
    g_rb_pool.push_back(r);
    return r;
}
struct ReadbackLease {
    Readback *r;
    ReadbackLease() : r(readback_acquire()) {}
    ~ReadbackLease()
    {
        if (!r) return;
        std::lock_guard<std::mutex> lk(g_rb_mu);
        r->busy = false;
    }
};

// ---- the count gsr_forward_count returned, per geom workspace ----
// gsr_forward_render trusts nothing about GsrBinning.D that it can check: the expansion, both partition passes and the
// range scan are sized by D, and a D that is not the count of the depth-sorted items in geom_ws would make them write out
// of bounds (too large) or truncate the list silently (too small).  The true count never leaves the host, so it is
// remembered here, keyed by (device, geom_ws); a small LRU table, guarded by a mutex.
struct CountNote {
    int dev;
    const void *ws;
    int64_t N, D;
    uint64_t stamp;
    int depth_passes; // how many depth-sort passes the last frame counted in this workspace needed: the next frame's launch guess
};
std::mutex g_note_mu;
std::vector<CountNote> g_notes;
uint64_t g_note_clock = 0;
constexpr size_t MAX_NOTES = 256;
void note_count(const void *ws, int64_t N, int64_t D, int depth_passes)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_note_mu);
    CountNote *slot = nullptr;
    for (CountNote &c : g_notes)
        if (c.dev == dev && c.ws == ws) slot = &c;
    if (!slot) {
        if (g_notes.size() < MAX_NOTES) {
            g_notes.push_back(CountNote{});
            slot = &g_notes.back();
        } else {
            slot = &*std::min_element(g_notes.begin(), g_notes.end(), [](const CountNote &a, const CountNote &b) { return a.stamp < b.stamp; });
        }
    }
    *slot = CountNote{dev, ws, N, D, ++g_note_clock, depth_passes};
}
// the depth-sort launch guess for a workspace: what its last frame needed, four if it has none
int depth_pass_guess(const void *ws)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_note_mu);
    for (CountNote &c : g_notes)
        if (c.dev == dev && c.ws == ws) return c.depth_passes >= 1 && c.depth_passes <= 4 ? c.depth_passes : 4;
    return 4;
}
// 1 = matches, 0 = mismatch, -1 = this workspace has no recorded count
int check_count(const void *ws, int64_t N, int64_t D)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_note_mu);
    for (CountNote &c : g_notes)
        if (c.dev == dev && c.ws == ws) {
            c.stamp = ++g_note_clock;
            return (c.N == N && c.D == D) ? 1 : 0;
        }
    return -1;
}

bool fwd_order_wanted(int tiles) { return tiles > 0 && tiles <= GSR_FO_MAX_TILES && !gsr_fwd_no_order && !gsr_fwd_xcd_map; }

std::once_flag g_tuning_once;
void read_tuning()
{
    std::call_once(g_tuning_once, [] {
        if (const char *e = getenv("GSR_DEBUG")) gsr_debug_flags = atoi(e) & GSR_DEBUG_ALLOWED;
        if (const char *e = getenv("GSR_BWD_BLOCK")) gsr_bwd_block = atoi(e);
        if (const char *e = getenv("GSR_BWD_XCD")) gsr_bwd_xcd_map = atoi(e);
        if (const char *e = getenv("GSR_FWD_XCD")) gsr_fwd_xcd_map = atoi(e) != 0;
        if (const char *e = getenv("GSR_BWD_NO_ORDER")) gsr_bwd_no_order = atoi(e) != 0;
        if (const char *e = getenv("GSR_NO_DEPTH_PACK")) gsr_no_depth_pack = atoi(e) != 0;
        if (const char *e = getenv("GSR_NO_NARROWING")) gsr_no_narrowing = atoi(e) != 0;
        if (const char *e = getenv("GSR_FWD_NO_ORDER")) gsr_fwd_no_order = atoi(e) != 0;
    });
}

} // namespace

GeomWs gsr_carve_geom(void *base, int64_t N)
{
    Carver c(base);
    GeomWs w;
    w.rec = c.take<BlendRec>((size_t)N); // FIRST: a caller may hand the start of geom_ws back to gsr_backward as GsrGeom.blend_records
    w.rect = c.take<TileRect>((size_t)N);
    w.depth_item = c.take<uint64_t>((size_t)N);
    w.sort_tmp = c.take<uint64_t>((size_t)N);
    w.id_sorted = c.take<uint32_t>((size_t)N);
    w.blk_minmax = c.take<uint32_t>(4 * (size_t)gsr_div_up(N, 256) + 4);
    w.depth_ctl = w.blk_minmax ? w.blk_minmax + 4 * (size_t)gsr_div_up(N, 256) : nullptr; // the uint4 behind the last block's (preprocess clears it)
    w.rect_sorted = c.take<TileRect>((size_t)N);
    w.cnt_sorted = c.take<int32_t>((size_t)N);
    w.doff = c.take<int32_t>((size_t)N);
    w.scan_tmp = c.take<int32_t>((size_t)gsr_div_up(N, 256) + 4);
    w.hist = c.take<int32_t>(256 * ((size_t)gsr_radix_blocks(N) + 1));
    w.acc[0] = c.take<int32_t>(3 * gsr_radix_acc_ints(N)); // the accumulators of both pass parities and of the first active pass, contiguous: preprocess clears them in one go
    w.acc[1] = w.acc[0] ? w.acc[0] + gsr_radix_acc_ints(N) : nullptr;
    w.acc_first = w.acc[0] ? w.acc[0] + 2 * gsr_radix_acc_ints(N) : nullptr;
    w.sum4096 = c.take<int32_t>((size_t)gsr_div_up(N, 4096) + 4);
    w.fwd_cost = c.take<int32_t>(4 * (size_t)GSR_FO_MAX_TILES); // (behind everything else: they move when N changes, and the order
    w.fwd_order = c.take<int32_t>((size_t)GSR_FO_MAX_TILES);    // made from a moved -- i.e. arbitrary -- cost table is still a permutation)
    w.bytes = c.off + 256;
    return w;
}

extern "C" {

int gsr_abi_version(void) { return GSR_ABI_VERSION; }

int gsr_build_flags(void)
{
#ifdef GSR_ABLATE
    return GSR_BUILD_ABLATE;
#else
    return 0;
#endif
}

const char *gsr_strerror(int code)
{
    switch (code) {
    case GSR_OK: return "ok";
    case GSR_E_NULL: return "required pointer is null";
    case GSR_E_DIMS: return "invalid dimensions or SH degree";
    case GSR_E_OVERFLOW: return "Number of rendered points exceeds the maximum supported (2^30)";
    case GSR_E_WORKSPACE: return "workspace missing or too small";
    case GSR_E_HIP: return "HIP runtime error (see stderr)";
    case GSR_E_CAPACITY: return "GsrBinning.D is not the count gsr_forward_count returned for this geom workspace";
    case GSR_E_ALIGN: return "an array pointer is not 16-byte aligned";
    default: return "unknown error";
    }
}

size_t gsr_geom_workspace_bytes(int64_t N) { return gsr_carve_geom(nullptr, N < 0 ? 0 : N).bytes; }
size_t gsr_binning_workspace_bytes(int64_t N, int64_t D, int32_t, int32_t) { return carve_bin(nullptr, N < 0 ? 0 : N, D < 0 ? 0 : D).bytes; }
size_t gsr_backward_workspace_bytes(int64_t N, int64_t, int32_t, int32_t) { return carve_bwd(nullptr, N < 0 ? 0 : N).bytes; }
size_t gsr_backward_accumulators_offset(int64_t N) { return gsr_align((size_t)(N < 0 ? 0 : N) * sizeof(BlendRec)); } // carve_bwd: the records come first
size_t gsr_block_order_ints(int32_t W, int32_t H)
{
    if (W <= 0 || H <= 0) return 0;
    const int64_t tiles = (int64_t)((W + GSR_TILE - 1) / GSR_TILE) * ((H + GSR_TILE - 1) / GSR_TILE);
    // images of more than GSR_BO_MAX_TILES tiles are never filed (gsr_internal.h): only the header (counters + the `filed` flag) is touched
    return tiles > GSR_BO_MAX_TILES ? (size_t)GSR_BO_HEADER : gsr_bo_ints((int)tiles, (W + GSR_TILE - 1) / GSR_TILE);
}

int gsr_forward_count(const GsrScene *scene, const GsrCamera *camera, const GsrGeom *geom, void *geom_ws, size_t geom_ws_bytes,
                      int64_t *num_rendered, void *stream)
{
    read_tuning();
    if (int rc = check_scene_cam(scene, camera)) return rc;
    if (!num_rendered) return GSR_E_NULL;
    *num_rendered = 0;
    const int64_t N = scene->N;
    if (N == 0) return GSR_OK; // reference behaviour undefined (quirk Q10): empty buffers, D = 0
    if (!geom_ok(geom)) return GSR_E_NULL;
    if (!geom_aligned(geom) || !gsr_aligned16(geom_ws)) return GSR_E_ALIGN;
    if (!geom_ws || geom_ws_bytes < gsr_geom_workspace_bytes(N)) return GSR_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const CamK cam = make_cam(camera);
    const GeomWs ws = gsr_carve_geom(geom_ws, N);
    const int st = t_fwd_record = timer_open(true);
    mark(st, 0, s);
    HIP_TRY(gsr_launch_preprocess(*scene, cam, *geom, ws, s, fwd_order_wanted(cam.grid_x * cam.grid_y)));
    mark(st, 1, s);
    ReadbackLease lease;
    Readback *rb = lease.r;
    if (!rb) return GSR_E_HIP;
    // the scan's last wave stores D = point_offsets[N-1] straight into the pinned host word
    // (preprocess left one partial sum per 256 Gaussians in scan_tmp: one launch)
    // (its first wave also turns the per-block depth extremes preprocess left into the depth sort's pass plan, on the device)
    // (words 2 .. : the visible depth extremes, one pair per control workgroup of the scan's launch -- the host derives from them
    // how many depth passes this frame needed, its launch guess for the next)
    if (gsr_small_depth_path(N)) HIP_TRY(gsr_launch_scan(geom->tiles_touched, nullptr, geom->point_offsets, ws.scan_tmp, N, 0, rb->pinned, true, s, ws.blk_minmax, ws.depth_ctl));
    else HIP_TRY(gsr_launch_scan_ctl_hist(geom->tiles_touched, geom->point_offsets, ws, N, rb->pinned, s)); // + the first active depth pass's histogram
    mark(st, 2, s);
    HIP_TRY(hipEventRecord(rb->ev, s));
    // Work that does not need D goes out before the host waits: Gaussians by depth bits (stable from id order, four 8-bit
    // passes over the high word, ending back in depth_item; the last one also carries each Gaussian's tile rectangle and
    // tile count to its sorted position) and the depth-order offsets (exclusive scan of those counts).
    // How many of the four 8-bit passes this frame needs is decided on the device (DepthCtl); the host launches as many as the
    // previous frame in this workspace needed (its guess; four the first time).  If the guess turns out too low the launched
    // passes leave the data alone and all four are launched once the readback has said so.
    const int guess = (gsr_debug_flags & 256) ? 4 : depth_pass_guess(geom_ws);
    // (packed depth items, scan_sort.hip: the tile grid at 6 bits per coordinate, the ids in 24)
    const int pack_ok = (!gsr_no_depth_pack && cam.grid_x <= 63 && cam.grid_y <= 63 && N <= (1 << 24)) ? 1 : 0;
    HIP_TRY(gsr_launch_depth_sort(ws, N, s, guess, pack_ok));
    HIP_TRY(hipEventSynchronize(rb->ev)); // D (and the pass count) are on the host; the GPU keeps sorting
    const int32_t last = *rb->pinned;
    int needed = 4;
    if (!gsr_small_depth_path(N)) {
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
        for (int k = 0, K = gsr_depth_ctl_wgs(N); k < K; ++k) {
            lo = std::min(lo, (uint32_t)rb->pinned[2 + 2 * k]);
            hi = std::max(hi, (uint32_t)rb->pinned[3 + 2 * k]);
        }
        needed = gsr_depth_plan(lo, hi, (gsr_debug_flags & 256) ? 4 : 0).npass;
    }
    if (needed > guess && !gsr_small_depth_path(N)) HIP_TRY(gsr_launch_depth_sort(ws, N, s, 4, pack_ok));
    mark(st, 3, s);
    // (the depth-order offsets are made by gsr_forward_render, next to their one reader -- the expansion; the alternative path of
    // GSR_DEBUG bit 9 scans them here, into ws.doff)
    if ((gsr_debug_flags & 512) && !gsr_small_depth_path(N)) HIP_TRY(gsr_launch_scan(ws.cnt_sorted, nullptr, ws.doff, ws.scan_tmp, N, 2, nullptr, false, s));
    *num_rendered = (int64_t)last;
    if (last < 0 || (int64_t)last > GSR_MAX_RENDERED) return GSR_E_OVERFLOW;
    note_count(geom_ws, N, (int64_t)last, needed);
    return GSR_OK;
}

int gsr_forward_render(const GsrScene *scene, const GsrCamera *camera, const GsrGeom *geom, const GsrBinning *binning,
                       const GsrImage *image, void *geom_ws, size_t geom_ws_bytes, void *bin_ws, size_t bin_ws_bytes, void *stream)
{
    read_tuning();
    if (int rc = check_scene_cam(scene, camera)) return rc;
    if (!binning || !image || !image->image || !image->inv_depth || !image->final_T || !image->n_contrib || !binning->ranges)
        return GSR_E_NULL;
    const int64_t N = scene->N, D = binning->D;
    if (D < 0 || D > GSR_MAX_RENDERED) return GSR_E_OVERFLOW;
    hipStream_t s = (hipStream_t)stream;
    const CamK cam = make_cam(camera);
    const size_t P = (size_t)cam.W * cam.H;
    const int tiles = cam.grid_x * cam.grid_y;
    if (D == 0 || N == 0) { // reference skips the blend: zeros, not background (forward.py:830, quirk Q10)
        t_fwd_record = -1;  // a sampled record that ends here stays incomplete and is dropped by gsr_stage_times
        HIP_TRY(hipMemsetAsync(binning->ranges, 0, sizeof(int32_t) * 2 * tiles, s));
        HIP_TRY(hipMemsetAsync(image->image, 0, P * 3 * sizeof(float), s));
        HIP_TRY(hipMemsetAsync(image->inv_depth, 0, P * sizeof(float), s));
        HIP_TRY(hipMemsetAsync(image->final_T, 0, P * sizeof(float), s));
        HIP_TRY(hipMemsetAsync(image->n_contrib, 0, P * sizeof(int32_t), s));
        // the accumulator clear of GsrBinning.backward_ws is promised whenever the workspace is handed over, blend or no blend
        if (binning->backward_ws && N > 0) {
            if (!gsr_aligned16(binning->backward_ws)) return GSR_E_ALIGN;
            HIP_TRY(hipMemsetAsync(carve_bwd(binning->backward_ws, N).acc, 0, sizeof(GradRec) * (size_t)N, s));
        }
        return GSR_OK;
    }
    if (!geom_ok(geom) || !binning->point_list) return GSR_E_NULL;
    if (!geom_aligned(geom) || !gsr_aligned16(geom_ws) || !gsr_aligned16(bin_ws) || !gsr_aligned16(binning->point_list) ||
        !gsr_aligned16(binning->ranges) || !gsr_aligned16(binning->block_order) || !gsr_aligned16(image->image) || !gsr_aligned16(image->inv_depth) ||
        !gsr_aligned16(image->final_T) || !gsr_aligned16(image->n_contrib))
        return GSR_E_ALIGN;
    if (!geom_ws || geom_ws_bytes < gsr_geom_workspace_bytes(N)) return GSR_E_WORKSPACE;
    if (!bin_ws || bin_ws_bytes < gsr_binning_workspace_bytes(N, D, cam.W, cam.H)) return GSR_E_WORKSPACE;
    // D must be the count gsr_forward_count returned for the items now in geom_ws (see CountNote above)
    if (check_count(geom_ws, N, D) != 1) return GSR_E_CAPACITY;
    const GeomWs gw = gsr_carve_geom(geom_ws, N);
    const BinWs bw = carve_bin(bin_ws, N, D);

    const int st = t_fwd_record;
    t_fwd_record = -1;
    mark(st, 4, s); // (stage 3 -> 4: the host between the two calls -- the wait for D, the caller's allocations)
    // 3. expansion of the depth-sorted Gaussians (gsr_forward_count) to (tile << id_shift | id) items.  When the tile
    //    bits and the id bits fit one 32-bit word (800x800 with 1M Gaussians: 12 + 20) the items are uint32, which
    //    halves the traffic of the expansion, both partition passes and the range scan.
    const int tb = tile_bits(tiles);
    int id_bits = 1;
    while ((1LL << id_bits) < N) ++id_bits;
    const bool narrow = tb + id_bits <= 32 && !(gsr_debug_flags & 32); // GSR_DEBUG bit 5: 64-bit tile items at any size (tests)
    const int id_shift = narrow ? id_bits : 32, item_bytes = narrow ? 4 : 8;
    // the stable partition by tile id below: ceil(tb/8) passes over the tile-id bits, split as evenly as possible (12 bits -> 6+6, 13 -> 7+6)
    const int npass = (tb + 7) / 8;
    auto pass_bits = [&](int pass, int shift) { return std::max(4, (tb - shift + (npass - pass) - 1) / (npass - pass)); };
    // the block order (forward -> backward scratch): header cleared here; filed by the blend below unless the image is large
    int32_t *order = binning->block_masks ? binning->block_order : nullptr;
    // (and not for a frame whose backward will take 8x8 blocks, which run in band order: the `filed` flag then stays 0)
    const bool file_order = order && tiles <= GSR_BO_MAX_TILES && gsr_bwd_block_px(N, D, tiles) == 32;
    const bool by_gaussian = (gsr_debug_flags & 512) != 0; // GSR_DEBUG bit 9: the expansion by Gaussian + the first pass's own histogram kernel (tests, A/B)
    // the forward blend's tiles by last frame's cost classes (gsr_internal.h "forward tile order"): the table was made by the spare
    // workgroup of this frame's preprocess (gsr_forward_count, same condition), so it is never stale or foreign
    const bool use_fwd_order = fwd_order_wanted(tiles);
    if (by_gaussian) {
        mark(st, 5, s);
        HIP_TRY(gsr_launch_expand(gw.id_sorted, gw.doff, gw.rect_sorted, bw.tile_a, N, cam.grid_x, D, id_shift, item_bytes, binning->ranges, 2 * tiles, bw.acc[0],
                                  (int)gsr_radix_acc_ints(D), order, order ? GSR_BO_HEADER : 0, file_order ? 1 : 0, s));
    } else {
        // one offset per 256 depth-sorted Gaussians (stage "depth_scan"), then one workgroup per radix block of the item array,
        // which also leaves the first partition pass's block histograms
        HIP_TRY(gsr_launch_depth_block_offsets(gw, N, binning->ranges, 2 * tiles, bw.acc[0], (int)gsr_radix_acc_ints(D), order, order ? GSR_BO_HEADER : 0,
                                               file_order ? 1 : 0, s));
        mark(st, 5, s);
        HIP_TRY(gsr_launch_expand_blocks(gw, bw.tile_a, N, cam.grid_x, D, id_shift, item_bytes, pass_bits(0, 0), bw.hist, bw.acc[0], s));
    }
    mark(st, 6, s);
    // 4. stable partition by tile id
    void *tsrc = bw.tile_a, *tdst = bw.tile_b;
    // Two passes over 64-bit items whose remaining tile bits + id bits fit a word after the first one (1080p with 5 M Gaussians:
    // 6 + 23): the first pass writes 32-bit items and the second recovers the first digit from the item's position
    // (scan_sort.hip ScatterFinal) -- 4 instead of 8 bytes per item through the second histogram and the final scatter.
    const int bits0 = pass_bits(0, 0);
    const bool narrowing = !narrow && npass == 2 && (tb - bits0) + id_bits <= 32 && !gsr_no_narrowing;
    for (int pass = 0, shift = 0; pass < npass; ++pass) {
        const int bits = pass_bits(pass, shift);
        const bool hist_ready = pass == 0 && !by_gaussian;
        if (pass + 1 < npass) {
            HIP_TRY(gsr_launch_radix_pass(tsrc, tdst, bw.hist, bw.acc[pass & 1], D, id_shift + shift, bits, item_bytes, bw.acc[(pass + 1) & 1], s, hist_ready,
                                          narrowing ? id_bits : 0, narrowing ? bw.acc[pass & 1] : nullptr));
        } else if (narrowing) {
            HIP_TRY(gsr_launch_radix_final_pass(tsrc, bw.hist, bw.acc[pass & 1], D, id_bits, bits, 4, id_bits, binning->point_list, binning->ranges, bw.edge, s,
                                                false, bw.acc[(pass + 1) & 1], bits0));
        } else {
            // 5. the last pass writes point_list and the tile ranges itself (reference forward.py:806-824, :561-586) instead of
            //    sorted items that a further kernel would re-read
            HIP_TRY(gsr_launch_radix_final_pass(tsrc, bw.hist, bw.acc[pass & 1], D, id_shift + shift, bits, item_bytes, id_shift, binning->point_list,
                                                binning->ranges, bw.edge, s, hist_ready));
        }
        shift += bits;
        void *t = tsrc; tsrc = tdst; tdst = t;
    }
    mark(st, 7, s);
    mark(st, 8, s); // (stage slot "ranges": nothing left in it)
    // 6. blend
        // the backward's accumulator records are cleared by the blend kernel's spare workgroups when the caller hands its backward
    // workspace over (GsrBinning.backward_ws); it then tells gsr_backward so (GsrBinning.backward_ws_cleared)
    void *clear = nullptr;
    size_t clear_bytes = 0;
    if (binning->backward_ws) {
        if (!gsr_aligned16(binning->backward_ws)) return GSR_E_ALIGN;
        clear = carve_bwd(binning->backward_ws, N).acc;
        clear_bytes = sizeof(GradRec) * (size_t)N;
    }
    HIP_TRY(gsr_launch_blend_forward(cam, binning->ranges, binning->point_list, geom->blend_records ? (const BlendRec *)geom->blend_records : gw.rec, *image, binning->block_masks, file_order ? order : nullptr,
                                     clear, clear_bytes, s, use_fwd_order ? gw.fwd_order : nullptr, tiles <= GSR_FO_MAX_TILES ? gw.fwd_cost : nullptr));
    mark(st, 9, s);
    return GSR_OK;
}

// first half: accumulator clear, record (re)pack, blend backward, optional view payload
static int backward_blend_impl(const GsrScene *scene, const GsrCamera *camera, const GsrGeom *geom, const GsrBinning *binning,
                               const GsrImage *image, const float *dL_dpixels, float *payload, void *ws, size_t ws_bytes, hipStream_t s, int st)
{
    const int64_t N = scene->N;
    if (!geom || !geom->radii || !geom->clamped_state) return GSR_E_NULL; // (cov3D may be NULL: gsr.h GsrGeom)
    if (!geom->blend_records && (!geom->xy || !geom->rgb || !geom->conic_opacity)) return GSR_E_NULL; // the records, or what they are rebuilt from
    if (!binning || !image || !dL_dpixels) return GSR_E_NULL;
    if (!geom_aligned(geom) || !gsr_aligned16(ws) || !gsr_aligned16(binning->point_list) || !gsr_aligned16(binning->ranges) ||
        !gsr_aligned16(binning->block_masks) || !gsr_aligned16(binning->block_order) ||
        !gsr_aligned16(image->final_T) || !gsr_aligned16(image->n_contrib) || !gsr_aligned16(dL_dpixels) || !gsr_aligned16(payload))
        return GSR_E_ALIGN;
    const int64_t D = binning->D;
    if (D < 0 || D > GSR_MAX_RENDERED) return GSR_E_OVERFLOW;
    if (D > 0 && (!binning->point_list || !binning->ranges || !image->final_T || !image->n_contrib)) return GSR_E_NULL;
    if (!ws || ws_bytes < gsr_backward_workspace_bytes(N, D, camera->W, camera->H)) return GSR_E_WORKSPACE;
    const CamK cam = make_cam(camera);
    const BwdWs bw = carve_bwd(ws, N);
    mark(st, 10, s);
    // (unless gsr_forward_render cleared this very workspace's accumulators in its blend kernel and nothing has used it since)
    if (!(binning->backward_ws_cleared && binning->backward_ws == ws)) HIP_TRY(hipMemsetAsync(bw.acc, 0, sizeof(GradRec) * (size_t)N, s));
    const BlendRec *records = (const BlendRec *)geom->blend_records;
    if (D > 0 && !records) {
        HIP_TRY(gsr_launch_pack_records(*geom, bw.rec, N, s));
        records = bw.rec;
    }
    mark(st, 11, s);
    if (D > 0) HIP_TRY(gsr_launch_blend_backward_splat(cam, binning->ranges, binning->point_list, records, *image, dL_dpixels, binning->block_masks,
                                                       binning->block_masks ? binning->block_order : nullptr, bw.acc, N, D, s));
    mark(st, 12, s);
    if (payload) HIP_TRY(gsr_launch_view_payload(*scene, cam, *geom, bw.acc, payload, s));
    return GSR_OK;
}

// second half: the four per-Gaussian kernels of backward_preprocess, fused
static int backward_geom_impl(const GsrScene *scene, const GsrCamera *camera, const GsrGeom *geom, const GsrGrads *grads, void *ws,
                              size_t ws_bytes, hipStream_t s, int st)
{
    const int64_t N = scene->N;
    // dL_dshs and dL_drgb may both be NULL here: the payload was taken from the blend half and the SH gradient is rebuilt later
    // (dL_dcolor / dL_dmean2D / dL_dconic may each be NULL: they are columns of the accumulator records in `ws`, gsr.h GsrGrads)
    if (!grads || !grads->dL_dmean3D || !grads->dL_dscale || !grads->dL_drot || !grads->dL_dopacity) return GSR_E_NULL;
    if (!geom || !geom->radii || !geom->clamped_state) return GSR_E_NULL; // cov3D NULL: recomputed from scales / rotations (gsr.h GsrGeom)
    if (!geom_aligned(geom) || !grads_aligned(grads) || !gsr_aligned16(ws)) return GSR_E_ALIGN;
    if (!ws || ws_bytes < gsr_backward_workspace_bytes(N, 0, camera->W, camera->H)) return GSR_E_WORKSPACE;
    const CamK cam = make_cam(camera);
    const BwdWs bw = carve_bwd(ws, N);
    HIP_TRY(gsr_launch_geom_backward(*scene, cam, *geom, bw.acc, *grads, s));
    mark(st, 13, s);
    return GSR_OK;
}

int gsr_backward(const GsrScene *scene, const GsrCamera *camera, const GsrGeom *geom, const GsrBinning *binning, const GsrImage *image,
                 const float *dL_dpixels, const GsrGrads *grads, void *ws, size_t ws_bytes, void *stream)
{
    read_tuning();
    if (int rc = check_scene_cam(scene, camera)) return rc;
    if (scene->N == 0) return GSR_OK;
    if (!grads || !grads->dL_dmean3D || !grads->dL_dscale || !grads->dL_drot || !grads->dL_dopacity || (!grads->dL_dshs && !grads->dL_drgb))
        return GSR_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    const int st = timer_open(false);
    if (int rc = backward_blend_impl(scene, camera, geom, binning, image, dL_dpixels, nullptr, ws, ws_bytes, s, st)) return rc;
    if (int rc = backward_geom_impl(scene, camera, geom, grads, ws, ws_bytes, s, st)) return rc;
    return GSR_OK;
}

int gsr_backward_blend(const GsrScene *scene, const GsrCamera *camera, const GsrGeom *geom, const GsrBinning *binning, const GsrImage *image,
                       const float *dL_dpixels, float *payload, void *ws, size_t ws_bytes, void *stream)
{
    read_tuning();
    if (int rc = check_scene_cam(scene, camera)) return rc;
    if (scene->N == 0) return GSR_OK;
    // stage events: the two halves of one backward share a record; it is opened here and closed by gsr_backward_geom
    t_bwd_record = timer_open(false);
    return backward_blend_impl(scene, camera, geom, binning, image, dL_dpixels, payload, ws, ws_bytes, (hipStream_t)stream, t_bwd_record);
}

int gsr_backward_geom(const GsrScene *scene, const GsrCamera *camera, const GsrGeom *geom, const GsrGrads *grads, void *ws, size_t ws_bytes,
                      void *stream)
{
    read_tuning();
    if (int rc = check_scene_cam(scene, camera)) return rc;
    if (scene->N == 0) return GSR_OK;
    const int st = t_bwd_record;
    t_bwd_record = -1;
    return backward_geom_impl(scene, camera, geom, grads, ws, ws_bytes, (hipStream_t)stream, st);
}

int gsr_stage_timing(int enable, int max_steps)
{
    std::lock_guard<std::mutex> lk(g_timer_mu);
    if (g_timer.ev) {
        for (size_t i = 0; i < (size_t)g_timer.max_steps * StageTimer::PER; ++i) (void)hipEventDestroy(g_timer.ev[i]);
        free(g_timer.ev);
        g_timer.ev = nullptr;
    }
    g_timer.on = false;
    g_timer.max_steps = g_timer.fwd_step = g_timer.bwd_step = g_timer.fwd_calls = g_timer.bwd_calls = 0;
    g_timer.every = 1;
    if (!enable) return GSR_OK;
    if (max_steps <= 0 || max_steps > 4096) return GSR_E_DIMS;
    g_timer.ev = (hipEvent_t *)calloc((size_t)max_steps * StageTimer::PER, sizeof(hipEvent_t));
    if (!g_timer.ev) return GSR_E_WORKSPACE;
    for (size_t i = 0; i < (size_t)max_steps * StageTimer::PER; ++i) HIP_TRY(hipEventCreate(&g_timer.ev[i]));
    g_timer.max_steps = max_steps;
    g_timer.on = true;
    return GSR_OK;
}

int gsr_stage_sampling(int every)
{
    if (every < 0) return GSR_E_DIMS;
    std::lock_guard<std::mutex> lk(g_timer_mu);
    g_timer.every = every; // 0 = paused (events stay allocated), k >= 1 = one forward/backward pair in k
    g_timer.fwd_calls = g_timer.bwd_calls = 0;
    return GSR_OK;
}

int gsr_stage_times(float *avg_ms, int *steps)
{
    if (!avg_ms || !steps) return GSR_E_NULL;
    std::lock_guard<std::mutex> lk(g_timer_mu);
    for (int k = 0; k < GSR_NSTAGES; ++k) avg_ms[k] = 0.0f;
    int n = g_timer.fwd_step < g_timer.bwd_step ? g_timer.fwd_step : g_timer.bwd_step;
    if (n > g_timer.max_steps) n = g_timer.max_steps;
    *steps = n;
    if (!g_timer.on.load() || n == 0) return GSR_OK;
    // stage k of the forward lies between event slots k and k+1 (k = 0..8); backward stages 9..11 between 10+j and 11+j.
    // A record whose call ended early (an error return, D == 0) has unrecorded events: it is left out.
    int used = 0;
    for (int st = 0; st < n; ++st) {
        float ms[GSR_NSTAGES];
        bool whole = true;
        for (int k = 0; k < 9 && whole; ++k) whole = hipEventElapsedTime(&ms[k], g_timer.at(st, k), g_timer.at(st, k + 1)) == hipSuccess;
        for (int j = 0; j < 3 && whole; ++j) whole = hipEventElapsedTime(&ms[9 + j], g_timer.at(st, 10 + j), g_timer.at(st, 11 + j)) == hipSuccess;
        if (!whole) {
            (void)hipGetLastError();
            continue;
        }
        for (int k = 0; k < GSR_NSTAGES; ++k) avg_ms[k] += ms[k];
        ++used;
    }
    *steps = n = used;
    if (n == 0) return GSR_OK;
    for (int k = 0; k < GSR_NSTAGES; ++k) avg_ms[k] /= (float)n;
    g_timer.fwd_step = g_timer.bwd_step = 0;
    return GSR_OK;
}

} // extern "C"
