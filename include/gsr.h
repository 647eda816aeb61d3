/*
 * gsr.h -- C ABI of libgsr_hip.so, the MI355X (gfx950) 3D-Gaussian-Splatting rasterizer.
 *
 * This is the drop-in boundary for the hot path of zhujinchong/3DGS-native.  The reference has no
 * native FFI of its own: its "operator API" is two Python functions,
 *     render_gaussians(...)   reference forward.py:629-894
 *     backward(...)           reference backward.py:955-1196
 * which launch Warp kernels.  The entry points below are what a binding for that pair calls instead
 * of wp.launch (INTEGRATION.md shows the ctypes stub); 3dgs-native_amd/forward.py and backward.py are
 * that binding, with the reference's keyword arguments and return-dict keys.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer unless marked host.  The caller owns all memory, including
 *     scratch (one byte buffer per call, sized by the *_workspace_bytes functions).  The library
 *     never allocates or frees device memory, so it is re-entrant for distinct buffers.  One process
 *     per GPU for multi-GPU use.
 *   - All work is enqueued on `stream` (a hipStream_t).  Only gsr_forward_count waits on the device: for
 *     the number of (tile, Gaussian) pairs D, which sizes GsrBinning -- the reference's one unavoidable
 *     readback, forward.py:764.  It waits on an event behind the 4-byte copy only, so the depth sort it
 *     has already enqueued keeps running.  State the library keeps (host side, all of it behind mutexes, so host
 *     threads may drive distinct buffers / streams concurrently): a pool of pinned 4-byte readback slots + events, one
 *     leased per gsr_forward_count in flight; the count each geom_ws was last given (checked by gsr_forward_render ->
 *     GSR_E_CAPACITY); the optional profiling aid at the end of this header.
 *   - Layouts are the reference's packed AoS: vec3 = 3 floats, vec4 = 4 floats, VEC6 = 6 floats
 *     (xx,xy,xz,yy,yz,zz; reference forward.py:186), images row-major [y][x].  Quaternions are
 *     (x,y,z,w) (forward.py:177).  Matrices are 16 floats row-major AS STORED by the reference's
 *     callers, used under the row-vector convention p' = p * M.
 *   - Alignment: every array pointer handed to the library must be 16-byte aligned (hipMalloc and torch give 256).  The
 *     kernels move vec4 / vec2 / SH rows as 16- and 8-byte accesses, so this matters for slices of a larger allocation: a
 *     gradient arena [3N | 3N | 4N | N | 48N] must start each segment on a multiple of 4 floats (3dgs-native_amd/dist.py
 *     arena_offsets pads it so).  Entry points check it and return GSR_E_ALIGN.
 *   - Tiles are 16x16 pixels (reference config.py:21-22).
 *   - Return value: GSR_OK or a negative GSR_E_* code; gsr_strerror() names it.
 */
#ifndef GSR_H
#define GSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_ABI_VERSION 7
#define GSR_TILE 16
#define GSR_SH_STRIDE 16 /* SH coefficients per Gaussian, always 16 (reference forward.py:310) */
#define GSR_MAX_RENDERED (1LL << 30) /* reference forward.py:765-767 */

enum {
    GSR_OK = 0,
    GSR_E_NULL = -1,      /* a required pointer is null */
    GSR_E_DIMS = -2,      /* N < 0, W/H <= 0, sh_degree outside 0..3, ... */
    GSR_E_OVERFLOW = -3,  /* D > GSR_MAX_RENDERED (reference raises ValueError, forward.py:765) */
    GSR_E_WORKSPACE = -4, /* workspace null or smaller than *_workspace_bytes(...) */
    GSR_E_HIP = -5,       /* a HIP runtime call or kernel launch failed */
    GSR_E_CAPACITY = -6,  /* GsrBinning.D is not the count gsr_forward_count returned for this geom_ws (or it was never counted) */
    GSR_E_ALIGN = -7      /* an array pointer is not 16-byte aligned (see "Alignment" above) */
};

/* Inputs of wp_preprocess (reference forward.py:190-211). */
typedef struct GsrScene {
    int64_t N;              /* number of Gaussians */
    const float *means;     /* [N*3]  world positions */
    const float *scales;    /* [N*3]  raw scales (no activation, quirk Q5) */
    const float *rotations; /* [N*4]  quaternions (x,y,z,w) */
    const float *opacity;   /* [N]    raw opacities */
    const float *sh;        /* [N*16*3] SH coefficients, stride 16 whatever the degree (quirk Q6) */
    int32_t sh_degree;      /* 0..3 */
    float scale_modifier;
    int32_t clamped;        /* clamp SH colour at 0 (forward.py:364-370) */
} GsrScene;

/* Camera of one view (reference forward.py:694-700, :734-739). */
typedef struct GsrCamera {
    float view[16];   /* `viewmatrix` flattened row-major, as the caller stores it */
    float proj[16];   /* `projmatrix` (full projection) flattened row-major */
    float campos[3];
    float bg[3];
    float tan_fovx, tan_fovy;
    float focal_x, focal_y; /* backward only: W/(2 tan_fovx), H/(2 tan_fovy) formed in float64 on the
                               host and rounded once (reference backward.py:1044-1045, quirk Q8);
                               the forward recomputes focal lengths in float32 (forward.py:115-116) */
    int32_t W, H;
} GsrCamera;

/* Per-Gaussian outputs of the forward pass: the reference's returned dict entries
 * radii / point_offsets / points_xy_image / depths / cov3Ds / colors / conic_opacity / clamped_state
 * (forward.py:881-894) plus tiles_touched.  Culled Gaussians get zeros (quirk Q11). */
typedef struct GsrGeom {
    int32_t *radii;         /* [N] */
    int32_t *tiles_touched; /* [N] */
    int32_t *point_offsets; /* [N] inclusive scan of tiles_touched (utils/wp_utils.py:47-60) */
    float *xy;              /* [N*2] */
    float *depths;          /* [N] */
    float *cov3D;           /* [N*6].  gsr_backward: may be NULL when it would be the forward's own output for this scene's scales,
                               rotations and scale_modifier -- the kernel then recomputes it (the same instructions, the same bits)
                               instead of reading 24 bytes per Gaussian; a caller with its own Sigma3D passes it as before */
    float *rgb;             /* [N*3] */
    float *conic_opacity;   /* [N*4] */
    float *clamped_state;   /* [N*3] */
    void *blend_records;    /* optional [N*16 floats = N 64-byte records, 64-byte rows]: per Gaussian (x, y, conic a, b, c, opacity,
                               r, g, b, 1/depth, 0 ...) -- xy, conic_opacity and rgb of the same Gaussian in one row, the unit the
                               blend kernels gather.
                               gsr_forward_count / gsr_forward_render (ABI 7): given a buffer, the forward keeps its records THERE
                               instead of in geom_ws, and `xy`, `conic_opacity`, `rgb` may then each be NULL: their values are
                               columns 0-1, 2-5 and 6-8 of the records (a caller exposes them as strided views), and the forward
                               saves writing them twice (36 of its 200 bytes per Gaussian).  Arrays that are given are still written.
                               gsr_backward / gsr_backward_blend: the records of the forward call (this buffer, or -- for a
                               forward that was given none -- the START of its geom_ws, if still unmodified); `xy`,
                               `conic_opacity`, `rgb` are then not read and may be NULL.  NULL -> rebuilt from xy / conic_opacity
                               / rgb (same values). */
    float *sh_dir_grad;     /* optional [N*9], not part of the reference's dict.  gsr_forward_count, given a buffer, writes per
                               visible Gaussian the nine sums d(colour before clamping)/d(view direction) that the SH backward
                               (backward.py:120-244) forms from the 48 coefficients: (d/dx, d/dy, d/dz) x (r, g, b), the same
                               float operations.  gsr_backward / gsr_backward_geom, given that buffer back, use it and do not
                               read scene->sh (36 instead of 192 bytes per Gaussian) -- valid only while scene->sh, scene->means,
                               sh_degree and camera->campos are those of the forward call that filled it.  NULL on either
                               side is fine: the backward then reads the coefficients itself.  Same results either way. */
} GsrGeom;

/* Sorted (tile, depth) list: dict entries point_list / ranges. */
typedef struct GsrBinning {
    int64_t D;           /* number of (tile, Gaussian) pairs, from gsr_forward_count */
    int32_t *point_list; /* [D] Gaussian ids sorted by (tile, depth bits, id) */
    int32_t *ranges;     /* [tiles*2] (start,end) per tile, (0,0) for untouched tiles */
    uint8_t *block_masks; /* optional [D] (allocate D rounded up to a multiple of 16 bytes: they are read 16 at a time), not part of
                             the reference's dict: bit k of byte i = list entry i may reach
                             alpha >= 1/255 inside 8x4-pixel block k of its tile (k & 1 = x half, k >> 1 = 4-row band).
                             gsr_forward_render writes it (entries up to each tile's saturation point) when not NULL;
                             gsr_backward, given the SAME array back unmodified, compacts each block's list from these
                             bytes instead of re-deriving the test from the records.  NULL on either side is fine. */
    int32_t *block_order; /* optional [gsr_block_order_ints(W, H)], used only together with block_masks: gsr_forward_render files
                             every 8x4-pixel block under (XCD band of its tile, cost class), the cost being the number of list
                             entries the backward will keep for the block; gsr_backward, given the SAME array back unmodified,
                             starts the heaviest blocks first (its blend kernel's last-started waves decide when it ends:
                             164 -> 154 us at 800x800 / 1 M Gaussians).  Read-only on the backward side.  Execution order
                             only: results are the same up to float-atomic order.  Nothing is filed (a flag in the array says
                             so) for images of more than 4 096 tiles, nor for a frame of large splats (D >= 20 N), whose
                             backward blend runs 8x8-pixel blocks in plain band order. */
    void *backward_ws;    /* optional: the workspace the caller will give gsr_backward (>= gsr_backward_workspace_bytes).  Handed to
                             gsr_forward_render, the forward blend kernel's spare workgroups clear the accumulator records in it
                             while that kernel drains -- the 64 bytes per Gaussian gsr_backward otherwise clears in a launch of
                             its own (12 us at 1 M Gaussians).  Every successful gsr_forward_render that is handed a
                             backward_ws leaves those records zero, also when it launches no blend (D == 0) or cannot host
                             the spare workgroups (it then clears with a memset of its own).  The size is the caller's
                             promise: the library has no byte count to check it against. */
    int32_t backward_ws_cleared; /* gsr_backward only: non-zero = `ws` is that `backward_ws`, and nothing has written to it since the
                             gsr_forward_render that cleared it (no other gsr_backward in particular): the clear is skipped */
} GsrBinning;

/* Per-pixel outputs: image, inverse-depth image, dict entries final_Ts / n_contrib. */
typedef struct GsrImage {
    float *image;       /* [H*W*3] */
    float *inv_depth;   /* [H*W] */
    float *final_T;     /* [H*W] */
    int32_t *n_contrib; /* [H*W] */
} GsrImage;

/* Gradients: the returned dict of backward() (reference backward.py:1185-1196).  The first five may
 * alias one contiguous arena [3N | 3N | 4N | N | 48N] (a single RCCL all-reduce covers it).  Every
 * array is fully written by gsr_backward (no pre-zeroing needed). */
typedef struct GsrGrads {
    float *dL_dmean3D;  /* [N*3] */
    float *dL_dscale;   /* [N*3] */
    float *dL_drot;     /* [N*4] (x,y,z,w) */
    float *dL_dopacity; /* [N] */
    float *dL_dshs;     /* [N*16*3] */
    /* The three blend-stage gradients below may each be NULL (ABI 7).  They are plain copies of what the blend backward accumulates
     * per Gaussian in its 64-byte accumulator records, which start gsr_backward_accumulators_offset(N) bytes into `ws` as N rows of
     * 16 floats laid out like the API arrays: columns 0-2 = dL_dcolor, 3-5 = dL_dmean2D (z = 0), 6-9 = dL_dconic (a, b, 0, c),
     * 10 = dL_dopacity, the rest zero.  A caller that gives every gsr_backward call a `ws` of its own exposes those columns as
     * strided views (3dgs-native_amd/backward.py does) and the per-Gaussian kernel writes 40 bytes per Gaussian less. */
    float *dL_dcolor;   /* [N*3] or NULL */
    float *dL_dmean2D;  /* [N*3] (z = 0) or NULL */
    float *dL_dconic;   /* [N*4] (a, b, 0, c) or NULL */
    /* Optional (view-parallel training, see gsr_sh_grad_from_views): this view's PAYLOAD, [N*3 + 4] floats =
     * N rows of the colour gradient the SH backward starts from, dL_dcolor * (1 - clamped) (backward.py:88-92; zero for
     * Gaussians that get no SH gradient), then the camera position (3 floats) and a zero.  When it is given, dL_dshs may
     * be NULL and the 192-byte-per-Gaussian SH gradient is then not written at all. */
    float *dL_drgb;     /* [N*3 + 4] or NULL */
} GsrGrads;

int gsr_abi_version(void);
const char *gsr_strerror(int code);
/* 0 for the product library.  Bit 0 (GSR_BUILD_ABLATE) marks the separate timing-ablation build (libgsr_hip_ablate.so,
 * `make ablate`), the only build in which GSR_DEBUG bits 0-3 -- switches that skip work and give WRONG results -- exist. */
#define GSR_BUILD_ABLATE 1
int gsr_build_flags(void);

/* Scratch sizes in bytes (host-side arithmetic only). */
size_t gsr_geom_workspace_bytes(int64_t N);
size_t gsr_binning_workspace_bytes(int64_t N, int64_t D, int32_t W, int32_t H);
size_t gsr_backward_workspace_bytes(int64_t N, int64_t D, int32_t W, int32_t H);
size_t gsr_backward_accumulators_offset(int64_t N); /* byte offset of the accumulator records inside that workspace (GsrGrads) */
size_t gsr_block_order_ints(int32_t W, int32_t H); /* int32 elements of GsrBinning.block_order for a W x H image */

/* Stage 1 of render_gaussians: wp_preprocess + wp_prefix_sum + the D readback
 * (reference forward.py:719-767), plus the part of the sort that does not depend on D (Gaussians by
 * depth).  Fills *geom, leaves per-Gaussian blend records, depth-sorted ids and offsets in geom_ws
 * (which must be passed unchanged to gsr_forward_render), and returns D in *num_rendered (host
 * pointer).  GSR_E_OVERFLOW if D > GSR_MAX_RENDERED.
 *
 * A hint, not a requirement: a caller that keeps ONE geom_ws per stream from frame to frame (instead of a fresh one per call) gets
 * a faster forward blend.  The last 80 KB of the workspace hold what every tile of the previous frame cost, and this call turns that
 * into the order in which gsr_forward_render's blend dispatches its tiles (heaviest first; images of up to 4 096 tiles).  Whatever
 * those bytes hold -- a fresh allocation's garbage, another camera's or another image size's costs -- the order is a permutation of
 * the tiles and every output is the same, bit for bit; only the blend's duration changes (-9 % at 800 x 800 with 1 M Gaussians). */
int gsr_forward_count(const GsrScene *scene, const GsrCamera *camera, const GsrGeom *geom,
                      void *geom_ws, size_t geom_ws_bytes, int64_t *num_rendered, void *stream);

/* Stage 2 of render_gaussians: key duplication, sort, tile ranges, blend
 * (reference forward.py:770-879).  binning->D must equal the count from stage 1.  With D == 0 the
 * image and per-pixel buffers are zero-filled, not background (quirk Q10). */
int gsr_forward_render(const GsrScene *scene, const GsrCamera *camera, const GsrGeom *geom,
                       const GsrBinning *binning, const GsrImage *image, void *geom_ws,
                       size_t geom_ws_bytes, void *bin_ws, size_t bin_ws_bytes, void *stream);

/* backward(): wp_render_backward_kernel + the four per-Gaussian kernels
 * (reference backward.py:890-953, :770-888).  dL_dpixels is [H*W*3].  Reads xy / conic_opacity / rgb /
 * radii / cov3D / clamped_state from *geom and point_list / ranges / final_T / n_contrib from
 * *binning / *image. */
int gsr_backward(const GsrScene *scene, const GsrCamera *camera, const GsrGeom *geom,
                 const GsrBinning *binning, const GsrImage *image, const float *dL_dpixels,
                 const GsrGrads *grads, void *ws, size_t ws_bytes, void *stream);

/* The same in two halves, for a caller that wants to start exchanging the view payload (GsrGrads.dL_drgb) while the
 * per-Gaussian half still runs: gsr_backward_blend = backward_render (backward.py:890-953) and, if `payload` is not
 * NULL, the [N*3 + 4] view payload; gsr_backward_geom = backward_preprocess (backward.py:770-888) from the accumulators
 * the first half left in `ws` (same ws, unmodified in between; grads->dL_dshs and grads->dL_drgb may both be NULL when the
 * payload was taken from the first half).  gsr_backward(...) == blend(payload = NULL) + geom. */
int gsr_backward_blend(const GsrScene *scene, const GsrCamera *camera, const GsrGeom *geom, const GsrBinning *binning,
                       const GsrImage *image, const float *dL_dpixels, float *payload, void *ws, size_t ws_bytes, void *stream);
int gsr_backward_geom(const GsrScene *scene, const GsrCamera *camera, const GsrGeom *geom, const GsrGrads *grads, void *ws,
                      size_t ws_bytes, void *stream);

/* ---- "next" rows of SURVEY.md section 8(f): the steps either side of the rasterizer in a training iteration ----
 *
 * f2  L1 loss + pixel gradient (reference loss.py: l1_loss :148-176 -> l1_loss_kernel :12-31;
 *     compute_image_gradients :217-244 -> backprop_l1_pixel_gradients :122-146).  One pass:
 *       *loss_sum   = sum over pixels and channels of |rendered - target|   (device float, overwritten)
 *       pixel_grad  = l1_weight * sign(rendered - target), sign(0) = +1 as wp.sign   (may be NULL)
 *     The caller divides loss_sum by W*H*3 (loss.py:174) and passes l1_weight = (1-lambda_dssim)/(H*W*3). */
int gsr_l1_loss_grad(const float *rendered, const float *target, float *pixel_grad, float *loss_sum, int32_t W, int32_t H,
                     float l1_weight, void *stream);

/* The other two entry points of the reference's loss.py (evaluation helpers; train.py:968-972 has the SSIM term commented
 * out and calls compute_image_gradients with lambda_dssim = 0):
 *   gsr_ssim        loss.py:178-215 (ssim -> gaussian_kernel :33-45 + ssim_kernel :47-119): *ssim_sum = sum over pixels of the
 *                   channel-averaged SSIM in an 11x11 window; the caller divides by W*H.  Window weights as the reference
 *                   applies them: indexed by distance into a Gaussian centred on index 5 (so the rim weighs most).
 *   gsr_depth_loss  loss.py:271-303 (depth_loss -> depth_loss_kernel :247-269): *loss_sum = sum |rendered - target| * mask
 *                   over [H*W] inverse-depth images; the caller divides by W*H. */
int gsr_ssim(const float *rendered, const float *target, float *ssim_sum /* device, overwritten */, int32_t W, int32_t H, void *stream);
int gsr_depth_loss(const float *rendered_depth, const float *target_depth, const float *depth_mask, float *loss_sum /* device, overwritten */,
                   int32_t W, int32_t H, void *stream);

/* f3  Fused Adam step with the reference's clamps (reference optimizer.py:7-139, launched at train.py:750-794):
 *     scale >= 1e-3, quaternion renormalised, opacity clamped to [0,1], +1e-9 in the vec3 divisions
 *     (utils/wp_utils.py:15-20).  grad pointers may be the slices of the backward's gradient arena. */
typedef struct GsrAdamGroup {
    float *param;      /* updated in place */
    const float *grad;
    float *m, *v;      /* first / second moments, updated in place */
    float lr;
} GsrAdamGroup;
typedef struct GsrAdam {
    int64_t N;
    GsrAdamGroup pos;     /* [N*3] */
    GsrAdamGroup scale;   /* [N*3] */
    GsrAdamGroup rot;     /* [N*4] */
    GsrAdamGroup opacity; /* [N] */
    GsrAdamGroup sh;      /* [N*16*3] */
    float beta1, beta2, epsilon;
    int32_t iteration;    /* 0-based; bias correction uses iteration + 1 (optimizer.py:47-48) */
} GsrAdam;
int gsr_adam_update(const GsrAdam *adam, void *stream);

/* ---- view-parallel gradient exchange -------------------------------------------------
 * The SH gradient of one view is an outer product: dL_dsh[k] = basis_k(normalize(mean - campos)) * dL_drgb
 * (backward.py:95-255), so V ranks need not all-reduce 48 floats per Gaussian: they all-gather each view's payload
 * (GsrGrads.dL_drgb: 3 floats per Gaussian + the camera position) and every rank rebuilds
 *     dL_dshs[i][k][c] = scale * sum_v basis_k(dir_v(i)) * drgb_v[i][c]       (views summed in order v = 0..V-1)
 * with the same per-view products the single-view backward forms.  `payloads` is a HOST array of V DEVICE pointers, each
 * to one [N*3 + 4] payload; coefficients above `sh_degree` are written as zero.  1 <= V <= GSR_MAX_VIEWS. */
#define GSR_MAX_VIEWS 16
int gsr_sh_grad_from_views(int64_t N, const float *means, int32_t sh_degree, int32_t V, const float *const *payloads, float scale,
                           float *dL_dshs /* [N*16*3] out */, void *stream);
/* gsr_adam_update with the SH gradient formed on the fly: adam->sh.grad is ignored (may be NULL) and the SH group is updated
 * with scale * sum_v basis_k(dir_v) * drgb_v -- bit for bit what gsr_sh_grad_from_views(adam->N, adam->pos.param, ...) followed
 * by gsr_adam_update would apply (the same products in the same order), without the 192-byte-per-Gaussian gradient ever being
 * written or read.  The directions use adam->pos.param BEFORE this call updates it (the positions the views were rendered
 * with).  V = 1 with the payload of backward(..., sh_gradient="factored") is the single-GPU trainer's step
 * (reference optimizer.py:128-139 + backward.py:95-255). */
int gsr_adam_update_views(const GsrAdam *adam, int32_t sh_degree, int32_t V, const float *const *payloads, float scale, void *stream);

/* ---- row f4: adaptive density control --------------------------------
 * Replaces the Warp kernels the reference trainer launches in densification_and_pruning()
 * (train.py:351-713): compute_grad_norms (train.py:398-406), mark_clone_candidates /
 * mark_split_candidates (optimizer.py:180-239), wp.utils.array_scan(..., inclusive=False)
 * (train.py:432,496,580,640), clone_gaussians (optimizer.py:312-365), split_gaussians
 * (optimizer.py:242-309), mark_split_originals_for_removal + invert_mask (train.py:547-576),
 * prune_gaussians (optimizer.py:367-385), compact_gaussians (optimizer.py:387-416) and
 * reset_opacities (optimizer.py:141-156).  The five parameter arrays are the trainer's `params` dict. */
typedef struct GsrParams {
    int64_t N;
    float *positions; /* [N*3]    */
    float *scales;    /* [N*3]    */
    float *rotations; /* [N*4]    */
    float *opacities; /* [N]      */
    float *shs;       /* [N*16*3] */
} GsrParams;
enum { GSR_MARK_CLONE = 0, GSR_MARK_SPLIT = 1 };
/* mask[i] = (|pos_grad[i]| >= grad_threshold) && (max(scale_i) <= / > percent_dense*scene_extent).
 * Rows i >= n_grad have no gradient (the reference reads past its avg_grads array there after a clone,
 * train.py:478-494: undefined; here their norm is 0). */
int gsr_densify_mark(const GsrParams *p, const float *pos_grad /* [n_grad*3] */, int64_t n_grad, float grad_threshold,
                     float scene_extent, float percent_dense, int mode, int32_t *mask /* [N] out */, void *stream);
/* valid[i] = opacities[i] > opacity_threshold */
int gsr_prune_mark(const GsrParams *p, float opacity_threshold, int32_t *valid /* [N] out */, void *stream);
/* valid[i] = !(i < offset && split_mask[i] == 1), i < n_total; split_mask has `offset` entries */
int gsr_split_removal_mask(int64_t n_total, int64_t offset, const int32_t *split_mask, int32_t *valid /* [n_total] out */, void *stream);
/* prefix = exclusive scan of mask; *count_host = prefix[N-1], the count the reference takes
 * (`int(prefix_sum.numpy()[-1])`: the last row's own flag is NOT counted).  Synchronises the stream.
 * scratch: gsr_mask_scan_workspace_bytes(N) device bytes. */
size_t gsr_mask_scan_workspace_bytes(int64_t N);
int gsr_mask_scan(int64_t N, const int32_t *mask, int32_t *prefix /* [N] out */, int32_t *count_host, void *scratch, size_t scratch_bytes,
                  void *stream);
/* out rows [0,N) = in rows; flagged row i is also written to row N + prefix[i] with
 * positions + randf(3i+k)*noise_scale.  Rows that would land at or past out->N are dropped. */
int gsr_clone_gaussians(const GsrParams *in, const int32_t *mask, const int32_t *prefix, float noise_scale, const GsrParams *out, void *stream);
/* out rows [0,N) = in rows; flagged row i spawns n_split rows at N + prefix[i]*n_split + j with scales*scale_factor and
 * positions + (randf(3*new_idx+k)*2-1)*0.01; rows at or past out->N are dropped (optimizer.py:288). */
int gsr_split_gaussians(const GsrParams *in, const int32_t *mask, const int32_t *prefix, int32_t n_split, float scale_factor,
                        const GsrParams *out, void *stream);
/* out row prefix[i] = in row i where valid[i] != 0; rows at or past out->N are dropped */
int gsr_compact_gaussians(const GsrParams *in, const int32_t *valid, const int32_t *prefix, const GsrParams *out, void *stream);
int gsr_reset_opacities(int64_t N, float max_opacity, float *opacities, void *stream);
/* init_gaussian_params (train.py:37-92, launched at train.py:193-214): positions randf(3i+k)*2.6-1.3, scales init_scale,
 * rotations (1,0,0,0) as stored, opacities 0.1, SH DC -0.007 and zeros above. */
int gsr_init_gaussians(const GsrParams *out, float init_scale, void *stream);

/* ---- profiling aid (process-wide: every thread's sampled calls land in the one record) ----------------
 * With timing enabled every stage boundary of the three entry points records a hipEvent on the
 * caller's stream (about 3 us each, and kernels no longer dispatch back to back across a record: ~6 % of a
 * C3 step if every step is recorded, so bench.py samples one step in five); up to `max_steps` pairs are kept.
 * gsr_stage_times() -- call it after synchronising the stream -- returns the average milliseconds per
 * stage over the steps recorded since enabling, and clears the record. */
enum {
    GSR_ST_PREPROCESS = 0, /* preprocess_kernel */
    GSR_ST_SCAN,           /* id-order scan of tiles_touched (2 kernels); its last wave stores D to a pinned host word */
    GSR_ST_DEPTH_SORT,     /* 4 radix passes over N items, the last one carrying rectangles + counts (overlaps the host wait for D) */
    GSR_ST_DEPTH_SCAN,     /* depth-order offsets: exclusive scan of the carried counts */
    GSR_ST_HOST_GAP,       /* stream idle between gsr_forward_count and gsr_forward_render (host round trip) */
    GSR_ST_EXPAND,         /* (tile,id) item expansion */
    GSR_ST_TILE_SORT,      /* radix passes over D items; the last one writes point_list and ranges */
    GSR_ST_RANGES,         /* (empty since the last tile pass took this work over; slot kept so stage indices stay stable) */
    GSR_ST_BLEND_FWD,      /* blend_forward_kernel */
    GSR_ST_BWD_PREP,       /* accumulator memset + record packing */
    GSR_ST_BLEND_BWD,      /* blend_backward_kernel */
    GSR_ST_GEOM_BWD,       /* geom_backward_kernel */
    GSR_NSTAGES
};
int gsr_stage_timing(int enable, int max_steps);
int gsr_stage_sampling(int every); /* after enabling: record only one forward/backward pair in `every` (default 1); 0 = paused (the
                                      events stay allocated and nothing is recorded until a later call with every >= 1) */
int gsr_stage_times(float *avg_ms /* [GSR_NSTAGES] host */, int *steps /* host */);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H */
