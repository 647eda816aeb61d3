/* gsr_cpu.c -- libgsr_cpu.so: the CPU oracle behind the SAME C ABI as libgsr_hip.so (include/gsr.h; SURVEY.md section 8(b):
 * "identical symbol set").  TEST INFRASTRUCTURE ONLY: the product package never loads it (tests/test_abi_and_host.py greps
 * for that); it exists so that one plain-C client (tests/c_abi/gsr_client.c, built twice) can drive both libraries through
 * include/gsr.h and the two dumps can be diffed -- byte for byte on the integer outputs.
 *
 * Every pointer is a HOST pointer here, `stream` is ignored, and each entry point is the sequence of reference-kernel
 * restatements of gsr_oracle.c (gsro_*: one function per Warp kernel, each citing the reference lines it follows) that
 * oracle/oracle.py runs for the same call:
 *   gsr_forward_count  = gsro_preprocess + gsro_prefix_sum                                  (forward.py:719-767)
 *   gsr_forward_render = gsro_duplicate_with_keys + gsro_sort_pairs + gsro_identify_tile_ranges + gsro_render_rows   (:770-879)
 *   gsr_backward       = gsro_render_backward_rows + cov2d / projection / sh / cov3d backward                      (backward.py:890-953, :770-888)
 * The "next"-row entry points with a C restatement (L1 loss + gradient, SSIM, depth loss, Adam) are wired too; density
 * control and the view-exchange rebuild have numpy oracles only (oracle/densify.py, tests), so those symbols exist and
 * return GSR_E_HIP ("not in the CPU library").  So "same ABI" means the same SYMBOL SET and struct layouts; the same BEHAVIOUR
 * holds for the rasterizer entry points only:
 *   real:  gsr_abi_version, gsr_strerror, gsr_build_flags, gsr_*_workspace_bytes, gsr_forward_count, gsr_forward_render,
 *          gsr_backward, gsr_l1_loss_grad, gsr_ssim, gsr_depth_loss, gsr_adam_update
 *   (records-only mode -- GsrGeom.blend_records as the forward's output with xy / conic_opacity / rgb NULL, ABI 7 -- is not
 *   offered here: the three arrays are required, GSR_E_NULL otherwise; a given blend_records buffer is ignored by the forward)
 *   stubs (return GSR_E_HIP, or a constant for the sizing / timing helpers):  gsr_backward_blend, gsr_backward_geom,
 *          gsr_sh_grad_from_views, gsr_adam_update_views, gsr_densify_mark, gsr_prune_mark, gsr_split_removal_mask, gsr_mask_scan,
 *          gsr_mask_scan_workspace_bytes, gsr_block_order_ints, gsr_backward_accumulators_offset, gsr_clone_gaussians, gsr_split_gaussians, gsr_compact_gaussians,
 *          gsr_reset_opacities, gsr_init_gaussians, gsr_stage_timing, gsr_stage_sampling, gsr_stage_times  The workspace functions return the bytes this library really uses (the
 * int64 sort keys live in the binning workspace; the backward needs none beyond 16 bytes). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "gsr.h"

/* gsr_oracle.c */
void gsro_preprocess(int N, const float *means, const float *scales, float scale_modifier, const float *rots, const float *opac,
                     const float *shs, int degree, int clamped, const float *view, const float *proj, const float *campos, int W, int H,
                     float tan_fovx, float tan_fovy, int32_t *radii, float *xy, float *depths, float *cov3Ds, float *rgb,
                     float *conic_opacity, int32_t *tiles_touched, float *clamped_state);
void gsro_prefix_sum(int N, const int32_t *in, int32_t *out);
void gsro_duplicate_with_keys(int N, const float *xy, const float *depths, const int32_t *point_offsets, int64_t *keys, int32_t *vals,
                              const int32_t *radii, int W, int H);
void gsro_sort_pairs(int64_t count, int64_t *keys, int32_t *vals);
void gsro_identify_tile_ranges(int64_t num_rendered, const int64_t *keys, int32_t *ranges);
void gsro_render_rows(int W, int H, int tile_y0, int tile_y1, const int32_t *ranges, const int32_t *point_list, const float *xy,
                      const float *colors, const float *conic_opacity, const float *depths, const float *bg, float *image,
                      float *depth_image, float *final_Ts, int32_t *n_contrib);
void gsro_render_backward_rows(int W, int H, int tile_y0, int tile_y1, const int32_t *ranges, const int32_t *point_list, const float *bg,
                               const float *xy, const float *conic_opacity, const float *colors, const float *final_Ts,
                               const int32_t *n_contrib, const float *dL_dpixels, float *dL_dmean2D, float *dL_dconic2D,
                               float *dL_dopacity, float *dL_dcolors);
void gsro_cov2d_backward(int N, const float *means, const float *cov3Ds, const int32_t *radii, float h_x, float h_y, float tan_fovx,
                         float tan_fovy, const float *view, const float *dL_dconics, float *dL_dmeans, float *dL_dcov3Ds);
void gsro_projection_backward(int N, const float *means, const int32_t *radii, const float *proj, const float *dL_dmean2D, float *dL_dmeans);
void gsro_sh_backward(int N, int degree, const float *means, const float *shs, const int32_t *radii, const float *campos,
                      const float *clamped, const float *dL_dcolor, float *dL_dmeans, float *dL_dshs);
void gsro_cov3d_backward(int N, const float *scales, const float *rots, const int32_t *radii, float scale_modifier,
                         const float *dL_dcov3Ds, float *dL_dscales, float *dL_drots);
float gsro_l1_loss_sum(int W, int H, const float *rendered, const float *target);
float gsro_ssim_sum(int W, int H, const float *rendered, const float *target);
float gsro_depth_loss_sum(int W, int H, const float *rendered, const float *target, const float *mask);
void gsro_l1_pixel_grad(int W, int H, const float *rendered, const float *target, float l1_weight, float *pixel_grad);
void gsro_adam_update(int N, const float *gpos, const float *gscl, const float *grot, const float *gopa, const float *gsh, float lr_pos,
                      float lr_scl, float lr_rot, float lr_opa, float lr_sh, float beta1, float beta2, float eps, int iteration,
                      float *pos, float *scl, float *rot, float *opa, float *sh, float *m_pos, float *m_scl, float *m_rot, float *m_opa,
                      float *m_sh, float *v_pos, float *v_scl, float *v_rot, float *v_opa, float *v_sh);

#define GSR_BUILD_CPU 2 /* gsr_build_flags() of this library */

static int tiles_of(int W, int H) { return ((W + GSR_TILE - 1) / GSR_TILE) * ((H + GSR_TILE - 1) / GSR_TILE); }
static int check(const GsrScene *sc, const GsrCamera *cam)
{
    if (!sc || !cam) return GSR_E_NULL;
    if (sc->N < 0 || sc->N > 0x7FFFFFFFLL || cam->W <= 0 || cam->H <= 0 || sc->sh_degree < 0 || sc->sh_degree > 3) return GSR_E_DIMS;
    if (sc->N > 0 && (!sc->means || !sc->scales || !sc->rotations || !sc->opacity || !sc->sh)) return GSR_E_NULL;
    return GSR_OK;
}

int gsr_abi_version(void) { return GSR_ABI_VERSION; }
int gsr_build_flags(void) { return GSR_BUILD_CPU; }
const char *gsr_strerror(int code)
{
    switch (code) {
    case GSR_OK: return "ok";
    case GSR_E_NULL: return "required pointer is null";
    case GSR_E_DIMS: return "invalid dimensions or SH degree";
    case GSR_E_OVERFLOW: return "Number of rendered points exceeds the maximum supported (2^30)";
    case GSR_E_WORKSPACE: return "workspace missing or too small";
    case GSR_E_HIP: return "not in the CPU library";
    case GSR_E_CAPACITY: return "GsrBinning.D is not the count gsr_forward_count returned for this geom workspace";
    case GSR_E_ALIGN: return "an array pointer is not 16-byte aligned";
    default: return "unknown error";
    }
}

/* geom workspace: [0] = the count (int64), checked by gsr_forward_render like the HIP library's count note */
size_t gsr_geom_workspace_bytes(int64_t N) { (void)N; return 64; }
size_t gsr_binning_workspace_bytes(int64_t N, int64_t D, int32_t W, int32_t H) { (void)N; (void)W; (void)H; return 64 + (size_t)(D > 0 ? D : 0) * sizeof(int64_t); }
size_t gsr_backward_workspace_bytes(int64_t N, int64_t D, int32_t W, int32_t H) { (void)D; (void)W; (void)H; return 64 + (size_t)(N > 0 ? N : 0) * 6 * sizeof(float); }

int gsr_forward_count(const GsrScene *sc, const GsrCamera *cam, const GsrGeom *g, void *geom_ws, size_t geom_ws_bytes, int64_t *num_rendered,
                      void *stream)
{
    (void)stream;
    int rc = check(sc, cam);
    if (rc) return rc;
    if (!num_rendered) return GSR_E_NULL;
    *num_rendered = 0;
    if (sc->N == 0) return GSR_OK;
    if (!g || !g->radii || !g->tiles_touched || !g->point_offsets || !g->xy || !g->depths || !g->cov3D || !g->rgb || !g->conic_opacity ||
        !g->clamped_state)
        return GSR_E_NULL;
    if (!geom_ws || geom_ws_bytes < gsr_geom_workspace_bytes(sc->N)) return GSR_E_WORKSPACE;
    const int N = (int)sc->N;
    /* culled Gaussians keep zero-initialised outputs (quirk Q11): the reference allocates zeroed arrays (forward.py:679-710) */
    memset(g->radii, 0, sizeof(int32_t) * N); memset(g->tiles_touched, 0, sizeof(int32_t) * N);
    memset(g->xy, 0, sizeof(float) * 2 * N); memset(g->depths, 0, sizeof(float) * N); memset(g->cov3D, 0, sizeof(float) * 6 * N);
    memset(g->rgb, 0, sizeof(float) * 3 * N); memset(g->conic_opacity, 0, sizeof(float) * 4 * N);
    memset(g->clamped_state, 0, sizeof(float) * 3 * N);
    gsro_preprocess(N, sc->means, sc->scales, sc->scale_modifier, sc->rotations, sc->opacity, sc->sh, sc->sh_degree, sc->clamped, cam->view,
                    cam->proj, cam->campos, cam->W, cam->H, cam->tan_fovx, cam->tan_fovy, g->radii, g->xy, g->depths, g->cov3D, g->rgb,
                    g->conic_opacity, g->tiles_touched, g->clamped_state);
    gsro_prefix_sum(N, g->tiles_touched, g->point_offsets);
    const int64_t D = g->point_offsets[N - 1];
    *num_rendered = D;
    *(int64_t *)geom_ws = D;
    return (D < 0 || D > GSR_MAX_RENDERED) ? GSR_E_OVERFLOW : GSR_OK;
}

int gsr_forward_render(const GsrScene *sc, const GsrCamera *cam, const GsrGeom *g, const GsrBinning *b, const GsrImage *img, void *geom_ws,
                       size_t geom_ws_bytes, void *bin_ws, size_t bin_ws_bytes, void *stream)
{
    (void)stream; (void)geom_ws_bytes;
    int rc = check(sc, cam);
    if (rc) return rc;
    if (!b || !img || !img->image || !img->inv_depth || !img->final_T || !img->n_contrib || !b->ranges) return GSR_E_NULL;
    const int64_t D = b->D;
    if (D < 0 || D > GSR_MAX_RENDERED) return GSR_E_OVERFLOW;
    const int W = cam->W, H = cam->H, tiles = tiles_of(W, H);
    const size_t P = (size_t)W * H;
    memset(b->ranges, 0, sizeof(int32_t) * 2 * tiles);
    memset(img->image, 0, sizeof(float) * 3 * P); memset(img->inv_depth, 0, sizeof(float) * P);
    memset(img->final_T, 0, sizeof(float) * P); memset(img->n_contrib, 0, sizeof(int32_t) * P);
    if (D == 0 || sc->N == 0) return GSR_OK; /* zeros, not background (forward.py:830, quirk Q10) */
    if (!g || !g->radii || !g->point_offsets || !g->xy || !g->depths || !g->rgb || !g->conic_opacity || !b->point_list) return GSR_E_NULL;
    if (!geom_ws || *(const int64_t *)geom_ws != D) return GSR_E_CAPACITY;
    if (!bin_ws || bin_ws_bytes < gsr_binning_workspace_bytes(sc->N, D, W, H)) return GSR_E_WORKSPACE;
    int64_t *keys = (int64_t *)((char *)bin_ws + 64);
    gsro_duplicate_with_keys((int)sc->N, g->xy, g->depths, g->point_offsets, keys, b->point_list, g->radii, W, H);
    gsro_sort_pairs(D, keys, b->point_list);
    gsro_identify_tile_ranges(D, keys, b->ranges);
    gsro_render_rows(W, H, 0, (H + GSR_TILE - 1) / GSR_TILE, b->ranges, b->point_list, g->xy, g->rgb, g->conic_opacity, g->depths, cam->bg,
                     img->image, img->inv_depth, img->final_T, img->n_contrib);
    if (b->block_masks) memset(b->block_masks, 0xFF, (size_t)D); /* every block "may be hit": a valid (trivial) mask set */
    return GSR_OK;
}

static int backward_blend(const GsrScene *sc, const GsrCamera *cam, const GsrGeom *g, const GsrBinning *b, const GsrImage *img,
                          const float *dpix, float *dcolor, float *dmean2D, float *dconic, float *dopacity)
{
    const int N = (int)sc->N, W = cam->W, H = cam->H;
    if (!g || !g->xy || !g->conic_opacity || !g->rgb || !b || !img || !dpix) return GSR_E_NULL;
    memset(dcolor, 0, sizeof(float) * 3 * N); memset(dmean2D, 0, sizeof(float) * 3 * N);
    memset(dconic, 0, sizeof(float) * 4 * N); memset(dopacity, 0, sizeof(float) * N);
    if (b->D > 0) {
        if (!b->point_list || !b->ranges || !img->final_T || !img->n_contrib) return GSR_E_NULL;
        gsro_render_backward_rows(W, H, 0, (H + GSR_TILE - 1) / GSR_TILE, b->ranges, b->point_list, cam->bg, g->xy, g->conic_opacity, g->rgb,
                                  img->final_T, img->n_contrib, dpix, dmean2D, dconic, dopacity, dcolor);
    }
    return GSR_OK;
}

static int backward_geom(const GsrScene *sc, const GsrCamera *cam, const GsrGeom *g, const GsrGrads *gr, float *dcov3D)
{
    const int N = (int)sc->N;
    if (!g || !g->radii || !g->cov3D || !g->clamped_state) return GSR_E_NULL;
    memset(gr->dL_dmean3D, 0, sizeof(float) * 3 * N);
    memset(dcov3D, 0, sizeof(float) * 6 * N);
    gsro_cov2d_backward(N, sc->means, g->cov3D, g->radii, cam->focal_x, cam->focal_y, cam->tan_fovx, cam->tan_fovy, cam->view, gr->dL_dconic,
                        gr->dL_dmean3D, dcov3D);
    gsro_projection_backward(N, sc->means, g->radii, cam->proj, gr->dL_dmean2D, gr->dL_dmean3D);
    if (gr->dL_dshs) {
        memset(gr->dL_dshs, 0, sizeof(float) * 48 * N);
        gsro_sh_backward(N, sc->sh_degree, sc->means, sc->sh, g->radii, cam->campos, g->clamped_state, gr->dL_dcolor, gr->dL_dmean3D, gr->dL_dshs);
    } else {
        return GSR_E_HIP; /* the payload-only mode (dL_drgb without dL_dshs) is a product optimisation, not a reference function */
    }
    memset(gr->dL_dscale, 0, sizeof(float) * 3 * N); memset(gr->dL_drot, 0, sizeof(float) * 4 * N);
    gsro_cov3d_backward(N, sc->scales, sc->rotations, g->radii, 1.0f /* quirk Q16 */, dcov3D, gr->dL_dscale, gr->dL_drot);
    return GSR_OK;
}

int gsr_backward(const GsrScene *sc, const GsrCamera *cam, const GsrGeom *g, const GsrBinning *b, const GsrImage *img, const float *dpix,
                 const GsrGrads *gr, void *ws, size_t ws_bytes, void *stream)
{
    (void)stream;
    int rc = check(sc, cam);
    if (rc) return rc;
    if (sc->N == 0) return GSR_OK;
    if (!gr || !gr->dL_dmean3D || !gr->dL_dscale || !gr->dL_drot || !gr->dL_dopacity || !gr->dL_dcolor || !gr->dL_dmean2D || !gr->dL_dconic)
        return GSR_E_NULL;
    if (!ws || ws_bytes < gsr_backward_workspace_bytes(sc->N, 0, cam->W, cam->H)) return GSR_E_WORKSPACE;
    rc = backward_blend(sc, cam, g, b, img, dpix, gr->dL_dcolor, gr->dL_dmean2D, gr->dL_dconic, gr->dL_dopacity);
    if (rc) return rc;
    return backward_geom(sc, cam, g, gr, (float *)((char *)ws + 64));
}

/* the split form needs the blend accumulators to persist in `ws` between the halves: not kept here */
int gsr_backward_blend(const GsrScene *a, const GsrCamera *b, const GsrGeom *c, const GsrBinning *d, const GsrImage *e, const float *f, float *g,
                       void *h, size_t i, void *j) { (void)a; (void)b; (void)c; (void)d; (void)e; (void)f; (void)g; (void)h; (void)i; (void)j; return GSR_E_HIP; }
int gsr_backward_geom(const GsrScene *a, const GsrCamera *b, const GsrGeom *c, const GsrGrads *d, void *e, size_t f, void *g)
{ (void)a; (void)b; (void)c; (void)d; (void)e; (void)f; (void)g; return GSR_E_HIP; }

/* ---- rows f2 / f3 ---- */
int gsr_l1_loss_grad(const float *rendered, const float *target, float *pixel_grad, float *loss_sum, int32_t W, int32_t H, float l1_weight, void *stream)
{
    (void)stream;
    if (!rendered || !target || !loss_sum) return GSR_E_NULL;
    if (W <= 0 || H <= 0) return GSR_E_DIMS;
    *loss_sum = gsro_l1_loss_sum(W, H, rendered, target);
    if (pixel_grad) gsro_l1_pixel_grad(W, H, rendered, target, l1_weight, pixel_grad);
    return GSR_OK;
}
int gsr_ssim(const float *rendered, const float *target, float *ssim_sum, int32_t W, int32_t H, void *stream)
{
    (void)stream;
    if (!rendered || !target || !ssim_sum) return GSR_E_NULL;
    if (W <= 0 || H <= 0) return GSR_E_DIMS;
    *ssim_sum = gsro_ssim_sum(W, H, rendered, target);
    return GSR_OK;
}
int gsr_depth_loss(const float *rd, const float *td, const float *mask, float *loss_sum, int32_t W, int32_t H, void *stream)
{
    (void)stream;
    if (!rd || !td || !mask || !loss_sum) return GSR_E_NULL;
    if (W <= 0 || H <= 0) return GSR_E_DIMS;
    *loss_sum = gsro_depth_loss_sum(W, H, rd, td, mask);
    return GSR_OK;
}
int gsr_adam_update(const GsrAdam *a, void *stream)
{
    (void)stream;
    if (!a) return GSR_E_NULL;
    if (a->N < 0) return GSR_E_DIMS;
    if (a->N == 0) return GSR_OK;
    gsro_adam_update((int)a->N, a->pos.grad, a->scale.grad, a->rot.grad, a->opacity.grad, a->sh.grad, a->pos.lr, a->scale.lr, a->rot.lr,
                     a->opacity.lr, a->sh.lr, a->beta1, a->beta2, a->epsilon, a->iteration, a->pos.param, a->scale.param, a->rot.param,
                     a->opacity.param, a->sh.param, a->pos.m, a->scale.m, a->rot.m, a->opacity.m, a->sh.m, a->pos.v, a->scale.v, a->rot.v,
                     a->opacity.v, a->sh.v);
    return GSR_OK;
}

/* ---- symbols whose oracle is numpy (oracle/densify.py) or that are product-only: present, not implemented ---- */
int gsr_sh_grad_from_views(int64_t N, const float *means, int32_t deg, int32_t V, const float *const *p, float s, float *o, void *st)
{ (void)N; (void)means; (void)deg; (void)V; (void)p; (void)s; (void)o; (void)st; return GSR_E_HIP; }
int gsr_adam_update_views(const GsrAdam *a, int32_t deg, int32_t V, const float *const *p, float s, void *st)
{ (void)a; (void)deg; (void)V; (void)p; (void)s; (void)st; return GSR_E_HIP; }
int gsr_densify_mark(const GsrParams *p, const float *g, int64_t n, float a, float b, float c, int m, int32_t *mask, void *s)
{ (void)p; (void)g; (void)n; (void)a; (void)b; (void)c; (void)m; (void)mask; (void)s; return GSR_E_HIP; }
int gsr_prune_mark(const GsrParams *p, float t, int32_t *v, void *s) { (void)p; (void)t; (void)v; (void)s; return GSR_E_HIP; }
int gsr_split_removal_mask(int64_t n, int64_t o, const int32_t *m, int32_t *v, void *s) { (void)n; (void)o; (void)m; (void)v; (void)s; return GSR_E_HIP; }
size_t gsr_mask_scan_workspace_bytes(int64_t N) { (void)N; return 256; }
size_t gsr_backward_accumulators_offset(int64_t N) { (void)N; return 0; } /* the CPU backward keeps no accumulator records: its GsrGrads arrays are all required */
size_t gsr_block_order_ints(int32_t W, int32_t H) { (void)W; (void)H; return 4; } /* never written or read on the CPU */
int gsr_mask_scan(int64_t N, const int32_t *m, int32_t *p, int32_t *c, void *s, size_t b, void *st) { (void)N; (void)m; (void)p; (void)c; (void)s; (void)b; (void)st; return GSR_E_HIP; }
int gsr_clone_gaussians(const GsrParams *i, const int32_t *m, const int32_t *p, float n, const GsrParams *o, void *s) { (void)i; (void)m; (void)p; (void)n; (void)o; (void)s; return GSR_E_HIP; }
int gsr_split_gaussians(const GsrParams *i, const int32_t *m, const int32_t *p, int32_t n, float f, const GsrParams *o, void *s) { (void)i; (void)m; (void)p; (void)n; (void)f; (void)o; (void)s; return GSR_E_HIP; }
int gsr_compact_gaussians(const GsrParams *i, const int32_t *v, const int32_t *p, const GsrParams *o, void *s) { (void)i; (void)v; (void)p; (void)o; (void)s; return GSR_E_HIP; }
int gsr_reset_opacities(int64_t N, float m, float *o, void *s) { (void)N; (void)m; (void)o; (void)s; return GSR_E_HIP; }
int gsr_init_gaussians(const GsrParams *o, float i, void *s) { (void)o; (void)i; (void)s; return GSR_E_HIP; }
int gsr_stage_timing(int enable, int max_steps) { (void)enable; (void)max_steps; return GSR_OK; }
int gsr_stage_sampling(int every) { return every < 0 ? GSR_E_DIMS : GSR_OK; }
int gsr_stage_times(float *avg_ms, int *steps)
{
    if (!avg_ms || !steps) return GSR_E_NULL;
    for (int k = 0; k < GSR_NSTAGES; ++k) avg_ms[k] = 0.0f;
    *steps = 0;
    return GSR_OK;
}
