/*
 * gsr_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, single thread, IEEE float32,
 * no FMA contraction) of the 3DGS rasterizer hot path of zhujinchong/3DGS-native, one C function per
 * reference Warp kernel.  It is the checker for the HIP library in 3dgs-native_amd/csrc; nothing in
 * the product path may link, import or call it (only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do).
 *
 * PARITY PINNING.  The reference's numeric engine is the third-party module warp-lang==1.7.0
 * (reference readme.md:31), which is not installed here and cannot be fetched; no stand-in for it is
 * written.  The only reference-produced numeric output in the reference tree is
 * assets/example_render.png (render.py:84-137).  The FORWARD half of this oracle is pinned against
 * that image (tests/test_oracle_golden.py).  The reference holds no backward output of any kind, so
 * the BACKWARD half is "parity unpinned": it is a literal transcription, cross-checked only by
 * finite differences on the sub-expressions that are true derivatives.
 *
 * Warp semantics assumed (warp/native/{vec,mat,quat}.h as published
 * for 1.7.0):
 *   A1  mat33(s0..s8)/mat44(flat16) fill row-major; m[i][j] == m[i,j]; m[i] is row i.
 *   A2  v*M is row-vector times matrix: r = M.row(0)*v[0]; r += M.row(i)*v[i] (i ascending).
 *       M*N: t[i][j] = 0; t[i][j] += M[i][k]*N[k][j] (k ascending).
 *   A3  literals are float32, int is int32, float->int truncates toward zero.
 *   A4  quat_to_matrix(q): columns are quat_rotate(q, e_i);
 *       quat_rotate(q,v) = v*(2w^2-1) + cross(q.xyz,v)*w*2 + q.xyz*dot(q.xyz,v)*2.
 *   A5  normalize(v) = v/length(v) if length>0 else 0; length = sqrt(dot); dot sums in index order.
 *   A6  radix_sort_pairs: stable ascending sort of the first `count` (int64 key, int32 value) pairs.
 *   A7  a CPU launch runs the grid serially with the last dimension fastest.
 *   A8  atomic_add on vec3/vec4 is component-wise.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE_M 16 /* config.py:21 */
#define TILE_N 16 /* config.py:22 */

typedef struct { float m[3][3]; } mat33;

static mat33 m33_mul(mat33 a, mat33 b) /* A2 */
{
    mat33 t;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = 0.0f;
            for (int k = 0; k < 3; ++k) s += a.m[i][k] * b.m[k][j];
            t.m[i][j] = s;
        }
    return t;
}
static mat33 m33_T(mat33 a)
{
    mat33 t;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t.m[i][j] = a.m[j][i];
    return t;
}
/* vec4(p,1) * M  (A2); M is 16 floats row-major. */
static void v4_mul_m44(const float p[4], const float *M, float out[4])
{
    for (int j = 0; j < 4; ++j) {
        float r = M[0 * 4 + j] * p[0];
        r += M[1 * 4 + j] * p[1];
        r += M[2 * 4 + j] * p[2];
        r += M[3 * 4 + j] * p[3];
        out[j] = r;
    }
}
static float dot3(const float a[3], const float b[3])
{
    float r = a[0] * b[0];
    r += a[1] * b[1];
    r += a[2] * b[2];
    return r;
}
static float fminf_(float a, float b) { return a < b ? a : b; }
static float fmaxf_(float a, float b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* forward.py:60-61 */
static float ndc2pix(float x, float size) { return ((x + 1.0f) * size - 1.0f) * 0.5f; }

/* forward.py:64-76.  tile_grid holds float(ceil(W/16)), float(ceil(H/16)) (forward.py:698-700). */
static void get_rect(const float p[2], float max_radius, int grid_x, int grid_y, int r[4])
{
    r[0] = imin(grid_x, imax(0, (int)((p[0] - max_radius) / (float)TILE_M)));
    r[1] = imin(grid_y, imax(0, (int)((p[1] - max_radius) / (float)TILE_N)));
    r[2] = imin(grid_x, imax(0, (int)((p[0] + max_radius + (float)TILE_M - 1.0f) / (float)TILE_M)));
    r[3] = imin(grid_y, imax(0, (int)((p[1] + max_radius + (float)TILE_N - 1.0f) / (float)TILE_N)));
}

/* forward.py:147-186 (compute_cov3d), A4 for quat_to_matrix. */
static void compute_cov3d(const float scale[3], float scale_mod, const float rot[4], float cov[6])
{
    mat33 S = {{{scale_mod * scale[0], 0.0f, 0.0f}, {0.0f, scale_mod * scale[1], 0.0f}, {0.0f, 0.0f, scale_mod * scale[2]}}};
    const float qx = rot[0], qy = rot[1], qz = rot[2], qw = rot[3];
    mat33 R;
    for (int c = 0; c < 3; ++c) {
        float v[3] = {c == 0 ? 1.0f : 0.0f, c == 1 ? 1.0f : 0.0f, c == 2 ? 1.0f : 0.0f};
        float cs = 2.0f * qw * qw - 1.0f;
        float cr[3] = {qy * v[2] - qz * v[1], qz * v[0] - qx * v[2], qx * v[1] - qy * v[0]};
        float q[3] = {qx, qy, qz};
        float d = dot3(q, v);
        for (int i = 0; i < 3; ++i) R.m[i][c] = v[i] * cs + cr[i] * qw * 2.0f + q[i] * d * 2.0f;
    }
    mat33 M = m33_mul(R, S);
    mat33 sigma = m33_mul(M, m33_T(M));
    cov[0] = sigma.m[0][0]; cov[1] = sigma.m[0][1]; cov[2] = sigma.m[0][2];
    cov[3] = sigma.m[1][1]; cov[4] = sigma.m[1][2]; cov[5] = sigma.m[2][2];
}

/* forward.py:80-144 (compute_cov2d).  NOTE quirk Q1: T = J*W with W = view[0:3,0:3] as stored. */
static void compute_cov2d(const float p[3], const float cov3d[6], const float *view, float tan_fovx,
                          float tan_fovy, float width, float height, float out[3])
{
    float ph[4] = {p[0], p[1], p[2], 1.0f}, t[4];
    v4_mul_m44(ph, view, t);
    float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    float txtz = t[0] / t[2], tytz = t[1] / t[2];
    t[0] = fminf_(limx, fmaxf_(-limx, txtz)) * t[2];
    t[1] = fminf_(limy, fmaxf_(-limy, tytz)) * t[2];
    float focal_x = width / (2.0f * tan_fovx);
    float focal_y = height / (2.0f * tan_fovy);
    mat33 J = {{{focal_x / t[2], 0.0f, -(focal_x * t[0]) / (t[2] * t[2])},
                {0.0f, focal_y / t[2], -(focal_y * t[1]) / (t[2] * t[2])},
                {0.0f, 0.0f, 0.0f}}};
    mat33 W = {{{view[0], view[1], view[2]}, {view[4], view[5], view[6]}, {view[8], view[9], view[10]}}};
    mat33 T = m33_mul(J, W);
    mat33 Vrk = {{{cov3d[0], cov3d[1], cov3d[2]}, {cov3d[1], cov3d[3], cov3d[4]}, {cov3d[2], cov3d[4], cov3d[5]}}};
    mat33 cov = m33_mul(m33_mul(T, m33_T(Vrk)), m33_T(T));
    out[0] = cov.m[0][0]; out[1] = cov.m[0][1]; out[2] = cov.m[1][1];
}

static const float SH_C0 = 0.28209479177387814f; /* forward.py:44 */
static const float SH_C1 = 0.4886025119029199f;  /* forward.py:45 */

/* forward.py:190-382 (wp_preprocess).  All outputs must be zero-initialised by the caller
 * (forward.py:703-710); culled Gaussians leave them untouched (quirk Q11). */
void gsro_preprocess(int N, const float *means, const float *scales, float scale_modifier,
                     const float *rots, const float *opac, const float *shs, int degree, int clamped,
                     const float *view, const float *proj, const float *campos, int W, int H,
                     float tan_fovx, float tan_fovy, int32_t *radii, float *xy, float *depths,
                     float *cov3Ds, float *rgb, float *conic_opacity, int32_t *tiles_touched,
                     float *clamped_state)
{
    const int grid_x = (W + TILE_M - 1) / TILE_M, grid_y = (H + TILE_N - 1) / TILE_N;
    for (int i = 0; i < N; ++i) {
        const float *p = means + 3 * i;
        float ph[4] = {p[0], p[1], p[2], 1.0f}, p_view[4], p_hom[4];
        v4_mul_m44(ph, view, p_view);
        if (p_view[2] < 0.2f) continue; /* :250 */
        v4_mul_m44(ph, proj, p_hom);
        float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        float p_proj[3] = {p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w};
        float cov3d[6];
        compute_cov3d(scales + 3 * i, scale_modifier, rots + 4 * i, cov3d);
        memcpy(cov3Ds + 6 * i, cov3d, sizeof(cov3d)); /* :260 */
        float cov2d[3];
        compute_cov2d(p, cov3d, view, tan_fovx, tan_fovy, (float)W, (float)H, cov2d);
        const float h_var = 0.3f;
        float cb0 = cov2d[0] + h_var, cb1 = cov2d[1], cb2 = cov2d[2] + h_var;
        float det = cb0 * cb2 - cb1 * cb1;
        if (det == 0.0f) continue; /* :278 */
        float det_inv = 1.0f / det;
        float conic[3] = {cb2 * det_inv, -cb1 * det_inv, cb0 * det_inv};
        float mid = 0.5f * (cb0 + cb2);
        float lambda1 = mid + sqrtf(fmaxf_(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf_(0.1f, mid * mid - det));
        float my_radius = ceilf(3.0f * sqrtf(fmaxf_(lambda1, lambda2)));
        float pim[2] = {ndc2pix(p_proj[0], (float)W), ndc2pix(p_proj[1], (float)H)};
        int rect[4];
        get_rect(pim, my_radius, grid_x, grid_y, rect);
        if ((rect[2] - rect[0]) * (rect[3] - rect[1]) == 0) continue; /* :301 */

        float dir_orig[3] = {p[0] - campos[0], p[1] - campos[1], p[2] - campos[2]};
        float len = sqrtf(dot3(dir_orig, dir_orig));
        float x = 0.0f, y = 0.0f, z = 0.0f;
        if (len > 0.0f) { x = dir_orig[0] / len; y = dir_orig[1] / len; z = dir_orig[2] / len; }
        const float *sh = shs + (size_t)i * 16 * 3; /* stride always 16 (quirk Q6) */
        float res[3];
        for (int c = 0; c < 3; ++c) {
#define SH(k) sh[(k) * 3 + c]
            float r = SH_C0 * SH(0);
            if (degree > 0) {
                r = r - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
                if (degree > 1) {
                    float xx = x * x, yy = y * y, zz = z * z, xy_ = x * y, yz = y * z, xz = x * z;
                    r = r + 1.0925484305920792f * xy_ * SH(4);
                    r = r + (-1.0925484305920792f) * yz * SH(5);
                    r = r + 0.31539156525252005f * (2.0f * zz - xx - yy) * SH(6);
                    r = r + (-1.0925484305920792f) * xz * SH(7);
                    r = r + 0.5462742152960396f * (xx - yy) * SH(8);
                    if (degree > 2) {
                        r = r + (-0.5900435899266435f) * y * (3.0f * xx - yy) * SH(9);
                        r = r + 2.890611442640554f * xy_ * z * SH(10);
                        r = r + (-0.4570457994644658f) * y * (4.0f * zz - xx - yy) * SH(11);
                        r = r + 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12);
                        r = r + (-0.4570457994644658f) * x * (4.0f * zz - xx - yy) * SH(13);
                        r = r + 1.445305721320277f * z * (xx - yy) * SH(14);
                        r = r + (-0.5900435899266435f) * x * (xx - 3.0f * yy) * SH(15);
                    }
                }
            }
#undef SH
            res[c] = r + 0.5f;
        }
        for (int c = 0; c < 3; ++c) {
            clamped_state[3 * i + c] = res[c] < 0.0f ? 1.0f : 0.0f;
            if (clamped) res[c] = fmaxf_(res[c], 0.0f);
            rgb[3 * i + c] = res[c];
        }
        depths[i] = p_view[2];
        radii[i] = (int)my_radius;
        xy[2 * i] = pim[0]; xy[2 * i + 1] = pim[1];
        conic_opacity[4 * i] = conic[0]; conic_opacity[4 * i + 1] = conic[1];
        conic_opacity[4 * i + 2] = conic[2]; conic_opacity[4 * i + 3] = opac[i];
        tiles_touched[i] = (rect[3] - rect[1]) * (rect[2] - rect[0]);
    }
}

/* utils/wp_utils.py:47-60 (wp_prefix_sum): serial inclusive scan. */
void gsro_prefix_sum(int N, const int32_t *in, int32_t *out)
{
    if (N <= 0) return;
    out[0] = in[0];
    for (int i = 1; i < N; ++i) out[i] = out[i - 1] + in[i];
}

/* forward.py:518-558 (wp_duplicate_with_keys). */
void gsro_duplicate_with_keys(int N, const float *xy, const float *depths, const int32_t *point_offsets,
                              int64_t *keys, int32_t *vals, const int32_t *radii, int W, int H)
{
    const int grid_x = (W + TILE_M - 1) / TILE_M, grid_y = (H + TILE_N - 1) / TILE_N;
    for (int tid = 0; tid < N; ++tid) {
        int r = radii[tid];
        if (r <= 0) continue;
        int64_t offset = tid > 0 ? point_offsets[tid - 1] : 0;
        int rect[4];
        get_rect(xy + 2 * tid, (float)r, grid_x, grid_y, rect);
        uint32_t bits;
        memcpy(&bits, depths + tid, 4); /* forward.py:51-57 */
        for (int y = rect[1]; y < rect[3]; ++y)
            for (int x = rect[0]; x < rect[2]; ++x) {
                int64_t tile_id = (int64_t)(y * grid_x + x);
                keys[offset] = (tile_id << 32) | (int64_t)bits;
                vals[offset] = tid;
                ++offset;
            }
    }
}

/* forward.py:799-803 -> wp.utils.radix_sort_pairs (A6): stable ascending by int64 key.
 * Restated as a stable bottom-up merge sort (same permutation as any stable sort). */
void gsro_sort_pairs(int64_t count, int64_t *keys, int32_t *vals)
{
    if (count <= 1) return;
    int64_t *k2 = (int64_t *)malloc(sizeof(int64_t) * count);
    int32_t *v2 = (int32_t *)malloc(sizeof(int32_t) * count);
    int64_t *ka = keys, *kb = k2;
    int32_t *va = vals, *vb = v2;
    for (int64_t w = 1; w < count; w *= 2) {
        for (int64_t lo = 0; lo < count; lo += 2 * w) {
            int64_t mid = lo + w < count ? lo + w : count, hi = lo + 2 * w < count ? lo + 2 * w : count;
            int64_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) {
                if (ka[j] < ka[i]) { kb[o] = ka[j]; vb[o++] = va[j++]; }
                else { kb[o] = ka[i]; vb[o++] = va[i++]; }
            }
            while (i < mid) { kb[o] = ka[i]; vb[o++] = va[i++]; }
            while (j < hi) { kb[o] = ka[j]; vb[o++] = va[j++]; }
        }
        int64_t *tk = ka; ka = kb; kb = tk;
        int32_t *tv = va; va = vb; vb = tv;
    }
    if (ka != keys) { memcpy(keys, ka, sizeof(int64_t) * count); memcpy(vals, va, sizeof(int32_t) * count); }
    free(k2); free(v2);
}

/* forward.py:561-586 (wp_identify_tile_ranges).  ranges is (tiles,2) int32, zero-initialised. */
void gsro_identify_tile_ranges(int64_t num_rendered, const int64_t *keys, int32_t *ranges)
{
    for (int64_t idx = 0; idx < num_rendered; ++idx) {
        int curr = (int)(keys[idx] >> 32);
        if (idx == 0) ranges[2 * curr] = 0;
        else {
            int prev = (int)(keys[idx - 1] >> 32);
            if (curr != prev) { ranges[2 * prev + 1] = (int32_t)idx; ranges[2 * curr] = (int32_t)idx; }
        }
        if (idx == num_rendered - 1) ranges[2 * curr + 1] = (int32_t)num_rendered;
    }
}

/* forward.py:385-515 (wp_render_gaussians), rows [y0,y1) of tiles only (y0=0,y1=grid_y for all;
 * the sub-range form lets bench.py time a bounded sample).  image (H,W,3), others (H,W). */
void gsro_render_rows(int W, int H, int tile_y0, int tile_y1, const int32_t *ranges,
                      const int32_t *point_list, const float *xy, const float *colors,
                      const float *conic_opacity, const float *depths, const float *bg, float *image,
                      float *depth_image, float *final_Ts, int32_t *n_contrib)
{
    const int grid_x = (W + TILE_M - 1) / TILE_M;
    for (int tile_x = 0; tile_x < grid_x; ++tile_x)
        for (int tile_y = tile_y0; tile_y < tile_y1; ++tile_y)
            for (int tid_x = 0; tid_x < TILE_M; ++tid_x)
                for (int tid_y = 0; tid_y < TILE_N; ++tid_y) {
                    int pix_x = tile_x * TILE_M + tid_x, pix_y = tile_y * TILE_N + tid_y;
                    if (!(pix_x < W && pix_y < H)) continue;
                    float pixf_x = (float)pix_x, pixf_y = (float)pix_y;
                    int tile_id = tile_y * grid_x + tile_x;
                    int range_start = ranges[2 * tile_id], range_end = ranges[2 * tile_id + 1];
                    float T = 1.0f, r = 0.0f, g = 0.0f, b = 0.0f, expected_inv_depth = 0.0f;
                    int contributor_count = 0, last_contributor = 0;
                    for (int i = range_start; i < range_end; ++i) {
                        int gid = point_list[i];
                        const float *pxy = xy + 2 * gid, *con_o = conic_opacity + 4 * gid, *color = colors + 3 * gid;
                        float d_x = pxy[0] - pixf_x, d_y = pxy[1] - pixf_y;
                        contributor_count += 1;
                        float power = -0.5f * (con_o[0] * d_x * d_x + con_o[2] * d_y * d_y) - con_o[1] * d_x * d_y;
                        if (power > 0.0f) continue;
                        float alpha = fminf_(0.99f, con_o[3] * expf(power));
                        if (alpha < (1.0f / 255.0f)) continue;
                        float test_T = T * (1.0f - alpha);
                        if (test_T < 0.0001f) break;
                        r += color[0] * alpha * T;
                        g += color[1] * alpha * T;
                        b += color[2] * alpha * T;
                        expected_inv_depth += (1.0f / depths[gid]) * alpha * T;
                        T = test_T;
                        last_contributor = contributor_count;
                    }
                    size_t px = (size_t)pix_y * W + pix_x;
                    final_Ts[px] = T;
                    n_contrib[px] = last_contributor;
                    image[3 * px] = r + T * bg[0];
                    image[3 * px + 1] = g + T * bg[1];
                    image[3 * px + 2] = b + T * bg[2];
                    depth_image[px] = expected_inv_depth;
                }
}

/* backward.py:559-706 (wp_render_backward_kernel), launch order per A7 (backward.py:932-934):
 * tile_x, tile_y, tid_x, tid_y with the last fastest, so float accumulation order matches Warp-CPU.
 * dL_dmean2D (N,3), dL_dconic (N,4), dL_dopacity (N), dL_dcolors (N,3): accumulated into. */
void gsro_render_backward_rows(int W, int H, int tile_y0, int tile_y1, const int32_t *ranges,
                               const int32_t *point_list, const float *bg, const float *xy,
                               const float *conic_opacity, const float *colors, const float *final_Ts,
                               const int32_t *n_contrib, const float *dL_dpixels, float *dL_dmean2D,
                               float *dL_dconic2D, float *dL_dopacity, float *dL_dcolors)
{
    const int grid_x = (W + TILE_M - 1) / TILE_M;
    for (int tile_x = 0; tile_x < grid_x; ++tile_x)
        for (int tile_y = tile_y0; tile_y < tile_y1; ++tile_y)
            for (int tid_x = 0; tid_x < TILE_M; ++tid_x)
                for (int tid_y = 0; tid_y < TILE_N; ++tid_y) {
                    int pix_x = tile_x * TILE_M + tid_x, pix_y = tile_y * TILE_N + tid_y;
                    if (!(pix_x < W && pix_y < H)) continue;
                    float pixf_x = (float)pix_x, pixf_y = (float)pix_y;
                    int tile_id = tile_y * grid_x + tile_x;
                    int range_start = ranges[2 * tile_id], range_end = ranges[2 * tile_id + 1];
                    size_t px = (size_t)pix_y * W + pix_x;
                    float T_final = final_Ts[px];
                    int last_contributor = n_contrib[px];
                    int last_kept = imin(range_end, range_start + last_contributor);
                    float T = T_final;
                    float accum_rec[3] = {0.0f, 0.0f, 0.0f};
                    float last_alpha = 0.0f, last_color[3] = {0.0f, 0.0f, 0.0f};
                    const float *dL_dpixel = dL_dpixels + 3 * px;
                    float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
                    for (int i = last_kept - 1; i > range_start - 1; --i) {
                        int gid = point_list[i];
                        const float *pxy = xy + 2 * gid, *con_o = conic_opacity + 4 * gid, *color = colors + 3 * gid;
                        float d_x = pxy[0] - pixf_x, d_y = pxy[1] - pixf_y;
                        float power = -0.5f * (con_o[0] * d_x * d_x + con_o[2] * d_y * d_y) - con_o[1] * d_x * d_y;
                        if (power > 0.0f) continue;
                        float G = expf(power);
                        float alpha = fminf_(0.99f, con_o[3] * G);
                        if (alpha < (1.0f / 255.0f)) continue;
                        T = T / (1.0f - alpha);
                        float dchannel_dcolor = alpha * T;
                        float tmp[3];
                        for (int c = 0; c < 3; ++c) {
                            accum_rec[c] = last_alpha * last_color[c] + (1.0f - last_alpha) * accum_rec[c];
                            last_color[c] = color[c];
                            tmp[c] = color[c] - accum_rec[c];
                        }
                        float dL_dalpha = dot3(tmp, dL_dpixel);
                        for (int c = 0; c < 3; ++c) dL_dcolors[3 * gid + c] += dchannel_dcolor * dL_dpixel[c];
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        float bg_dot_dpixel = dot3(bg, dL_dpixel);
                        dL_dalpha += (-T_final / (1.0f - alpha)) * bg_dot_dpixel;
                        float dL_dG = con_o[3] * dL_dalpha;
                        float gdx = G * d_x, gdy = G * d_y;
                        float dG_ddelx = -gdx * con_o[0] - gdy * con_o[1];
                        float dG_ddely = -gdy * con_o[2] - gdx * con_o[1];
                        dL_dmean2D[3 * gid] += dL_dG * dG_ddelx * ddelx_dx;
                        dL_dmean2D[3 * gid + 1] += dL_dG * dG_ddely * ddely_dy;
                        dL_dmean2D[3 * gid + 2] += 0.0f;
                        dL_dconic2D[4 * gid] += -0.5f * gdx * d_x * dL_dG;
                        dL_dconic2D[4 * gid + 1] += -0.5f * gdx * d_y * dL_dG;
                        dL_dconic2D[4 * gid + 2] += 0.0f;
                        dL_dconic2D[4 * gid + 3] += -0.5f * gdy * d_y * dL_dG;
                        dL_dopacity[gid] += G * dL_dalpha;
                    }
                }
}

/* SECOND CHECKER for ill-conditioned (needle-like) splats -- not a restatement of any reference function.
 * Every per-(pixel, entry) term is formed exactly as gsro_render_backward_rows forms it (same float32 expressions, same
 * replay of alpha and T), but the per-Gaussian sums over pixels are kept in float64, so the result carries no
 * accumulation-order rounding.  tests/test_gpu_fuzz.py uses it to tell two things apart that the plain comparison mixes:
 * how far the reference's own serial float32 sum is from the exactly accumulated one, and how far the HIP kernel is. */
void gsro_render_backward_rows_acc64(int W, int H, int tile_y0, int tile_y1, const int32_t *ranges,
                                     const int32_t *point_list, const float *bg, const float *xy,
                                     const float *conic_opacity, const float *colors, const float *final_Ts,
                                     const int32_t *n_contrib, const float *dL_dpixels, double *dL_dmean2D,
                                     double *dL_dconic2D, double *dL_dopacity, double *dL_dcolors)
{
    const int grid_x = (W + TILE_M - 1) / TILE_M;
    for (int tile_x = 0; tile_x < grid_x; ++tile_x)
        for (int tile_y = tile_y0; tile_y < tile_y1; ++tile_y)
            for (int tid_x = 0; tid_x < TILE_M; ++tid_x)
                for (int tid_y = 0; tid_y < TILE_N; ++tid_y) {
                    int pix_x = tile_x * TILE_M + tid_x, pix_y = tile_y * TILE_N + tid_y;
                    if (!(pix_x < W && pix_y < H)) continue;
                    float pixf_x = (float)pix_x, pixf_y = (float)pix_y;
                    int tile_id = tile_y * grid_x + tile_x;
                    int range_start = ranges[2 * tile_id], range_end = ranges[2 * tile_id + 1];
                    size_t px = (size_t)pix_y * W + pix_x;
                    float T_final = final_Ts[px];
                    int last_kept = imin(range_end, range_start + n_contrib[px]);
                    float T = T_final;
                    float accum_rec[3] = {0.0f, 0.0f, 0.0f};
                    float last_alpha = 0.0f, last_color[3] = {0.0f, 0.0f, 0.0f};
                    const float *dL_dpixel = dL_dpixels + 3 * px;
                    float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
                    for (int i = last_kept - 1; i > range_start - 1; --i) {
                        int gid = point_list[i];
                        const float *pxy = xy + 2 * gid, *con_o = conic_opacity + 4 * gid, *color = colors + 3 * gid;
                        float d_x = pxy[0] - pixf_x, d_y = pxy[1] - pixf_y;
                        float power = -0.5f * (con_o[0] * d_x * d_x + con_o[2] * d_y * d_y) - con_o[1] * d_x * d_y;
                        if (power > 0.0f) continue;
                        float G = expf(power);
                        float alpha = fminf_(0.99f, con_o[3] * G);
                        if (alpha < (1.0f / 255.0f)) continue;
                        T = T / (1.0f - alpha);
                        float dchannel_dcolor = alpha * T;
                        float tmp[3];
                        for (int c = 0; c < 3; ++c) {
                            accum_rec[c] = last_alpha * last_color[c] + (1.0f - last_alpha) * accum_rec[c];
                            last_color[c] = color[c];
                            tmp[c] = color[c] - accum_rec[c];
                        }
                        float dL_dalpha = dot3(tmp, dL_dpixel);
                        for (int c = 0; c < 3; ++c) dL_dcolors[3 * gid + c] += (double)(dchannel_dcolor * dL_dpixel[c]);
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        float bg_dot_dpixel = dot3(bg, dL_dpixel);
                        dL_dalpha += (-T_final / (1.0f - alpha)) * bg_dot_dpixel;
                        float dL_dG = con_o[3] * dL_dalpha;
                        float gdx = G * d_x, gdy = G * d_y;
                        float dG_ddelx = -gdx * con_o[0] - gdy * con_o[1];
                        float dG_ddely = -gdy * con_o[2] - gdx * con_o[1];
                        dL_dmean2D[3 * gid] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                        dL_dmean2D[3 * gid + 1] += (double)(dL_dG * dG_ddely * ddely_dy);
                        dL_dconic2D[4 * gid] += (double)(-0.5f * gdx * d_x * dL_dG);
                        dL_dconic2D[4 * gid + 1] += (double)(-0.5f * gdx * d_y * dL_dG);
                        dL_dconic2D[4 * gid + 3] += (double)(-0.5f * gdy * d_y * dL_dG);
                        dL_dopacity[gid] += (double)(G * dL_dalpha);
                    }
                }
}

/* backward.py:259-435 (compute_cov2d_backward_kernel).  dL_dconics (N,4); dL_dmeans (N,3) +=;
 * dL_dcov3Ds (N,6) written.  NOTE quirk Q1: uses T = W*J, cov2D = T^T Vrk^T T. */
void gsro_cov2d_backward(int N, const float *means, const float *cov3Ds, const int32_t *radii, float h_x,
                         float h_y, float tan_fovx, float tan_fovy, const float *view,
                         const float *dL_dconics, float *dL_dmeans, float *dL_dcov3Ds)
{
    for (int idx = 0; idx < N; ++idx) {
        float *dcov = dL_dcov3Ds + 6 * idx;
        if (radii[idx] <= 0) { for (int k = 0; k < 6; ++k) dcov[k] = 0.0f; continue; }
        const float *mean = means + 3 * idx, *c3 = cov3Ds + 6 * idx;
        float dL_dconic[3] = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
        float mh[4] = {mean[0], mean[1], mean[2], 1.0f}, t[4];
        v4_mul_m44(mh, view, t);
        float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
        float tz = t[2], inv_tz = 1.0f / tz;
        float txtz = t[0] * inv_tz, tytz = t[1] * inv_tz;
        int xcl = (txtz < -limx) || (txtz > limx), ycl = (tytz < -limy) || (tytz > limy);
        float x_grad_mul = 1.0f - (float)xcl, y_grad_mul = 1.0f - (float)ycl;
        float tx = fminf_(limx, fmaxf_(-limx, txtz)) * tz;
        float ty = fminf_(limy, fmaxf_(-limy, tytz)) * tz;
        float inv_tz2 = inv_tz * inv_tz, inv_tz3 = inv_tz2 * inv_tz;
        float J00 = h_x * inv_tz, J11 = h_y * inv_tz;
        float J02 = -h_x * tx * inv_tz2, J12 = -h_y * ty * inv_tz2;
        mat33 Jpre = {{{J00, 0.0f, J02}, {0.0f, J11, J12}, {0.0f, 0.0f, 0.0f}}};
        mat33 J = m33_T(Jpre);
        mat33 Wm = {{{view[0], view[1], view[2]}, {view[4], view[5], view[6]}, {view[8], view[9], view[10]}}};
        mat33 T = m33_mul(Wm, J);
        mat33 Vrk = {{{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}}};
        mat33 c2 = m33_mul(m33_mul(m33_T(T), m33_T(Vrk)), T);
        float a = c2.m[0][0] + 0.3f, b = c2.m[0][1], c = c2.m[1][1] + 0.3f;
        float denom = a * c - b * b;
        float dL_da = 0.0f, dL_db = 0.0f, dL_dc = 0.0f;
        if (denom != 0.0f) {
            float denom2inv = 1.0f / (denom * denom + 1e-7f);
            dL_da = denom2inv * (-c * c * dL_dconic[0] + 2.0f * b * c * dL_dconic[1] + (denom - a * c) * dL_dconic[2]);
            dL_dc = denom2inv * (-a * a * dL_dconic[2] + 2.0f * a * b * dL_dconic[1] + (denom - a * c) * dL_dconic[0]);
            dL_db = denom2inv * 2.0f * (b * c * dL_dconic[0] - (denom + 2.0f * b * b) * dL_dconic[1] + a * b * dL_dconic[2]);
        }
#define Tm(i, j) T.m[i][j]
#define V(i, j) Vrk.m[i][j]
        dcov[0] = Tm(0,0) * Tm(0,0) * dL_da + Tm(0,0) * Tm(0,1) * dL_db + Tm(0,1) * Tm(0,1) * dL_dc;
        dcov[1] = 2.0f * Tm(0,0) * Tm(1,0) * dL_da + (Tm(0,0) * Tm(1,1) + Tm(1,0) * Tm(0,1)) * dL_db + 2.0f * Tm(0,1) * Tm(1,1) * dL_dc;
        dcov[2] = 2.0f * Tm(0,0) * Tm(2,0) * dL_da + (Tm(0,0) * Tm(2,1) + Tm(2,0) * Tm(0,1)) * dL_db + 2.0f * Tm(0,1) * Tm(2,1) * dL_dc;
        dcov[3] = Tm(1,0) * Tm(1,0) * dL_da + Tm(1,0) * Tm(1,1) * dL_db + Tm(1,1) * Tm(1,1) * dL_dc;
        dcov[4] = 2.0f * Tm(2,0) * Tm(1,0) * dL_da + (Tm(1,0) * Tm(2,1) + Tm(2,0) * Tm(1,1)) * dL_db + 2.0f * Tm(1,1) * Tm(2,1) * dL_dc;
        dcov[5] = Tm(2,0) * Tm(2,0) * dL_da + Tm(2,0) * Tm(2,1) * dL_db + Tm(2,1) * Tm(2,1) * dL_dc;
        float dL_dT00 = 2.0f * (Tm(0,0) * V(0,0) + Tm(1,0) * V(1,0) + Tm(2,0) * V(2,0)) * dL_da + (Tm(0,1) * V(0,0) + Tm(1,1) * V(1,0) + Tm(2,1) * V(2,0)) * dL_db;
        float dL_dT01 = 2.0f * (Tm(0,0) * V(0,1) + Tm(1,0) * V(1,1) + Tm(2,0) * V(2,1)) * dL_da + (Tm(0,1) * V(0,1) + Tm(1,1) * V(1,1) + Tm(2,1) * V(2,1)) * dL_db;
        float dL_dT02 = 2.0f * (Tm(0,0) * V(0,2) + Tm(1,0) * V(1,2) + Tm(2,0) * V(2,2)) * dL_da + (Tm(0,1) * V(0,2) + Tm(1,1) * V(1,2) + Tm(2,1) * V(2,2)) * dL_db;
        float dL_dT10 = 2.0f * (Tm(0,1) * V(0,0) + Tm(1,1) * V(1,0) + Tm(2,1) * V(2,0)) * dL_dc + (Tm(0,0) * V(0,0) + Tm(1,0) * V(1,0) + Tm(2,0) * V(2,0)) * dL_db;
        float dL_dT11 = 2.0f * (Tm(0,1) * V(0,1) + Tm(1,1) * V(1,1) + Tm(2,1) * V(2,1)) * dL_dc + (Tm(0,0) * V(0,1) + Tm(1,0) * V(1,1) + Tm(2,0) * V(2,1)) * dL_db;
        float dL_dT12 = 2.0f * (Tm(0,1) * V(0,2) + Tm(1,1) * V(1,2) + Tm(2,1) * V(2,2)) * dL_dc + (Tm(0,0) * V(0,2) + Tm(1,0) * V(1,2) + Tm(2,0) * V(2,2)) * dL_db;
#undef Tm
#undef V
        float dL_dJ00 = Wm.m[0][0] * dL_dT00 + Wm.m[1][0] * dL_dT01 + Wm.m[2][0] * dL_dT02;
        float dL_dJ02 = Wm.m[0][2] * dL_dT00 + Wm.m[1][2] * dL_dT01 + Wm.m[2][2] * dL_dT02;
        float dL_dJ11 = Wm.m[0][1] * dL_dT10 + Wm.m[1][1] * dL_dT11 + Wm.m[2][1] * dL_dT12;
        float dL_dJ12 = Wm.m[0][2] * dL_dT10 + Wm.m[1][2] * dL_dT11 + Wm.m[2][2] * dL_dT12;
        float dL_dtx = -h_x * inv_tz2 * dL_dJ02;
        float dL_dty = -h_y * inv_tz2 * dL_dJ12;
        float dL_dtz = -h_x * inv_tz2 * dL_dJ00 - h_y * inv_tz2 * dL_dJ11 + 2.0f * h_x * tx * inv_tz3 * dL_dJ02 + 2.0f * h_y * ty * inv_tz3 * dL_dJ12;
        float dL_dt[4] = {dL_dtx * x_grad_mul, dL_dty * y_grad_mul, dL_dtz, 1.0f};
        /* (dL_dt,1) * transpose(view): out[j] = sum_i dL_dt[i] * view[j][i]  (quirk Q3) */
        for (int j = 0; j < 3; ++j) {
            float r = view[j * 4 + 0] * dL_dt[0];
            r += view[j * 4 + 1] * dL_dt[1];
            r += view[j * 4 + 2] * dL_dt[2];
            r += view[j * 4 + 3] * dL_dt[3];
            dL_dmeans[3 * idx + j] += r;
        }
    }
}

/* backward.py:709-768 (compute_projection_backward_kernel). */
void gsro_projection_backward(int N, const float *means, const int32_t *radii, const float *proj,
                              const float *dL_dmean2D, float *dL_dmeans)
{
#define P(r, c) proj[(r) * 4 + (c)]
    for (int idx = 0; idx < N; ++idx) {
        if (radii[idx] <= 0) continue;
        const float *m = means + 3 * idx, *d2 = dL_dmean2D + 3 * idx;
        float mh[4] = {m[0], m[1], m[2], 1.0f}, m_hom[4];
        v4_mul_m44(mh, proj, m_hom);
        float m_w = 1.0f / (m_hom[3] + 0.0000001f);
        float mul1 = (P(0,0) * m[0] + P(1,0) * m[1] + P(2,0) * m[2] + P(3,0)) * m_w * m_w;
        float mul2 = (P(0,1) * m[0] + P(1,1) * m[1] + P(2,1) * m[2] + P(3,1)) * m_w * m_w;
        for (int k = 0; k < 3; ++k) {
            float g = (P(k,0) * m_w - P(k,3) * mul1) * d2[0] + (P(k,1) * m_w - P(k,3) * mul2) * d2[1];
            dL_dmeans[3 * idx + k] += g;
        }
    }
#undef P
}

/* backward.py:43-64 (dnormvdv). */
static void dnormvdv(const float v[3], const float dv[3], float out[3])
{
    float sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    if (sum2 < 1e-10f) { out[0] = out[1] = out[2] = 0.0f; return; }
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    out[0] = ((sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * invsum32;
    out[1] = (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * invsum32;
    out[2] = (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * invsum32;
}

/* backward.py:69-255 (sh_backward_kernel).  dL_dshs (N*16,3) written (stride 16, quirk Q6);
 * dL_dmeans +=. */
void gsro_sh_backward(int N, int degree, const float *means, const float *shs, const int32_t *radii,
                      const float *campos, const float *clamped_state, const float *dL_dcolor,
                      float *dL_dmeans, float *dL_dshs)
{
    for (int idx = 0; idx < N; ++idx) {
        if (radii[idx] <= 0) continue;
        const float *mean = means + 3 * idx;
        float dir_orig[3] = {mean[0] - campos[0], mean[1] - campos[1], mean[2] - campos[2]};
        float dir_len = sqrtf(dot3(dir_orig, dir_orig));
        if (dir_len < 1e-8f) continue;
        float x = dir_orig[0] / dir_len, y = dir_orig[1] / dir_len, z = dir_orig[2] / dir_len;
        float dL_dRGB[3];
        for (int c = 0; c < 3; ++c)
            dL_dRGB[c] = dL_dcolor[3 * idx + c] * (1.0f + (-1.0f * clamped_state[3 * idx + c]));
        float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
        const float *sh = shs + (size_t)idx * 48;
        float *out = dL_dshs + (size_t)idx * 48;
#define SHV(k, c) sh[(k) * 3 + (c)]
#define OUT(k, coef) for (int c = 0; c < 3; ++c) out[(k) * 3 + c] = (coef) * dL_dRGB[c]
        OUT(0, SH_C0);
        if (degree > 0) {
            float d1 = -SH_C1 * y, d2 = SH_C1 * z, d3 = -SH_C1 * x;
            OUT(1, d1); OUT(2, d2); OUT(3, d3);
            for (int c = 0; c < 3; ++c) {
                dRGBdx[c] = -SH_C1 * SHV(3, c);
                dRGBdy[c] = -SH_C1 * SHV(1, c);
                dRGBdz[c] = SH_C1 * SHV(2, c);
            }
            if (degree > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                const float C2_0 = 1.0925484305920792f, C2_1 = -1.0925484305920792f, C2_2 = 0.31539156525252005f,
                            C2_3 = -1.0925484305920792f, C2_4 = 0.5462742152960396f;
                float d4 = C2_0 * xy, d5 = C2_1 * yz, d6 = C2_2 * (2.0f * zz - xx - yy), d7 = C2_3 * xz, d8 = C2_4 * (xx - yy);
                OUT(4, d4); OUT(5, d5); OUT(6, d6); OUT(7, d7); OUT(8, d8);
                for (int c = 0; c < 3; ++c) {
                    dRGBdx[c] += C2_0 * y * SHV(4, c) + C2_2 * 2.0f * -x * SHV(6, c) + C2_3 * z * SHV(7, c) + C2_4 * 2.0f * x * SHV(8, c);
                    dRGBdy[c] += C2_0 * x * SHV(4, c) + C2_1 * z * SHV(5, c) + C2_2 * 2.0f * -y * SHV(6, c) + C2_4 * 2.0f * -y * SHV(8, c);
                    dRGBdz[c] += C2_1 * y * SHV(5, c) + C2_2 * 2.0f * 2.0f * z * SHV(6, c) + C2_3 * x * SHV(7, c);
                }
                if (degree > 2) {
                    const float C3_0 = -0.5900435899266435f, C3_1 = 2.890611442640554f, C3_2 = -0.4570457994644658f,
                                C3_3 = 0.3731763325901154f, C3_4 = -0.4570457994644658f, C3_5 = 1.445305721320277f,
                                C3_6 = -0.5900435899266435f;
                    float d9 = C3_0 * y * (3.0f * xx - yy), d10 = C3_1 * xy * z, d11 = C3_2 * y * (4.0f * zz - xx - yy),
                          d12 = C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), d13 = C3_4 * x * (4.0f * zz - xx - yy),
                          d14 = C3_5 * z * (xx - yy), d15 = C3_6 * x * (xx - 3.0f * yy);
                    OUT(9, d9); OUT(10, d10); OUT(11, d11); OUT(12, d12); OUT(13, d13); OUT(14, d14); OUT(15, d15);
                    for (int c = 0; c < 3; ++c) {
                        dRGBdx[c] += (C3_0 * SHV(9, c) * 3.0f * 2.0f * xy + C3_1 * SHV(10, c) * yz + C3_2 * SHV(11, c) * -2.0f * xy +
                                      C3_3 * SHV(12, c) * -3.0f * 2.0f * xz + C3_4 * SHV(13, c) * (-3.0f * xx + 4.0f * zz - yy) +
                                      C3_5 * SHV(14, c) * 2.0f * xz + C3_6 * SHV(15, c) * 3.0f * (xx - yy));
                        dRGBdy[c] += (C3_0 * SHV(9, c) * 3.0f * (xx - yy) + C3_1 * SHV(10, c) * xz +
                                      C3_2 * SHV(11, c) * (-3.0f * yy + 4.0f * zz - xx) + C3_3 * SHV(12, c) * -3.0f * 2.0f * yz +
                                      C3_4 * SHV(13, c) * -2.0f * xy + C3_5 * SHV(14, c) * -2.0f * yz + C3_6 * SHV(15, c) * -3.0f * 2.0f * xy);
                        dRGBdz[c] += (C3_1 * SHV(10, c) * xy + C3_2 * SHV(11, c) * 4.0f * 2.0f * yz +
                                      C3_3 * SHV(12, c) * 3.0f * (2.0f * zz - xx - yy) + C3_4 * SHV(13, c) * 4.0f * 2.0f * xz +
                                      C3_5 * SHV(14, c) * (xx - yy));
                    }
                }
            }
        }
#undef SHV
#undef OUT
        float dL_ddir[3] = {dot3(dRGBdx, dL_dRGB), dot3(dRGBdy, dL_dRGB), dot3(dRGBdz, dL_dRGB)};
        float loc[3];
        dnormvdv(dir_orig, dL_ddir, loc);
        for (int c = 0; c < 3; ++c) dL_dmeans[3 * idx + c] += loc[c];
    }
}

/* backward.py:439-556 (compute_cov3d_backward_kernel).  NOTE quirk Q2: M = S*R with the
 * 1-2(y^2+z^2) matrix form; rots are (x,y,z,w). */
void gsro_cov3d_backward(int N, const float *scales, const float *rots, const int32_t *radii,
                         float scale_modifier, const float *dL_dcov3Ds, float *dL_dscales, float *dL_drots)
{
    for (int idx = 0; idx < N; ++idx) {
        if (radii[idx] <= 0) {
            for (int k = 0; k < 3; ++k) dL_dscales[3 * idx + k] = 0.0f;
            for (int k = 0; k < 4; ++k) dL_drots[4 * idx + k] = 0.0f;
            continue;
        }
        const float *sv = scales + 3 * idx, *q = rots + 4 * idx, *dc = dL_dcov3Ds + 6 * idx;
        float r = q[3], x = q[0], y = q[1], z = q[2];
        mat33 R = {{{1.0f - 2.0f * (y * y + z * z), 2.0f * (x * y - r * z), 2.0f * (x * z + r * y)},
                    {2.0f * (x * y + r * z), 1.0f - 2.0f * (x * x + z * z), 2.0f * (y * z - r * x)},
                    {2.0f * (x * z - r * y), 2.0f * (y * z + r * x), 1.0f - 2.0f * (x * x + y * y)}}};
        float s_vec[3] = {scale_modifier * sv[0], scale_modifier * sv[1], scale_modifier * sv[2]};
        mat33 S = {{{s_vec[0], 0, 0}, {0, s_vec[1], 0}, {0, 0, s_vec[2]}}};
        mat33 M = m33_mul(S, R);
        mat33 dSig = {{{dc[0], 0.5f * dc[1], 0.5f * dc[2]}, {0.5f * dc[1], dc[3], 0.5f * dc[4]}, {0.5f * dc[2], 0.5f * dc[4], dc[5]}}};
        mat33 twoM;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) twoM.m[i][j] = 2.0f * M.m[i][j];
        mat33 dL_dM = m33_mul(twoM, dSig); /* 2.0 * M * dL_dSigma, left-assoc */
        mat33 Rt = m33_T(R), dMt = m33_T(dL_dM);
        for (int k = 0; k < 3; ++k) dL_dscales[3 * idx + k] = dot3(Rt.m[k], dMt.m[k]) * scale_modifier;
        float ds[3][3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) ds[i][j] = dMt.m[i][j] * s_vec[i];
        float dL_dr = 2.0f * (z * (ds[0][1] - ds[1][0]) + y * (ds[2][0] - ds[0][2]) + x * (ds[1][2] - ds[2][1]));
        float dL_dx = 2.0f * (y * (ds[1][0] + ds[0][1]) + z * (ds[2][0] + ds[0][2]) + r * (ds[1][2] - ds[2][1])) - 4.0f * x * (ds[2][2] + ds[1][1]);
        float dL_dy = 2.0f * (x * (ds[1][0] + ds[0][1]) + r * (ds[2][0] - ds[0][2]) + z * (ds[1][2] + ds[2][1])) - 4.0f * y * (ds[2][2] + ds[0][0]);
        float dL_dz = 2.0f * (r * (ds[0][1] - ds[1][0]) + x * (ds[2][0] + ds[0][2]) + y * (ds[1][2] + ds[2][1])) - 4.0f * z * (ds[1][1] + ds[0][0]);
        dL_drots[4 * idx] = dL_dx; dL_drots[4 * idx + 1] = dL_dy; dL_drots[4 * idx + 2] = dL_dz; dL_drots[4 * idx + 3] = dL_dr;
    }
}

/* ===================== "next" rows f2 / f3 ===================== */

/* loss.py:12-31 (l1_loss_kernel) -- launch dim=(width,height): i = x outer, j = y fastest (A7); float
 * accumulation in that order.  Returns the raw sum; the caller divides by W*H*3 (loss.py:174). */
float gsro_l1_loss_sum(int W, int H, const float *rendered, const float *target)
{
    float acc = 0.0f;
    for (int i = 0; i < W; ++i)
        for (int j = 0; j < H; ++j) {
            const float *r = rendered + 3 * ((size_t)j * W + i), *t = target + 3 * ((size_t)j * W + i);
            float d0 = fabsf(r[0] - t[0]), d1 = fabsf(r[1] - t[1]), d2 = fabsf(r[2] - t[2]);
            float l1 = d0 + d1 + d2;
            acc += l1;
        }
    return acc;
}

/* loss.py:33-45 (gaussian_kernel) + :47-119 (ssim_kernel) + :178-215 (ssim).  Quirk Q21: the 11 window weights are
 * kernel[k] = exp(-(k-5)^2 / (2 sigma^2)) -- a Gaussian centred on INDEX 5 -- but the kernel indexes them by DISTANCE
 * (gaussian_weights[|x-i|], loss.py:81-84), so a tap at distance d weighs exp(-(d-5)^2/4.5): largest at the rim of the
 * window, smallest at the centre.  Reproduced as written.  Launch dim=(width,height): i = x outer, j = y fastest (A7).
 * Returns the raw sum of per-pixel SSIM values; the caller divides by W*H (loss.py:214). */
float gsro_ssim_sum(int W, int H, const float *rendered, const float *target)
{
    const int window_size = 11, half_window = window_size / 2;
    const float sigma = 1.5f;
    float gw[11];
    for (int i = 0; i < window_size; ++i) {
        const int x = i - window_size / 2;
        gw[i] = expf(-1.0f * (float)(x * x) / (2.0f * sigma * sigma));
    }
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    float total = 0.0f;
    for (int i = 0; i < W; ++i)
        for (int j = 0; j < H; ++j) {
            float mu1[3] = {0, 0, 0}, mu2[3] = {0, 0, 0}, s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0}, s12[3] = {0, 0, 0}, weight_sum = 0.0f;
            const int y0 = j - half_window > 0 ? j - half_window : 0, y1 = j + half_window + 1 < H ? j + half_window + 1 : H;
            const int x0 = i - half_window > 0 ? i - half_window : 0, x1 = i + half_window + 1 < W ? i + half_window + 1 : W;
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) {
                    const int wy = abs(y - j), wx = abs(x - i);
                    if (wx <= half_window && wy <= half_window) {
                        const float w = gw[wx] * gw[wy];
                        const float *p1 = rendered + 3 * ((size_t)y * W + x), *p2 = target + 3 * ((size_t)y * W + x);
                        for (int c = 0; c < 3; ++c) {
                            mu1[c] += p1[c] * w;
                            mu2[c] += p2[c] * w;
                            s1[c] += (p1[c] * p1[c]) * w;
                            s2[c] += (p2[c] * p2[c]) * w;
                            s12[c] += (p1[c] * p2[c]) * w;
                        }
                        weight_sum += w;
                    }
                }
            if (weight_sum > 0.0f)
                for (int c = 0; c < 3; ++c) { mu1[c] /= weight_sum; mu2[c] /= weight_sum; s1[c] /= weight_sum; s2[c] /= weight_sum; s12[c] /= weight_sum; }
            float ss[3];
            for (int c = 0; c < 3; ++c) {
                const float v1 = s1[c] - mu1[c] * mu1[c], v2 = s2[c] - mu2[c] * mu2[c], v12 = s12[c] - mu1[c] * mu2[c];
                ss[c] = ((2.0f * mu1[c] * mu2[c] + c1) * (2.0f * v12 + c2)) / ((mu1[c] * mu1[c] + mu2[c] * mu2[c] + c1) * (v1 + v2 + c2));
            }
            total += (ss[0] + ss[1] + ss[2]) / 3.0f;
        }
    return total;
}

/* loss.py:247-269 (depth_loss_kernel) + :271-303 (depth_loss): sum |rendered - target| * mask, x outer / y fastest; the
 * caller divides by W*H (loss.py:302). */
float gsro_depth_loss_sum(int W, int H, const float *rendered, const float *target, const float *mask)
{
    float acc = 0.0f;
    for (int i = 0; i < W; ++i)
        for (int j = 0; j < H; ++j) {
            const size_t k = (size_t)j * W + i;
            acc += fabsf(rendered[k] - target[k]) * mask[k];
        }
    return acc;
}

/* loss.py:122-146 (backprop_l1_pixel_gradients); wp.sign(x) = -1 if x < 0 else +1 (assumption A9). */
void gsro_l1_pixel_grad(int W, int H, const float *rendered, const float *target, float l1_weight, float *pixel_grad)
{
    for (size_t k = 0; k < (size_t)W * H * 3; ++k) {
        float d = rendered[k] - target[k];
        pixel_grad[k] = l1_weight * (d < 0.0f ? -1.0f : 1.0f);
    }
}

/* optimizer.py:7-139 (adam_update). */
void gsro_adam_update(int N, const float *gpos, const float *gscl, const float *grot, const float *gopa, const float *gsh,
                      float lr_pos, float lr_scale, float lr_rot, float lr_opac, float lr_sh, float beta1, float beta2,
                      float epsilon, int iteration, float *pos, float *scl, float *rot, float *opa, float *sh, float *mpos,
                      float *mscl, float *mrot, float *mopa, float *msh, float *vpos, float *vscl, float *vrot, float *vopa,
                      float *vsh)
{
    const float bc1 = 1.0f - powf(beta1, (float)(iteration + 1));
    const float bc2 = 1.0f - powf(beta2, (float)(iteration + 1));
    const float omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
    for (int i = 0; i < N; ++i) {
        for (int c = 0; c < 3; ++c) { /* positions :51-58 */
            int k = 3 * i + c;
            mpos[k] = beta1 * mpos[k] + omb1 * gpos[k];
            vpos[k] = beta2 * vpos[k] + omb2 * (gpos[k] * gpos[k]);
            float mc = mpos[k] / bc1, vc = vpos[k] / bc2;
            float den = sqrtf(vc) + epsilon;
            pos[k] = pos[k] - lr_pos * (mc / (den + 1e-9f));
        }
        for (int c = 0; c < 3; ++c) { /* scales :61-75 */
            int k = 3 * i + c;
            mscl[k] = beta1 * mscl[k] + omb1 * gscl[k];
            vscl[k] = beta2 * vscl[k] + omb2 * (gscl[k] * gscl[k]);
            float mc = mscl[k] / bc1, vc = vscl[k] / bc2;
            float den = sqrtf(vc) + epsilon;
            float upd = lr_scale * (mc / (den + 1e-9f));
            scl[k] = fmaxf_(scl[k] - upd, 0.001f);
        }
        for (int c = 0; c < 4; ++c) { /* rotations :78-101 */
            int k = 4 * i + c;
            mrot[k] = beta1 * mrot[k] + omb1 * grot[k];
            vrot[k] = beta2 * vrot[k] + omb2 * (grot[k] * grot[k]);
            float mc = mrot[k] / bc1, vc = vrot[k] / bc2;
            float den = sqrtf(vc) + epsilon;
            rot[k] = rot[k] - lr_rot * mc / den;
        }
        { /* :103-115 */
            float *q = rot + 4 * i;
            float len = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            if (len > 0.0f) { q[0] = q[0] / len; q[1] = q[1] / len; q[2] = q[2] / len; q[3] = q[3] / len; }
        }
        { /* opacity :118-126 */
            mopa[i] = beta1 * mopa[i] + omb1 * gopa[i];
            vopa[i] = beta2 * vopa[i] + omb2 * (gopa[i] * gopa[i]);
            float mc = mopa[i] / bc1, vc = vopa[i] / bc2;
            float upd = lr_opac * mc / (sqrtf(vc) + epsilon);
            opa[i] = fmaxf_(fminf_(opa[i] - upd, 1.0f), 0.0f);
        }
        for (int j = 0; j < 48; ++j) { /* SH :128-139 */
            size_t k = (size_t)i * 48 + j;
            msh[k] = beta1 * msh[k] + omb1 * gsh[k];
            vsh[k] = beta2 * vsh[k] + omb2 * (gsh[k] * gsh[k]);
            float mc = msh[k] / bc1, vc = vsh[k] / bc2;
            float den = sqrtf(vc) + epsilon;
            sh[k] = sh[k] - lr_sh * (mc / (den + 1e-9f));
        }
    }
}
