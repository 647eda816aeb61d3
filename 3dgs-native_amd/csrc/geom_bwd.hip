// geom_bwd.hip -- fused per-Gaussian backward for gfx950.
//
// The reference runs four kernels in a fixed order, each re-reading the Gaussian and adding into the
// same dL_dmeans array, with two pointless host syncs in between (backward.py:770-888):
//   compute_cov2d_backward_kernel      :259-435   conic grad -> dL/dSigma3D, dL/dmean (via T = W*J)
//   compute_projection_backward_kernel :709-768   dL/dmean2D -> dL/dmean
//   sh_backward_kernel                 :69-255    dL/dcolor  -> dL/dSH, dL/dmean (via view direction)
//   compute_cov3d_backward_kernel      :439-556   dL/dSigma3D -> dL/dscale, dL/drot
// This is ONE kernel, one thread per Gaussian: dL/dSigma3D never leaves registers, the three mean
// contributions are added in the reference's order (cov2d, projection, SH), and every output array is
// written for every Gaussian (zeros for culled ones), so the caller pre-zeroes nothing.  It also
// unpacks the 64-byte blend-gradient accumulator into the API arrays dL_dcolor / dL_dmean2D /
// dL_dconic / dL_dopacity.
//
// Reference quirks kept on purpose: Q1 (backward uses T = W*J, cov2D = T^T Vrk^T T), Q2 (M = S*R with
// the 1-2(y^2+z^2) matrix), Q3 ((dL_dt,1) * view^T adds view[j][3]), and Q16: backward() never passes
// its scale_modifier down (backward.py:1155-1182; default 1.0 at :805), so the cov3d part runs with 1.0.
#include "gsr_internal.h"
#include "sh_stage.h"
#include "sigma3d.h"

namespace {

struct M33 {
    float m[3][3];
};
__device__ __forceinline__ M33 mul33(const M33 &a, const M33 &b)
{
    M33 t;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) s += a.m[i][k] * b.m[k][j];
            t.m[i][j] = s;
        }
    return t;
}
__device__ __forceinline__ M33 tr33(const M33 &a)
{
    M33 t;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) t.m[i][j] = a.m[j][i];
    return t;
}
__device__ __forceinline__ float dot3(const float a[3], const float b[3])
{
    float r = a[0] * b[0];
    r += a[1] * b[1];
    r += a[2] * b[2];
    return r;
}

// DIRGRAD: the forward left, per visible Gaussian, the nine sums d(colour before clamping)/d(direction) that the SH backward
// forms from the 48 coefficients (GsrGeom.sh_dir_grad, written by preprocess_kernel with these very expressions).  The kernel
// then does not read the SH array at all: 36 instead of 192 bytes per Gaussian in, 48 fewer staging registers; the LDS image
// only transposes the OUTPUT rows.  Without it (a caller that kept no forward state) the coefficients are read as before.
template <bool DIRGRAD>
__global__ __launch_bounds__(256) void geom_backward_kernel(
    int64_t N, const float *__restrict__ means, const float *__restrict__ scales, const float *__restrict__ rots,
    const float *__restrict__ shs, int degree, CamK cam, float h_x, float h_y, const int32_t *__restrict__ radii,
    const float *__restrict__ cov3Ds, const float *__restrict__ clamped_state, const GradRec *__restrict__ acc,
    float *__restrict__ dL_dmean3D, float *__restrict__ dL_dscale, float *__restrict__ dL_drot, float *__restrict__ dL_dopacity,
    float *__restrict__ dL_dshs, float *__restrict__ dL_dcolor, float *__restrict__ dL_dmean2D, float *__restrict__ dL_dconic,
    float *__restrict__ dL_drgb, const float *__restrict__ sh_dir_grad, float scale_mod)
{
    // SH rows (input coefficients, then in place the output gradients) live in LDS; moved cooperatively
    __shared__ float4 s_rows[4 * SH_WAVE_F4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave_row0 = (int64_t)blockIdx.x * blockDim.x + wv * 64;
    const int rows_valid = (int)min((int64_t)64, max((int64_t)0, N - wave_row0));
    float4 *lds_wave = s_rows + wv * SH_WAVE_F4;
    const int64_t idx = wave_row0 + lane;
    const bool in_range = idx < N;
    // only the SH rows of visible Gaussians are read (the radii are known up front here)
    const int my_radius = in_range ? radii[idx] : 0;
    const unsigned long long row_mask = (degree > 0) ? __ballot(my_radius > 0) : 0ull;
    ShRegs sh_regs;
    float4 dg4[3]; // DIRGRAD: the wave's 64 x 9 floats as 144 coalesced float4 (lane, lane + 64, lane + 128)
    if constexpr (DIRGRAD) {
        const float *g = sh_dir_grad + 9 * wave_row0;
        const int nfl = rows_valid * 9;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int j = k * 64 + lane;
            dg4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row_mask != 0ull && j < 144) {
                if (4 * j + 3 < nfl) dg4[k] = gsr_ld4<GSR_NT_INPUTS != 0>(reinterpret_cast<const float4 *>(g) + j);
                else {
                    if (4 * j < nfl) dg4[k].x = g[4 * j];
                    if (4 * j + 1 < nfl) dg4[k].y = g[4 * j + 1];
                    if (4 * j + 2 < nfl) dg4[k].z = g[4 * j + 2];
                }
            }
        }
    } else {
        sh_rows_fetch(reinterpret_cast<const float4 *>(shs) + wave_row0 * 12, sh_regs, lane, row_mask);
    }
    float *row = reinterpret_cast<float *>(lds_wave + lane * SH_ROW_F4);

    // phase 1 (while the SH rows are in flight): blend-gradient unpacking, cov2d and projection backward
    float g_col[3] = {0.f, 0.f, 0.f}, g_m2d[2] = {0.f, 0.f}, g_con[3] = {0.f, 0.f, 0.f};
    float o_mean[3] = {0.f, 0.f, 0.f}, o_scale[3] = {0.f, 0.f, 0.f}, o_rot[4] = {0.f, 0.f, 0.f, 0.f};
    float mean[3] = {0.f, 0.f, 0.f}, dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int sh_written = 0; // number of leading SH coefficients whose gradient was written into the LDS row
    float o_rgb[3] = {0.f, 0.f, 0.f}; // dL_dcolor * (1 - clamped) where the SH backward runs, else 0 (optional output)
    bool vis = false;
    float sv[3] = {0.f, 0.f, 0.f};
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in_range) {

    const float4 *ap = reinterpret_cast<const float4 *>(acc + idx);
    const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2]; // GradRec: r g b | x y 0 | a b 0 c | opacity
    const float a8 = a2.z;
    g_col[0] = a0.x; g_col[1] = a0.y; g_col[2] = a0.z;
    g_m2d[0] = a0.w; g_m2d[1] = a1.x;
    g_con[0] = a1.z; g_con[1] = a1.w; g_con[2] = a2.y; // d/da, d/db, d/dc
    // API-layout copies of the blend-stage gradients, for a caller that wants packed arrays (NULL: it reads the record's columns)
    // (dL_dcolor and dL_dmean2D: 12-byte rows, written wave-cooperatively at the end of the kernel)
    if (dL_dconic) *reinterpret_cast<float4 *>(dL_dconic + 4 * idx) = make_float4(g_con[0], g_con[1], 0.0f, g_con[2]);
    gsr_st1<GSR_NT_MISC_STORE != 0>(dL_dopacity + idx, a8);

    vis = my_radius > 0;
    if (vis) {
        mean[0] = gsr_ld1<GSR_NT_INPUTS != 0>(means + 3 * idx); mean[1] = gsr_ld1<GSR_NT_INPUTS != 0>(means + 3 * idx + 1); mean[2] = gsr_ld1<GSR_NT_INPUTS != 0>(means + 3 * idx + 2);
        // (scale and quaternion: the cov3d backward at the end needs them, and so does Sigma3D when it is recomputed)
        sv[0] = gsr_ld1<GSR_NT_INPUTS != 0>(scales + 3 * idx); sv[1] = gsr_ld1<GSR_NT_INPUTS != 0>(scales + 3 * idx + 1); sv[2] = gsr_ld1<GSR_NT_INPUTS != 0>(scales + 3 * idx + 2);
        q = gsr_ld4<GSR_NT_INPUTS != 0>(reinterpret_cast<const float4 *>(rots + 4 * idx));
        // ---------------- cov2d backward (backward.py:259-435) ----------------
        {
            float c3[6];
            if (cov3Ds) { // (uniform)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    c3[2 * k] = gsr_ld1<GSR_NT_INPUTS != 0>(cov3Ds + 6 * idx + 2 * k); c3[2 * k + 1] = gsr_ld1<GSR_NT_INPUTS != 0>(cov3Ds + 6 * idx + 2 * k + 1);
                }
            } else {
                // the caller vouches that GsrGeom.cov3D would be the forward's own output for these scales / rotations / scale
                // modifier: the same instructions give the same bits (sigma3d.h) and 24 bytes per Gaussian are not read
                gsr_sigma3d(scale_mod * sv[0], scale_mod * sv[1], scale_mod * sv[2], q, c3);
            }
            float t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float r = cam.view[j] * mean[0];
                r += cam.view[4 + j] * mean[1];
                r += cam.view[8 + j] * mean[2];
                r += cam.view[12 + j] * 1.0f;
                t[j] = r;
            }
            const float limx = 1.3f * cam.tan_fovx, limy = 1.3f * cam.tan_fovy;
            const float tz = t[2], inv_tz = 1.0f / tz;
            const float txtz = t[0] * inv_tz, tytz = t[1] * inv_tz;
            const float x_grad_mul = 1.0f - (((txtz < -limx) || (txtz > limx)) ? 1.0f : 0.0f);
            const float y_grad_mul = 1.0f - (((tytz < -limy) || (tytz > limy)) ? 1.0f : 0.0f);
            const float tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
            const float ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
            const float inv_tz2 = inv_tz * inv_tz, inv_tz3 = inv_tz2 * inv_tz;
            const float J00 = h_x * inv_tz, J11 = h_y * inv_tz;
            const float J02 = -h_x * tx * inv_tz2, J12 = -h_y * ty * inv_tz2;
            const M33 J = {{{J00, 0.0f, 0.0f}, {0.0f, J11, 0.0f}, {J02, J12, 0.0f}}}; // transpose of the projection Jacobian
            const M33 Wm = {{{cam.view[0], cam.view[1], cam.view[2]}, {cam.view[4], cam.view[5], cam.view[6]}, {cam.view[8], cam.view[9], cam.view[10]}}};
            const M33 T = mul33(Wm, J);
            const M33 Vrk = {{{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}}};
            const M33 c2 = mul33(mul33(tr33(T), tr33(Vrk)), T);
            const float a = c2.m[0][0] + 0.3f, b = c2.m[0][1], c = c2.m[1][1] + 0.3f;
            const float denom = a * c - b * b;
            float dL_da = 0.0f, dL_db = 0.0f, dL_dc = 0.0f;
            if (denom != 0.0f) {
                const float denom2inv = 1.0f / (denom * denom + 1e-7f);
                dL_da = denom2inv * (-c * c * g_con[0] + 2.0f * b * c * g_con[1] + (denom - a * c) * g_con[2]);
                dL_dc = denom2inv * (-a * a * g_con[2] + 2.0f * a * b * g_con[1] + (denom - a * c) * g_con[0]);
                dL_db = denom2inv * 2.0f * (b * c * g_con[0] - (denom + 2.0f * b * b) * g_con[1] + a * b * g_con[2]);
            }
#define Tm(i, j) T.m[i][j]
#define V(i, j) Vrk.m[i][j]
            dcov[0] = Tm(0,0) * Tm(0,0) * dL_da + Tm(0,0) * Tm(0,1) * dL_db + Tm(0,1) * Tm(0,1) * dL_dc;
            dcov[1] = 2.0f * Tm(0,0) * Tm(1,0) * dL_da + (Tm(0,0) * Tm(1,1) + Tm(1,0) * Tm(0,1)) * dL_db + 2.0f * Tm(0,1) * Tm(1,1) * dL_dc;
            dcov[2] = 2.0f * Tm(0,0) * Tm(2,0) * dL_da + (Tm(0,0) * Tm(2,1) + Tm(2,0) * Tm(0,1)) * dL_db + 2.0f * Tm(0,1) * Tm(2,1) * dL_dc;
            dcov[3] = Tm(1,0) * Tm(1,0) * dL_da + Tm(1,0) * Tm(1,1) * dL_db + Tm(1,1) * Tm(1,1) * dL_dc;
            dcov[4] = 2.0f * Tm(2,0) * Tm(1,0) * dL_da + (Tm(1,0) * Tm(2,1) + Tm(2,0) * Tm(1,1)) * dL_db + 2.0f * Tm(1,1) * Tm(2,1) * dL_dc;
            dcov[5] = Tm(2,0) * Tm(2,0) * dL_da + Tm(2,0) * Tm(2,1) * dL_db + Tm(2,1) * Tm(2,1) * dL_dc;
            const float dL_dT00 = 2.0f * (Tm(0,0) * V(0,0) + Tm(1,0) * V(1,0) + Tm(2,0) * V(2,0)) * dL_da + (Tm(0,1) * V(0,0) + Tm(1,1) * V(1,0) + Tm(2,1) * V(2,0)) * dL_db;
            const float dL_dT01 = 2.0f * (Tm(0,0) * V(0,1) + Tm(1,0) * V(1,1) + Tm(2,0) * V(2,1)) * dL_da + (Tm(0,1) * V(0,1) + Tm(1,1) * V(1,1) + Tm(2,1) * V(2,1)) * dL_db;
            const float dL_dT02 = 2.0f * (Tm(0,0) * V(0,2) + Tm(1,0) * V(1,2) + Tm(2,0) * V(2,2)) * dL_da + (Tm(0,1) * V(0,2) + Tm(1,1) * V(1,2) + Tm(2,1) * V(2,2)) * dL_db;
            const float dL_dT10 = 2.0f * (Tm(0,1) * V(0,0) + Tm(1,1) * V(1,0) + Tm(2,1) * V(2,0)) * dL_dc + (Tm(0,0) * V(0,0) + Tm(1,0) * V(1,0) + Tm(2,0) * V(2,0)) * dL_db;
            const float dL_dT11 = 2.0f * (Tm(0,1) * V(0,1) + Tm(1,1) * V(1,1) + Tm(2,1) * V(2,1)) * dL_dc + (Tm(0,0) * V(0,1) + Tm(1,0) * V(1,1) + Tm(2,0) * V(2,1)) * dL_db;
            const float dL_dT12 = 2.0f * (Tm(0,1) * V(0,2) + Tm(1,1) * V(1,2) + Tm(2,1) * V(2,2)) * dL_dc + (Tm(0,0) * V(0,2) + Tm(1,0) * V(1,2) + Tm(2,0) * V(2,2)) * dL_db;
#undef Tm
#undef V
            const float dL_dJ00 = Wm.m[0][0] * dL_dT00 + Wm.m[1][0] * dL_dT01 + Wm.m[2][0] * dL_dT02;
            const float dL_dJ02 = Wm.m[0][2] * dL_dT00 + Wm.m[1][2] * dL_dT01 + Wm.m[2][2] * dL_dT02;
            const float dL_dJ11 = Wm.m[0][1] * dL_dT10 + Wm.m[1][1] * dL_dT11 + Wm.m[2][1] * dL_dT12;
            const float dL_dJ12 = Wm.m[0][2] * dL_dT10 + Wm.m[1][2] * dL_dT11 + Wm.m[2][2] * dL_dT12;
            const float dL_dtx = -h_x * inv_tz2 * dL_dJ02;
            const float dL_dty = -h_y * inv_tz2 * dL_dJ12;
            const float dL_dtz = -h_x * inv_tz2 * dL_dJ00 - h_y * inv_tz2 * dL_dJ11 + 2.0f * h_x * tx * inv_tz3 * dL_dJ02 + 2.0f * h_y * ty * inv_tz3 * dL_dJ12;
            const float dt[4] = {dL_dtx * x_grad_mul, dL_dty * y_grad_mul, dL_dtz, 1.0f};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float r = cam.view[j * 4 + 0] * dt[0];
                r += cam.view[j * 4 + 1] * dt[1];
                r += cam.view[j * 4 + 2] * dt[2];
                r += cam.view[j * 4 + 3] * dt[3];
                o_mean[j] += r;
            }
        }
        // ---------------- projection backward (backward.py:709-768) ----------------
        {
#define PM(r, c) cam.proj[(r) * 4 + (c)]
            float mw = PM(0, 3) * mean[0];
            mw += PM(1, 3) * mean[1];
            mw += PM(2, 3) * mean[2];
            mw += PM(3, 3) * 1.0f;
            const float m_w = 1.0f / (mw + 0.0000001f);
            const float mul1 = (PM(0,0) * mean[0] + PM(1,0) * mean[1] + PM(2,0) * mean[2] + PM(3,0)) * m_w * m_w;
            const float mul2 = (PM(0,1) * mean[0] + PM(1,1) * mean[1] + PM(2,1) * mean[2] + PM(3,1)) * m_w * m_w;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                o_mean[k] += (PM(k,0) * m_w - PM(k,3) * mul1) * g_m2d[0] + (PM(k,1) * m_w - PM(k,3) * mul2) * g_m2d[1];
#undef PM
        }
    } // vis (phase 1)
    } // in_range (phase 1)

    // phase 2: the SH rows are parked in LDS (block barrier: must sit outside the divergent code)
    float dg[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (DIRGRAD) {
        // through the (still unused) LDS image: 144 float4 in, every lane takes its own nine floats (stride 9: conflict-free)
        float4 *st = lds_wave;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k * 64 + lane < 144) st[k * 64 + lane] = dg4[k];
        wave_lds_fence();
        const float *mine = reinterpret_cast<const float *>(lds_wave) + 9 * lane;
#pragma unroll
        for (int k = 0; k < 9; ++k) dg[k] = mine[k];
        wave_lds_fence(); // all lanes have read before any lane writes its output row over the staging area
    } else {
        sh_rows_commit(sh_regs, lds_wave, lane);
        __syncthreads();
    }
    if (in_range) {
    if (vis) {
        // ---------------- SH backward (backward.py:69-255) ----------------
        {
            const float dir_orig[3] = {mean[0] - cam.campos[0], mean[1] - cam.campos[1], mean[2] - cam.campos[2]};
            const float dir_len = sqrtf(dot3(dir_orig, dir_orig));
            if (!(dir_len < 1e-8f)) {
                const float x = dir_orig[0] / dir_len, y = dir_orig[1] / dir_len, z = dir_orig[2] / dir_len;
                float dRGB[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) dRGB[c] = g_col[c] * (1.0f + (-1.0f * gsr_ld1<GSR_NT_INPUTS != 0>(clamped_state + 3 * idx + c)));
#pragma unroll
                for (int c = 0; c < 3; ++c) o_rgb[c] = dRGB[c];
                // d(colour)/d(direction): nine sums over the coefficients (backward.py:120-244), either handed over by the forward
                // (DIRGRAD) or formed here from the row in LDS -- the same function, the same float operations either way
                float dx_[3] = {dg[0], dg[1], dg[2]}, dy_[3] = {dg[3], dg[4], dg[5]}, dz_[3] = {dg[6], dg[7], dg[8]};
                if constexpr (!DIRGRAD) sh_direction_sums(row, degree, x, y, z, dx_, dy_, dz_);
                // the gradient rows replace the coefficients in place (all reads above are done)
                const float SH_C0 = 0.28209479177387814f, SH_C1 = 0.4886025119029199f;
#define OUT(k, coef)                                                                                                           \
    {                                                                                                                         \
        const float cf = (coef);                                                                                              \
        _Pragma("unroll") for (int c = 0; c < 3; ++c) row[(k) * 3 + c] = cf * dRGB[c];                                         \
    }
                OUT(0, SH_C0);
                sh_written = 1;
                if (degree > 0) {
                    OUT(1, -SH_C1 * y); OUT(2, SH_C1 * z); OUT(3, -SH_C1 * x);
                    sh_written = 4;
                    if (degree > 1) {
                        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        const float C2_0 = 1.0925484305920792f, C2_1 = -1.0925484305920792f, C2_2 = 0.31539156525252005f,
                                    C2_3 = -1.0925484305920792f, C2_4 = 0.5462742152960396f;
                        OUT(4, C2_0 * xy); OUT(5, C2_1 * yz); OUT(6, C2_2 * (2.0f * zz - xx - yy)); OUT(7, C2_3 * xz); OUT(8, C2_4 * (xx - yy));
                        sh_written = 9;
                        if (degree > 2) {
                            const float C3_0 = -0.5900435899266435f, C3_1 = 2.890611442640554f, C3_2 = -0.4570457994644658f,
                                        C3_3 = 0.3731763325901154f, C3_4 = -0.4570457994644658f, C3_5 = 1.445305721320277f,
                                        C3_6 = -0.5900435899266435f;
                            OUT(9, C3_0 * y * (3.0f * xx - yy)); OUT(10, C3_1 * xy * z); OUT(11, C3_2 * y * (4.0f * zz - xx - yy));
                            OUT(12, C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy)); OUT(13, C3_4 * x * (4.0f * zz - xx - yy));
                            OUT(14, C3_5 * z * (xx - yy)); OUT(15, C3_6 * x * (xx - 3.0f * yy));
                            sh_written = 16;
                        }
                    }
                }
#undef OUT
                const float dL_ddir[3] = {dot3(dx_, dRGB), dot3(dy_, dRGB), dot3(dz_, dRGB)};
                // dnormvdv (backward.py:43-64)
                const float *v = dir_orig;
                const float sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
                if (!(sum2 < 1e-10f)) {
                    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
                    o_mean[0] += ((sum2 - v[0] * v[0]) * dL_ddir[0] - v[1] * v[0] * dL_ddir[1] - v[2] * v[0] * dL_ddir[2]) * invsum32;
                    o_mean[1] += (-v[0] * v[1] * dL_ddir[0] + (sum2 - v[1] * v[1]) * dL_ddir[1] - v[2] * v[1] * dL_ddir[2]) * invsum32;
                    o_mean[2] += (-v[0] * v[2] * dL_ddir[0] - v[1] * v[2] * dL_ddir[1] + (sum2 - v[2] * v[2]) * dL_ddir[2]) * invsum32;
                }
            }
        }
        // ---------------- cov3d backward (backward.py:439-556), scale_modifier = 1.0 (Q16) ----------------
        {
            const float scale_modifier = 1.0f;
            const float r = q.w, x = q.x, y = q.y, z = q.z;
            const M33 R = {{{1.0f - 2.0f * (y * y + z * z), 2.0f * (x * y - r * z), 2.0f * (x * z + r * y)},
                            {2.0f * (x * y + r * z), 1.0f - 2.0f * (x * x + z * z), 2.0f * (y * z - r * x)},
                            {2.0f * (x * z - r * y), 2.0f * (y * z + r * x), 1.0f - 2.0f * (x * x + y * y)}}};
            const float s_vec[3] = {scale_modifier * sv[0], scale_modifier * sv[1], scale_modifier * sv[2]};
            const M33 S = {{{s_vec[0], 0.f, 0.f}, {0.f, s_vec[1], 0.f}, {0.f, 0.f, s_vec[2]}}};
            const M33 M = mul33(S, R);
            const M33 dSig = {{{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]}, {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]}, {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}}};
            M33 twoM;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) twoM.m[i][j] = 2.0f * M.m[i][j];
            const M33 dL_dM = mul33(twoM, dSig);
            const M33 Rt = tr33(R), dMt = tr33(dL_dM);
#pragma unroll
            for (int k = 0; k < 3; ++k) o_scale[k] = dot3(Rt.m[k], dMt.m[k]) * scale_modifier;
            float ds[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) ds[i][j] = dMt.m[i][j] * s_vec[i];
            const float dL_dr = 2.0f * (z * (ds[0][1] - ds[1][0]) + y * (ds[2][0] - ds[0][2]) + x * (ds[1][2] - ds[2][1]));
            const float dL_dx = 2.0f * (y * (ds[1][0] + ds[0][1]) + z * (ds[2][0] + ds[0][2]) + r * (ds[1][2] - ds[2][1])) - 4.0f * x * (ds[2][2] + ds[1][1]);
            const float dL_dy = 2.0f * (x * (ds[1][0] + ds[0][1]) + r * (ds[2][0] - ds[0][2]) + z * (ds[1][2] + ds[2][1])) - 4.0f * y * (ds[2][2] + ds[0][0]);
            const float dL_dz = 2.0f * (r * (ds[0][1] - ds[1][0]) + x * (ds[2][0] + ds[0][2]) + y * (ds[1][2] + ds[2][1])) - 4.0f * z * (ds[1][1] + ds[0][0]);
            o_rot[0] = dL_dx; o_rot[1] = dL_dy; o_rot[2] = dL_dz; o_rot[3] = dL_dr;
        }
    }

    gsr_st4<GSR_NT_MISC_STORE != 0>(reinterpret_cast<float4 *>(dL_drot + 4 * idx), make_float4(o_rot[0], o_rot[1], o_rot[2], o_rot[3]));
    if (dL_drgb) {
        if (idx == 0) { // the payload's trailer: where this view was taken from
            dL_drgb[3 * N] = cam.campos[0]; dL_drgb[3 * N + 1] = cam.campos[1]; dL_drgb[3 * N + 2] = cam.campos[2]; dL_drgb[3 * N + 3] = 0.0f;
        }
    }
    // coefficients that got no gradient (culled Gaussian, lower degree) are zero, as in the reference's zero-initialised array
    for (int k = sh_written * 3; k < 48; ++k) row[k] = 0.0f;
    } // in_range
    // the 12-byte-row outputs leave through the pad float4 of the wave's LDS rows as whole float4 lines (sh_stage.h): as
    // per-lane scalar stores each of these arrays cost three instructions over 64 partial lines
    if (rows_valid > 0) {
        if (dL_dcolor) wave_store_vec3_in_pad(dL_dcolor + 3 * wave_row0, lds_wave, lane, rows_valid, g_col[0], g_col[1], g_col[2]);
        if (dL_dmean2D) wave_store_vec3_in_pad(dL_dmean2D + 3 * wave_row0, lds_wave, lane, rows_valid, g_m2d[0], g_m2d[1], 0.0f);
        wave_store_vec3_in_pad(dL_dmean3D + 3 * wave_row0, lds_wave, lane, rows_valid, o_mean[0], o_mean[1], o_mean[2]);
        wave_store_vec3_in_pad(dL_dscale + 3 * wave_row0, lds_wave, lane, rows_valid, o_scale[0], o_scale[1], o_scale[2]);
        if (dL_drgb) wave_store_vec3_in_pad(dL_drgb + 3 * wave_row0, lds_wave, lane, rows_valid, o_rgb[0], o_rgb[1], o_rgb[2]);
    }
    __syncthreads();
    if (dL_dshs && rows_valid > 0) sh_rows_store(reinterpret_cast<float4 *>(dL_dshs) + wave_row0 * 12, lds_wave, lane, rows_valid);
}

// ---- the view payload on its own (gsr_backward_blend): what geom_backward_kernel writes to dL_drgb, from the same inputs
// with the same expressions, so the exchange can start before the per-Gaussian half runs ----
__global__ __launch_bounds__(256) void view_payload_kernel(int64_t N, const float *__restrict__ means, CamK cam, const int32_t *__restrict__ radii,
                                                           const float *__restrict__ clamped_state, const GradRec *__restrict__ acc,
                                                           float *__restrict__ payload)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N) return;
    float o_rgb[3] = {0.f, 0.f, 0.f};
    if (radii[idx] > 0) {
        const float mean[3] = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
        const float dir_orig[3] = {mean[0] - cam.campos[0], mean[1] - cam.campos[1], mean[2] - cam.campos[2]};
        const float dir_len = sqrtf(dot3(dir_orig, dir_orig));
        if (!(dir_len < 1e-8f)) {
            const float4 a0 = *reinterpret_cast<const float4 *>(acc + idx);
            const float g_col[3] = {a0.x, a0.y, a0.z};
#pragma unroll
            for (int c = 0; c < 3; ++c) o_rgb[c] = g_col[c] * (1.0f + (-1.0f * clamped_state[3 * idx + c]));
        }
    }
    payload[3 * idx] = o_rgb[0]; payload[3 * idx + 1] = o_rgb[1]; payload[3 * idx + 2] = o_rgb[2];
    if (idx == 0) { payload[3 * N] = cam.campos[0]; payload[3 * N + 1] = cam.campos[1]; payload[3 * N + 2] = cam.campos[2]; payload[3 * N + 3] = 0.0f; }
}

// ---- SH gradient rebuilt from V views' colour gradients (gsr_sh_grad_from_views) ----
// One lane per Gaussian: per view it normalises the direction once, forms the 16 basis values with the expressions (and
// rounding) of the single-view kernel above, and adds basis_k * drgb_c into 48 accumulators; the finished row goes out
// through the same LDS image as the SH rows everywhere else (12 coalesced 1-KiB stores per wave).
__global__ __launch_bounds__(256) void sh_grad_from_views_kernel(int64_t N, const float *__restrict__ means, int degree, int V, ShViewSet vs,
                                                                 float scale, float *__restrict__ dL_dshs)
{
    __shared__ float4 s_rows[4 * SH_WAVE_F4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave_row0 = (int64_t)blockIdx.x * blockDim.x + wv * 64;
    const int rows_valid = (int)min((int64_t)64, max((int64_t)0, N - wave_row0));
    float4 *lds_wave = s_rows + wv * SH_WAVE_F4;
    const int64_t i = wave_row0 + lane;
    float acc[48];
#pragma unroll
    for (int k = 0; k < 48; ++k) acc[k] = 0.0f;
    if (i < N) {
        const float m[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
        sh_grad_sum_over_views(vs, V, N, i, m, degree, acc);
    }
    float *row = reinterpret_cast<float *>(lds_wave + lane * SH_ROW_F4);
#pragma unroll
    for (int k = 0; k < 48; ++k) row[k] = acc[k] * scale;
    __syncthreads();
    if (rows_valid > 0) sh_rows_store(reinterpret_cast<float4 *>(dL_dshs) + wave_row0 * 12, lds_wave, lane, rows_valid);
}

} // namespace

hipError_t gsr_launch_geom_backward(const GsrScene &sc, const CamK &cam, const GsrGeom &g, const GradRec *acc, const GsrGrads &gr,
                                    hipStream_t s)
{
    if (sc.N <= 0) return hipSuccess;
    // focal lengths come from the host, formed in float64 and rounded once (reference backward.py:1044-1045, quirk Q8)
    const float h_x = cam.focal_x, h_y = cam.focal_y;
#define GEOM_BWD(DG)                                                                                                          \
    hipLaunchKernelGGL(geom_backward_kernel<DG>, dim3((unsigned)gsr_div_up(sc.N, 256)), dim3(256), 0, s, sc.N, sc.means, sc.scales, \
                       sc.rotations, sc.sh, sc.sh_degree, cam, h_x, h_y, g.radii, g.cov3D, g.clamped_state, acc, gr.dL_dmean3D,  \
                       gr.dL_dscale, gr.dL_drot, gr.dL_dopacity, gr.dL_dshs, gr.dL_dcolor, gr.dL_dmean2D, gr.dL_dconic, gr.dL_drgb, \
                       g.sh_dir_grad, sc.scale_modifier)
    if (g.sh_dir_grad) GEOM_BWD(true);
    else GEOM_BWD(false);
#undef GEOM_BWD
    return hipGetLastError();
}

hipError_t gsr_launch_view_payload(const GsrScene &sc, const CamK &cam, const GsrGeom &g, const GradRec *acc, float *payload, hipStream_t s)
{
    if (sc.N <= 0) return hipSuccess;
    hipLaunchKernelGGL(view_payload_kernel, dim3((unsigned)gsr_div_up(sc.N, 256)), dim3(256), 0, s, sc.N, sc.means, cam, g.radii, g.clamped_state,
                       acc, payload);
    return hipGetLastError();
}

extern "C" int gsr_sh_grad_from_views(int64_t N, const float *means, int32_t sh_degree, int32_t V, const float *const *payloads, float scale,
                                      float *dL_dshs, void *stream)
{
    if (N < 0 || V < 1 || V > GSR_MAX_VIEWS || sh_degree < 0 || sh_degree > 3 || N > ((int64_t)1 << 27)) return GSR_E_DIMS;
    if (N == 0) return GSR_OK;
    if (!means || !payloads || !dL_dshs) return GSR_E_NULL;
    if (!gsr_aligned16(dL_dshs)) return GSR_E_ALIGN; // payload rows are read as scalars: rows of a gathered [V][3N+4] block are fine
    ShViewSet vs;
    for (int v = 0; v < GSR_MAX_VIEWS; ++v) {
        vs.payload[v] = v < V ? payloads[v] : nullptr;
        if (v < V && !payloads[v]) return GSR_E_NULL;
    }
    hipLaunchKernelGGL(sh_grad_from_views_kernel, dim3((unsigned)gsr_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, N, means,
                       (int)sh_degree, (int)V, vs, scale, dL_dshs);
    return hipGetLastError() == hipSuccess ? GSR_OK : GSR_E_HIP;
}
