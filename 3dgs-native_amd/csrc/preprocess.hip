// preprocess.hip -- per-Gaussian projection stage for gfx950.
//
// Computes what the reference's wp_preprocess does (forward.py:190-382: near cull, clip-space
// projection, Sigma3D = R S S^T R^T, EWA Sigma2D, conic, 3-sigma radius, tile rectangle, SH colour)
// with float operations in the same order as the CPU oracle (compiled with -ffp-contract=off), so the
// integer outputs (radii, tile counts, depth bits -> sort order) agree exactly.  On top of the
// reference's outputs it emits three internal products for the later stages:
//   rec[i]        64-byte blend record (xy, conic, opacity, rgb, 1/depth): one sector per gather
//   rect[i]       packed tile rectangle (so key duplication does not redo the float math)
//   depth_item[i] (depth bits << 32 | i), the item the depth sort works on
// One thread per Gaussian, 256 per workgroup; every output is written for every i (zeros for culled
// Gaussians, quirk Q11), so no buffer needs pre-zeroing.
#include "gsr_internal.h"
#include "sh_stage.h"
#include "fwd_order.h"
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
// (p,1) * M under the row-vector convention, rows accumulated in ascending order.
__device__ __forceinline__ void rowvec_mul44(float px, float py, float pz, const float *M, float out[4])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float r = M[j] * px;
        r += M[4 + j] * py;
        r += M[8 + j] * pz;
        r += M[12 + j] * 1.0f;
        out[j] = r;
    }
}
__device__ __forceinline__ float ndc2pix(float x, float size) { return ((x + 1.0f) * size - 1.0f) * 0.5f; }

__device__ __forceinline__ int f2i(float v) { return (int)v; } // v_cvt_i32_f32: truncation

__global__ __launch_bounds__(256) void preprocess_kernel(
    int64_t N, const float *__restrict__ means, const float *__restrict__ scales, const float *__restrict__ rots,
    const float *__restrict__ opac, const float *__restrict__ shs, int degree, int clamped, float scale_mod, CamK cam,
    int32_t *__restrict__ radii, float *__restrict__ xy, float *__restrict__ depths, float *__restrict__ cov3Ds,
    float *__restrict__ rgb, float *__restrict__ conic_opacity, int32_t *__restrict__ tiles_touched,
    float *__restrict__ clamped_state, BlendRec *__restrict__ rec, TileRect *__restrict__ rect,
    uint64_t *__restrict__ depth_item, int32_t *__restrict__ zero_acc, int zero_n, int32_t *__restrict__ block_tile_sums,
    float *__restrict__ sh_dir_grad, uint32_t *__restrict__ blk_minmax, int dbg, const int32_t *__restrict__ fwd_cost,
    int32_t *__restrict__ fwd_order, int n_tiles)
{
    const unsigned nblk = gridDim.x - (fwd_order ? 1u : 0u); // the workgroups that hold Gaussians
    __shared__ int s_tiles[4];
    __shared__ uint32_t s_dmin[4], s_dmax[4], s_dvis[4];
    // SH rows are fetched wave-cooperatively (coalesced) into LDS while the geometry math runs
    __shared__ float4 s_rows[4 * SH_WAVE_F4];
    // the spare workgroup behind the Gaussians' (launched when the forward blend wants a tile order): see fwd_order.h
    if (fwd_order && blockIdx.x == nblk) {
        fwd_order_block(fwd_cost, fwd_order, n_tiles, reinterpret_cast<int *>(s_rows));
        return;
    }
    // the accumulators of the first depth-sort pass (scan_sort.hip, radix_hist_kernel) are cleared here: saves a memset launch
    for (int64_t z = (int64_t)blockIdx.x * 256 + threadIdx.x; z < zero_n; z += (int64_t)nblk * 256) zero_acc[z] = 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave_row0 = (int64_t)blockIdx.x * blockDim.x + wv * 64;
    float4 *lds_wave = s_rows + wv * SH_WAVE_F4;
    const int64_t i = min(wave_row0 + lane, N - 1); // tail lanes redo the last Gaussian and store nothing
    const bool in_range = wave_row0 + lane < N;

    // outputs, defaulting to the culled values
    int o_radius = 0, o_tiles = 0;
    float o_xy[2] = {0.0f, 0.0f}, o_depth = 0.0f, o_cov[6] = {0, 0, 0, 0, 0, 0}, o_rgb[3] = {0, 0, 0};
    float o_con[4] = {0, 0, 0, 0}, o_cl[3] = {0, 0, 0};
    float o_dg[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; // d(colour)/d(direction) for the backward (GsrGeom.sh_dir_grad), when asked for
    TileRect o_rect = {0, 0, 0, 0};
    bool visible = false, need_sh = false;

    constexpr bool NTI = GSR_NT_INPUTS != 0, NTM = GSR_NT_MISC_STORE != 0;
    const float px = gsr_ld1<NTI>(means + 3 * i), py = gsr_ld1<NTI>(means + 3 * i + 1), pz = gsr_ld1<NTI>(means + 3 * i + 2);
    float p_view[4], p_hom[4];
    rowvec_mul44(px, py, pz, cam.view, p_view);
    rowvec_mul44(px, py, pz, cam.proj, p_hom);
    const float p_w = 1.0f / (p_hom[3] + 0.0000001f);
    const float ndc_x = p_hom[0] * p_w, ndc_y = p_hom[1] * p_w;
    // SH rows are fetched (under the covariance math) only for Gaussians that are likely to be drawn: in front of the
    // near plane with the centre within 1.1x the image.  The rare visible Gaussian that fails this guess (a big one
    // centred far outside) gets its row in a second, late fetch below, so the guess only affects speed.
    const bool likely = in_range && !(p_view[2] < 0.2f) && fabsf(ndc_x) <= 1.1f && fabsf(ndc_y) <= 1.1f;
    const unsigned long long early_mask = GSR_ABL(dbg, 4) ? 0ull : __ballot(likely);
    ShRegs sh_regs;
    sh_rows_fetch(reinterpret_cast<const float4 *>(shs) + wave_row0 * 12, sh_regs, lane, early_mask);
    if (!(p_view[2] < 0.2f)) {

        // Sigma3D = (R S)(R S)^T, R's columns = quat_rotate(q, e_c)
        // Scale, rotation AND opacity are requested here, together: the opacity is only used behind the rectangle test further
        // down, where its load was a third memory round trip behind the first two (the kernel ran 44 us with neither the SH
        // fetch nor any store, i.e. on latency): 87 -> 82 us at C3, 312 -> 304 us at C5.  Hoisting all three above the near-plane
        // test as well measured the same, and would read 32 B for every Gaussian behind the camera.
        float sc_x = gsr_ld1<NTI>(scales + 3 * i), sc_y = gsr_ld1<NTI>(scales + 3 * i + 1), sc_z = gsr_ld1<NTI>(scales + 3 * i + 2);
        float4 q = gsr_ld4<NTI>(reinterpret_cast<const float4 *>(rots + 4 * i));
        float opacity_i = gsr_ld1<NTI>(opac + i);
        asm volatile("" : "+v"(sc_x), "+v"(sc_y), "+v"(sc_z), "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w), "+v"(opacity_i)); // not to be sunk back
        const float sx = scale_mod * sc_x, sy = scale_mod * sc_y, sz = scale_mod * sc_z;
        gsr_sigma3d(sx, sy, sz, q, o_cov); // (sigma3d.h: shared with the geometry backward, which may recompute it)

        // EWA projection.  T = J * W with W = view[0:3,0:3] as stored (quirk Q1: forward convention).
        float t0, t1;
        const float t2 = p_view[2];
        {
            const float limx = 1.3f * cam.tan_fovx, limy = 1.3f * cam.tan_fovy;
            const float txtz = p_view[0] / t2, tytz = p_view[1] / t2;
            t0 = fminf(limx, fmaxf(-limx, txtz)) * t2;
            t1 = fminf(limy, fmaxf(-limy, tytz)) * t2;
        }
        const float Wf = (float)cam.W, Hf = (float)cam.H;
        const float focal_x = Wf / (2.0f * cam.tan_fovx), focal_y = Hf / (2.0f * cam.tan_fovy);
        const M33 J = {{{focal_x / t2, 0.0f, -(focal_x * t0) / (t2 * t2)}, {0.0f, focal_y / t2, -(focal_y * t1) / (t2 * t2)}, {0.0f, 0.0f, 0.0f}}};
        const M33 Wm = {{{cam.view[0], cam.view[1], cam.view[2]}, {cam.view[4], cam.view[5], cam.view[6]}, {cam.view[8], cam.view[9], cam.view[10]}}};
        const M33 T = mul33(J, Wm);
        const M33 Vrk = {{{o_cov[0], o_cov[1], o_cov[2]}, {o_cov[1], o_cov[3], o_cov[4]}, {o_cov[2], o_cov[4], o_cov[5]}}};
        const M33 c2 = mul33(mul33(T, tr33(Vrk)), tr33(T));

        const float cb0 = c2.m[0][0] + 0.3f, cb1 = c2.m[0][1], cb2 = c2.m[1][1] + 0.3f;
        const float det = cb0 * cb2 - cb1 * cb1;
        if (det != 0.0f) {
            const float det_inv = 1.0f / det;
            const float mid = 0.5f * (cb0 + cb2);
            const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lambda1 = mid + sq, lambda2 = mid - sq;
            const float my_radius = ceilf(3.0f * sqrtf(fmaxf(lambda1, lambda2)));
            const float pim_x = ndc2pix(ndc_x, Wf), pim_y = ndc2pix(ndc_y, Hf);
            const int rx0 = min(cam.grid_x, max(0, f2i((pim_x - my_radius) / 16.0f)));
            const int ry0 = min(cam.grid_y, max(0, f2i((pim_y - my_radius) / 16.0f)));
            const int rx1 = min(cam.grid_x, max(0, f2i((pim_x + my_radius + 16.0f - 1.0f) / 16.0f)));
            const int ry1 = min(cam.grid_y, max(0, f2i((pim_y + my_radius + 16.0f - 1.0f) / 16.0f)));
            const int tiles = (ry1 - ry0) * (rx1 - rx0);
            need_sh = tiles != 0;
            if (tiles != 0) {
                visible = true;
                o_depth = p_view[2];
                o_radius = f2i(my_radius);
                o_xy[0] = pim_x; o_xy[1] = pim_y;
                o_con[0] = cb2 * det_inv; o_con[1] = -cb1 * det_inv; o_con[2] = cb0 * det_inv; o_con[3] = opacity_i;
                o_tiles = tiles;
                o_rect.x0 = (uint16_t)rx0; o_rect.y0 = (uint16_t)ry0; o_rect.x1 = (uint16_t)rx1; o_rect.y1 = (uint16_t)ry1;
            }
        }
    }

    // the block's tile count, for the id-order scan (scan_sort.hip: its reduce pass is this)
    {
        int t = in_range ? o_tiles : 0; // tail lanes redo the last Gaussian
        // DPP adds (no LDS round trips, unlike six ds_bpermute steps): lane 63 ends up with the wave's sum
        t += __builtin_amdgcn_update_dpp(0, t, 0x111, 0xF, 0xF, false); // row_shr:1
        t += __builtin_amdgcn_update_dpp(0, t, 0x112, 0xF, 0xF, false); // row_shr:2
        t += __builtin_amdgcn_update_dpp(0, t, 0x114, 0xF, 0xF, false); // row_shr:4
        t += __builtin_amdgcn_update_dpp(0, t, 0x118, 0xF, 0xF, false); // row_shr:8
        t += __builtin_amdgcn_update_dpp(0, t, 0x142, 0xA, 0xF, false); // row_bcast:15 into rows 1 and 3
        t += __builtin_amdgcn_update_dpp(0, t, 0x143, 0xC, 0xF, false); // row_bcast:31 into rows 2 and 3
        if (lane == 63) s_tiles[wv] = t;
        // the block's smallest and largest VISIBLE depth bits, for the depth sort's pass plan (scan_sort.hip, DepthCtl)
        uint32_t lo = (visible && in_range) ? __float_as_uint(o_depth) : 0xFFFFFFFFu, hi = (visible && in_range) ? __float_as_uint(o_depth) : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo = min(lo, (uint32_t)__shfl_xor((int)lo, d, 64));
            hi = max(hi, (uint32_t)__shfl_xor((int)hi, d, 64));
        }
        const unsigned long long vis_bits = __ballot(visible && in_range);
        if (lane == 0) { s_dmin[wv] = lo; s_dmax[wv] = hi; s_dvis[wv] = (uint32_t)__popcll(vis_bits); }
    }
    sh_rows_commit(sh_regs, lds_wave, lane);
    // in_range: tail lanes redo the last Gaussian; their rows do not exist (found by tests/test_gpu_fuzz.py: N = 1, one big
    // splat centred outside the frustum -> 63 rows read past the end of the SH array)
    const unsigned long long late_mask = GSR_ABL(dbg, 4) ? 0ull : (__ballot(need_sh && in_range) & ~early_mask);
    if (late_mask) { // wave-uniform and rare
        sh_rows_fetch(reinterpret_cast<const float4 *>(shs) + wave_row0 * 12, sh_regs, lane, late_mask);
        sh_rows_commit_masked(sh_regs, lds_wave, lane, late_mask);
    }
    __syncthreads(); // SH rows have landed in LDS
    if (threadIdx.x == 0) {
        block_tile_sums[blockIdx.x] = s_tiles[0] + s_tiles[1] + s_tiles[2] + s_tiles[3];
        reinterpret_cast<uint4 *>(blk_minmax)[blockIdx.x] = make_uint4(min(min(s_dmin[0], s_dmin[1]), min(s_dmin[2], s_dmin[3])),
                                                                       max(max(s_dmax[0], s_dmax[1]), max(s_dmax[2], s_dmax[3])),
                                                                       s_dvis[0] + s_dvis[1] + s_dvis[2] + s_dvis[3], 0u);
        // the slot behind the last block is the frame's DepthCtlRaw (scan_sort.hip): cleared here, combined into by the scan's launch
        if (blockIdx.x == 0) reinterpret_cast<uint4 *>(blk_minmax)[nblk] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (need_sh) {
                // SH colour (forward.py:304-372), stride 16 coefficients per Gaussian
                const float dx = px - cam.campos[0], dy = py - cam.campos[1], dz = pz - cam.campos[2];
                float l2 = dx * dx;
                l2 += dy * dy;
                l2 += dz * dz;
                const float len = sqrtf(l2);
                float x = 0.0f, y = 0.0f, z = 0.0f;
                if (len > 0.0f) { x = dx / len; y = dy / len; z = dz / len; }
                const float *sh = reinterpret_cast<const float *>(lds_wave + lane * SH_ROW_F4);
                const float SH_C0 = 0.28209479177387814f, SH_C1 = 0.4886025119029199f;
                const float xx = x * x, yy = y * y, zz = z * z, xy_ = x * y, yz = y * z, xz = x * z;
                if (sh_dir_grad) {
                    // the nine sums the SH backward would form from these same 48 coefficients (sh_stage.h): 36 bytes out here
                    // save geom_backward_kernel 192 bytes in.  (x, y, z) are the backward's too: it divides the same
                    // differences by the same length, and skips the Gaussian where that length is below 1e-8.
                    float gx[3] = {0.f, 0.f, 0.f}, gy[3] = {0.f, 0.f, 0.f}, gz[3] = {0.f, 0.f, 0.f};
                    sh_direction_sums(sh, degree, x, y, z, gx, gy, gz);
#pragma unroll
                    for (int c = 0; c < 3; ++c) { o_dg[c] = gx[c]; o_dg[3 + c] = gy[c]; o_dg[6 + c] = gz[c]; }
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
#define SHC(k) sh[(k) * 3 + c]
                    float r = SH_C0 * SHC(0);
                    if (degree > 0) {
                        r = r - SH_C1 * y * SHC(1) + SH_C1 * z * SHC(2) - SH_C1 * x * SHC(3);
                        if (degree > 1) {
                            r = r + 1.0925484305920792f * xy_ * SHC(4);
                            r = r + (-1.0925484305920792f) * yz * SHC(5);
                            r = r + 0.31539156525252005f * (2.0f * zz - xx - yy) * SHC(6);
                            r = r + (-1.0925484305920792f) * xz * SHC(7);
                            r = r + 0.5462742152960396f * (xx - yy) * SHC(8);
                            if (degree > 2) {
                                r = r + (-0.5900435899266435f) * y * (3.0f * xx - yy) * SHC(9);
                                r = r + 2.890611442640554f * xy_ * z * SHC(10);
                                r = r + (-0.4570457994644658f) * y * (4.0f * zz - xx - yy) * SHC(11);
                                r = r + 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SHC(12);
                                r = r + (-0.4570457994644658f) * x * (4.0f * zz - xx - yy) * SHC(13);
                                r = r + 1.445305721320277f * z * (xx - yy) * SHC(14);
                                r = r + (-0.5900435899266435f) * x * (xx - 3.0f * yy) * SHC(15);
                            }
                        }
                    }
#undef SHC
                    r = r + 0.5f;
                    o_cl[c] = r < 0.0f ? 1.0f : 0.0f;
                    if (clamped) r = fmaxf(r, 0.0f);
                    o_rgb[c] = r;
                }
    }
    // ---- outputs.  Everything that is a whole 4-, 8- or 16-byte element per Gaussian is one coalesced store per array; the
    // 12-, 24- and 64-byte rows (colour, clamp flags, Sigma3D, blend record) go out through the wave's LDS area -- the SH image
    // is dead by now -- as whole float4 lines (sh_stage.h, wave_store_rows): as per-lane strided stores they were 9 of the
    // kernel's 17 store instructions, each touching 64 partial lines.
    const int rows_valid = (int)min((int64_t)64, N - wave_row0); // > 0: the grid covers exactly ceil(N / 256) blocks... a wave past N has <= 0
    float *stage = reinterpret_cast<float *>(lds_wave);
    if (rows_valid > 0 && !GSR_ABL(dbg, 8)) {
        const float v_rgb[3] = {o_rgb[0], o_rgb[1], o_rgb[2]}, v_cl[3] = {o_cl[0], o_cl[1], o_cl[2]};
        if (rgb) wave_store_rows<3>(rgb + 3 * wave_row0, stage, lane, rows_valid, v_rgb); // (absent: the caller reads the record's columns)
        wave_store_rows<3>(clamped_state + 3 * wave_row0, stage + 256, lane, rows_valid, v_cl);
        wave_store_rows<6>(cov3Ds + 6 * wave_row0, stage + 512, lane, rows_valid, o_cov);
        const float inv_depth = visible ? 1.0f / o_depth : 0.0f;
        const float v_rec[16] = {o_xy[0], o_xy[1], o_con[0], o_con[1], o_con[2], o_con[3], o_rgb[0], o_rgb[1], o_rgb[2], inv_depth, 0.0f, 0.0f,
                                 0.0f, 0.0f, 0.0f, 0.0f};
        wave_store_rows<16>(reinterpret_cast<float *>(rec + wave_row0), stage + 1024, lane, rows_valid, v_rec);
        if (sh_dir_grad) wave_store_rows<9>(sh_dir_grad + 9 * wave_row0, stage + 2048, lane, rows_valid, o_dg);
    }
    if (!in_range) return;
    if (GSR_ABL(dbg, 8)) { if (o_rgb[0] + o_cov[0] + o_con[0] + o_xy[0] + o_depth == 123.456f) radii[i] = 1; } else {
    gsr_st1i<NTM>(radii + i, o_radius);
    tiles_touched[i] = o_tiles;
    if (xy) *reinterpret_cast<float2 *>(xy + 2 * i) = make_float2(o_xy[0], o_xy[1]);
    gsr_st1<NTM>(depths + i, o_depth);
    if (conic_opacity) *reinterpret_cast<float4 *>(conic_opacity + 4 * i) = make_float4(o_con[0], o_con[1], o_con[2], o_con[3]);
    }

    // internal products
    rect[i] = o_rect;
    const uint32_t dbits = visible ? __float_as_uint(o_depth) : 0xFFFFFFFFu;
    depth_item[i] = ((uint64_t)dbits << 32) | (uint64_t)(uint32_t)i;
}

} // namespace

hipError_t gsr_launch_preprocess(const GsrScene &sc, const CamK &cam, const GsrGeom &g, const GeomWs &ws, hipStream_t s, bool make_fwd_order)
{
    if (sc.N == 0) return hipSuccess;
    const int threads = 256;
    const unsigned blocks = (unsigned)gsr_div_up(sc.N, threads) + (make_fwd_order ? 1u : 0u);
    hipLaunchKernelGGL(preprocess_kernel, dim3(blocks), dim3(threads), 0, s, sc.N, sc.means, sc.scales, sc.rotations,
                       sc.opacity, sc.sh, sc.sh_degree, sc.clamped, sc.scale_modifier, cam, g.radii, g.xy, g.depths,
                       g.cov3D, g.rgb, g.conic_opacity, g.tiles_touched, g.clamped_state,
                       g.blend_records ? (BlendRec *)g.blend_records : ws.rec /* the caller's record buffer, else the workspace's */, ws.rect, ws.depth_item, ws.acc[0],
                       3 * (int)gsr_radix_acc_ints(sc.N) /* acc[0], acc[1] (the depth sort may start at a later pass) and acc_first */, ws.scan_tmp, g.sh_dir_grad,
                       ws.blk_minmax, gsr_debug_flags, ws.fwd_cost, make_fwd_order ? ws.fwd_order : nullptr, cam.grid_x * cam.grid_y);
    return hipGetLastError();
}
