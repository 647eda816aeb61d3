// blend_bwd.hip -- per-tile back-to-front gradient replay for gfx950.
//
// The reference's wp_render_backward_kernel (backward.py:559-706) issues four float-vector atomics
// per (pixel, Gaussian) pair.  Here a workgroup owns a 16x16 tile, replays the tile's list from the
// deepest contributing entry back to the front in LDS-staged batches, and reduces the nine gradient
// components of an entry ON CHIP before anything leaves the CU:
//   lane   : sums its P pixels (same column) in registers
//   wave   : six v_add_f32 DPP steps (row_shr 1,2,4,8 + row_bcast 15,31) leave the wave sum in lane 63
//   tile   : lane 63 of each wave adds into an LDS slot (ds_add_f32)
//   memory : one 36-byte atomic burst per (tile, contributing entry) into a 64-byte-aligned
//            per-Gaussian accumulator record, 16 lanes per record so a wave instruction covers four
//            whole 64-B atomic requests
// Entries that contribute to no pixel of a wave are skipped with one ballot.  Per-pixel arithmetic
// keeps the reference's operation order; only the order of the float sums differs (quirk Q15).
#include "gsr_internal.h"

namespace {

constexpr int BBATCH = 128;

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(moved);
}
// Sum over the 64 lanes; the total is valid in lane 63 only.
__device__ __forceinline__ float wave_sum_lane63(float v)
{
    v = dpp_add<0x111, 0xF>(v); // row_shr:1
    v = dpp_add<0x112, 0xF>(v); // row_shr:2
    v = dpp_add<0x114, 0xF>(v); // row_shr:4
    v = dpp_add<0x118, 0xF>(v); // row_shr:8   -> lane 15 of each row = row sum
    v = dpp_add<0x142, 0xA>(v); // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xC>(v); // row_bcast:31 into rows 2,3 -> lane 63 = wave sum
    return v;
}

template <int P>
__global__ __launch_bounds__(256 / P) void blend_backward_kernel(int W, int H, int grid_x, float bg0, float bg1, float bg2,
                                                                 const int32_t *__restrict__ ranges,
                                                                 const int32_t *__restrict__ point_list,
                                                                 const BlendRec *__restrict__ rec,
                                                                 const float *__restrict__ final_T,
                                                                 const int32_t *__restrict__ n_contrib,
                                                                 const float *__restrict__ dL_dpixels, GradRec *__restrict__ acc)
{
    constexpr int NT = 256 / P;
    constexpr int NW = NT / 64;
    constexpr int ROWS = 16 / P;
    __shared__ float4 s_a[BBATCH];       // xy.x xy.y con.a con.b
    __shared__ float4 s_b[BBATCH];       // con.c opacity r g
    __shared__ float s_c[BBATCH];        // b
    __shared__ int s_id[BBATCH];
    __shared__ float s_part[BBATCH][12]; // per-entry tile sums (9 used)
    __shared__ int s_any[BBATCH];
    __shared__ int s_max;

    const int tid = threadIdx.x, lane = tid & 63;
    const int tile = blockIdx.x;
    const int tile_x = tile % grid_x, tile_y = tile / grid_x;
    const int pix_x = tile_x * 16 + (tid & 15);
    const int row0 = tile_y * 16 + (tid >> 4);
    const float pixf_x = (float)pix_x;
    const int2 range = *reinterpret_cast<const int2 *>(ranges + 2 * tile);
    const int start = range.x, end = range.y;
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;

    float pixf_y[P], T[P], Tfin[P], ar[P][3], lc[P][3], la[P], dp[P][3], bgdot[P];
    int kept[P]; // exclusive upper bound of list indices this pixel replays
    int my_max = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int py = row0 + p * ROWS;
        pixf_y[p] = (float)py;
        kept[p] = start;
        T[p] = Tfin[p] = 0.0f;
        la[p] = 0.0f; bgdot[p] = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { ar[p][c] = 0.0f; lc[p][c] = 0.0f; dp[p][c] = 0.0f; }
        if (pix_x < W && py < H) {
            const size_t px = (size_t)py * W + pix_x;
            Tfin[p] = T[p] = final_T[px];
            const int nc = n_contrib[px];
            kept[p] = min(end, start + nc);
            my_max = max(my_max, kept[p]);
            dp[p][0] = dL_dpixels[3 * px]; dp[p][1] = dL_dpixels[3 * px + 1]; dp[p][2] = dL_dpixels[3 * px + 2];
            float t = bg0 * dp[p][0];
            t += bg1 * dp[p][1];
            t += bg2 * dp[p][2];
            bgdot[p] = t;
        }
    }
    if (tid == 0) s_max = start;
    __syncthreads();
    atomicMax(&s_max, my_max);
    __syncthreads();
    const int hi_all = s_max;

    for (int hi = hi_all; hi > start; hi -= BBATCH) {
        const int lo = max(start, hi - BBATCH);
        const int cnt = hi - lo;
        // stage entries [lo, hi) ; slot k holds list index hi-1-k (replay order)
        for (int k = tid; k < cnt; k += NT) {
            const int id = point_list[hi - 1 - k];
            const float4 *rp = reinterpret_cast<const float4 *>(rec + id);
            const float4 a = rp[0], b = rp[1], c = rp[2];
            s_a[k] = a;
            s_b[k] = b;
            s_c[k] = c.x;
            s_id[k] = id;
            s_any[k] = 0;
#pragma unroll
            for (int c9 = 0; c9 < 9; ++c9) s_part[k][c9] = 0.0f;
        }
        __syncthreads();

        bool wave_live = false;
#pragma unroll
        for (int p = 0; p < P; ++p) wave_live = wave_live || (kept[p] > lo);
        if (__any(wave_live)) {
            for (int k = 0; k < cnt; ++k) {
                const int idx = hi - 1 - k;
                const float4 a = s_a[k];
                const float4 b = s_b[k];
                const float colb = s_c[k];
                const float col[3] = {b.z, b.w, colb};
                const float d_x = a.x - pixf_x;
                const float axx = a.z * d_x * d_x;
                const float bdx = a.w * d_x;
                float g_col[3] = {0.f, 0.f, 0.f}, g_mx = 0.f, g_my = 0.f, g_ca = 0.f, g_cb = 0.f, g_cc = 0.f, g_op = 0.f;
                bool any = false;
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const float d_y = a.y - pixf_y[p];
                    const float power = -0.5f * (axx + b.x * d_y * d_y) - bdx * d_y;
                    const float G = fast_exp(power);
                    const float alpha = fminf(0.99f, b.y * G);
                    const bool live = (idx < kept[p]) && !(power > 0.0f) && !(alpha < (1.0f / 255.0f));
                    if (live) {
                        any = true;
                        const float Tn = T[p] / (1.0f - alpha);
                        T[p] = Tn;
                        const float dchannel_dcolor = alpha * Tn;
                        float dL_dalpha;
                        {
                            float diff[3];
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                ar[p][c] = la[p] * lc[p][c] + (1.0f - la[p]) * ar[p][c];
                                lc[p][c] = col[c];
                                diff[c] = col[c] - ar[p][c];
                                g_col[c] += dchannel_dcolor * dp[p][c];
                            }
                            float t = diff[0] * dp[p][0];
                            t += diff[1] * dp[p][1];
                            t += diff[2] * dp[p][2];
                            dL_dalpha = t;
                        }
                        dL_dalpha *= Tn;
                        la[p] = alpha;
                        dL_dalpha += (-Tfin[p] / (1.0f - alpha)) * bgdot[p];
                        const float dL_dG = b.y * dL_dalpha;
                        const float gdx = G * d_x, gdy = G * d_y;
                        const float dG_ddelx = -gdx * a.z - gdy * a.w;
                        const float dG_ddely = -gdy * b.x - gdx * a.w;
                        g_mx += dL_dG * dG_ddelx * ddelx_dx;
                        g_my += dL_dG * dG_ddely * ddely_dy;
                        g_ca += -0.5f * gdx * d_x * dL_dG;
                        g_cb += -0.5f * gdx * d_y * dL_dG;
                        g_cc += -0.5f * gdy * d_y * dL_dG;
                        g_op += G * dL_dalpha;
                    }
                }
                if (__any(any)) { // wave-uniform: at least one lane contributed
                    float v[9] = {g_col[0], g_col[1], g_col[2], g_mx, g_my, g_ca, g_cb, g_cc, g_op};
#pragma unroll
                    for (int c = 0; c < 9; ++c) v[c] = wave_sum_lane63(v[c]);
                    if (lane == 63) {
                        if (NW == 1) {
#pragma unroll
                            for (int c = 0; c < 9; ++c) s_part[k][c] = v[c];
                        } else {
#pragma unroll
                            for (int c = 0; c < 9; ++c) atomicAdd(&s_part[k][c], v[c]);
                        }
                        s_any[k] = 1;
                    }
                }
            }
        }
        __syncthreads();
        // flush: 16 lanes per entry -> one 64-byte accumulator record each
        for (int q = tid; q < cnt * 16; q += NT) {
            const int e = q >> 4, c = q & 15;
            if (c < 9 && s_any[e]) unsafeAtomicAdd(&acc[s_id[e]].f[c], s_part[e][c]);
        }
        __syncthreads();
    }
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

int gsr_bwd_p_override = 4;
int gsr_bwd_mode = 0;

hipError_t gsr_launch_pack_records(const GsrGeom &g, BlendRec *rec, int64_t N, hipStream_t s)
{
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_records_kernel, dim3((unsigned)gsr_div_up(N, 256)), dim3(256), 0, s, g.xy, g.conic_opacity, g.rgb,
                       g.depths, rec, N);
    return hipGetLastError();
}

hipError_t gsr_launch_blend_backward(const CamK &cam, const int32_t *ranges, const int32_t *point_list, const BlendRec *rec,
                                     const GsrImage &img, const float *dL_dpixels, GradRec *acc, hipStream_t s)
{
    const int tiles = cam.grid_x * cam.grid_y;
    if (tiles <= 0) return hipSuccess;
    int P = gsr_bwd_p_override;
    if (P != 1 && P != 2 && P != 4) P = 4;
#define LAUNCH(PP)                                                                                                            \
    hipLaunchKernelGGL(blend_backward_kernel<PP>, dim3(tiles), dim3(256 / PP), 0, s, cam.W, cam.H, cam.grid_x, cam.bg[0],     \
                       cam.bg[1], cam.bg[2], ranges, point_list, rec, img.final_T, img.n_contrib, dL_dpixels, acc)
    if (P == 1) LAUNCH(1);
    else if (P == 2) LAUNCH(2);
    else LAUNCH(4);
#undef LAUNCH
    return hipGetLastError();
}
