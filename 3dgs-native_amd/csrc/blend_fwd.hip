// blend_fwd.hip -- per-tile front-to-back alpha blending for gfx950.
//
// What the reference's wp_render_gaussians does per pixel (forward.py:385-515), restructured for
// CDNA4: one workgroup per 16x16 tile; the tile's sorted list is staged through LDS in batches of
// 256 entries, each entry gathered ONCE per tile as a single 64-byte record (the reference gathers
// four arrays per pixel per entry); every lane then reads the staged entry as an LDS broadcast.
// A lane owns P pixels of one column (P = 1, 2 or 4), which shares dx, a*dx*dx and b*dx between them
// and divides the LDS broadcast traffic by P.  Wave ballots end a wave's work as soon as its 64 lanes
// are saturated and __syncthreads_and ends the tile.
//
// Float operations are in the reference's order (no contraction) so the discrete tests (power > 0,
// alpha < 1/255, T < 1e-4) agree with the CPU oracle except where exp() itself rounds differently:
// exp is v_exp_f32(power * log2 e), relative error < 5e-7 for power in [-5.6, 0].
#include "gsr_internal.h"

namespace {

constexpr int BATCH = 256;

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

template <int P>
__global__ __launch_bounds__(256 / P) void blend_forward_kernel(int W, int H, int grid_x, float bg0, float bg1, float bg2,
                                                                const int32_t *__restrict__ ranges,
                                                                const int32_t *__restrict__ point_list,
                                                                const BlendRec *__restrict__ rec, float *__restrict__ image,
                                                                float *__restrict__ inv_depth, float *__restrict__ final_T,
                                                                int32_t *__restrict__ n_contrib)
{
    constexpr int NT = 256 / P;  // threads per tile
    constexpr int ROWS = 16 / P; // pixel rows covered by one sweep of the threads
    __shared__ float4 s_a[BATCH]; // xy.x xy.y con.a con.b
    __shared__ float4 s_b[BATCH]; // con.c opacity r g
    __shared__ float2 s_c[BATCH]; // b 1/depth

    const int tid = threadIdx.x;
    const int tile = blockIdx.x;
    const int tile_x = tile % grid_x, tile_y = tile / grid_x;
    const int pix_x = tile_x * 16 + (tid & 15);
    const int row0 = tile_y * 16 + (tid >> 4);
    const float pixf_x = (float)pix_x;

    const int2 range = *reinterpret_cast<const int2 *>(ranges + 2 * tile);
    const int start = range.x, end = range.y;

    float pixf_y[P], T[P], cr[P], cg[P], cb[P], cd[P];
    int last[P];
    bool done[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int py = row0 + p * ROWS;
        pixf_y[p] = (float)py;
        T[p] = 1.0f; cr[p] = cg[p] = cb[p] = cd[p] = 0.0f;
        last[p] = 0;
        done[p] = !(pix_x < W && py < H);
    }

    for (int base = start; base < end; base += BATCH) {
        bool mine_done = true;
#pragma unroll
        for (int p = 0; p < P; ++p) mine_done = mine_done && done[p];
        if (__syncthreads_and(mine_done)) break; // whole tile saturated (also fences LDS reuse)

        const int cnt = min(BATCH, end - base);
        for (int k = tid; k < cnt; k += NT) {
            const int id = point_list[base + k];
            const float4 *rp = reinterpret_cast<const float4 *>(rec + id);
            const float4 a = rp[0], b = rp[1], c = rp[2];
            s_a[k] = a;
            s_b[k] = b;
            s_c[k] = make_float2(c.x, c.y);
        }
        __syncthreads();

        if (__all(mine_done)) continue; // this wave has nothing left; keep serving the barriers

        for (int j = 0; j < cnt; ++j) {
            const float4 a = s_a[j];
            const float4 b = s_b[j];
            const float2 c = s_c[j];
            const float dx = a.x - pixf_x;
            const float axx = a.z * dx * dx;
            const float bdx = a.w * dx;
            const int contributor = base - start + j + 1;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const float dy = a.y - pixf_y[p];
                const float power = -0.5f * (axx + b.x * dy * dy) - bdx * dy;
                const float alpha = fminf(0.99f, b.y * fast_exp(power));
                const float test_T = T[p] * (1.0f - alpha);
                const bool live = !done[p] && !(power > 0.0f) && !(alpha < (1.0f / 255.0f));
                const bool sat = live && (test_T < 0.0001f);
                const bool acc = live && !sat;
                done[p] = done[p] || sat;
                if (acc) {
                    cr[p] += b.z * alpha * T[p];
                    cg[p] += b.w * alpha * T[p];
                    cb[p] += c.x * alpha * T[p];
                    cd[p] += c.y * alpha * T[p];
                    T[p] = test_T;
                    last[p] = contributor;
                }
            }
            if ((j & 15) == 15) {
                bool d = true;
#pragma unroll
                for (int p = 0; p < P; ++p) d = d && done[p];
                if (__all(d)) break;
            }
        }
    }

#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int py = row0 + p * ROWS;
        if (pix_x < W && py < H) {
            const size_t px = (size_t)py * W + pix_x;
            final_T[px] = T[p];
            n_contrib[px] = last[p];
            image[3 * px] = cr[p] + T[p] * bg0;
            image[3 * px + 1] = cg[p] + T[p] * bg1;
            image[3 * px + 2] = cb[p] + T[p] * bg2;
            inv_depth[px] = cd[p];
        }
    }
}

} // namespace

int gsr_blend_p_override = 1; // pixels per lane (GSR_BLEND_P)

hipError_t gsr_launch_blend_forward(const CamK &cam, const int32_t *ranges, const int32_t *point_list, const BlendRec *rec,
                                    const GsrImage &img, hipStream_t s)
{
    const int tiles = cam.grid_x * cam.grid_y;
    if (tiles <= 0) return hipSuccess;
    int P = gsr_blend_p_override;
    if (P != 1 && P != 2 && P != 4) P = 1;
#define LAUNCH(PP)                                                                                                            \
    hipLaunchKernelGGL(blend_forward_kernel<PP>, dim3(tiles), dim3(256 / PP), 0, s, cam.W, cam.H, cam.grid_x, cam.bg[0],      \
                       cam.bg[1], cam.bg[2], ranges, point_list, rec, img.image, img.inv_depth, img.final_T, img.n_contrib)
    if (P == 1) LAUNCH(1);
    else if (P == 2) LAUNCH(2);
    else LAUNCH(4);
#undef LAUNCH
    return hipGetLastError();
}
