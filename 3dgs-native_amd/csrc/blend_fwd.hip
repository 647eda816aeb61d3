// blend_fwd.hip -- per-tile front-to-back alpha blending for gfx950.
//
// What the reference's wp_render_gaussians does per pixel (forward.py:385-515), restructured for
// CDNA4: one 256-thread workgroup per 16x16 tile, each of its four waves owning one 8x8 pixel block.
// The tile's sorted list is staged through LDS in batches of 256 entries, each entry gathered ONCE per
// tile as a single 64-byte record (the reference gathers four arrays per pixel per entry).  While
// staging, the thread that fetched an entry also tests it against the four 8x8 blocks (exact convex
// minimum of the conic over the block rectangle vs ln(255 o), conservative) and stores a 4-bit hit
// mask; a wave then skips, with one scalar test, every entry that cannot reach alpha >= 1/255 inside
// its block -- most of a tile's list.  Live entries are read as LDS broadcasts.  A wave stops when its
// 64 pixels are saturated (ballot), the tile when all four are (__syncthreads_and).
//
// Float operations are in the reference's order (no contraction) so the discrete tests (power > 0,
// alpha < 1/255, T < 1e-4) agree with the CPU oracle except where exp() itself rounds differently:
// exp is v_exp_f32(power * log2 e), relative error < 5e-7 for power in [-5.6, 0].  `n_contrib` is the
// 1-based LIST position of the last contributing entry, so skipping dead entries does not change it.
#include "gsr_internal.h"

namespace {

constexpr int BATCH = 256;

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// minimise 0.5*qa*u^2 + 0.5*qc*v^2 + qb*u*v over v in [vlo, vhi]
__device__ __forceinline__ float edge_min_q(float qa, float qb, float qc, float u, float vlo, float vhi)
{
    float v = -qb * u * fast_rcp(qc);
    v = fminf(vhi, fmaxf(vlo, v));
    return 0.5f * (qa * u * u + qc * v * v) + qb * u * v;
}
// Can alpha reach 1/255 anywhere in pixels [x0,x0+7] x [y0,y0+7]?  Conservative (small slack).
__device__ __forceinline__ bool block_may_hit(float gx, float gy, float ca, float cb, float cc, float lim, float x0, float y0)
{
    const float dxl = gx - (x0 + 7.0f), dxh = gx - x0, dyl = gy - (y0 + 7.0f), dyh = gy - y0;
    float qmin;
    if (dxl <= 0.0f && dxh >= 0.0f && dyl <= 0.0f && dyh >= 0.0f) qmin = 0.0f;
    else {
        qmin = edge_min_q(ca, cb, cc, dxl, dyl, dyh);
        qmin = fminf(qmin, edge_min_q(ca, cb, cc, dxh, dyl, dyh));
        qmin = fminf(qmin, edge_min_q(cc, cb, ca, dyl, dxl, dxh));
        qmin = fminf(qmin, edge_min_q(cc, cb, ca, dyh, dxl, dxh));
    }
    return qmin <= lim;
}

__global__ __launch_bounds__(256) void blend_forward_kernel(int W, int H, int grid_x, float bg0, float bg1, float bg2,
                                                            const int32_t *__restrict__ ranges,
                                                            const int32_t *__restrict__ point_list,
                                                            const BlendRec *__restrict__ rec, float *__restrict__ image,
                                                            float *__restrict__ inv_depth, float *__restrict__ final_T,
                                                            int32_t *__restrict__ n_contrib)
{
    __shared__ float4 s_a[BATCH]; // xy.x xy.y con.a con.b
    __shared__ float4 s_b[BATCH]; // con.c opacity r g
    __shared__ float2 s_c[BATCH]; // b 1/depth
    __shared__ int s_mask[BATCH]; // bit w: entry may touch block w

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tile = blockIdx.x;
    const int tile_x = tile % grid_x, tile_y = tile / grid_x;
    const int pix_x = tile_x * 16 + (wv & 1) * 8 + (lane & 7);
    const int pix_y = tile_y * 16 + (wv >> 1) * 8 + (lane >> 3);
    float pixf_x = (float)pix_x, pixf_y = (float)pix_y;
    asm volatile("" : "+v"(pixf_x), "+v"(pixf_y)); // keep the converted coordinates in registers (hipcc re-converts them per entry otherwise)
    const float tx0 = (float)(tile_x * 16), ty0 = (float)(tile_y * 16);
    const int mybit = 1 << wv;

    const int2 range = *reinterpret_cast<const int2 *>(ranges + 2 * tile);
    const int start = range.x, end = range.y;

    float T = 1.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f, cd = 0.0f;
    int last = 0;
    bool done = !(pix_x < W && pix_y < H);

    // software pipeline: the gather of batch k+1 (id, then its 64-byte record) is in flight while batch k is blended
    int nid = (start + tid < end) ? point_list[start + tid] : -1;
    float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na, ncd = na;
    if (nid >= 0) {
        const float4 *rp = reinterpret_cast<const float4 *>(rec + nid);
        na = rp[0]; nb = rp[1]; ncd = rp[2];
    }
    for (int base = start; base < end; base += BATCH) {
        if (__syncthreads_and(done)) break; // whole tile saturated (also fences LDS reuse)

        const int cnt = min(BATCH, end - base);
        if (tid < cnt) {
            const float4 a = na, b = nb, c = ncd;
            s_a[tid] = a;
            s_b[tid] = b;
            s_c[tid] = make_float2(c.x, c.y);
            int m = 0;
            if (b.y * 255.0f >= 1.0f) {
                const float lim = __builtin_amdgcn_logf(b.y * 255.0f) * 0.6931471805599453f * 1.0001f + 1e-3f; // ln via v_log_f32, argument >= 1
                // axis-aligned box of the ellipse q <= lim (half-widths sqrt(2 lim c / det), sqrt(2 lim a / det), padded): a block
                // outside it cannot be hit, and only the blocks inside get the exact rectangle test
                const float det = a.z * b.x - a.w * a.w;
                const bool boxless = !(det > 0.0f); // not a proper ellipse (never for a valid conic): exact tests for every block
                const float kdet = 2.0f * lim * fast_rcp(det);
                const float hx = __builtin_amdgcn_sqrtf(kdet * b.x) * 1.001f + 0.05f, hy = __builtin_amdgcn_sqrtf(kdet * a.z) * 1.001f + 0.05f;
                const float lx = a.x - hx - tx0, rx = a.x + hx - tx0, ly = a.y - hy - ty0, ry = a.y + hy - ty0; // box relative to the tile origin
                const bool xin[2] = {boxless || (lx <= 7.0f && rx >= 0.0f), boxless || (lx <= 15.0f && rx >= 8.0f)};
                const bool yin[2] = {boxless || (ly <= 7.0f && ry >= 0.0f), boxless || (ly <= 15.0f && ry >= 8.0f)};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (xin[k & 1] && yin[k >> 1] &&
                        block_may_hit(a.x, a.y, a.z, a.w, b.x, lim, tx0 + (float)((k & 1) * 8), ty0 + (float)((k >> 1) * 8))) m |= 1 << k;
            }
            s_mask[tid] = m;
        }
        // issue the next batch's gather now; it completes under the blend loop below
        {
            const int nidx = base + BATCH + tid;
            nid = (nidx < end) ? point_list[nidx] : -1;
            if (nid >= 0) {
                const float4 *rp = reinterpret_cast<const float4 *>(rec + nid);
                na = rp[0]; nb = rp[1]; ncd = rp[2];
            }
        }
        __syncthreads();

        if (__all(done)) continue; // this wave has nothing left; keep serving the barriers

        // each wave walks only the entries whose mask has its bit: 64 mask words -> one ballot -> scalar bit loop.
        // Two register sets alternate so the LDS broadcast reads of the next live entry are in flight while the
        // current one is blended (no register shuffling); saturation of the whole wave is re-checked once per
        // 64-entry group, not per entry.
#define GSR_BLEND(A, B, C, J)                                                                                                  \
    {                                                                                                                         \
        const float dx = A.x - pixf_x, dy = A.y - pixf_y;                                                                     \
        const float power = -0.5f * (A.z * dx * dx + B.x * dy * dy) - A.w * dx * dy;                                          \
        const float alpha = fminf(0.99f, B.y * fast_exp(power));                                                              \
        const float test_T = T * (1.0f - alpha);                                                                              \
        const bool live = !done && !(power > 0.0f) && !(alpha < (1.0f / 255.0f));                                             \
        const bool sat = live && (test_T < 0.0001f);                                                                          \
        done = done || sat;                                                                                                   \
        if (live && !sat) {                                                                                                   \
            cr += B.z * alpha * T;                                                                                            \
            cg += B.w * alpha * T;                                                                                            \
            cb += C.x * alpha * T;                                                                                            \
            cd += C.y * alpha * T;                                                                                            \
            T = test_T;                                                                                                       \
            last = base - start + (J) + 1;                                                                                    \
        }                                                                                                                     \
    }
        for (int g = 0; g < cnt; g += 64) {
            const int mv = (g + lane < cnt) ? s_mask[g + lane] : 0;
            unsigned long long bits = __ballot((mv & mybit) != 0);
            if (!bits) continue;
            int j0 = g + __builtin_ctzll(bits);
            bits &= bits - 1;
            float4 a0 = s_a[j0], b0 = s_b[j0];
            float2 c0 = s_c[j0];
            for (;;) {
                int j1 = j0;
                const bool more1 = bits != 0;
                if (more1) { j1 = g + __builtin_ctzll(bits); bits &= bits - 1; }
                const float4 a1 = s_a[j1], b1 = s_b[j1];
                const float2 c1 = s_c[j1];
                GSR_BLEND(a0, b0, c0, j0);
                if (!more1) break;
                const bool more0 = bits != 0;
                if (more0) { j0 = g + __builtin_ctzll(bits); bits &= bits - 1; }
                a0 = s_a[j0]; b0 = s_b[j0]; c0 = s_c[j0];
                GSR_BLEND(a1, b1, c1, j1);
                if (!more0) break;
            }
            if (__all(done)) break; // the wave's 64 pixels are saturated: nothing later in the list can contribute
        }
#undef GSR_BLEND
    }

    if (pix_x < W && pix_y < H) {
        const size_t px = (size_t)pix_y * W + pix_x;
        final_T[px] = T;
        n_contrib[px] = last;
        image[3 * px] = cr + T * bg0;
        image[3 * px + 1] = cg + T * bg1;
        image[3 * px + 2] = cb + T * bg2;
        inv_depth[px] = cd;
    }
}

} // namespace

hipError_t gsr_launch_blend_forward(const CamK &cam, const int32_t *ranges, const int32_t *point_list, const BlendRec *rec,
                                    const GsrImage &img, hipStream_t s)
{
    const int tiles = cam.grid_x * cam.grid_y;
    if (tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(blend_forward_kernel, dim3(tiles), dim3(256), 0, s, cam.W, cam.H, cam.grid_x, cam.bg[0], cam.bg[1], cam.bg[2],
                       ranges, point_list, rec, img.image, img.inv_depth, img.final_T, img.n_contrib);
    return hipGetLastError();
}
