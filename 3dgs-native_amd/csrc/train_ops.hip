// train_ops.hip -- the two HBM-bound steps either side of the rasterizer in a training iteration
// (SURVEY.md section 8(f) rows f2 and f3): L1 loss + pixel gradient, and the fused Adam update.
#include <math.h>

#include "gsr_internal.h"
#include "sh_stage.h"

namespace {

// ---- f2: reference loss.py:12-31 (l1_loss_kernel) + :122-146 (backprop_l1_pixel_gradients) in one pass ----
// Flat over H*W*3 floats, float4 per lane (the image is packed vec3, so any 16-byte group is valid).
__global__ __launch_bounds__(256) void l1_loss_grad_kernel(const float *__restrict__ rendered, const float *__restrict__ target,
                                                           float *__restrict__ pixel_grad, float *__restrict__ loss_sum, int64_t n,
                                                           float l1_weight)
{
    __shared__ float s_part[4];
    float acc = 0.0f;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 r = reinterpret_cast<const float4 *>(rendered)[i];
        const float4 t = reinterpret_cast<const float4 *>(target)[i];
        const float d0 = r.x - t.x, d1 = r.y - t.y, d2 = r.z - t.z, d3 = r.w - t.w;
        acc += fabsf(d0) + fabsf(d1) + fabsf(d2) + fabsf(d3);
        if (pixel_grad)
            reinterpret_cast<float4 *>(pixel_grad)[i] = make_float4(l1_weight * (d0 < 0.0f ? -1.0f : 1.0f), l1_weight * (d1 < 0.0f ? -1.0f : 1.0f),
                                                                    l1_weight * (d2 < 0.0f ? -1.0f : 1.0f), l1_weight * (d3 < 0.0f ? -1.0f : 1.0f));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { // tail (< 4 floats)
        const int64_t i = (n4 << 2) + threadIdx.x;
        const float d = rendered[i] - target[i];
        acc += fabsf(d);
        if (pixel_grad) pixel_grad[i] = l1_weight * (d < 0.0f ? -1.0f : 1.0f);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(loss_sum, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

// ---- loss.py:47-119 (ssim_kernel): one 16x16 pixel tile per workgroup, the tile and its 5-pixel halo of both images staged
// in LDS (26 x 26 x 6 floats), each thread then walks its 11 x 11 window with the reference's accumulation order (row by
// row, x fastest).  Window weights are indexed by DISTANCE into a Gaussian centred on index 5 (quirk Q21): w[d] as passed.
struct SsimW { float w[6]; };
__global__ __launch_bounds__(256) void ssim_kernel(const float *__restrict__ rendered, const float *__restrict__ target, float *__restrict__ ssim_sum,
                                                   int W, int H, SsimW gw)
{
    constexpr int HALF = 5, SIDE = 16 + 2 * HALF;
    __shared__ float s_r[SIDE * SIDE * 3], s_t[SIDE * SIDE * 3];
    __shared__ float s_part[4];
    const int tx0 = blockIdx.x * 16, ty0 = blockIdx.y * 16;
    for (int k = threadIdx.x; k < SIDE * SIDE; k += 256) {
        const int ly = k / SIDE, lx = k - ly * SIDE;
        const int y = ty0 + ly - HALF, x = tx0 + lx - HALF;
        const bool in = x >= 0 && x < W && y >= 0 && y < H;
        const size_t g = in ? 3 * ((size_t)y * W + x) : 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            s_r[3 * k + c] = in ? rendered[g + c] : 0.0f;
            s_t[3 * k + c] = in ? target[g + c] : 0.0f;
        }
    }
    __syncthreads();
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int i = tx0 + lx, j = ty0 + ly;
    float val = 0.0f;
    if (i < W && j < H) {
        float mu1[3] = {0.f, 0.f, 0.f}, mu2[3] = {0.f, 0.f, 0.f}, s1[3] = {0.f, 0.f, 0.f}, s2[3] = {0.f, 0.f, 0.f}, s12[3] = {0.f, 0.f, 0.f};
        float weight_sum = 0.0f;
        const int y0 = max(0, j - HALF), y1 = min(H, j + HALF + 1), x0 = max(0, i - HALF), x1 = min(W, i + HALF + 1);
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                const float w = gw.w[abs(x - i)] * gw.w[abs(y - j)];
                const int k = 3 * ((y - ty0 + HALF) * SIDE + (x - tx0 + HALF));
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float p1 = s_r[k + c], p2 = s_t[k + c];
                    mu1[c] += p1 * w;
                    mu2[c] += p2 * w;
                    s1[c] += (p1 * p1) * w;
                    s2[c] += (p2 * p2) * w;
                    s12[c] += (p1 * p2) * w;
                }
                weight_sum += w;
            }
        const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
        float ss[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float m1 = mu1[c], m2 = mu2[c], a1 = s1[c], a2 = s2[c], a12 = s12[c];
            if (weight_sum > 0.0f) { m1 /= weight_sum; m2 /= weight_sum; a1 /= weight_sum; a2 /= weight_sum; a12 /= weight_sum; }
            const float v1 = a1 - m1 * m1, v2 = a2 - m2 * m2, v12 = a12 - m1 * m2;
            ss[c] = ((2.0f * m1 * m2 + c1) * (2.0f * v12 + c2)) / ((m1 * m1 + m2 * m2 + c1) * (v1 + v2 + c2));
        }
        val = (ss[0] + ss[1] + ss[2]) / 3.0f;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) val += __shfl_xor(val, d, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = val;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(ssim_sum, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

// ---- loss.py:247-269 (depth_loss_kernel): sum |rendered - target| * mask ----
__global__ __launch_bounds__(256) void depth_loss_kernel(const float *__restrict__ rendered, const float *__restrict__ target,
                                                         const float *__restrict__ mask, float *__restrict__ loss_sum, int64_t n)
{
    __shared__ float s_part[4];
    float acc = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        acc += fabsf(rendered[i] - target[i]) * mask[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(loss_sum, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

// ---- f3: reference optimizer.py:7-139 (adam_update) ----
// The SH rows of parameter and both moments (3 x 192 bytes in, 3 x 192 out per Gaussian: all of the update's traffic that matters)
// are use-once streams like preprocess' and geom_bwd's: non-temporal (sh_stage.h; GSR_NT_ADAM=0 for A/B).
#ifndef GSR_NT_ADAM
#define GSR_NT_ADAM 1
#endif
struct AdamK {
    float beta1, beta2, omb1, omb2, eps, bc1, bc2;
};

// one element of the vec3 groups (positions, scales, SH): wp_vec3_div_element adds 1e-9 to the denominator
__device__ __forceinline__ float adam_vec3_elem(float p, float g, float &m, float &v, float lr, const AdamK &k)
{
    m = k.beta1 * m + k.omb1 * g;
    v = k.beta2 * v + k.omb2 * (g * g);
    const float mc = m / k.bc1, vc = v / k.bc2;
    const float denom = sqrtf(vc) + k.eps;
    return p - lr * (mc / (denom + 1e-9f));
}

// positions, scales, rotations, opacities: one thread per Gaussian
__global__ __launch_bounds__(256) void adam_small_kernel(int64_t N, float *__restrict__ pos, const float *__restrict__ gpos, float *__restrict__ mpos,
                                                         float *__restrict__ vpos, float lr_pos, float *__restrict__ scl,
                                                         const float *__restrict__ gscl, float *__restrict__ mscl, float *__restrict__ vscl,
                                                         float lr_scale, float *__restrict__ rot, const float *__restrict__ grot,
                                                         float *__restrict__ mrot, float *__restrict__ vrot, float lr_rot, float *__restrict__ opa,
                                                         const float *__restrict__ gopa, float *__restrict__ mopa, float *__restrict__ vopa,
                                                         float lr_opac, AdamK k)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float m = mpos[3 * i + c], v = vpos[3 * i + c];
        pos[3 * i + c] = adam_vec3_elem(pos[3 * i + c], gpos[3 * i + c], m, v, lr_pos, k);
        mpos[3 * i + c] = m; vpos[3 * i + c] = v;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g = gscl[3 * i + c], s = scl[3 * i + c];
        const float m = k.beta1 * mscl[3 * i + c] + k.omb1 * g;
        const float v = k.beta2 * vscl[3 * i + c] + k.omb2 * (g * g);
        mscl[3 * i + c] = m; vscl[3 * i + c] = v;
        const float mc = m / k.bc1, vc = v / k.bc2;
        const float denom = sqrtf(vc) + k.eps;
        const float update = lr_scale * (mc / (denom + 1e-9f)); // scale_update = lr * div_element(...)   (optimizer.py:70)
        scl[3 * i + c] = fmaxf(s - update, 0.001f);
    }
    {
        const float4 g = *reinterpret_cast<const float4 *>(grot + 4 * i);
        float4 m = *reinterpret_cast<float4 *>(mrot + 4 * i), v = *reinterpret_cast<float4 *>(vrot + 4 * i);
        float4 q = *reinterpret_cast<float4 *>(rot + 4 * i);
        float *gm[4] = {&m.x, &m.y, &m.z, &m.w}, *gv[4] = {&v.x, &v.y, &v.z, &v.w}, *gq[4] = {&q.x, &q.y, &q.z, &q.w};
        const float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            *gm[c] = k.beta1 * *gm[c] + k.omb1 * gg[c];
            *gv[c] = k.beta2 * *gv[c] + k.omb2 * (gg[c] * gg[c]);
            const float mc = *gm[c] / k.bc1, vc = *gv[c] / k.bc2;
            const float denom = sqrtf(vc) + k.eps;
            *gq[c] = *gq[c] - lr_rot * mc / denom; // (lr * m_hat) / denom, optimizer.py:96-101
        }
        const float len = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
        if (len > 0.0f) q = make_float4(q.x / len, q.y / len, q.z / len, q.w / len);
        *reinterpret_cast<float4 *>(mrot + 4 * i) = m;
        *reinterpret_cast<float4 *>(vrot + 4 * i) = v;
        *reinterpret_cast<float4 *>(rot + 4 * i) = q;
    }
    {
        const float g = gopa[i];
        const float m = k.beta1 * mopa[i] + k.omb1 * g;
        const float v = k.beta2 * vopa[i] + k.omb2 * (g * g);
        mopa[i] = m; vopa[i] = v;
        const float mc = m / k.bc1, vc = v / k.bc2;
        const float upd = lr_opac * mc / (sqrtf(vc) + k.eps);
        opa[i] = fmaxf(fminf(opa[i] - upd, 1.0f), 0.0f);
    }
}

// SH coefficients: purely element-wise over N*48 floats -> float4, fully coalesced (optimizer.py:128-139)
__global__ __launch_bounds__(256) void adam_sh_kernel(int64_t n4, float4 *__restrict__ p, const float4 *__restrict__ g, float4 *__restrict__ m,
                                                      float4 *__restrict__ v, float lr, AdamK k)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pp = gsr_ld4<GSR_NT_ADAM != 0>(p + i), mm = gsr_ld4<GSR_NT_ADAM != 0>(m + i), vv = gsr_ld4<GSR_NT_ADAM != 0>(v + i);
        const float4 gg = gsr_ld4<GSR_NT_ADAM != 0>(g + i);
        pp.x = adam_vec3_elem(pp.x, gg.x, mm.x, vv.x, lr, k);
        pp.y = adam_vec3_elem(pp.y, gg.y, mm.y, vv.y, lr, k);
        pp.z = adam_vec3_elem(pp.z, gg.z, mm.z, vv.z, lr, k);
        pp.w = adam_vec3_elem(pp.w, gg.w, mm.w, vv.w, lr, k);
        gsr_st4<GSR_NT_ADAM != 0>(p + i, pp); gsr_st4<GSR_NT_ADAM != 0>(m + i, mm); gsr_st4<GSR_NT_ADAM != 0>(v + i, vv);
    }
}

// The same SH update with the gradient formed on the fly from V view payloads (gsr_adam_update_views): lane i builds the 48
// sums basis_k(dir_v) * drgb_v of ITS Gaussian exactly as gsr_sh_grad_from_views does (sh_stage.h: same function, same order,
// then the same `* scale`), parks them in the wave's LDS image, and the wave then streams the parameter and moment rows
// through as whole float4 lines, applying adam_vec3_elem element by element.  The dense gradient (192 bytes per Gaussian
// written by the backward or the rebuild kernel, then read here) never exists.  Must run BEFORE the position update: the
// directions are those the forward rendered with.
__global__ __launch_bounds__(256) void adam_sh_views_kernel(int64_t N, const float *__restrict__ means, int degree, int V, ShViewSet vs,
                                                            float scale, float4 *__restrict__ p, float4 *__restrict__ m, float4 *__restrict__ v,
                                                            float lr, AdamK k)
{
    __shared__ float4 s_rows[4 * SH_WAVE_F4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave_row0 = (int64_t)blockIdx.x * blockDim.x + wv * 64;
    const int rows_valid = (int)min((int64_t)64, max((int64_t)0, N - wave_row0));
    if (rows_valid <= 0) return; // whole wave (no block barrier below)
    float4 *lds_wave = s_rows + wv * SH_WAVE_F4;
    const int64_t i = wave_row0 + lane;
    float acc[48];
#pragma unroll
    for (int q = 0; q < 48; ++q) acc[q] = 0.0f;
    if (i < N) {
        const float mean[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
        sh_grad_sum_over_views(vs, V, N, i, mean, degree, acc);
    }
    float *row = reinterpret_cast<float *>(lds_wave + lane * SH_ROW_F4);
#pragma unroll
    for (int q = 0; q < 48; ++q) row[q] = acc[q] * scale;
    wave_lds_fence();
    // the wave's 64 rows of parameter, first and second moment stream through as whole float4 lines, four at a time per lane
    // (all twelve at once cost 144 staging registers on top of the 48 sums: one wave per SIMD)
#pragma unroll 1
    for (int q0 = 0; q0 < 12; q0 += 4) {
        float4 pp[4], mm[4], vv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = (q0 + q) * 64 + lane, r = e / 12;
            const int64_t g = wave_row0 * 12 + (r < rows_valid ? e : 0);
            pp[q] = gsr_ld4<GSR_NT_ADAM != 0>(p + g); mm[q] = gsr_ld4<GSR_NT_ADAM != 0>(m + g); vv[q] = gsr_ld4<GSR_NT_ADAM != 0>(v + g);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = (q0 + q) * 64 + lane, r = e / 12, c = e - r * 12;
            if (r < rows_valid) {
                const float4 gg = lds_wave[r * SH_ROW_F4 + c];
                float4 a = pp[q], b = mm[q], d = vv[q];
                a.x = adam_vec3_elem(a.x, gg.x, b.x, d.x, lr, k);
                a.y = adam_vec3_elem(a.y, gg.y, b.y, d.y, lr, k);
                a.z = adam_vec3_elem(a.z, gg.z, b.z, d.z, lr, k);
                a.w = adam_vec3_elem(a.w, gg.w, b.w, d.w, lr, k);
                const int64_t g = wave_row0 * 12 + e;
                gsr_st4<GSR_NT_ADAM != 0>(p + g, a); gsr_st4<GSR_NT_ADAM != 0>(m + g, b); gsr_st4<GSR_NT_ADAM != 0>(v + g, d);
            }
        }
    }
}

bool group_ok(const GsrAdamGroup &g) { return g.param && g.grad && g.m && g.v; }

} // namespace

extern "C" {

int gsr_l1_loss_grad(const float *rendered, const float *target, float *pixel_grad, float *loss_sum, int32_t W, int32_t H, float l1_weight,
                     void *stream)
{
    if (!rendered || !target || !loss_sum) return GSR_E_NULL;
    if (W <= 0 || H <= 0) return GSR_E_DIMS;
    if (!gsr_aligned16(rendered) || !gsr_aligned16(target) || !gsr_aligned16(pixel_grad)) return GSR_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(loss_sum, 0, sizeof(float), s) != hipSuccess) return GSR_E_HIP;
    const int64_t n = (int64_t)W * H * 3;
    // at most two workgroups per CU, grid stride: every workgroup ends with one atomic on the ONE loss word, and 1 875 of them (an
    // 800 x 800 image at a float4 per thread) queued there for ~20 us behind 6 us of streaming (27.4 -> 8 us in the trainer's trace)
    const unsigned blocks = (unsigned)std::min<int64_t>(512, gsr_div_up(gsr_div_up(n, 4), 256));
    hipLaunchKernelGGL(l1_loss_grad_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, rendered, target, pixel_grad, loss_sum, n, l1_weight);
    return hipGetLastError() == hipSuccess ? GSR_OK : GSR_E_HIP;
}

int gsr_ssim(const float *rendered, const float *target, float *ssim_sum, int32_t W, int32_t H, void *stream)
{
    if (!rendered || !target || !ssim_sum) return GSR_E_NULL;
    if (W <= 0 || H <= 0) return GSR_E_DIMS;
    hipStream_t s = (hipStream_t)stream;
    SsimW gw; // loss.py:33-45 with sigma = 1.5, window 11: kernel[k] = exp(-(k-5)^2 / 4.5), used at k = distance (Q21)
    const float sigma = 1.5f;
    for (int k = 0; k < 6; ++k) {
        const int x = k - 5;
        gw.w[k] = expf(-1.0f * (float)(x * x) / (2.0f * sigma * sigma));
    }
    if (hipMemsetAsync(ssim_sum, 0, sizeof(float), s) != hipSuccess) return GSR_E_HIP;
    hipLaunchKernelGGL(ssim_kernel, dim3((W + 15) / 16, (H + 15) / 16), dim3(256), 0, s, rendered, target, ssim_sum, W, H, gw);
    return hipGetLastError() == hipSuccess ? GSR_OK : GSR_E_HIP;
}

int gsr_depth_loss(const float *rendered_depth, const float *target_depth, const float *depth_mask, float *loss_sum, int32_t W, int32_t H,
                   void *stream)
{
    if (!rendered_depth || !target_depth || !depth_mask || !loss_sum) return GSR_E_NULL;
    if (W <= 0 || H <= 0) return GSR_E_DIMS;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)W * H;
    if (hipMemsetAsync(loss_sum, 0, sizeof(float), s) != hipSuccess) return GSR_E_HIP;
    hipLaunchKernelGGL(depth_loss_kernel, dim3((unsigned)std::min<int64_t>(1024, gsr_div_up(n, 256))), dim3(256), 0, s, rendered_depth, target_depth,
                       depth_mask, loss_sum, n);
    return hipGetLastError() == hipSuccess ? GSR_OK : GSR_E_HIP;
}

static int adam_impl(const GsrAdam *a, int32_t sh_degree, int32_t V, const float *const *payloads, float scale, void *stream)
{
    const bool views = payloads != nullptr;
    if (!a) return GSR_E_NULL;
    if (a->N < 0) return GSR_E_DIMS;
    if (views && (V < 1 || V > GSR_MAX_VIEWS || sh_degree < 0 || sh_degree > 3 || a->N > ((int64_t)1 << 27))) return GSR_E_DIMS;
    if (a->N == 0) return GSR_OK;
    if (!group_ok(a->pos) || !group_ok(a->scale) || !group_ok(a->rot) || !group_ok(a->opacity)) return GSR_E_NULL;
    if (!a->sh.param || !a->sh.m || !a->sh.v || (!views && !a->sh.grad)) return GSR_E_NULL;
    for (const GsrAdamGroup *g : {&a->pos, &a->scale, &a->rot, &a->opacity, &a->sh})
        if (!gsr_aligned16(g->param) || !gsr_aligned16(g->grad) || !gsr_aligned16(g->m) || !gsr_aligned16(g->v)) return GSR_E_ALIGN;
    ShViewSet vs;
    for (int q = 0; q < GSR_MAX_VIEWS; ++q) {
        vs.payload[q] = (views && q < V) ? payloads[q] : nullptr;
        if (views && q < V && !payloads[q]) return GSR_E_NULL;
    }
    hipStream_t s = (hipStream_t)stream;
    AdamK k;
    k.beta1 = a->beta1; k.beta2 = a->beta2; k.eps = a->epsilon;
    k.omb1 = 1.0f - a->beta1; k.omb2 = 1.0f - a->beta2;
    // bias corrections in float32 on the host: 1 - pow(beta, float(iteration + 1))   (optimizer.py:47-48)
    k.bc1 = 1.0f - powf(a->beta1, (float)(a->iteration + 1));
    k.bc2 = 1.0f - powf(a->beta2, (float)(a->iteration + 1));
    if (views) // first: it needs the positions the views were rendered with
        hipLaunchKernelGGL(adam_sh_views_kernel, dim3((unsigned)gsr_div_up(a->N, 256)), dim3(256), 0, s, a->N, a->pos.param, (int)sh_degree, (int)V, vs,
                           scale, (float4 *)a->sh.param, (float4 *)a->sh.m, (float4 *)a->sh.v, a->sh.lr, k);
    hipLaunchKernelGGL(adam_small_kernel, dim3((unsigned)gsr_div_up(a->N, 256)), dim3(256), 0, s, a->N, a->pos.param, a->pos.grad, a->pos.m,
                       a->pos.v, a->pos.lr, a->scale.param, a->scale.grad, a->scale.m, a->scale.v, a->scale.lr, a->rot.param, a->rot.grad,
                       a->rot.m, a->rot.v, a->rot.lr, a->opacity.param, a->opacity.grad, a->opacity.m, a->opacity.v, a->opacity.lr, k);
    if (!views) {
        const int64_t n4 = a->N * 12; // 48 floats per Gaussian
        const unsigned blocks = (unsigned)std::min<int64_t>(4096, gsr_div_up(n4, 256));
        hipLaunchKernelGGL(adam_sh_kernel, dim3(blocks), dim3(256), 0, s, n4, (float4 *)a->sh.param, (const float4 *)a->sh.grad, (float4 *)a->sh.m,
                           (float4 *)a->sh.v, a->sh.lr, k);
    }
    return hipGetLastError() == hipSuccess ? GSR_OK : GSR_E_HIP;
}

int gsr_adam_update(const GsrAdam *a, void *stream) { return adam_impl(a, 0, 0, nullptr, 1.0f, stream); }

int gsr_adam_update_views(const GsrAdam *a, int32_t sh_degree, int32_t V, const float *const *payloads, float scale, void *stream)
{
    if (!payloads) return GSR_E_NULL;
    return adam_impl(a, sh_degree, V, payloads, scale, stream);
}

} // extern "C"
