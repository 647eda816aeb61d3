// densify.hip -- adaptive density control: the Warp kernels the reference
// trainer launches from densification_and_pruning() (train.py:351-713), as HBM-bound row movers.
//
// A Gaussian is five rows (12 + 12 + 16 + 4 + 192 = 236 bytes).  Every mover gives one Gaussian to 16
// consecutive lanes: lanes 0..11 carry one float4 of the SH row each (192 B, 16-byte aligned), lane 12 the
// position, 13 the scale, 14 the quaternion, 15 the opacity -- so a wave moves four Gaussians with every SH
// access a full 64-byte segment per quarter-wave, and the source/destination row indices are computed once
// per Gaussian instead of 48 times (the reference copies SH with a 16-iteration vec3 loop per thread,
// optimizer.py:276-277, 346-347, 415-416: stride-192-byte accesses across the wave).
#include <math.h>

#include "gsr_internal.h"

namespace {

// Warp's stateless generator behind wp.randf(wp.uint32(x)) (optimizer.py:297-299, 353-355): one PCG hash round of
// the state, top 24 bits scaled to [0,1).  Restated from NVIDIA Warp's published native/rand.h (warp-lang is not
// installed here: parity unpinned).
__device__ __forceinline__ uint32_t rand_pcg(uint32_t state)
{
    const uint32_t b = state * 747796405u + 2891336453u;
    const uint32_t c = ((b >> ((b >> 28u) + 4u)) ^ b) * 277803737u;
    return (c >> 22u) ^ c;
}
__device__ __forceinline__ float randf(uint32_t state) { return (float)(rand_pcg(state) >> 8) * (1.0f / 16777216.0f); }

struct RowEdit {
    bool add_pos, mul_scale;
    float px, py, pz, sm;
};

// lane `part` of a 16-lane group moves its piece of Gaussian `src` of `in` to row `dst` of `out`
__device__ __forceinline__ void move_part(const GsrParams &in, int64_t src, const GsrParams &out, int64_t dst, int part, const RowEdit &e)
{
    if (part < 12) {
        reinterpret_cast<float4 *>(out.shs)[dst * 12 + part] = reinterpret_cast<const float4 *>(in.shs)[src * 12 + part];
    } else if (part == 12) {
        float x = in.positions[src * 3 + 0], y = in.positions[src * 3 + 1], z = in.positions[src * 3 + 2];
        if (e.add_pos) { x += e.px; y += e.py; z += e.pz; }
        out.positions[dst * 3 + 0] = x; out.positions[dst * 3 + 1] = y; out.positions[dst * 3 + 2] = z;
    } else if (part == 13) {
        float x = in.scales[src * 3 + 0], y = in.scales[src * 3 + 1], z = in.scales[src * 3 + 2];
        if (e.mul_scale) { x *= e.sm; y *= e.sm; z *= e.sm; }
        out.scales[dst * 3 + 0] = x; out.scales[dst * 3 + 1] = y; out.scales[dst * 3 + 2] = z;
    } else if (part == 14) {
        reinterpret_cast<float4 *>(out.rotations)[dst] = reinterpret_cast<const float4 *>(in.rotations)[src];
    } else {
        out.opacities[dst] = in.opacities[src];
    }
}

// compute_grad_norms (train.py:398-406) fused into mark_clone_candidates / mark_split_candidates (optimizer.py:180-239)
__global__ __launch_bounds__(256) void mark_kernel(GsrParams p, const float *__restrict__ pos_grad, int64_t n_grad, float grad_threshold,
                                                   float scale_threshold, int mode, int32_t *__restrict__ mask)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.N) return;
    float norm = 0.0f;
    if (i < n_grad) {
        const float gx = pos_grad[i * 3 + 0], gy = pos_grad[i * 3 + 1], gz = pos_grad[i * 3 + 2];
        norm = sqrtf(gx * gx + gy * gy + gz * gz);
    }
    const float max_scale = fmaxf(fmaxf(p.scales[i * 3 + 0], p.scales[i * 3 + 1]), p.scales[i * 3 + 2]);
    const bool high_grad = norm >= grad_threshold;
    const bool size_ok = mode == GSR_MARK_SPLIT ? (max_scale > scale_threshold) : (max_scale <= scale_threshold);
    mask[i] = (high_grad && size_ok) ? 1 : 0;
}

// prune_gaussians (optimizer.py:367-385)
__global__ __launch_bounds__(256) void prune_mark_kernel(int64_t N, const float *__restrict__ opacities, float threshold, int32_t *__restrict__ valid)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) valid[i] = opacities[i] > threshold ? 1 : 0;
}

// mark_split_originals_for_removal + invert_mask (train.py:547-576) in one pass
__global__ __launch_bounds__(256) void split_removal_kernel(int64_t n_total, int64_t offset, const int32_t *__restrict__ split_mask,
                                                            int32_t *__restrict__ valid)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_total) valid[i] = (i < offset && split_mask[i] == 1) ? 0 : 1;
}

// clone_gaussians (optimizer.py:312-365)
__global__ __launch_bounds__(256) void clone_kernel(GsrParams in, const int32_t *__restrict__ mask, const int32_t *__restrict__ prefix,
                                                    float noise_scale, GsrParams out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t >> 4;
    const int part = (int)(t & 15);
    if (i >= in.N) return;
    RowEdit e{false, false, 0.f, 0.f, 0.f, 1.f};
    move_part(in, i, out, i, part, e);
    if (mask[i] != 1) return;
    const int64_t dst = (int64_t)prefix[i] + in.N;
    if (dst >= out.N) return; // the reference writes past its arrays here (the last row's flag is not in its count)
    if (part == 12) {
        const int32_t i3 = (int32_t)i * 3;
        e.add_pos = true;
        e.px = randf((uint32_t)i3) * noise_scale;
        e.py = randf((uint32_t)(i3 + 1)) * noise_scale;
        e.pz = randf((uint32_t)(i3 + 2)) * noise_scale;
    }
    move_part(in, i, out, dst, part, e);
}

// split_gaussians (optimizer.py:242-309)
__global__ __launch_bounds__(256) void split_kernel(GsrParams in, const int32_t *__restrict__ mask, const int32_t *__restrict__ prefix, int n_split,
                                                    float scale_factor, GsrParams out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t >> 4;
    const int part = (int)(t & 15);
    if (i >= in.N) return;
    RowEdit e{false, false, 0.f, 0.f, 0.f, 1.f};
    move_part(in, i, out, i, part, e);
    if (mask[i] != 1) return;
    const int64_t first = in.N + (int64_t)prefix[i] * n_split;
    e.mul_scale = true;
    e.sm = scale_factor;
    for (int j = 0; j < n_split; ++j) {
        const int64_t dst = first + j;
        if (dst >= out.N) break; // optimizer.py:288
        if (part == 12) {
            const int32_t d3 = (int32_t)dst * 3;
            e.add_pos = true;
            e.px = (randf((uint32_t)d3) * 2.0f - 1.0f) * 0.01f;
            e.py = (randf((uint32_t)(d3 + 1)) * 2.0f - 1.0f) * 0.01f;
            e.pz = (randf((uint32_t)(d3 + 2)) * 2.0f - 1.0f) * 0.01f;
        }
        move_part(in, i, out, dst, part, e);
    }
}

// compact_gaussians (optimizer.py:387-416)
__global__ __launch_bounds__(256) void compact_kernel(GsrParams in, const int32_t *__restrict__ valid, const int32_t *__restrict__ prefix, GsrParams out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t >> 4;
    if (i >= in.N || valid[i] == 0) return;
    const int64_t dst = prefix[i];
    if (dst >= out.N) return; // same undercount as the clone: the reference's output array is one row short
    const RowEdit e{false, false, 0.f, 0.f, 0.f, 1.f};
    move_part(in, i, out, dst, (int)(t & 15), e);
}

// init_gaussian_params (train.py:37-92): positions = randf(3i+k) * 2.6 - 1.3, scales = init_scale, rotation = (1, 0, 0, 0) as
// stored (x, y, z, w -- not the identity of that convention; kept), opacity 0.1, SH DC = -0.007, higher bands 0.
// 16 lanes per Gaussian like the movers, so the SH row is written as whole 64-byte segments.
__global__ __launch_bounds__(256) void init_kernel(GsrParams out, float init_scale)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t >> 4;
    const int part = (int)(t & 15);
    if (i >= out.N) return;
    if (part < 12) {
        reinterpret_cast<float4 *>(out.shs)[i * 12 + part] = part == 0 ? make_float4(-0.007f, -0.007f, -0.007f, 0.0f) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else if (part == 12) {
        const int32_t i3 = (int32_t)i * 3;
        out.positions[i * 3 + 0] = randf((uint32_t)i3) * 2.6f - 1.3f;
        out.positions[i * 3 + 1] = randf((uint32_t)(i3 + 1)) * 2.6f - 1.3f;
        out.positions[i * 3 + 2] = randf((uint32_t)(i3 + 2)) * 2.6f - 1.3f;
    } else if (part == 13) {
        out.scales[i * 3 + 0] = init_scale; out.scales[i * 3 + 1] = init_scale; out.scales[i * 3 + 2] = init_scale;
    } else if (part == 14) {
        reinterpret_cast<float4 *>(out.rotations)[i] = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
    } else {
        out.opacities[i] = 0.1f;
    }
}

// reset_opacities (optimizer.py:141-156)
__global__ __launch_bounds__(256) void fill_kernel(int64_t N, float v, float *__restrict__ p)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) p[i] = v;
}

bool params_ok(const GsrParams *p) { return p && (p->N == 0 || (p->positions && p->scales && p->rotations && p->opacities && p->shs)); }
int done() { return hipGetLastError() == hipSuccess ? GSR_OK : GSR_E_HIP; }
unsigned blocks_for(int64_t threads) { return (unsigned)gsr_div_up(threads, 256); }
constexpr int64_t MAX_ROWS = (int64_t)1 << 27; // 16 lanes per row must fit a 32-bit grid; row*3 must fit the int32 the generator is seeded with

} // namespace

extern "C" {

int gsr_densify_mark(const GsrParams *p, const float *pos_grad, int64_t n_grad, float grad_threshold, float scene_extent, float percent_dense,
                     int mode, int32_t *mask, void *stream)
{
    if (!p) return GSR_E_NULL;
    if (p->N < 0 || n_grad < 0 || p->N > MAX_ROWS || (mode != GSR_MARK_CLONE && mode != GSR_MARK_SPLIT)) return GSR_E_DIMS;
    if (p->N == 0) return GSR_OK;
    if (!p->scales || !mask || (n_grad > 0 && !pos_grad)) return GSR_E_NULL;
    const float scale_threshold = percent_dense * scene_extent; // float32 product, as in the kernels (optimizer.py:199, 232)
    hipLaunchKernelGGL(mark_kernel, dim3(blocks_for(p->N)), dim3(256), 0, (hipStream_t)stream, *p, pos_grad, n_grad, grad_threshold,
                       scale_threshold, mode, mask);
    return done();
}

int gsr_prune_mark(const GsrParams *p, float opacity_threshold, int32_t *valid, void *stream)
{
    if (!p) return GSR_E_NULL;
    if (p->N < 0 || p->N > MAX_ROWS) return GSR_E_DIMS;
    if (p->N == 0) return GSR_OK;
    if (!p->opacities || !valid) return GSR_E_NULL;
    hipLaunchKernelGGL(prune_mark_kernel, dim3(blocks_for(p->N)), dim3(256), 0, (hipStream_t)stream, p->N, p->opacities, opacity_threshold, valid);
    return done();
}

int gsr_split_removal_mask(int64_t n_total, int64_t offset, const int32_t *split_mask, int32_t *valid, void *stream)
{
    if (n_total < 0 || offset < 0 || offset > n_total || n_total > MAX_ROWS) return GSR_E_DIMS;
    if (n_total == 0) return GSR_OK;
    if (!valid || (offset > 0 && !split_mask)) return GSR_E_NULL;
    hipLaunchKernelGGL(split_removal_kernel, dim3(blocks_for(n_total)), dim3(256), 0, (hipStream_t)stream, n_total, offset, split_mask, valid);
    return done();
}

size_t gsr_mask_scan_workspace_bytes(int64_t N) { return N <= 0 ? 256 : gsr_align(((size_t)gsr_div_up(N, GSR_SCAN_WAVE_ITEMS) + 4) * sizeof(int32_t)); } // + 4: read as int4

int gsr_mask_scan(int64_t N, const int32_t *mask, int32_t *prefix, int32_t *count_host, void *scratch, size_t scratch_bytes, void *stream)
{
    if (!count_host) return GSR_E_NULL;
    if (N < 0 || N > MAX_ROWS) return GSR_E_DIMS;
    *count_host = 0;
    if (N == 0) return GSR_OK;
    if (!mask || !prefix || !scratch) return GSR_E_NULL;
    if (!gsr_aligned16(scratch)) return GSR_E_ALIGN;
    if (scratch_bytes < gsr_mask_scan_workspace_bytes(N)) return GSR_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (gsr_launch_scan(mask, nullptr, prefix, (int32_t *)scratch, N, 2, nullptr, false, s) != hipSuccess) return GSR_E_HIP;
    // the reference's count is the LAST ENTRY of the exclusive scan (train.py:433, 497, 581, 641)
    if (hipMemcpyAsync(count_host, prefix + (N - 1), sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess) return GSR_E_HIP;
    return hipStreamSynchronize(s) == hipSuccess ? GSR_OK : GSR_E_HIP;
}

int gsr_clone_gaussians(const GsrParams *in, const int32_t *mask, const int32_t *prefix, float noise_scale, const GsrParams *out, void *stream)
{
    if (!params_ok(in) || !params_ok(out)) return GSR_E_NULL;
    if (in->N < 0 || out->N < in->N || out->N > MAX_ROWS) return GSR_E_DIMS;
    if (in->N == 0) return GSR_OK;
    if (!mask || !prefix) return GSR_E_NULL;
    hipLaunchKernelGGL(clone_kernel, dim3(blocks_for(in->N * 16)), dim3(256), 0, (hipStream_t)stream, *in, mask, prefix, noise_scale, *out);
    return done();
}

int gsr_split_gaussians(const GsrParams *in, const int32_t *mask, const int32_t *prefix, int32_t n_split, float scale_factor, const GsrParams *out,
                        void *stream)
{
    if (!params_ok(in) || !params_ok(out)) return GSR_E_NULL;
    if (in->N < 0 || out->N < in->N || out->N > MAX_ROWS || n_split < 0) return GSR_E_DIMS;
    if (in->N == 0) return GSR_OK;
    if (!mask || !prefix) return GSR_E_NULL;
    hipLaunchKernelGGL(split_kernel, dim3(blocks_for(in->N * 16)), dim3(256), 0, (hipStream_t)stream, *in, mask, prefix, (int)n_split, scale_factor,
                       *out);
    return done();
}

int gsr_compact_gaussians(const GsrParams *in, const int32_t *valid, const int32_t *prefix, const GsrParams *out, void *stream)
{
    if (!params_ok(in) || !params_ok(out)) return GSR_E_NULL;
    if (in->N < 0 || out->N < 0 || in->N > MAX_ROWS) return GSR_E_DIMS;
    if (in->N == 0 || out->N == 0) return GSR_OK;
    if (!valid || !prefix) return GSR_E_NULL;
    hipLaunchKernelGGL(compact_kernel, dim3(blocks_for(in->N * 16)), dim3(256), 0, (hipStream_t)stream, *in, valid, prefix, *out);
    return done();
}

int gsr_init_gaussians(const GsrParams *out, float init_scale, void *stream)
{
    if (!params_ok(out)) return GSR_E_NULL;
    if (out->N < 0 || out->N > MAX_ROWS) return GSR_E_DIMS;
    if (out->N == 0) return GSR_OK;
    hipLaunchKernelGGL(init_kernel, dim3(blocks_for(out->N * 16)), dim3(256), 0, (hipStream_t)stream, *out, init_scale);
    return done();
}

int gsr_reset_opacities(int64_t N, float max_opacity, float *opacities, void *stream)
{
    if (N < 0) return GSR_E_DIMS;
    if (N == 0) return GSR_OK;
    if (!opacities) return GSR_E_NULL;
    hipLaunchKernelGGL(fill_kernel, dim3(blocks_for(N)), dim3(256), 0, (hipStream_t)stream, N, max_opacity, opacities);
    return done();
}

} // extern "C"
