// sh_stage.h -- wave-cooperative movement of SH rows (16 coefficients x 3 floats = 192 B per Gaussian).
//
// A lane that reads or writes its own 192-byte row touches 64 different cache lines per wave
// instruction.  Instead the wave moves its 64 consecutive rows (12 KiB, contiguous in memory) with 12
// fully coalesced 1-KiB float4 instructions through an LDS image, and each lane then works on its own
// row in LDS.  Rows are padded from 12 to 13 float4 so the per-lane ds_read_b128/ds_write_b128 at a
// 52-dword lane stride are bank-conflict free.
#pragma once
#include <hip/hip_runtime.h>

// Non-temporal loads and stores for the use-once streams of the two per-Gaussian kernels (compile-time switches, all ON in the
// product; -DGSR_NT_...=0 for A/B: DESIGN "Round 4: streaming policy").  On this pool a float4 copy runs at 5.0-5.9 TB/s with the
// default cache policy and 6.1-6.4 TB/s with nt loads and stores (tools/copy_bw.hip): a line that will not be read again should
// not push one that will out of L2 / the Infinity Cache, and a kernel should pay for its own write-back instead of leaving 200 MB
// of dirty lines to its successor.  Round 3 had tried two of these streams alone and found a zero-sum (preprocess faster, the
// geometry backward slower); with ALL of them nt, preprocess goes 87.9 -> 73.5 us, the geometry backward 85.3 -> 76.2 and the step
// 0.5756 -> 0.5462 ms (profiles/r04_h_ab_nontemporal_streams.txt).  What stays on the default policy is what a later kernel
// re-reads soon: tile counts, rectangles and depth items (scan, sort), the sort's items, point_list, block masks, accumulators.
typedef float gsr_f4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 gsr_ld4(const float4 *p)
{
    if (NT) { const gsr_f4 q = __builtin_nontemporal_load(reinterpret_cast<const gsr_f4 *>(p)); return make_float4(q.x, q.y, q.z, q.w); }
    return *p;
}
template <bool NT> __device__ __forceinline__ void gsr_st4(float4 *p, const float4 v)
{
    if (NT) { const gsr_f4 q = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(q, reinterpret_cast<gsr_f4 *>(p)); }
    else *p = v;
}
template <bool NT> __device__ __forceinline__ float gsr_ld1(const float *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ int gsr_ld1i(const int32_t *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void gsr_st1(float *p, float v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT> __device__ __forceinline__ void gsr_st1i(int32_t *p, int32_t v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
#ifndef GSR_NT_INPUTS
#define GSR_NT_INPUTS 1       // the per-Gaussian inputs of preprocess / geom_bwd read once per kernel (positions, scales, rotations, opacity, Sigma3D ...)
#endif
#ifndef GSR_NT_MISC_STORE
#define GSR_NT_MISC_STORE 1   // per-lane outputs nobody reads soon (radii, depths; dL_drot, dL_dopacity)
#endif
#ifndef GSR_NT_SH_LOAD
#define GSR_NT_SH_LOAD 1      // the 192-byte SH rows (preprocess; geom_bwd when it has no direction sums)
#endif
#ifndef GSR_NT_ROW_STORE
#define GSR_NT_ROW_STORE 1    // preprocess' AoS outputs (Sigma3D, clamp flags, direction sums, blend records)
#endif
#ifndef GSR_NT_GRAD_STORE
#define GSR_NT_GRAD_STORE 1   // geom_bwd's gradient rows (192-byte SH gradient rows, the 12-byte rows)
#endif

#define SH_ROW_F4 13                      // padded row length in float4
#define SH_WAVE_F4 (64 * SH_ROW_F4)       // LDS float4 per wave

// Split form: issue the 12 coalesced loads early (fetch), do unrelated math, then park them in LDS (commit),
// so the HBM latency of the 12 KiB hides under that math instead of in front of it.
struct ShRegs {
    float4 v[12];
};
// row_mask: bit r set = row r of the wave is wanted (rows of culled Gaussians are not read at all)
__device__ __forceinline__ void sh_rows_fetch(const float4 *__restrict__ g4, ShRegs &regs, int lane, unsigned long long row_mask)
{
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int i = k * 64 + lane;
        regs.v[k] = ((row_mask >> (i / 12)) & 1ull) ? gsr_ld4<GSR_NT_SH_LOAD != 0>(g4 + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__device__ __forceinline__ unsigned long long sh_rows_all(int rows_valid) { return rows_valid >= 64 ? ~0ull : ((1ull << rows_valid) - 1ull); }
__device__ __forceinline__ void sh_rows_commit(const ShRegs &regs, float4 *lds_wave, int lane)
{
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int i = k * 64 + lane;
        const int r = i / 12, c = i - r * 12;
        lds_wave[r * SH_ROW_F4 + c] = regs.v[k];
    }
}

// commit only the rows named in row_mask (used for a second, partial fetch)
__device__ __forceinline__ void sh_rows_commit_masked(const ShRegs &regs, float4 *lds_wave, int lane, unsigned long long row_mask)
{
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int i = k * 64 + lane;
        const int r = i / 12, c = i - r * 12;
        if ((row_mask >> r) & 1ull) lds_wave[r * SH_ROW_F4 + c] = regs.v[k];
    }
}

// g4: first row of the wave (global, as float4); rows_valid: rows of this wave that exist (<= 64)
__device__ __forceinline__ void sh_rows_load(const float4 *__restrict__ g4, float4 *lds_wave, int lane, int rows_valid)
{
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int i = k * 64 + lane;
        const int r = i / 12, c = i - r * 12;
        if (r < rows_valid) lds_wave[r * SH_ROW_F4 + c] = g4[i];
    }
}
__device__ __forceinline__ void sh_rows_store(float4 *__restrict__ g4, const float4 *lds_wave, int lane, int rows_valid)
{
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int i = k * 64 + lane;
        const int r = i / 12, c = i - r * 12;
        if (r < rows_valid) gsr_st4<GSR_NT_GRAD_STORE != 0>(g4 + i, lds_wave[r * SH_ROW_F4 + c]);
    }
}

// Lanes of ONE wave handing data to each other through LDS need no s_barrier -- a wave's LDS operations execute in order --
// but the COMPILER reasons per thread: to it, "lane 5 writes, lane 40 reads" is a data race it may reorder across (seen in
// round 2: two consecutive wave_store_vec3_in_pad calls, the second call's stores of lanes 48-63 hoisted above the first
// call's loads, which only lanes 0-47 execute).  This is the fence for that: no instruction, just no motion of memory
// operations across it.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- wave-cooperative AoS stores -------------------------------------------------------------------------------------
// A lane that stores its own K-float row (K = 3, 6, 16 ...) with K scalar / vector stores makes every store instruction
// touch 64 different rows: partial 64-byte lines, K times over.  Instead the wave parks its 64 rows in LDS and streams the
// 64*K contiguous floats out as whole float4s (one 1-KiB fully coalesced instruction per 256 floats).  `lds` is the wave's
// own scratch (>= 64*K floats, 16-byte aligned; the dead SH image serves); `g` is the wave's first row in global memory
// (16-byte aligned because the wave starts on a multiple of 64 rows); rows_valid < 64 only in the array's last wave, where a
// trailing partial float4 is written float by float.  Row stride in LDS is K floats: for odd K (3) that is conflict-free,
// for K = 6 two-way, negligible next to the global-memory side.
template <int K>
__device__ __forceinline__ void wave_store_rows(float *__restrict__ g, float *lds, int lane, int rows_valid, const float (&v)[K])
{
    wave_lds_fence(); // whatever the wave last read from this scratch has been read
#pragma unroll
    for (int k = 0; k < K; ++k) lds[lane * K + k] = v[k];
    wave_lds_fence();
    const int nfl = rows_valid * K; // floats to write
    constexpr int NF4 = 64 * K / 4;
#pragma unroll
    for (int j0 = 0; j0 < NF4; j0 += 64) {
        const int j = j0 + lane;
        if (j < NF4) {
            if (4 * j + 3 < nfl) gsr_st4<GSR_NT_ROW_STORE != 0>(reinterpret_cast<float4 *>(g) + j, reinterpret_cast<const float4 *>(lds)[j]);
            else
                for (int e = 4 * j; e < nfl && e < 4 * j + 4; ++e) g[e] = lds[e];
        }
    }
}

// The same for 3-float rows when the wave's LDS image is still live (geom_bwd.hip keeps the SH rows, then their gradients,
// in it until the kernel ends): the image's rows are padded from 12 to 13 float4, and that 13th float4 of a lane's own row
// holds its 3 floats.  Output float4 j = floats 4j .. 4j+3 of the contiguous [64][3] block = pad slots of rows (4j)/3 ...
__device__ __forceinline__ void wave_store_vec3_in_pad(float *__restrict__ g, float4 *lds_wave, int lane, int rows_valid, float x, float y, float z)
{
    float *img = reinterpret_cast<float *>(lds_wave);
    float *mine = img + lane * (SH_ROW_F4 * 4) + 48;
    wave_lds_fence(); // the previous call's reads of the pad slots are done
    mine[0] = x; mine[1] = y; mine[2] = z;
    wave_lds_fence();
    const int nfl = rows_valid * 3;
    if (lane < 48) {
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = 4 * lane + c;
            o[c] = img[(e / 3) * (SH_ROW_F4 * 4) + 48 + (e % 3)];
        }
        if (4 * lane + 3 < nfl) gsr_st4<GSR_NT_GRAD_STORE != 0>(reinterpret_cast<float4 *>(g) + lane, make_float4(o[0], o[1], o[2], o[3]));
        else
            for (int c = 0; c < 4; ++c)
                if (4 * lane + c < nfl) g[4 * lane + c] = o[c];
    }
}

// ---- d(colour before clamping)/d(view direction): the nine sums of the reference's SH backward (backward.py:120-244) ------
// dx[c] = sum_k d basis_k/dx (x, y, z) * sh[k][c], likewise dy, dz, in the reference's statement order.  Used by
// geom_backward_kernel (from the coefficients) AND by preprocess_kernel, which hands the nine floats to the backward
// (GsrGeom.sh_dir_grad) so that the backward need not read 192 bytes of coefficients per Gaussian again: one function, so the
// two paths are the same float operations (both files are compiled with -ffp-contract=off).
// `sh`: the Gaussian's 16 x 3 coefficients (row-major [k][c]); dx, dy, dz are overwritten when degree > 0, else left alone.
__device__ __forceinline__ void sh_direction_sums(const float *sh, int degree, float x, float y, float z, float dx_[3], float dy_[3], float dz_[3])
{
#define SHV(k, c) sh[(k) * 3 + (c)]
    const float SH_C1 = 0.4886025119029199f;
    if (degree > 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            dx_[c] = -SH_C1 * SHV(3, c);
            dy_[c] = -SH_C1 * SHV(1, c);
            dz_[c] = SH_C1 * SHV(2, c);
        }
        if (degree > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            const float C2_0 = 1.0925484305920792f, C2_1 = -1.0925484305920792f, C2_2 = 0.31539156525252005f,
                        C2_3 = -1.0925484305920792f, C2_4 = 0.5462742152960396f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dx_[c] += C2_0 * y * SHV(4, c) + C2_2 * 2.0f * -x * SHV(6, c) + C2_3 * z * SHV(7, c) + C2_4 * 2.0f * x * SHV(8, c);
                dy_[c] += C2_0 * x * SHV(4, c) + C2_1 * z * SHV(5, c) + C2_2 * 2.0f * -y * SHV(6, c) + C2_4 * 2.0f * -y * SHV(8, c);
                dz_[c] += C2_1 * y * SHV(5, c) + C2_2 * 2.0f * 2.0f * z * SHV(6, c) + C2_3 * x * SHV(7, c);
            }
            if (degree > 2) {
                const float C3_0 = -0.5900435899266435f, C3_1 = 2.890611442640554f, C3_2 = -0.4570457994644658f,
                            C3_3 = 0.3731763325901154f, C3_4 = -0.4570457994644658f, C3_5 = 1.445305721320277f,
                            C3_6 = -0.5900435899266435f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    dx_[c] += (C3_0 * SHV(9, c) * 3.0f * 2.0f * xy + C3_1 * SHV(10, c) * yz + C3_2 * SHV(11, c) * -2.0f * xy +
                               C3_3 * SHV(12, c) * -3.0f * 2.0f * xz + C3_4 * SHV(13, c) * (-3.0f * xx + 4.0f * zz - yy) +
                               C3_5 * SHV(14, c) * 2.0f * xz + C3_6 * SHV(15, c) * 3.0f * (xx - yy));
                    dy_[c] += (C3_0 * SHV(9, c) * 3.0f * (xx - yy) + C3_1 * SHV(10, c) * xz +
                               C3_2 * SHV(11, c) * (-3.0f * yy + 4.0f * zz - xx) + C3_3 * SHV(12, c) * -3.0f * 2.0f * yz +
                               C3_4 * SHV(13, c) * -2.0f * xy + C3_5 * SHV(14, c) * -2.0f * yz + C3_6 * SHV(15, c) * -3.0f * 2.0f * xy);
                    dz_[c] += (C3_1 * SHV(10, c) * xy + C3_2 * SHV(11, c) * 4.0f * 2.0f * yz +
                               C3_3 * SHV(12, c) * 3.0f * (2.0f * zz - xx - yy) + C3_4 * SHV(13, c) * 4.0f * 2.0f * xz +
                               C3_5 * SHV(14, c) * (xx - yy));
                }
            }
        }
    }
#undef SHV
}

// ---- SH basis values of a unit direction (the coefficients' multipliers in forward.py:330-344 / backward.py:95-119) ---------
// bk[0 .. nb) are written, nb = (degree + 1)^2 is returned.  One function for gsr_sh_grad_from_views and the fused
// Adam-from-views update (train_ops.hip), so both form bit-identical per-view products basis_k * dL_drgb.
__device__ __forceinline__ int sh_basis(int degree, float x, float y, float z, float bk[16])
{
    const float SH_C0 = 0.28209479177387814f, SH_C1 = 0.4886025119029199f;
    bk[0] = SH_C0;
    int nb = 1;
    if (degree > 0) {
        bk[1] = -SH_C1 * y; bk[2] = SH_C1 * z; bk[3] = -SH_C1 * x;
        nb = 4;
        if (degree > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            const float C2_0 = 1.0925484305920792f, C2_1 = -1.0925484305920792f, C2_2 = 0.31539156525252005f,
                        C2_3 = -1.0925484305920792f, C2_4 = 0.5462742152960396f;
            bk[4] = C2_0 * xy; bk[5] = C2_1 * yz; bk[6] = C2_2 * (2.0f * zz - xx - yy); bk[7] = C2_3 * xz; bk[8] = C2_4 * (xx - yy);
            nb = 9;
            if (degree > 2) {
                const float C3_0 = -0.5900435899266435f, C3_1 = 2.890611442640554f, C3_2 = -0.4570457994644658f,
                            C3_3 = 0.3731763325901154f, C3_4 = -0.4570457994644658f, C3_5 = 1.445305721320277f,
                            C3_6 = -0.5900435899266435f;
                bk[9] = C3_0 * y * (3.0f * xx - yy); bk[10] = C3_1 * xy * z; bk[11] = C3_2 * y * (4.0f * zz - xx - yy);
                bk[12] = C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy); bk[13] = C3_4 * x * (4.0f * zz - xx - yy);
                bk[14] = C3_5 * z * (xx - yy); bk[15] = C3_6 * x * (xx - 3.0f * yy);
                nb = 16;
            }
        }
    }
    return nb;
}

// The per-Gaussian sum over views of basis_k(dir_v) * drgb_v[c] (views in order), UNSCALED: acc[3k + c].  `payload[v]` is a
// view's [N*3 + 4] payload (GsrGrads.dL_drgb): colour-gradient rows, then the camera position.
struct ShViewSet {
    const float *payload[GSR_MAX_VIEWS];
};
__device__ __forceinline__ void sh_grad_sum_over_views(const ShViewSet &vs, int V, int64_t N, int64_t i, const float m[3], int degree, float acc[48])
{
#pragma unroll
    for (int k = 0; k < 48; ++k) acc[k] = 0.0f;
    for (int v = 0; v < V; ++v) {
        const float *cp = vs.payload[v] + 3 * N; // wave-uniform: scalar loads
        const float d[3] = {m[0] - cp[0], m[1] - cp[1], m[2] - cp[2]};
        float l2 = d[0] * d[0];
        l2 += d[1] * d[1];
        l2 += d[2] * d[2];
        const float len = sqrtf(l2);
        const float *gp = vs.payload[v] + 3 * i;
        const float g[3] = {gp[0], gp[1], gp[2]};
        if (len < 1e-8f) continue; // backward.py:84-86: no SH gradient for a Gaussian at the camera centre
        const float x = d[0] / len, y = d[1] / len, z = d[2] / len;
        float bk[16];
        const int nb = sh_basis(degree, x, y, z, bk);
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < nb) {
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[3 * k + c] += bk[k] * g[c];
            }
    }
}
