// sigma3d.h -- the 3D covariance of a Gaussian from its (already modified) scale and its quaternion, upper triangle
// (xx, xy, xz, yy, yz, zz), with exactly the operations of the reference's compute_cov3d (forward.py:147-186: rotation matrix column by
// column from the quaternion as stored (x, y, z, w), M = R * S, Sigma = M * M^T).  ONE definition for the two kernels that need it:
// preprocess_kernel, which writes it (GsrGeom.cov3D), and geom_backward_kernel, which may recompute it instead of reading those 24
// bytes per Gaussian back when the caller says the array is the forward's own (GsrGeom.cov3D = NULL in gsr_backward) -- the same
// instructions in the same order, hence the same bits.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ void gsr_sigma3d(float sx, float sy, float sz, float4 q, float out[6])
{
#pragma clang fp contract(off)
    float R[3][3];
    {
        const float cs = 2.0f * q.w * q.w - 1.0f;
        const float qv[3] = {q.x, q.y, q.z};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v[3] = {c == 0 ? 1.0f : 0.0f, c == 1 ? 1.0f : 0.0f, c == 2 ? 1.0f : 0.0f};
            const float cr[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
            float d = qv[0] * v[0];
            d += qv[1] * v[1];
            d += qv[2] * v[2];
#pragma unroll
            for (int r = 0; r < 3; ++r) R[r][c] = v[r] * cs + cr[r] * q.w * 2.0f + qv[r] * d * 2.0f;
        }
    }
    const float S[3][3] = {{sx, 0.0f, 0.0f}, {0.0f, sy, 0.0f}, {0.0f, 0.0f, sz}};
    float M[3][3], sig[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) s += R[i][k] * S[k][j];
            M[i][j] = s;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) s += M[i][k] * M[j][k]; // M * M^T
            sig[i][j] = s;
        }
    out[0] = sig[0][0]; out[1] = sig[0][1]; out[2] = sig[0][2];
    out[3] = sig[1][1]; out[4] = sig[1][2]; out[5] = sig[2][2];
}
