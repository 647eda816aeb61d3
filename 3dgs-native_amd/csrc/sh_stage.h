// sh_stage.h -- wave-cooperative movement of SH rows (16 coefficients x 3 floats = 192 B per Gaussian).
//
// A lane that reads or writes its own 192-byte row touches 64 different cache lines per wave
// instruction.  Instead the wave moves its 64 consecutive rows (12 KiB, contiguous in memory) with 12
// fully coalesced 1-KiB float4 instructions through an LDS image, and each lane then works on its own
// row in LDS.  Rows are padded from 12 to 13 float4 so the per-lane ds_read_b128/ds_write_b128 at a
// 52-dword lane stride are bank-conflict free.
#pragma once
#include <hip/hip_runtime.h>

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
        regs.v[k] = ((row_mask >> (i / 12)) & 1ull) ? g4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
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
        if (r < rows_valid) g4[i] = lds_wave[r * SH_ROW_F4 + c];
    }
}
