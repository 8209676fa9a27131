// common.h -- shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/scarlet_hip.h"

#define SC_WAVE 64
#define SC_BLOCK 256            // every kernel here runs 256-thread workgroups (4 waves)
#define SC_NWAVES (SC_BLOCK / SC_WAVE)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- wave / block reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = SC_WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, SC_WAVE);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = SC_WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, SC_WAVE);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = SC_WAVE / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, SC_WAVE));
    return v;
}

// Block-wide sum of a double; `red` is SC_NWAVES doubles of LDS.  All threads get the result.
__device__ __forceinline__ double block_sum(double v, double *red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & (SC_WAVE - 1), wid = threadIdx.x / SC_WAVE;
    __syncthreads();                       // protect `red` from a previous use
    if (lane == 0) red[wid] = v;
    __syncthreads();
    double r = 0;
#pragma unroll
    for (int w = 0; w < SC_NWAVES; ++w) r += red[w];
    return r;
}
// numpy's max semantics: NaN propagates (np.max returns nan if any element is nan)
__device__ __forceinline__ float block_max_nan(float v, bool isnan_any, float *red) {
    v = wave_max(v);
    unsigned long long nanmask = __ballot(isnan_any);
    const int lane = threadIdx.x & (SC_WAVE - 1), wid = threadIdx.x / SC_WAVE;
    __syncthreads();
    if (lane == 0) red[wid] = nanmask ? __builtin_nanf("") : v;
    __syncthreads();
    float r = red[0];
    bool bad = r != r;
#pragma unroll
    for (int w = 1; w < SC_NWAVES; ++w) { float t = red[w]; bad |= (t != t); r = fmaxf(r, t); }
    return bad ? __builtin_nanf("") : r;
}

// ---------------------------------------------------------------- fast FFT lengths
// next_fast_len(n) for n < SC_NFL_MAX, filled on the host at first use (scarlet_hip.hip)
#define SC_NFL_MAX 2304
extern __constant__ unsigned short sc_nfl_table[SC_NFL_MAX];

__device__ __forceinline__ int dev_next_fast_len(int n) { return sc_nfl_table[n]; }

// LDS row stride for an image of width W: == 2 (mod 32) so that the 16x4 MFMA A-operand
// read pattern (16 rows x 2 consecutive columns per 32-lane group) is bank-conflict free.
__host__ __device__ __forceinline__ int tile_stride(int W) { return ((W - 2 + 31) / 32) * 32 + 2; }
// LDS row stride for the GEMM scratch (B-operand reads: 2 rows x 16 columns per group)
__host__ __device__ __forceinline__ int scratch_stride(int wp) { return ((wp + 31) / 32) * 32 + 16; }
__host__ __device__ __forceinline__ int round16(int v) { return (v + 15) & ~15; }
