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
// All-lane reductions over the 64-lane wave.  Within a 16-lane DPP row: two quad permutes
// (lane^1, lane^2), row_half_mirror and row_mirror -- VALU-rate moves, no LDS crossbar;
// across the four rows: v_readlane into SGPRs.  (__shfl_xor lowers to ds_bpermute, ~60+
// cycles of latency per step on the critical path of every phase end.)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float read_lane(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ double read_lane(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// wave-uniform values that the compiler cannot prove uniform (they come out of LDS / shuffles):
// moving them to SGPRs turns every branch on them into a scalar branch (no exec masking, no
// accumulator copies around MFMA chains)
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uniform(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                            __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
#define SC_DPP_XOR1 0xB1        // quad_perm [1,0,3,2]
#define SC_DPP_XOR2 0x4E        // quad_perm [2,3,0,1]
#define SC_DPP_HALF_MIRROR 0x141
#define SC_DPP_MIRROR 0x140
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
    v += dpp_mov<SC_DPP_XOR1>(v);
    v += dpp_mov<SC_DPP_XOR2>(v);
    v += dpp_mov<SC_DPP_HALF_MIRROR>(v);
    v += dpp_mov<SC_DPP_MIRROR>(v);
    return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<SC_DPP_XOR1>(v));
    v = fmaxf(v, dpp_mov<SC_DPP_XOR2>(v));
    v = fmaxf(v, dpp_mov<SC_DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_mov<SC_DPP_MIRROR>(v));
    return fmaxf(fmaxf(read_lane(v, 0), read_lane(v, 16)), fmaxf(read_lane(v, 32), read_lane(v, 48)));
}

// Sum of N independent float values over the wave, entirely in DPP adds (six per value):
// xor-butterfly inside each 16-lane row, then row_bcast:15 / row_bcast:31 chain the four
// rows.  Only the lanes of the LAST row (48..63) hold the totals afterwards.
#define SC_DPP_BCAST15 0x142
#define SC_DPP_BCAST31 0x143
template <int N>
__device__ __forceinline__ void wave_sum_lastrow(float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov<SC_DPP_XOR1>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov<SC_DPP_XOR2>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov<SC_DPP_HALF_MIRROR>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov<SC_DPP_MIRROR>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i)     // rows 1 and 3 receive lane 15 of the row before
        v[i] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), SC_DPP_BCAST15, 0xa, 0xf, false));
#pragma unroll
    for (int i = 0; i < N; ++i)     // rows 2 and 3 receive lane 31 (= rows 0 + 1)
        v[i] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), SC_DPP_BCAST31, 0xc, 0xf, false));
}

// Sums of N independent float values over the 16-lane ROWS of the wave with a third of the DPP adds
// of wave_sum_lastrow: the first two butterfly steps (lane ^ 1, lane ^ 2) also halve the number of
// live values -- after a step each lane keeps only the values whose index bit equals its lane bit
// -- and the two remaining in-row steps (rotations by 4 and 8 lanes, which keep lane & 3) run on
// ceil(N / 4) registers.  Afterwards EVERY lane of row R holds, in out[m], the sum over row R of
// value 4 m + (lane & 3); the caller adds the four rows (they go to LDS as four partials).
// (3 N / 2 + 3 N / 4 + N / 2 instructions instead of 6 N.)
#define SC_DPP_ROR4 0x124
#define SC_DPP_ROR8 0x128
template <int N>
__device__ __forceinline__ void wave_rowsum_quads(const float (&v)[N], float (&out)[(N + 3) / 4]) {
    constexpr int N2 = (N + 1) / 2, N4 = (N + 3) / 4;
    const int lane = threadIdx.x & (SC_WAVE - 1);
    const bool b0 = lane & 1, b1 = lane & 2;
    float r[N2];
#pragma unroll
    for (int i = 0; i < N2; ++i) {
        const float a = v[2 * i] + dpp_mov<SC_DPP_XOR1>(v[2 * i]);
        if (2 * i + 1 < N) {
            const float b = v[2 * i + 1] + dpp_mov<SC_DPP_XOR1>(v[2 * i + 1]);
            r[i] = b0 ? b : a;
        } else r[i] = a;
    }
#pragma unroll
    for (int i = 0; i < N4; ++i) {
        const float a = r[2 * i] + dpp_mov<SC_DPP_XOR2>(r[2 * i]);
        if (2 * i + 1 < N2) {
            const float b = r[2 * i + 1] + dpp_mov<SC_DPP_XOR2>(r[2 * i + 1]);
            out[i] = b1 ? b : a;
        } else out[i] = a;
    }
#pragma unroll
    for (int i = 0; i < N4; ++i) out[i] += dpp_mov<SC_DPP_ROR4>(out[i]);
#pragma unroll
    for (int i = 0; i < N4; ++i) out[i] += dpp_mov<SC_DPP_ROR8>(out[i]);
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

// ---------------------------------------------------------------- float4 access to LDS tiles
__device__ __forceinline__ void lds_store4(float *p, float4 v)
{   // rows are 8-byte aligned (stride == 2 mod 32 floats): two 8-byte stores
    reinterpret_cast<float2 *>(p)[0] = make_float2(v.x, v.y);
    reinterpret_cast<float2 *>(p)[1] = make_float2(v.z, v.w);
}
__device__ __forceinline__ float4 lds_load4(const float *p)
{
    const float2 a = reinterpret_cast<const float2 *>(p)[0], b = reinterpret_cast<const float2 *>(p)[1];
    return make_float4(a.x, a.y, b.x, b.y);
}

// ---------------------------------------------------------------- fast FFT lengths
// next_fast_len(n) for n < SC_NFL_MAX, filled on the host at first use (scarlet_hip.hip)
#define SC_NFL_MAX 2304
extern __constant__ unsigned short sc_nfl_table[SC_NFL_MAX];

__device__ __forceinline__ int dev_next_fast_len(int n) { return sc_nfl_table[n]; }

// LDS row stride for an image of width W: == 2 (mod 32) so that the 16x4 MFMA A-operand
// read pattern (16 rows x 2 consecutive columns per 32-lane group) is bank-conflict free.
__host__ __device__ __forceinline__ int tile_stride(int W) { return ((W - 2 + 31) / 32) * 32 + 2; }
// Row stride of the exact-shape fused instance (k_iterate2<4,5,64>, k_fit2x): == 4 (mod 32).  The sweep walks the
// down / up wedges along (row - 1, column + 2): stride LW - 2, which is 0 (mod 32) for the stride above -- the eight
// walkers of a wedge side then hit ONE bank -- and 2 for this one; the GEMM 2 epilogue (rows 4 lq + r, column lr)
// becomes conflict-free as well (banks lr + 16 lq), and rows are 16-byte aligned.
#ifndef SC_XS_STRIDE
#define SC_XS_STRIDE 68
#endif
// LDS row stride for the GEMM scratch (B-operand reads: 2 rows x 16 columns per group)
__host__ __device__ __forceinline__ int scratch_stride(int wp) { return ((wp + 31) / 32) * 32 + 16; }
__host__ __device__ __forceinline__ int round16(int v) { return (v + 15) & ~15; }
