// bigk.h -- the gradient step of Blend.fit for MANY components per scene (8 < K <= 32).
//
// The register-tiled kernels of engine.h keep sed[K][B], d loss/d sed[K][B] and the K (K+1)/2
// Gram accumulators per thread, which stops at K = 8.  Crowded scenes (BASELINE config 5: 30
// overlapping sources) run the same mathematics in passes over component chunks of eight:
//
//   k_bigk_resid      grid (T, S)        model, residual, loss; G_b = w^2 (model_b - image_b)
//                                        written once to a scratch plane set [S][B][HW]   [a1-a5]
//   k_bigk_gram       grid (T, pairs, S) 8 x 8 block of the morphology Gram S S^T per
//                                        chunk pair                                        [a6]
//   k_bigk_lipschitz  grid (S)           lambda_max(S S^T) by repeated squaring + Rayleigh
//                                        quotient on one wave, lambda_max(A^T A) by Jacobi [a6]
//   k_bigk_step       grid (T, chunks, S) d loss/d sed partials and the morphology step    [a5, a7]
//   k_bigk_sed        grid (S)           SED step                                          [a7]
//
// Partial sums use the layout of engine.h (`partials[S][T][P]`, P = 1 + K B + K (K+1) / 2), so
// the constraint and convergence kernels that follow are the same as for small K.
//
// Round 2: with exact Lipschitz constants and planes of a multiple of 64 pixels the five passes become
//   k_bigk_lmorph     grid (S)           lambda_max(A^T A): all the morphology step needs
//   k_bigk_fused      grid (T, S)        passes 1 + 4 in ONE pass over the morphologies on the matrix cores
//                                        (model, residual, loss, step, d loss/d sed sums; no residual planes)
//   k_bigk_gram_mfma  grid (T, S)        pass 2 as a GEMM over pixels, one pass instead of one per chunk pair
//   k_bigk_lipschitz  grid (S)           lambda_max(S S^T) only (sed_only), beside the fused pass on a second stream
//   k_bigk_sed        grid (S)           SED step + the loss record
// (scarlet_hip.hip: backward_impl; the chunked passes remain for approximate constants, other plane sizes, the PSF path
// and behind the NO_BIGK_FUSED / NO_GRAM_MFMA switches).
#pragma once
#include "common.h"
#include "engine.h"

#define SC_CHUNK 8

// ---- pass 1: model, residual, loss
__global__ __launch_bounds__(SC_BLOCK) void k_bigk_resid(GradArgs a, float *resid)
{
    const int s = blockIdx.y, tile = blockIdx.x;
    if (!a.active[s]) return;
    const int K = a.K, B = a.B, HW = a.HW;
    __shared__ float sed_s[SC_KBIG * SC_BMAX];
    __shared__ double red[SC_NWAVES];
    const int c0 = a.cur[s];
    for (int i = threadIdx.x; i < K * B; i += SC_BLOCK)
        sed_s[(i / B) * SC_BMAX + (i % B)] = a.sed[c0][(size_t)s * K * B + i];
    __syncthreads();
    const float *img = a.images + (size_t)s * B * HW;
    const float *wgt = a.weights ? a.weights + (size_t)s * B * HW : nullptr;
    const float *mor = a.morph[c0] + (size_t)s * K * HW;
    float *G = resid + (size_t)s * B * HW;
    double loss = 0;
    const int p_end = min(HW, (tile + 1) * SC_TILE_PIX);
    if ((HW & 3) == 0) {
        // 16 B per lane on every stream (K morphologies, B images [, B weights], B planes of G out)
        const int HW4 = HW >> 2, g_end = p_end >> 2;
        const float4 *mor4 = reinterpret_cast<const float4 *>(mor), *img4 = reinterpret_cast<const float4 *>(img);
        const float4 *wgt4 = reinterpret_cast<const float4 *>(wgt);
        float4 *G4 = reinterpret_cast<float4 *>(G);
        for (int g = tile * (SC_TILE_PIX >> 2) + threadIdx.x; g < g_end; g += SC_BLOCK) {
            float4 model[SC_BMAX];
#pragma unroll
            for (int b = 0; b < SC_BMAX; ++b) model[b] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < K; ++k) {
                const float4 m = mor4[(size_t)k * HW4 + g];
#pragma unroll
                for (int b = 0; b < SC_BMAX; ++b)
                    if (b < B) {
                        const float sk = sed_s[k * SC_BMAX + b];
                        model[b].x += sk * m.x; model[b].y += sk * m.y; model[b].z += sk * m.z; model[b].w += sk * m.w;
                    }
            }
#pragma unroll
            for (int b = 0; b < SC_BMAX; ++b)
                if (b < B) {
                    const float4 im = img4[(size_t)b * HW4 + g];
                    const float4 w = wgt ? wgt4[(size_t)b * HW4 + g] : make_float4(a.weight_scalar, a.weight_scalar, a.weight_scalar, a.weight_scalar);
                    const float d0 = w.x * (model[b].x - im.x), d1 = w.y * (model[b].y - im.y);
                    const float d2 = w.z * (model[b].z - im.z), d3 = w.w * (model[b].w - im.w);
                    loss += (double)d0 * (double)d0; loss += (double)d1 * (double)d1;
                    loss += (double)d2 * (double)d2; loss += (double)d3 * (double)d3;
                    G4[(size_t)b * HW4 + g] = make_float4(w.x * d0, w.y * d1, w.z * d2, w.w * d3);
                }
        }
    } else
    for (int p = tile * SC_TILE_PIX + threadIdx.x; p < p_end; p += SC_BLOCK) {
        float model[SC_BMAX];
#pragma unroll
        for (int b = 0; b < SC_BMAX; ++b) model[b] = 0.f;
        for (int k = 0; k < K; ++k) {
            const float m = mor[(size_t)k * HW + p];
#pragma unroll
            for (int b = 0; b < SC_BMAX; ++b)
                if (b < B) model[b] += sed_s[k * SC_BMAX + b] * m;
        }
#pragma unroll
        for (int b = 0; b < SC_BMAX; ++b)
            if (b < B) {
                const float w = wgt ? wgt[(size_t)b * HW + p] : a.weight_scalar;
                const float d = w * (model[b] - img[(size_t)b * HW + p]);
                loss += (double)d * (double)d;
                G[(size_t)b * HW + p] = w * d;
            }
    }
    loss = block_sum(0.5 * loss, red);
    if (threadIdx.x == 0) a.partials[((size_t)s * a.T + tile) * n_partials(K, B)] = loss;
}

// sum of 64 per-thread values over the workgroup; result[i] valid in threads i < 64
__device__ __forceinline__ double block_sum64(float (&acc)[64], float (*red)[64])
{
    wave_sum_lastrow(acc);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == SC_WAVE - 1) {
#pragma unroll
        for (int i = 0; i < 64; ++i) red[wid][i] = acc[i];
    }
    __syncthreads();
    double r = 0;
    if (threadIdx.x < 64) {
#pragma unroll
        for (int w = 0; w < SC_NWAVES; ++w) r += (double)red[w][threadIdx.x];
    }
    return r;
}

// ---- pass 2: Gram blocks.  blockIdx.y enumerates chunk pairs (c1 <= c2).
__global__ __launch_bounds__(SC_BLOCK) void k_bigk_gram(GradArgs a)
{
    const int s = blockIdx.z, tile = blockIdx.x;
    if (!a.active[s]) return;
    const int K = a.K, B = a.B, HW = a.HW;
    const int nch = (K + SC_CHUNK - 1) / SC_CHUNK;
    int c1 = 0, rem = blockIdx.y;
    while (rem >= nch - c1) { rem -= nch - c1; ++c1; }
    const int c2 = c1 + rem;
    __shared__ float red[SC_NWAVES][64];
    const float *mor = a.morph[a.cur[s]] + (size_t)s * K * HW;
    float acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.f;
    const int p_end = min(HW, (tile + 1) * SC_TILE_PIX);
    // (a 16 B per lane form of this loop was measured slower: 128 live accumulators and operands spill)
    for (int p = tile * SC_TILE_PIX + threadIdx.x; p < p_end; p += SC_BLOCK) {
        float m1[SC_CHUNK], m2[SC_CHUNK];
#pragma unroll
        for (int i = 0; i < SC_CHUNK; ++i) {
            const int k = c1 * SC_CHUNK + i, k2 = c2 * SC_CHUNK + i;
            m1[i] = k < K ? mor[(size_t)k * HW + p] : 0.f;
            m2[i] = k2 < K ? mor[(size_t)k2 * HW + p] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < SC_CHUNK; ++i)
#pragma unroll
            for (int j = 0; j < SC_CHUNK; ++j) acc[i * SC_CHUNK + j] += m1[i] * m2[j];
    }
    const double r = block_sum64(acc, red);
    if (threadIdx.x < 64) {
        const int k = c1 * SC_CHUNK + (threadIdx.x >> 3), k2 = c2 * SC_CHUNK + (threadIdx.x & 7);
        if (k < K && k2 < K && k <= k2) {
            const int go = k * K - (k * (k - 1)) / 2 + (k2 - k);        // packed upper triangle
            a.partials[((size_t)s * a.T + tile) * n_partials(K, B) + 1 + K * B + go] = r;
        }
    }
}

// ---- pass 2 on the matrix cores: the whole K x K Gram block of a tile in ONE pass over the morphologies.
// S S^T is a GEMM whose reduction index is the pixel: v_mfma_f32_32x32x2_f32 takes A[i][k] = B[k][i] =
// m_i(pixel k) from the SAME register -- lane (component i = lane & 31, half h = lane >> 5) loads 16 consecutive
// pixels of its component (4 x 16 B; the two halves of a component make a 128-byte line) and issues one MFMA per
// pixel pair.  Every morphology value is loaded once instead of once per chunk pair (4 x for K = 30), and the 64
// accumulators per thread of the chunk-pair form become 16.  float32 accumulation inside the MFMA is flushed to
// float64 every 256 pixels per wave (the chunk-pair form sums 64 float32 products per thread before its float64
// tree: the same class of rounding).  HW % 4 == 0, K <= 32.  grid (T, S).
typedef float bigk_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(SC_BLOCK) void k_bigk_gram_mfma(GradArgs a)
{
    const int s = blockIdx.y, tile = blockIdx.x;
    if (!a.active[s]) return;
    const int K = a.K, B = a.B, HW = a.HW;
    __shared__ double red[SC_NWAVES][16][SC_WAVE];                 // 32 KB
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, comp = lane & 31, half = lane >> 5;
    const float *mor = a.morph[a.cur[s]] + (size_t)s * K * HW + (size_t)(comp < K ? comp : 0) * HW;
    const int p_end = min(HW, (tile + 1) * SC_TILE_PIX);
    constexpr int WPIX = SC_TILE_PIX / SC_NWAVES, NBATCH = WPIX / 32;   // 1024 pixels per wave in 32 batches of 32
    const int p0 = tile * SC_TILE_PIX + wid * WPIX + half * 16;
    auto load = [&](int j, float4 (&v)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = p0 + j * 32 + 4 * q;
            v[q] = (comp < K && p < p_end) ? *reinterpret_cast<const float4 *>(mor + p) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    double accd[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) accd[r] = 0.0;
    bigk_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 cur[4], nxt[4];
    load(0, cur);
#pragma unroll 1
    for (int j = 0; j < NBATCH; ++j) {
        if (j + 1 < NBATCH) load(j + 1, nxt);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[q].x, cur[q].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[q].y, cur[q].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[q].z, cur[q].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[q].w, cur[q].w, acc, 0, 0, 0);
        }
        if ((j & 7) == 7) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { accd[r] += (double)acc[r]; acc[r] = 0.f; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
    }
    // C layout: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wid][r][lane] = accd[r];
    __syncthreads();
    const int P = n_partials(K, B);
    double *out = a.partials + ((size_t)s * a.T + tile) * P + 1 + K * B;
    for (int e = threadIdx.x; e < 16 * SC_WAVE; e += SC_BLOCK) {
        const int r = e >> 6, l = e & 63;
        const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        if (row <= col && col < K) {
            double v = 0;
#pragma unroll
            for (int w = 0; w < SC_NWAVES; ++w) v += red[w][r][l];
            out[row * K - (row * (row - 1)) / 2 + (col - row)] = v;
        }
    }
}

// ---- passes 1 + 4 in ONE pass over the morphologies, on the matrix cores (exact Lipschitz constants, HW % 64 == 0).
// Everything the gradient step does at a pixel needs that pixel only:
//     model_b = sum_k sed[k][b] m_k        G_b = w^2 (model_b - image_b)        m_k <- m_k - step sum_b sed[k][b] G_b
// so the residual planes need not exist in memory: a wave takes 32 consecutive pixels and runs three small GEMMs on
// v_mfma_f32_32x32x2_f32 whose operand / result layouts chain without moving data between lanes:
//   model (bands x pixels) = sed^T (bands x comps) . M (comps x pixels), 16 steps of two components.  Lane (pixel p,
//       half h) supplies at step t the component kappa(t, h) = (t & 3) + 8 (t >> 2) + 4 h -- exactly the row that
//       register t of a 32 x 32 RESULT holds in that lane --, and receives the model of bands 4 h .. 4 h + 3 at its
//       pixel in result registers 0 .. 3;
//   gm (comps x pixels) = sed (comps x bands) . G (bands x pixels), 4 steps of two bands: at step t the halves supply
//       bands t and 4 + t, which are the G values the lane has just computed; result register t is the gradient of
//       component kappa(t, h) at the lane's pixel -- next to m_kappa(t, h), loaded for the first product;
//   dsed (comps x bands) = M (comps x pixels) . G^T (pixels x bands): the reduction runs over pixels, the lanes must
//       hold components: the 32 x 32 tile of M and the 8 x 32 tile of G go through LDS once (wave-private).
// Against k_bigk_resid + k_bigk_step: no G planes (B written + 4 B read per chunk), the morphologies read once instead
// of twice.  The Gram matrix stays with k_bigk_gram_mfma on the second stream.  grid (T, S), four waves of 1024 pixels.
__global__ __launch_bounds__(SC_BLOCK, 2) void k_bigk_fused(GradArgs a)
{
    const int s = blockIdx.y, tile = blockIdx.x;
    if (!a.active[s]) return;
    const int K = a.K, B = a.B, HW = a.HW;
    // A wave takes 64 consecutive pixels per trip as TWO interleaved tiles of 32 (lane p holds pixels 2 p and 2 p + 1:
    // 8-byte loads and stores, and two independent MFMA chains in every product)
    constexpr int LDT = 68;                                             // row stride of the transposed tiles (64 + 4)
    __shared__ __align__(16) float Mt[SC_NWAVES][32][LDT];
    __shared__ __align__(16) float Gt[SC_NWAVES][8][LDT];
    __shared__ double dred[SC_NWAVES][256];
    __shared__ double red[SC_NWAVES];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, p = lane & 31, h = lane >> 5;
    const int c0 = a.cur[s];
    const float *sed = a.sed[c0] + (size_t)s * K * B;
    const float step_morph = 1.0f / (float)a.lipschitz[2 * s + 1];
    // operand constants: sed^T for the model (row = band = lane & 31, slot h <-> component kappa(t, h)) and sed for the
    // gradient (row = component = lane & 31, slot h <-> band 4 h + t)
    float asedT[16], ased[4];
    unsigned fixmask = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int kap = (t & 3) + 8 * (t >> 2) + 4 * h;
        asedT[t] = (kap < K && p < B) ? sed[kap * B + p] : 0.f;
        if (kap < K && a.fix_morph && a.fix_morph[(size_t)s * K + kap]) fixmask |= 1u << t;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) ased[t] = (p < K && 4 * h + t < B) ? sed[p * B + 4 * h + t] : 0.f;
    const float *mor = a.morph[c0] + (size_t)s * K * HW;
    float *mout = a.morph[1 - c0] + (size_t)s * K * HW;
    const float *img = a.images + (size_t)s * B * HW;
    const float *wgt = a.weights ? a.weights + (size_t)s * B * HW : nullptr;
    const int p_end = min(HW, (tile + 1) * SC_TILE_PIX);
    constexpr int WPIX = SC_TILE_PIX / SC_NWAVES, NTRIP = WPIX / 64;
    const int pw = tile * SC_TILE_PIX + wid * WPIX;
    double loss = 0;
    bigk_f32x16 accD0, accD1;                                           // dsed, float32 over the wave's 1024 pixels
#pragma unroll
    for (int r = 0; r < 16; ++r) { accD0[r] = 0.f; accD1[r] = 0.f; }
#pragma unroll 1
    for (int j = 0; j < NTRIP; ++j) {
        if (pw + j * 64 >= p_end) break;                               // (uniform: tiles end on multiples of 64)
        const int px = pw + j * 64 + 2 * p;
        float mA[16], mB[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int kap = (t & 3) + 8 * (t >> 2) + 4 * h;
            const float2 v = kap < K ? *reinterpret_cast<const float2 *>(mor + (size_t)kap * HW + px) : make_float2(0.f, 0.f);
            mA[t] = v.x; mB[t] = v.y;
        }
        float2 im[4], w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = 4 * h + r;
            im[r] = b < B ? *reinterpret_cast<const float2 *>(img + (size_t)b * HW + px) : make_float2(0.f, 0.f);
            w[r] = b < B ? (wgt ? *reinterpret_cast<const float2 *>(wgt + (size_t)b * HW + px) : make_float2(a.weight_scalar, a.weight_scalar))
                         : make_float2(0.f, 0.f);
        }
        bigk_f32x16 accA, accB;
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA[r] = 0.f; accB[r] = 0.f; }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            accA = __builtin_amdgcn_mfma_f32_32x32x2f32(asedT[t], mA[t], accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(asedT[t], mB[t], accB, 0, 0, 0);
        }
        float gA[4], gB[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dA = w[r].x * (accA[r] - im[r].x), dB = w[r].y * (accB[r] - im[r].y);   // (bands >= B: w = 0)
            loss += (double)dA * (double)dA; loss += (double)dB * (double)dB;
            gA[r] = w[r].x * dA; gB[r] = w[r].y * dB;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA[r] = 0.f; accB[r] = 0.f; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            accA = __builtin_amdgcn_mfma_f32_32x32x2f32(ased[t], gA[t], accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(ased[t], gB[t], accB, 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int kap = (t & 3) + 8 * (t >> 2) + 4 * h;
            const bool fx = (fixmask >> t) & 1u;
            const float oA = a.raw_gradient ? accA[t] : (fx ? mA[t] : mA[t] - step_morph * accA[t]);
            const float oB = a.raw_gradient ? accB[t] : (fx ? mB[t] : mB[t] - step_morph * accB[t]);
            if (kap < K) *reinterpret_cast<float2 *>(mout + (size_t)kap * HW + px) = make_float2(oA, oB);
            *reinterpret_cast<float2 *>(&Mt[wid][kap][2 * p]) = make_float2(mA[t], mB[t]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<float2 *>(&Gt[wid][4 * h + r][2 * p]) = make_float2(gA[r], gB[r]);
        wave_sync();
        // dsed: lane (row lane & 31, half h) takes pixels 32 h .. 32 h + 31 of its row, two chains of 16 pixel pairs
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float am[16], bg[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(&Mt[wid][p][32 * h + 16 * half + 4 * q]);
                am[4 * q] = v.x; am[4 * q + 1] = v.y; am[4 * q + 2] = v.z; am[4 * q + 3] = v.w;
                const float4 u = p < 8 ? *reinterpret_cast<const float4 *>(&Gt[wid][p & 7][32 * h + 16 * half + 4 * q]) : make_float4(0.f, 0.f, 0.f, 0.f);
                bg[4 * q] = u.x; bg[4 * q + 1] = u.y; bg[4 * q + 2] = u.z; bg[4 * q + 3] = u.w;
            }
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                accD0 = __builtin_amdgcn_mfma_f32_32x32x2f32(am[q], bg[q], accD0, 0, 0, 0);
                accD1 = __builtin_amdgcn_mfma_f32_32x32x2f32(am[q + 1], bg[q + 1], accD1, 0, 0, 0);
            }
        }
        wave_sync();
    }
    // dsed: result column = band = lane & 31 (< 8 used), row = component (r & 3) + 8 (r >> 2) + 4 h
    if (p < 8) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dred[wid][((r & 3) + 8 * (r >> 2) + 4 * h) * 8 + p] = (double)accD0[r] + (double)accD1[r];
    }
    loss = block_sum(0.5 * loss, red);                                  // (synchronises: dred is complete after it)
    const int P = n_partials(K, B);
    double *out = a.partials + ((size_t)s * a.T + tile) * P;
    if (threadIdx.x == 0) out[0] = loss;
    for (int e = threadIdx.x; e < 256; e += SC_BLOCK) {
        const int k = e >> 3, b = e & 7;
        if (k < K && b < B) {
            double v = 0;
#pragma unroll
            for (int w2 = 0; w2 < SC_NWAVES; ++w2) v += dred[w2][e];
            out[1 + k * B + b] = v;
        }
    }
}

// ---- pass 3: Lipschitz constants, one wave per scene (blend.py:186-223)
// lambda_max of the PSD Gram matrix G (n <= 32): M = G / tr G is squared SC_SQUARINGS times
// (renormalised by its trace each time), which leaves u1 u1^T up to terms (lambda_i /
// lambda_1)^(2^SC_SQUARINGS); the Rayleigh quotient of its heaviest column with the ORIGINAL G
// then misses lambda_1 by at most n / (e 2^(SC_SQUARINGS + 1)) relative (< 1e-8), whatever the
// spectral gaps.  Each lane owns a 4 x 4 block of the 32 x 32 product.
#define SC_SQUARINGS 30
// One WORKGROUP per scene: the 30 dependent 32 x 32 x 32 float64 products are the whole cost of this pass (a
// single wave took 0.24 ms for BASELINE config 5 whatever the number of scenes); four waves share each product,
// a thread owns a 2 x 2 block, operands are read from LDS four k at a time.
// `sed_only` = 2: as 1, and the loss partials do not exist yet (exact constants only): no loss record from here.
// `sed_only` = 1: lambda_max(A^T A) comes from k_bigk_lmorph (the morphology step waits for that one only; this
// kernel then runs beside the step on a second stream, scarlet_hip.hip) and lipschitz[2 s + 1] is not written here.
__global__ __launch_bounds__(SC_BLOCK) void k_bigk_lipschitz(GradArgs a, int sed_only)
{
    const int s = blockIdx.x;
    if (!a.active[s]) return;
    const int K = a.K, B = a.B, P = n_partials(K, B);
    constexpr int LD = SC_KBIG + 1;
    __shared__ double Gm[SC_KBIG][LD], M0[SC_KBIG][LD], M1[SC_KBIG][LD];
    __shared__ double ata[SC_BMAX * SC_BMAX];
    __shared__ float sed_s[SC_KBIG * SC_BMAX];
    __shared__ double red[SC_NWAVES];
    const int tid = threadIdx.x, lane = tid & 63;
    const bool w0 = tid < SC_WAVE;                      // the scalar parts run on the first wave, as before
    const int c0 = a.cur[s];
    for (int i = tid; i < K * B; i += SC_BLOCK)
        sed_s[(i / B) * SC_BMAX + (i % B)] = a.sed[c0][(size_t)s * K * B + i];
    for (int i = tid; i < SC_KBIG * SC_KBIG; i += SC_BLOCK) {
        const int k = i / SC_KBIG, k2 = i - k * SC_KBIG;
        double r = 0;
        if (k < K && k2 < K) {
            const int lo = k < k2 ? k : k2, hi = k < k2 ? k2 : k;
            const int go = lo * K - (lo * (lo - 1)) / 2 + (hi - lo);
            for (int t = 0; t < a.T; ++t) r += a.partials[((size_t)s * a.T + t) * P + 1 + K * B + go];
        }
        Gm[k][k2] = r;
    }
    double loss = 0;
    if (w0 && sed_only != 2) for (int t = 0; t < a.T; ++t) loss += a.partials[((size_t)s * a.T + t) * P];
    __syncthreads();
    double trace = 0;
    if (tid < SC_KBIG) trace = Gm[tid][tid];
    trace = block_sum(trace, red);
    const int it_new = a.it[s] + 1;
    double L_sed = 0, L_morph = 0;
    if (a.approximate_L) {
        double LS = 0;
        for (int i = tid; i < K * B; i += SC_BLOCK) { const float v = sed_s[(i / B) * SC_BMAX + (i % B)]; LS += (double)v * v; }
        LS = block_sum(LS, red);
        double LA = trace;
        if (it_new > 1 && loss > a.mse[(size_t)s * a.mse_capacity + it_new - 2]) { LA *= 2; LS *= 2; }
        L_sed = LA; L_morph = LS;          // (loss is valid on the first wave, which is the one that writes)
    } else {
        // ---- lambda_max(S S^T)
        const double inv = 1.0 / trace;
        for (int i = tid; i < SC_KBIG * SC_KBIG; i += SC_BLOCK) M0[i / SC_KBIG][i % SC_KBIG] = Gm[i / SC_KBIG][i % SC_KBIG] * inv;
        __syncthreads();
        const int bi = (tid >> 4) << 1, bj = (tid & 15) << 1;
        double (*src)[LD] = M0, (*dst)[LD] = M1;
        for (int q = 0; q < SC_SQUARINGS; ++q) {
            double a00 = 0, a01 = 0, a10 = 0, a11 = 0;
#pragma unroll 2
            for (int k = 0; k < SC_KBIG; k += 4) {
                double u0[4], u1[4], v0[4], v1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { u0[e] = src[bi][k + e]; u1[e] = src[bi + 1][k + e]; v0[e] = src[k + e][bj]; v1[e] = src[k + e][bj + 1]; }
#pragma unroll
                for (int e = 0; e < 4; ++e) { a00 += u0[e] * v0[e]; a01 += u0[e] * v1[e]; a10 += u1[e] * v0[e]; a11 += u1[e] * v1[e]; }
            }
            // renormalised by the trace every fourth squaring: lambda_1 >= 1/32 of the trace, so four squarings
            // without it shrink the dominant entries to no less than 2^-80 (float64: no underflow)
            double sc = 1.0;
            if ((q & 3) == 3 || q == SC_SQUARINGS - 1) {
                double tr = (bi == bj) ? a00 + a11 : 0.0;
                tr = block_sum(tr, red);
                sc = 1.0 / tr;
            }
            dst[bi][bj] = a00 * sc; dst[bi][bj + 1] = a01 * sc; dst[bi + 1][bj] = a10 * sc; dst[bi + 1][bj + 1] = a11 * sc;
            __syncthreads();
            double (*tmp)[LD] = src; src = dst; dst = tmp;
        }
        if (w0) {
            // heaviest column of the (rank-one) power, Rayleigh quotient with G
            double best = lane < SC_KBIG ? src[lane][lane] : -1.0;
            int bidx = lane;
            for (int o = 32; o > 0; o >>= 1) {
                const double b2 = __shfl_xor(best, o, SC_WAVE);
                const int i2 = __shfl_xor(bidx, o, SC_WAVE);
                if (b2 > best || (b2 == best && i2 < bidx)) { best = b2; bidx = i2; }
            }
            double v = lane < SC_KBIG ? src[lane][bidx] : 0.0, gv = 0;
            if (lane < SC_KBIG) dst[0][lane] = v;
            wave_sync();
            if (lane < SC_KBIG)
                for (int j = 0; j < SC_KBIG; ++j) gv += Gm[lane][j] * dst[0][j];
            const double num = wave_sum(v * gv), den = wave_sum(v * v);
            L_sed = num / den;
            // ---- lambda_max(A^T A): the smaller of the two Gram matrices of the SED matrix (<= 8 x 8)
            const int n = sed_only ? 0 : (K < B ? K : B);
            for (int i = lane; i < n * n; i += SC_WAVE) {
                const int x = i / n, y = i - x * n;
                double r = 0;
                if (K < B) for (int b = 0; b < B; ++b) r += (double)sed_s[x * SC_BMAX + b] * sed_s[y * SC_BMAX + b];
                else       for (int k = 0; k < K; ++k) r += (double)sed_s[k * SC_BMAX + x] * sed_s[k * SC_BMAX + y];
                ata[x * n + y] = r;
            }
            wave_sync();
            if (lane == 0 && !sed_only) L_morph = jacobi_lambda_max(ata, n, n);
        }
    }
    if (tid == 0) {
        if (sed_only != 2 && it_new <= a.mse_capacity) a.mse[(size_t)s * a.mse_capacity + it_new - 1] = loss;
        a.lipschitz[2 * s] = L_sed;
        if (!sed_only) a.lipschitz[2 * s + 1] = L_morph;
    }
}

// lambda_max(A^T A) alone (blend.py:205-218), one wave per scene: all the morphology step needs.  Power iteration
// by repeated squaring on the wave (wave_lambda_max8) instead of the single-lane Jacobi above: ~10 k cycles.
__global__ __launch_bounds__(SC_WAVE) void k_bigk_lmorph(GradArgs a)
{
    const int s = blockIdx.x;
    if (!a.active[s]) return;
    const int K = a.K, B = a.B, P = n_partials(K, B), lane = threadIdx.x;
    __shared__ double ata[SC_BMAX * SC_BMAX];
    __shared__ float sed_s[SC_KBIG * SC_BMAX];
    __shared__ double buf[2][64];
    const int c0 = a.cur[s];
    for (int i = lane; i < K * B; i += SC_WAVE)
        sed_s[(i / B) * SC_BMAX + (i % B)] = a.sed[c0][(size_t)s * K * B + i];
    wave_sync();
    double L_morph;
    if (a.approximate_L) {
        double LS = 0, loss = 0;
        for (int i = lane; i < K * B; i += SC_WAVE) { const float v = sed_s[(i / B) * SC_BMAX + (i % B)]; LS += (double)v * v; }
        LS = wave_sum(LS);
        for (int t = 0; t < a.T; ++t) loss += a.partials[((size_t)s * a.T + t) * P];
        const int it_new = a.it[s] + 1;
        if (it_new > 1 && loss > a.mse[(size_t)s * a.mse_capacity + it_new - 2]) LS *= 2;
        L_morph = LS;
    } else {
        const int n = K < B ? K : B;
        for (int i = lane; i < n * n; i += SC_WAVE) {
            const int x = i / n, y = i - x * n;
            double r = 0;
            if (K < B) for (int b = 0; b < B; ++b) r += (double)sed_s[x * SC_BMAX + b] * sed_s[y * SC_BMAX + b];
            else       for (int k = 0; k < K; ++k) r += (double)sed_s[k * SC_BMAX + x] * sed_s[k * SC_BMAX + y];
            ata[x * n + y] = r;
        }
        wave_sync();
        L_morph = wave_lambda_max8(ata, n, n, buf);
    }
    if (lane == 0) a.lipschitz[2 * s + 1] = L_morph;
}

// ---- pass 4: d loss / d sed partials and the morphology step for one chunk of components
// BM: bands the instance is built for (B <= BM).  The chunk's SEDs are scalars (SGPRs) and the accumulators of
// absent bands do not exist: ~140 VGPRs (three waves per SIMD) instead of the 304 (one wave) of the first form,
// which held every SED in a vector register across the loop -- a streaming pass needs the waves.
template <int BM>
__global__ __launch_bounds__(SC_BLOCK, 2) void k_bigk_step(GradArgs a, const float *resid)
{
    const int s = blockIdx.z, tile = blockIdx.x, ch = blockIdx.y;
    if (!a.active[s]) return;
    const int K = a.K, B = a.B, HW = a.HW;
    __shared__ float sed_s[SC_CHUNK * SC_BMAX];
    __shared__ float red[SC_NWAVES][64];
    const int c0 = a.cur[s];
    if (threadIdx.x < SC_CHUNK * SC_BMAX) {
        const int k = ch * SC_CHUNK + (threadIdx.x >> 3), b = threadIdx.x & 7;
        sed_s[threadIdx.x] = (k < K && b < B) ? a.sed[c0][((size_t)s * K + k) * B + b] : 0.f;
    }
    __syncthreads();
    const float step_morph = 1.0f / (float)a.lipschitz[2 * s + 1];
    const float *mor = a.morph[c0] + (size_t)s * K * HW;
    float *mout = a.morph[1 - c0] + (size_t)s * K * HW;
    const float *G = resid + (size_t)s * B * HW;
    bool fixm[SC_CHUNK];
#pragma unroll
    for (int i = 0; i < SC_CHUNK; ++i) {
        const int k = ch * SC_CHUNK + i;
        fixm[i] = k < K && a.fix_morph && a.fix_morph[(size_t)s * K + k];
    }
    float sk[SC_CHUNK][BM];
#pragma unroll
    for (int i = 0; i < SC_CHUNK; ++i)
#pragma unroll
        for (int b = 0; b < BM; ++b)
            sk[i][b] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sed_s[i * SC_BMAX + b])));
    float acc[64];                          // dsed[i][b], i = component of the chunk (b >= BM: stays 0)
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.f;
    const int p_end = min(HW, (tile + 1) * SC_TILE_PIX);
    if ((HW & 3) == 0) {
        const int HW4 = HW >> 2, g_end = p_end >> 2;
        const float4 *mor4 = reinterpret_cast<const float4 *>(mor), *G4 = reinterpret_cast<const float4 *>(G);
        float4 *mout4 = reinterpret_cast<float4 *>(mout);
        for (int g = tile * (SC_TILE_PIX >> 2) + threadIdx.x; g < g_end; g += SC_BLOCK) {
            float4 gb[BM];
#pragma unroll
            for (int b = 0; b < BM; ++b) gb[b] = b < B ? G4[(size_t)b * HW4 + g] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < SC_CHUNK; ++i) {
                const int k = ch * SC_CHUNK + i;
                if (k < K) {
                    const float4 m = mor4[(size_t)k * HW4 + g];
                    float4 gm = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int b = 0; b < BM; ++b) {
                        float r = acc[i * SC_BMAX + b];
                        r += gb[b].x * m.x; r += gb[b].y * m.y; r += gb[b].z * m.z; r += gb[b].w * m.w;
                        acc[i * SC_BMAX + b] = r;
                        gm.x += sk[i][b] * gb[b].x; gm.y += sk[i][b] * gb[b].y; gm.z += sk[i][b] * gb[b].z; gm.w += sk[i][b] * gb[b].w;
                    }
                    float4 o;
                    if (a.raw_gradient) o = gm;
                    else if (fixm[i]) o = m;
                    else o = make_float4(m.x - step_morph * gm.x, m.y - step_morph * gm.y, m.z - step_morph * gm.z, m.w - step_morph * gm.w);
                    mout4[(size_t)k * HW4 + g] = o;
                }
            }
        }
    } else
    for (int p = tile * SC_TILE_PIX + threadIdx.x; p < p_end; p += SC_BLOCK) {
        float gb[BM];
#pragma unroll
        for (int b = 0; b < BM; ++b) gb[b] = b < B ? G[(size_t)b * HW + p] : 0.f;
#pragma unroll
        for (int i = 0; i < SC_CHUNK; ++i) {
            const int k = ch * SC_CHUNK + i;
            if (k < K) {
                const float m = mor[(size_t)k * HW + p];
                float gm = 0.f;
#pragma unroll
                for (int b = 0; b < BM; ++b) {
                    acc[i * SC_BMAX + b] += gb[b] * m;
                    gm += sk[i][b] * gb[b];
                }
                mout[(size_t)k * HW + p] = a.raw_gradient ? gm : (fixm[i] ? m : m - step_morph * gm);
            }
        }
    }
    const double r = block_sum64(acc, red);
    if (threadIdx.x < 64) {
        const int k = ch * SC_CHUNK + (threadIdx.x >> 3), b = threadIdx.x & 7;
        if (k < K && b < B) a.partials[((size_t)s * a.T + tile) * n_partials(K, B) + 1 + k * B + b] = r;
    }
}

// ---- pass 5: SED step (blend.py:91-93)
// `write_mse`: the loss record of the iteration is written here (k_bigk_lipschitz ran with sed_only = 2, before the
// loss existed)
__global__ __launch_bounds__(SC_BLOCK) void k_bigk_sed(GradArgs a, int write_mse)
{
    const int s = blockIdx.x;
    if (!a.active[s]) return;
    const int K = a.K, B = a.B, P = n_partials(K, B);
    const int c0 = a.cur[s];
    if (write_mse && threadIdx.x == SC_BLOCK - 1) {
        double loss = 0;
        for (int t = 0; t < a.T; ++t) loss += a.partials[((size_t)s * a.T + t) * P];
        const int it_new = a.it[s] + 1;
        if (it_new <= a.mse_capacity) a.mse[(size_t)s * a.mse_capacity + it_new - 1] = loss;
    }
    const float step_sed = 1.0f / (float)a.lipschitz[2 * s];
    for (int i = threadIdx.x; i < K * B; i += SC_BLOCK) {
        double g = 0;
        for (int t = 0; t < a.T; ++t) g += a.partials[((size_t)s * a.T + t) * P + 1 + i];
        const float cur = a.sed[c0][(size_t)s * K * B + i];
        const bool fixed = a.fix_sed && a.fix_sed[(size_t)s * K + i / B];
        a.sed[1 - c0][(size_t)s * K * B + i] = a.raw_gradient ? (float)g : (fixed ? cur : cur - step_sed * (float)g);
    }
}

// PSF path with K > 8: the loss of the scene comes from k_psf_resid's per-plane sums; it rides in
// slot 0 of tile 0 like everywhere else
__global__ void k_bigk_loss_from_planes(GradArgs a, const double *loss_part)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.S || !a.active[s]) return;
    const int P = n_partials(a.K, a.B);
    double l = 0;
    for (int b = 0; b < a.B; ++b) l += loss_part[s * a.B + b];
    for (int t = 0; t < a.T; ++t) a.partials[((size_t)s * a.T + t) * P] = t == 0 ? l : 0.0;
}
