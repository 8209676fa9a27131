// fused.h -- one kernel per proximal-gradient iteration, one workgroup per scene.
//
// When the K morphologies of a scene fit in LDS (K * H * (W+2) * 4 B <= ~140 KiB, H, W <= 64,
// W % 4 == 0) the whole iteration of Blend.fit (blend.py:79-102) runs in one launch with the
// morphologies resident in LDS between the gradient step and the constraints:
//
//   phase 0  ALL global loads of the iteration are issued first (morph tiles and images,
//            16 B/lane); morph tiles -> LDS, morph Gram S S^T on the fly            [a6]
//   eig      Lipschitz constants: float32 Jacobi for the dominant eigenvector +
//            float64 Rayleigh quotient, one lane each for S S^T and A A^T; the image
//            loads are in flight meanwhile                                          [a6]
//   phase 1  model, residual, loss, G, SED gradient sums from the prefetched images;
//            morphology step applied in place in LDS                                [a1-a5, a7]
//   phase 2  wave k runs the constraint pipeline of component k on its LDS tile and
//            writes the result to the other HBM buffer                              [a8-a17]
//   phase 3  convergence flags, it/cur bookkeeping                                  [a18]
//
// HBM traffic per scene-iteration = images + morph read + morph write (+ a second,
// L2-resident read of the previous morph for the convergence sums) -- the algorithmic
// bytes of SURVEY.md 8d.  The unfused kernels of engine.h remain the general path
// (larger images, approximate_L).
#pragma once
#include "common.h"
#include "prox_ops.h"
#include "wave_ops.h"
#include "engine.h"

struct FusedArgs {
    int S, K, B, H, W;
    const float *images, *weights;
    float weight_scalar;
    float *sed[2], *morph[2];
    int *cur;
    const uint8_t *fix_sed, *fix_morph;
    int *centers; double *shifts; int *flags;
    double *lipschitz, *mse; int mse_capacity;
    int *it, *active, *status;
    int symmetric, monotonic;
    float l0_thresh, l1_thresh;
    const double *centroid_psf; int centroid_P;
    double e_rel2;
    long long *stamps;               // NULL, or [S][16] shader-clock stamps (diagnostics only)
};

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

// Largest eigenvalue of a symmetric PSD n x n matrix (n <= N <= 4) by ONE lane, all in
// registers: cyclic Jacobi in float32 accumulating the eigenvectors, then the Rayleigh
// quotient of the dominant eigenvector with the float64 matrix.  The quotient's error is
// quadratic in the eigenvector error, i.e. <= ~1e-7 relative in the worst (degenerate)
// case and ~1e-12 typically -- at float32 latencies instead of a float64 Jacobi chain.
// Replaces np.linalg.eigvals(...).max() of blend.py:216-218.
template <int N>
__device__ inline double lambda_max_rayleigh(const double *Ain, int n, int ld)
{
    float A[N][N], V[N][N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            A[i][j] = (i < n && j < n) ? (float)Ain[i * ld + j] : 0.f;
            V[i][j] = i == j ? 1.f : 0.f;
        }
    for (int sweep = 0; sweep < 8; ++sweep) {
        float off = 0.f, diag = 0.f;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            diag += A[i][i] * A[i][i];
#pragma unroll
            for (int j = i + 1; j < N; ++j) off += A[i][j] * A[i][j];
        }
        if (off <= 1e-13f * diag) break;
#pragma unroll
        for (int p = 0; p < N - 1; ++p)
#pragma unroll
            for (int q = p + 1; q < N; ++q) {
                const float apq = A[p][q];
                if (apq != 0.f) {
                    const float th = (A[q][q] - A[p][p]) / (2.f * apq);
                    const float tf = (th >= 0.f ? 1.f : -1.f) / (fabsf(th) + sqrtf(th * th + 1.f));
                    const float c = 1.0f / sqrtf(tf * tf + 1.f), sn = tf * c;
#pragma unroll
                    for (int r = 0; r < N; ++r) {
                        const float arp = A[r][p], arq = A[r][q];
                        A[r][p] = c * arp - sn * arq; A[r][q] = sn * arp + c * arq;
                    }
#pragma unroll
                    for (int r = 0; r < N; ++r) {
                        const float apr = A[p][r], aqr = A[q][r];
                        A[p][r] = c * apr - sn * aqr; A[q][r] = sn * apr + c * aqr;
                    }
#pragma unroll
                    for (int r = 0; r < N; ++r) {
                        const float vrp = V[r][p], vrq = V[r][q];
                        V[r][p] = c * vrp - sn * vrq; V[r][q] = sn * vrp + c * vrq;
                    }
                }
            }
    }
    int best = 0;
    float bestv = A[0][0];            // (static indexing only: dynamic indices would spill to scratch)
#pragma unroll
    for (int i = 1; i < N; ++i) if (A[i][i] > bestv) { bestv = A[i][i]; best = i; }
    double v[N];
#pragma unroll
    for (int r = 0; r < N; ++r) {
        float x = V[r][0];
#pragma unroll
        for (int j = 1; j < N; ++j) x = (best == j) ? V[r][j] : x;
        v[r] = (double)x;
    }
    double num = 0, den = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (i < n) {
            double row = 0;
#pragma unroll
            for (int j = 0; j < N; ++j) if (j < n) row += Ain[i * ld + j] * v[j];
            num += v[i] * row; den += v[i] * v[i];
        }
    }
    return num / den;
}

template <int KM, int BM>
__global__ __launch_bounds__(SC_BLOCK, 2) void k_iterate(FusedArgs a)
{
    extern __shared__ __align__(16) float lds[];
    const int s = blockIdx.x;
    if (!a.active[s]) return;
    const int K = a.K, B = a.B, H = a.H, W = a.W, HW = H * W, LW = tile_stride(W);
    const int tile_floats = H * LW;
    float *tiles = lds;
    float *vecs = lds + (size_t)K * tile_floats;
    constexpr int NG = KM * (KM + 1) / 2;
    constexpr int NP = 1 + KM * BM;
    constexpr int GPT = 4;                       // float4 groups per thread (H, W <= 64)
    constexpr bool PREFETCH = (KM <= 4);         // image prefetch needs 16*BM/4 more VGPRs
    __shared__ double red[SC_NWAVES][NP > NG ? NP : NG];
    __shared__ double tot[NP > NG ? NP : NG];
    __shared__ double mat[2][KM * KM > BM * BM ? KM * KM : BM * BM];
    __shared__ float sed_s[KM * BM], sed_new[KM * BM];
    __shared__ float step_s[2];
    __shared__ double conv_s[KM][4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c0 = a.cur[s];
    const float *min_g = a.morph[c0] + (size_t)s * K * HW;
    float *mout_g = a.morph[1 - c0] + (size_t)s * K * HW;
    const float *sed_in = a.sed[c0] + (size_t)s * K * B;
    float *sed_out = a.sed[1 - c0] + (size_t)s * K * B;
    const int it_new = a.it[s] + 1;
    const int ngroups = HW >> 2, gpr = W >> 2;           // float4 groups, groups per row
    const float *img = a.images + (size_t)s * B * HW;
    const float *wgt = a.weights ? a.weights + (size_t)s * B * HW : nullptr;
    // optional diagnostics: shader-clock stamps at the phase boundaries (never read back here)
#define STAMP(i) do { if (a.stamps && tid == 0) a.stamps[(size_t)s * 16 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    STAMP(0);

    // ---------------- phase 0: issue every global load, tiles -> LDS, Gram
    float4 mreg[GPT][KM];
#pragma unroll
    for (int j = 0; j < GPT; ++j) {
        const int g = tid + j * SC_BLOCK;
#pragma unroll
        for (int k = 0; k < KM; ++k)
            mreg[j][k] = (g < ngroups && k < K) ? reinterpret_cast<const float4 *>(min_g + (size_t)k * HW)[g]
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 ireg[PREFETCH ? GPT : 1][PREFETCH ? BM : 1];
    if (PREFETCH) {
#pragma unroll
        for (int j = 0; j < GPT; ++j) {
            const int g = tid + j * SC_BLOCK;
#pragma unroll
            for (int b = 0; b < BM; ++b)
                ireg[PREFETCH ? j : 0][PREFETCH ? b : 0] =
                    (g < ngroups && b < B) ? reinterpret_cast<const float4 *>(img + (size_t)b * HW)[g]
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    for (int i = tid; i < K * B; i += SC_BLOCK) sed_s[(i / B) * BM + (i % B)] = sed_in[i];
    float gram[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) gram[i] = 0.f;
#pragma unroll
    for (int j = 0; j < GPT; ++j) {
        const int g = tid + j * SC_BLOCK;
        if (g < ngroups) {
            const int y = g / gpr, x = (g - y * gpr) << 2;
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K) lds_store4(tiles + k * tile_floats + y * LW + x, mreg[j][k]);
            int gi = 0;
#pragma unroll
            for (int k = 0; k < KM; ++k)
#pragma unroll
                for (int k2 = k; k2 < KM; ++k2) {
                    const float4 p = mreg[j][k], q = mreg[j][k2];
                    gram[gi] += p.x * q.x + p.y * q.y + p.z * q.z + p.w * q.w;
                    ++gi;
                }
        }
    }
    {
        int gi = 0, go = 0;
#pragma unroll
        for (int k = 0; k < KM; ++k)
#pragma unroll
            for (int k2 = k; k2 < KM; ++k2) {
                if (k < K && k2 < K) {
                    const double v = wave_sum((double)gram[gi]);
                    if (lane == 0) red[wid][go] = v;
                    ++go;
                }
                ++gi;
            }
    }
    __syncthreads();
    STAMP(1);
    // assemble the two small Gram matrices in parallel (one entry per thread) ...
    const bool small_side = (K <= B);          // nonzero spectrum of A^T A == that of A A^T
    if (tid < K * K) {
        const int k = tid / K, k2 = tid - k * K;
        const int lo = k < k2 ? k : k2, hi = k < k2 ? k2 : k;
        const int go = lo * K - (lo * (lo - 1)) / 2 + (hi - lo);     // packed upper-triangle index
        double r = 0;
#pragma unroll
        for (int w = 0; w < SC_NWAVES; ++w) r += red[w][go];
        mat[0][k * KM + k2] = r;
    }
    if (tid >= SC_WAVE && tid < SC_WAVE + (small_side ? K * K : B * B)) {
        const int i = tid - SC_WAVE;
        double r = 0;
        if (small_side) {
            const int k = i / K, k2 = i - k * K;
            for (int b = 0; b < B; ++b) r += (double)sed_s[k * BM + b] * sed_s[k2 * BM + b];
            mat[1][k * KM + k2] = r;
        } else {
            const int b = i / B, b2 = i - b * B;
            for (int k = 0; k < K; ++k) r += (double)sed_s[k * BM + b] * sed_s[k * BM + b2];
            mat[1][b * BM + b2] = r;
        }
    }
    __syncthreads();
    // ... then one lane per matrix: blend.py:205-218, L_sed = lambda_max(S S^T) (wave 0),
    // L_morph = lambda_max(A^T A) (wave 1)
    if (tid == 0) {
        const double L = (KM <= 4) ? lambda_max_rayleigh<(KM <= 4 ? KM : 1)>(mat[0], K, KM)
                                   : jacobi_lambda_max(mat[0], K, KM);
        step_s[0] = 1.0f / (float)L;
        a.lipschitz[2 * s] = L;
    } else if (tid == SC_WAVE) {
        double L;
        if (small_side)
            L = (KM <= 4) ? lambda_max_rayleigh<(KM <= 4 ? KM : 1)>(mat[1], K, KM) : jacobi_lambda_max(mat[1], K, KM);
        else
            L = jacobi_lambda_max(mat[1], B, BM);
        step_s[1] = 1.0f / (float)L;
        a.lipschitz[2 * s + 1] = L;
    }
    __syncthreads();
    STAMP(2);
    const float step_sed = step_s[0], step_morph = step_s[1];

    // ---------------- phase 1: gradient + morphology step in LDS
    float sed[KM][BM];
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int b = 0; b < BM; ++b) sed[k][b] = (k < K && b < B) ? sed_s[k * BM + b] : 0.f;
    bool fixm[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) fixm[k] = (k < K) && a.fix_morph && a.fix_morph[(size_t)s * K + k];
    float dsed[KM][BM];
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int b = 0; b < BM; ++b) dsed[k][b] = 0.f;
    double loss = 0;
#pragma unroll
    for (int j = 0; j < GPT; ++j) {
        const int g = tid + j * SC_BLOCK;
        if (g < ngroups) {
            const int y = g / gpr, x = (g - y * gpr) << 2;
            float m[KM][4], gm[KM][4];
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < K) v = lds_load4(tiles + k * tile_floats + y * LW + x);
                m[k][0] = v.x; m[k][1] = v.y; m[k][2] = v.z; m[k][3] = v.w;
                gm[k][0] = gm[k][1] = gm[k][2] = gm[k][3] = 0.f;
            }
#pragma unroll
            for (int b = 0; b < BM; ++b) {
                if (b < B) {
                    const float4 iv = PREFETCH ? ireg[PREFETCH ? j : 0][PREFETCH ? b : 0]
                                               : reinterpret_cast<const float4 *>(img + (size_t)b * HW)[g];
                    float4 wv = make_float4(a.weight_scalar, a.weight_scalar, a.weight_scalar, a.weight_scalar);
                    if (wgt) wv = reinterpret_cast<const float4 *>(wgt + (size_t)b * HW)[g];
                    const float im[4] = {iv.x, iv.y, iv.z, iv.w}, ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float model = 0.f;
#pragma unroll
                        for (int k = 0; k < KM; ++k) model += sed[k][b] * m[k][e];
                        const float d = ww[e] * (model - im[e]);
                        loss += (double)d * (double)d;
                        const float gg = ww[e] * d;
#pragma unroll
                        for (int k = 0; k < KM; ++k) { dsed[k][b] += gg * m[k][e]; gm[k][e] += sed[k][b] * gg; }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K && !fixm[k])
                    lds_store4(tiles + k * tile_floats + y * LW + x,
                               make_float4(m[k][0] - step_morph * gm[k][0], m[k][1] - step_morph * gm[k][1],
                                           m[k][2] - step_morph * gm[k][2], m[k][3] - step_morph * gm[k][3]));
        }
    }
    {
        double v = wave_sum(0.5 * loss);
        if (lane == 0) red[wid][0] = v;
#pragma unroll
        for (int k = 0; k < KM; ++k)
#pragma unroll
            for (int b = 0; b < BM; ++b)
                if (k < K && b < B) {
                    v = wave_sum((double)dsed[k][b]);
                    if (lane == 0) red[wid][1 + k * B + b] = v;
                }
    }
    __syncthreads();
    STAMP(3);
    for (int i = tid; i < 1 + K * B; i += SC_BLOCK) {
        double r = 0;
#pragma unroll
        for (int w = 0; w < SC_NWAVES; ++w) r += red[w][i];
        tot[i] = r;
    }
    __syncthreads();
    for (int i = tid; i < K * B; i += SC_BLOCK) {
        const int k = i / B, b = i - k * B;
        const float curv = sed_s[k * BM + b];
        const bool fixed = a.fix_sed && a.fix_sed[(size_t)s * K + k];
        sed_new[k * BM + b] = fixed ? curv : curv - step_sed * (float)tot[1 + i];
    }
    if (tid == 0 && it_new <= a.mse_capacity) a.mse[(size_t)s * a.mse_capacity + it_new - 1] = tot[0];
    __syncthreads();
    STAMP(4);

    // ---------------- phase 2: constraints, one wave per component
    for (int k = wid; k < K; k += SC_NWAVES) {
        const int c = s * K + k;
        Tile t; t.H = H; t.W = W; t.LW = LW; t.m = tiles + k * tile_floats;
        float *vec = vecs + wid * SC_WAVE_VEC_FLOATS;
        int cy = a.centers[2 * c], cx = a.centers[2 * c + 1];
        int stat = 0;
        wave_max_pixel(t, cy, cx, stat);
        if (a.symmetric) {
            double dy = a.shifts[2 * c], dx = a.shifts[2 * c + 1];
            if (it_new % 5 == 0) {
                wave_centroid(t, a.centroid_psf, a.centroid_P, cy, cx, dy, dx, stat);
                if (lane == 0) { a.shifts[2 * c] = dy; a.shifts[2 * c + 1] = dx; }
            }
            STAMP(8);
            const bool none = (dy != dy);
            wave_symmetry(t, cy, cx, none ? SCARLET_SYM_SOFT : SCARLET_SYM_KSPACE, 1.0f, dy, dx, false, 0.f, vec);
        }
        STAMP(9);
        int lstop = 1 << 30;            // last sweep level computed; pixels beyond are <= 0 -> 0
        if (a.monotonic) wave_monotonic<float>(t, cy, cx, 0.f, &lstop);
        STAMP(10);
        if (lane == 0) { a.centers[2 * c] = cy; a.centers[2 * c + 1] = cx; }
        // sparsity, positivity, max (update.py:71-82, 27-32, 62-65): one pass over the LDS tile.
        // (Two short passes through LDS instead of 128 live registers: the register version
        // spilled and ran 2.5x slower.)
        float vmax = -INFINITY;
        bool anynan = false;
#pragma unroll 4
        for (int g = lane; g < ngroups; g += SC_WAVE) {
            const int y = g / gpr, x = (g - y * gpr) << 2;
            float *p = t.m + y * LW + x;
            const float4 v4 = lds_load4(p);
            float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (a.l0_thresh >= 0.f && fabsf(v[e]) < a.l0_thresh * step_morph) v[e] = 0.f;
                if (a.l1_thresh >= 0.f) {
                    const float mag = fabsf(v[e]) - a.l1_thresh * step_morph;
                    v[e] = (v[e] > 0.f ? 1.f : (v[e] < 0.f ? -1.f : 0.f)) * (mag < 0.f ? 0.f : mag);
                }
                if (v[e] < 0.f || sweep_level(y, x + e, cy, cx) > lstop) v[e] = 0.f;
                anynan |= (v[e] != v[e]);
                vmax = fmaxf(vmax, v[e]);
            }
            lds_store4(p, make_float4(v[0], v[1], v[2], v[3]));
        }
        float norm = wave_max(vmax);
        if (__any(anynan)) norm = __builtin_nanf("");
        if (!(norm > 0.f) || isinf(norm)) stat |= SCARLET_STATUS_NONFINITE;
        wave_sync();
        // normalise, store, convergence sums against the previous iteration (buffer c0, L2-resident)
        const float4 *last4 = reinterpret_cast<const float4 *>(min_g + (size_t)k * HW);
        float4 *out4 = reinterpret_cast<float4 *>(mout_g + (size_t)k * HW);
        float d2f = 0.f, n2f = 0.f;                      // 64 float terms per lane, then f64 across lanes
#pragma unroll 4
        for (int g = lane; g < ngroups; g += SC_WAVE) {
            const int y = g / gpr, x = (g - y * gpr) << 2;
            const float4 l = last4[g];
            const float4 v4 = lds_load4(t.m + y * LW + x);
            const float4 o = make_float4(v4.x / norm, v4.y / norm, v4.z / norm, v4.w / norm);
            out4[g] = o;
            const float e0 = l.x - o.x, e1 = l.y - o.y, e2 = l.z - o.z, e3 = l.w - o.w;
            d2f += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
            n2f += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
        }
        double d2 = (double)d2f, n2 = (double)n2f;
        d2 = wave_sum(d2); n2 = wave_sum(n2);
        double d2s = 0, n2s = 0;
        if (lane < B) {
            float v = sed_new[k * BM + lane];
            if (v < 0.f) v = 0.f;
            v = v * norm;
            sed_out[k * B + lane] = v;
            const float d = sed_s[k * BM + lane] - v;
            d2s = (double)(d * d);
            n2s = (double)(v * v);
        }
        d2s = wave_sum(d2s); n2s = wave_sum(n2s);
        if (lane == 0) {
            conv_s[k][0] = d2s; conv_s[k][1] = n2s; conv_s[k][2] = d2; conv_s[k][3] = n2;
            if (stat) atomicOr(&a.status[s], stat);
        }
    }
    __syncthreads();
    STAMP(5);

    // ---------------- phase 3: Blend._check_convergence + bookkeeping (blend.py:141-184)
    if (tid == 0) {
        a.it[s] = it_new;
        a.cur[s] = 1 - c0;
        if (it_new > 1) {
            bool done = true;
            for (int k = 0; k < K; ++k) {
                int f = a.flags[s * K + k];
                if (conv_s[k][0] <= a.e_rel2 * conv_s[k][1]) f &= ~SCARLET_FLAG_SED_NOT_CONVERGED;
                else { f |= SCARLET_FLAG_SED_NOT_CONVERGED; done = false; }
                if (conv_s[k][2] <= a.e_rel2 * conv_s[k][3]) f &= ~SCARLET_FLAG_MORPH_NOT_CONVERGED;
                else { f |= SCARLET_FLAG_MORPH_NOT_CONVERGED; done = false; }
                a.flags[s * K + k] = f;
            }
            if (done) a.active[s] = 0;
        }
    }
    STAMP(6);
#undef STAMP
}
