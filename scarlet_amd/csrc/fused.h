// fused.h -- one kernel per proximal-gradient iteration, one workgroup per scene.
//
// For K <= 4 components on frames up to 64 x 64 (W % 4 == 0) the whole iteration of Blend.fit
// (blend.py:79-102) runs in one launch with the morphologies resident in LDS between the gradient
// step and the constraints.  This file holds the four-wave kernel k_iterate<4, BM> (one wave per
// component; used for 5 < B <= 8) and the pieces shared with the eight-wave kernel of fused2.h
// (K <= 4, B <= 5: the headline workload).  Phases:
//
//   phase 0  ALL global loads of the iteration are issued first (morph tiles and images,
//            16 B/lane); morph tiles -> LDS, morph Gram S S^T on the fly            [a6]
//   eig      Lipschitz constants: largest root of the float64 characteristic polynomial
//            (n <= 4; Newton from the trace), lanes 0/1 of wave 0 for S S^T and A A^T;
//            the image loads are in flight meanwhile                                [a6]
//   phase 1  model, residual, loss, G, SED gradient sums from the prefetched images;
//            morphology step applied in place in LDS                                [a1-a5, a7]
//   phase 2  wave k runs the constraint pipeline of component k on its LDS tile and
//            writes the result to the other HBM buffer                              [a8-a17]
//   phase 3  convergence flags, it/cur bookkeeping                                  [a18]
//
// HBM traffic per scene-iteration = images + morph read + morph write (+ a second read of the
// previous morph for the convergence sums) -- the algorithmic bytes of SURVEY.md 8d plus that
// re-read.  The unfused kernels of engine.h remain the general path (K > 4, larger frames,
// approximate_L, PSF).
#pragma once
#include <type_traits>
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
    float *kscache;                  // NULL, or [S][K][2][SC_KSC_FLOATS]: Hankel vectors of the last k-space symmetry (k_iterate2)
};
// Per component and wave of its pair: 64 entries of av, bv, cv (this wave's half), then the header
// {H W cy cx, magic, dy (2 words), dx (2 words), s} the vectors were made for
#define SC_KSC_FLOATS 200
#define SC_KSC_MAGIC 0x5ca71e70u

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
    constexpr int GPW = 16;                      // float4 groups per lane in the per-wave passes
    constexpr bool PREFETCH = (KM <= 4);         // image prefetch needs 16*BM/4 more VGPRs
    __shared__ double red[SC_NWAVES][NP > NG ? NP : NG];
    __shared__ double tot[NP > NG ? NP : NG];
    __shared__ double mat[2][KM * KM > BM * BM ? KM * KM : BM * BM];
    __shared__ float sed_s[KM * BM], sed_new[KM * BM];
    __shared__ float step_s[2];
    __shared__ double conv_s[KM][4];
    const int tid = threadIdx.x, lane = tid & 63, wid = uniform(tid >> 6);    // wid in an SGPR: scalar branches
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
    const bool small_side = (K <= B);          // nonzero spectrum of A^T A == that of A A^T
    __syncthreads();                           // sed_s visible
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
    // Lipschitz constants (blend.py:205-218): L_sed = lambda_max(S S^T) on lane 0,
    // L_morph = lambda_max(A^T A) on lane 1 of wave 0 -- one instruction stream for both
    if (wid == 0) {
        if (lane < K * K) {
            const int k = lane / K, k2 = lane - k * K;
            const int lo = k < k2 ? k : k2, hi = k < k2 ? k2 : k;
            const int go = lo * K - (lo * (lo - 1)) / 2 + (hi - lo);     // packed upper-triangle index
            double r = 0;
#pragma unroll
            for (int w = 0; w < SC_NWAVES; ++w) r += red[w][go];
            mat[0][k * KM + k2] = r;
        }
        if (lane < (small_side ? K * K : B * B)) {
            double r = 0;
            if (small_side) {
                const int k = lane / K, k2 = lane - k * K;
                for (int b = 0; b < B; ++b) r += (double)sed_s[k * BM + b] * sed_s[k2 * BM + b];
                mat[1][k * KM + k2] = r;
            } else {
                const int b = lane / B, b2 = lane - b * B;
                for (int k = 0; k < K; ++k) r += (double)sed_s[k * BM + b] * sed_s[k * BM + b2];
                mat[1][b * BM + b2] = r;
            }
        }
        wave_sync();
        if (lane < 2) {
            const int n = (lane == 0 || small_side) ? K : B, ld = (lane == 0 || small_side) ? KM : BM;
            const double L = n <= 4 ? lambda_max_charpoly4(mat[lane], n, ld) : jacobi_lambda_max(mat[lane], n, ld);
            step_s[lane] = 1.0f / (float)L;
            a.lipschitz[2 * s + lane] = L;
        }
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
        cy = uniform(cy); cx = uniform(cx);
        if (a.symmetric) {
            double dy = a.shifts[2 * c], dx = a.shifts[2 * c + 1];
            if (it_new % 5 == 0) {
                wave_centroid(t, a.centroid_psf, a.centroid_P, cy, cx, dy, dx, stat);
                if (lane == 0) { a.shifts[2 * c] = dy; a.shifts[2 * c + 1] = dx; }
            }
            STAMP(8);
            cy = uniform(cy); cx = uniform(cx); dy = uniform(dy); dx = uniform(dx);
            const bool none = (dy != dy);
            wave_symmetry(t, cy, cx, none ? SCARLET_SYM_SOFT : SCARLET_SYM_KSPACE, 1.0f, dy, dx, false, 0.f, vec,
                          (a.stamps && wid == 0) ? a.stamps + (size_t)s * 16 + 12 : nullptr);
        }
        STAMP(9);
        int lstop = 1 << 30;            // last sweep level computed (early exit); pixels beyond are <= 0 -> 0
        if (a.monotonic) wave_monotonic<float>(t, cy, cx, 0.f, &lstop);
        STAMP(10);
        if (lane == 0) { a.centers[2 * c] = cy; a.centers[2 * c + 1] = cx; }
        // ---- sparsity, positivity (update.py:71-82, 27-32), normalisation (update.py:62-65),
        // store, convergence sums against the previous iteration: ONE pass over the LDS tile.
        // The previous morphology (buffer c0) is loaded up front, all 16 B/lane requests in
        // flight together: loads queued behind this pass's own stores would wait for them.
        const float4 *last4 = reinterpret_cast<const float4 *>(min_g + (size_t)k * HW);
        float4 lastv[GPW];
#pragma unroll
        for (int j = 0; j < GPW; ++j) {
            const int g = lane + j * SC_WAVE;
            lastv[j] = g < ngroups ? last4[g] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const bool cut = lstop < (1 << 30);
        float l0 = a.l0_thresh >= 0.f ? a.l0_thresh * step_morph : -1.f;
        float l1 = a.l1_thresh >= 0.f ? a.l1_thresh * step_morph : -1.f;
        auto sparse = [&](float v) {
            if (l0 >= 0.f && fabsf(v) < l0) v = 0.f;
            if (l1 >= 0.f) {
                const float mag = fabsf(v) - l1;
                v = (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f)) * (mag < 0.f ? 0.f : mag);
            }
            return v;
        };
        // lane -> (row, float4 group) walk without divisions: +64 groups per step
        const int dyq = SC_WAVE / gpr, dxq = SC_WAVE - dyq * gpr;
        const int y0 = lane / gpr, x0 = lane - y0 * gpr;
        float norm;
        if (a.monotonic) {
            // after the sweep no pixel exceeds the peak pixel (each is capped by a convex
            // combination of pixels closer to the peak) and the maps above are monotone:
            // morph.max() is the processed peak value (a NaN elsewhere: see below)
            norm = sparse(t.m[cy * LW + cx]);
            if (norm < 0.f) norm = 0.f;
        } else {
            float vmax = -INFINITY;
            bool anynan = false;
            int y = y0, xq = x0;
            for (int g = lane; g < ngroups; g += SC_WAVE) {
                float *p = t.m + y * LW + (xq << 2);
                float4 v = lds_load4(p);
                v.x = sparse(v.x); v.y = sparse(v.y); v.z = sparse(v.z); v.w = sparse(v.w);
                v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y;
                v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w;
                anynan |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
                vmax = fmaxf(fmaxf(vmax, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
                lds_store4(p, v);
                y += dyq; xq += dxq;
                if (xq >= gpr) { xq -= gpr; ++y; }
            }
            norm = wave_max(vmax);
            if (__any(anynan)) norm = __builtin_nanf("");
            l0 = -1.f; l1 = -1.f;                       // applied
            wave_sync();
        }
        const bool regular = norm > 0.f && !isinf(norm);             // else: the reference's 0/0, x/inf, NaN results
        const float rnorm = 1.0f / norm;
        float4 *out4 = reinterpret_cast<float4 *>(mout_g + (size_t)k * HW);
        float d2f = 0.f, n2f = 0.f;                      // <= 64 float terms per lane, then f64 across lanes
        // CUT: zero beyond the sweep's last level; GEN: thresholds and/or an irregular norm
        auto final_pass = [&](auto cut_c, auto gen_c) {
            constexpr bool CUT = decltype(cut_c)::value, GEN = decltype(gen_c)::value;
            int y = y0, xq = x0;
#pragma unroll
            for (int j = 0; j < GPW; ++j) {
                const int g = lane + j * SC_WAVE;
                if (g < ngroups) {
                    const float4 v4 = lds_load4(t.m + y * LW + (xq << 2));
                    const float4 l = lastv[j];
                    float v[4] = {v4.x, v4.y, v4.z, v4.w}, o[4];
                    const int ay = y < cy ? cy - y : y - cy;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (GEN) v[e] = sparse(v[e]);
                        v[e] = v[e] < 0.f ? 0.f : v[e];                       // NaN stays NaN
                        if (CUT) {
                            const int x = (xq << 2) + e, ax = x < cx ? cx - x : x - cx;
                            if (max(ax, ay) + ax + ay > lstop) v[e] = 0.f;
                        }
                        if (GEN && !regular) o[e] = v[e] / norm;
                        else {
                            // v / norm, correctly rounded (but for rare double roundings):
                            // one Newton step on v * (1 / norm)
                            const float q = v[e] * rnorm;
                            o[e] = fmaf(fmaf(-q, norm, v[e]), rnorm, q);
                        }
                    }
                    out4[g] = make_float4(o[0], o[1], o[2], o[3]);
                    const float e0 = l.x - o[0], e1 = l.y - o[1], e2 = l.z - o[2], e3 = l.w - o[3];
                    d2f += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
                    n2f += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
                }
                y += dyq; xq += dxq;
                if (xq >= gpr) { xq -= gpr; ++y; }
                if (j & 1) __builtin_amdgcn_sched_barrier(0);    // two groups in flight, not sixteen (VGPRs)
            }
        };
        using std::true_type; using std::false_type;
        if (regular && l0 < 0.f && l1 < 0.f) {
            if (cut) final_pass(true_type{}, false_type{}); else final_pass(false_type{}, false_type{});
        } else {
            final_pass(true_type{}, true_type{});
        }
        if (__any(n2f != n2f) && norm == norm) {
            // a NaN pixel away from the peak: np.max is NaN and the reference's morph becomes NaN everywhere
            norm = __builtin_nanf("");
            const float4 nan4 = make_float4(norm, norm, norm, norm);
            for (int g = lane; g < ngroups; g += SC_WAVE) out4[g] = nan4;
            d2f = norm; n2f = norm;
        }
        if (!(norm > 0.f) || isinf(norm)) stat |= SCARLET_STATUS_NONFINITE;
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
            if (a.stamps && wid == 0) a.stamps[(size_t)s * 16 + 11] = (long long)__builtin_amdgcn_s_memtime();
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
