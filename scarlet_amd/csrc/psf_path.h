// psf_path.h -- row a3b: Observation.render with a PSF difference kernel, and its adjoint.
//
// The reference renders a model by zero-pad -> ifftshift -> rfftn -> * K-hat -> irfftn ->
// fftshift -> centre crop (observation.py:198-201, fft.py:304-317, 264-279, 193-211,
// 138-181, 38-65, 7-35, 68-106) at the 5-smooth FFT shape of (N + P + 3) per axis (last axis
// even), and autograd runs the same chain with conj(K-hat) for the gradient.  With the pad
// start (dS+1)//2, the shift F//2 and the crop start (cur-new+1)//2 all three index maps
// collapse to ONE modular offset per array:
//     image pixel y  <->  padded, shifted index  (y + o_img) mod F,   o_img = (F-N+1)//2 - F//2
//     kernel pixel q <->                         (q + o_ker) mod F,   o_ker = (F-P+1)//2 - F//2
// so the kernels below read/write the FFT buffers directly at those indices -- no separate
// pad / shift / crop passes.  The offsets are taken from the REFERENCE's F (they decide which
// pixel of an even-sized kernel is its centre); the transforms run at the smallest 7-smooth
// length >= N + P - 1, which yields the same linear convolution inside the crop (psf_geom).  The transforms themselves are batched hipFFT (rocFFT) R2C / C2R
// plans over all (scene, band) planes; K-hat is computed once per batch.
//
// One iteration with a PSF:
//   k_psf_model   : model planes written at their padded positions (zeros elsewhere)
//   R2C, k_spec_mul (K-hat, 1/F scaling), C2R
//   k_psf_resid   : d = w (render - image), loss partials; writes w d back at padded positions
//   R2C, k_spec_mul (conj K-hat), C2R          -> G = render^T (w d)
//   k_grad_psf / k_step_psf : as k_grad / k_step but with G read from the FFT buffer
// That is the hipFFT chain of round 1 (planes beyond LDS, odd widths).  When the half-spectrum plane fits LDS the chain
// is k_psf_model4g -> k_psf_conv (fftconv.h) -> k_step_psf4f -> k_sed_step on compact planes: see "three-pass form" below.
#pragma once
#include "common.h"
#include "engine.h"

struct PsfGeom {
    int H, W, Fy, Fx, Fxh;       // device FFT shape, Fxh = Fx/2 + 1
    int oy, ox;                  // image offsets (mod F)
    int Fry, Frx;                // the REFERENCE's FFT shape: fixes where an even-sized kernel is centred
};

__host__ __device__ inline int pos_mod(int a, int n) { int r = a % n; return r < 0 ? r + n : r; }

struct PsfArgs {
    int S, K, B, T;
    PsfGeom g;
    const float *images, *weights;
    float weight_scalar;
    float *sed[2], *morph[2];
    const int *cur;
    const uint8_t *fix_sed, *fix_morph;
    float *real;                 // [S*B][Fy][Fx]
    float2 *spec;                // [S*B][Fy][Fxh]
    const float2 *khat;          // [B][Fy][Fxh] (or [S*B]...: khat_per_scene)
    int khat_per_scene;
    double *partials;            // [S][T][P] (sed gradient + Gram) ; loss in loss_part
    double *loss_part;           // [S][B]
    double *lipschitz, *mse;
    int mse_capacity;
    int *it;
    const int *active;
    int approximate_L;
    int raw_gradient;            // as GradArgs::raw_gradient
};

// model_b = sum_k sed[k][b] morph[k] written into the padded FFT input planes (a1, a2 + pad).
// One thread per padded pixel and scene: the K morphology values are read once and all B band
// planes are written from them.  grid (ceil(Fy Fx / 256), S).
__global__ __launch_bounds__(SC_BLOCK) void k_psf_model(PsfArgs a)
{
    const int s = blockIdx.y;
    if (!a.active[s]) return;
    const PsfGeom g = a.g;
    const int c0 = a.cur[s], HW = g.H * g.W, K = a.K, B = a.B;
    __shared__ float sed_s[SC_KBIG * SC_BMAX];
    for (int i = threadIdx.x; i < K * B; i += SC_BLOCK)
        sed_s[(i / B) * SC_BMAX + (i % B)] = a.sed[c0][(size_t)s * K * B + i];
    __syncthreads();
    const int n = g.Fy * g.Fx;
    const int i = blockIdx.x * SC_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int iy = i / g.Fx, ix = i - iy * g.Fx;
    const int y = pos_mod(iy - g.oy, g.Fy), x = pos_mod(ix - g.ox, g.Fx);
    float v[SC_BMAX];
#pragma unroll
    for (int b = 0; b < SC_BMAX; ++b) v[b] = 0.f;
    if (y < g.H && x < g.W) {
        const float *mor = a.morph[c0] + (size_t)s * K * HW + y * g.W + x;
        for (int k = 0; k < K; ++k) {
            const float m = mor[(size_t)k * HW];
#pragma unroll
            for (int b = 0; b < SC_BMAX; ++b)
                if (b < B) v[b] += sed_s[k * SC_BMAX + b] * m;
        }
    }
    float *out = a.real + (size_t)s * B * n + i;
#pragma unroll
    for (int b = 0; b < SC_BMAX; ++b)
        if (b < B) out[(size_t)b * n] = v[b];
}

// spectrum *= K-hat (or its conjugate) * scale ; K-hat has `nk` planes: B (shared by all scenes of a
// band) or S * B (one kernel set per scene)
__global__ void k_spec_mul(float2 *spec, const float2 *khat, int nk, int plane_elems, int64_t total,
                           int conj, float scale)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int plane = (int)(i / plane_elems), e = (int)(i - (int64_t)plane * plane_elems);
        const float2 k = khat[(size_t)(plane % nk) * plane_elems + e];
        const float2 v = spec[i];
        const float ki = conj ? -k.y : k.y;
        spec[i] = make_float2((v.x * k.x - v.y * ki) * scale, (v.x * ki + v.y * k.x) * scale);
    }
}

// d = w (render - image); loss; w d written back (padded layout, zeros outside the image)
__global__ __launch_bounds__(SC_BLOCK) void k_psf_resid(PsfArgs a)
{
    const int plane = blockIdx.x, s = plane / a.B, b = plane - s * a.B;
    if (!a.active[s]) return;
    const PsfGeom g = a.g;
    __shared__ double red[SC_NWAVES];
    const int HW = g.H * g.W;
    float *buf = a.real + (size_t)plane * g.Fy * g.Fx;
    const float *img = a.images + ((size_t)s * a.B + b) * HW;
    const float *wgt = a.weights ? a.weights + ((size_t)s * a.B + b) * HW : nullptr;
    const int n = g.Fy * g.Fx;
    double loss = 0;
    for (int i = threadIdx.x; i < n; i += SC_BLOCK) {
        const int iy = i / g.Fx, ix = i - iy * g.Fx;
        const int y = pos_mod(iy - g.oy, g.Fy), x = pos_mod(ix - g.ox, g.Fx);
        float v = 0.f;
        if (y < g.H && x < g.W) {
            const float w = wgt ? wgt[y * g.W + x] : a.weight_scalar;
            const float d = w * (buf[i] - img[y * g.W + x]);
            loss += (double)d * (double)d;
            v = w * d;
        }
        buf[i] = v;
    }
    loss = block_sum(loss, red);
    if (threadIdx.x == 0) a.loss_part[plane] = 0.5 * loss;
}

// G (B planes, padded layout) -> partial sums of d loss / d sed and of the morph Gram
template <int KM, int BM>
__global__ __launch_bounds__(SC_BLOCK) void k_grad_psf(PsfArgs a)
{
    const int s = blockIdx.y, tile = blockIdx.x;
    if (!a.active[s]) return;
    const PsfGeom g = a.g;
    const int K = a.K, B = a.B, HW = g.H * g.W;
    __shared__ double red[SC_NWAVES][KM * BM + KM * (KM + 1) / 2];
    const int c0 = a.cur[s];
    float dsed[KM][BM], gram[KM * (KM + 1) / 2];
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int b = 0; b < BM; ++b) dsed[k][b] = 0.f;
#pragma unroll
    for (int i = 0; i < KM * (KM + 1) / 2; ++i) gram[i] = 0.f;
    const float *mor = a.morph[c0] + (size_t)s * K * HW;
    const float *G = a.real + (size_t)s * B * g.Fy * g.Fx;
    const int p_end = min(HW, (tile + 1) * SC_TILE_PIX);
    for (int p = tile * SC_TILE_PIX + threadIdx.x; p < p_end; p += SC_BLOCK) {
        const int y = p / g.W, x = p - y * g.W;
        const int gi0 = pos_mod(y + g.oy, g.Fy) * g.Fx + pos_mod(x + g.ox, g.Fx);
        float m[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) m[k] = k < K ? mor[(size_t)k * HW + p] : 0.f;
#pragma unroll
        for (int b = 0; b < BM; ++b)
            if (b < B) {
                const float gg = G[(size_t)b * g.Fy * g.Fx + gi0];
#pragma unroll
                for (int k = 0; k < KM; ++k) dsed[k][b] += gg * m[k];
            }
        int gi = 0;
#pragma unroll
        for (int k = 0; k < KM; ++k)
#pragma unroll
            for (int k2 = k; k2 < KM; ++k2) gram[gi++] += m[k] * m[k2];
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int b = 0; b < BM; ++b)
            if (k < K && b < B) {
                const double v = wave_sum((double)dsed[k][b]);
                if (lane == 0) red[wid][k * B + b] = v;
            }
    {
        int gi = 0, go = 0;
#pragma unroll
        for (int k = 0; k < KM; ++k)
#pragma unroll
            for (int k2 = k; k2 < KM; ++k2) {
                if (k < K && k2 < K) {
                    const double v = wave_sum((double)gram[gi]);
                    if (lane == 0) red[wid][K * B + go] = v;
                    ++go;
                }
                ++gi;
            }
    }
    __syncthreads();
    const int P = n_partials(K, B);
    double *out = a.partials + ((size_t)s * a.T + tile) * P;
    for (int i = threadIdx.x; i < P - 1; i += SC_BLOCK) {
        double r = 0;
#pragma unroll
        for (int w = 0; w < SC_NWAVES; ++w) r += red[w][i];
        out[1 + i] = r;
    }
    if (threadIdx.x == 0) {
        // the loss of the scene (sum over its bands) rides in slot 0 of tile 0
        double l = 0;
        if (tile == 0) for (int b = 0; b < B; ++b) l += a.loss_part[s * B + b];
        out[0] = l;
    }
}

// Lipschitz constants + SED step + morphology step from G (same bookkeeping as k_step)
template <int KM, int BM>
__global__ __launch_bounds__(SC_BLOCK) void k_step_psf(PsfArgs a)
{
    const int s = blockIdx.y, tile = blockIdx.x;
    if (!a.active[s]) return;
    const PsfGeom g = a.g;
    const int K = a.K, B = a.B, HW = g.H * g.W, P = n_partials(K, B);
    __shared__ double tot[1 + KM * BM + KM * (KM + 1) / 2];
    __shared__ double mat[KM * KM + BM * BM];
    __shared__ float sed_s[KM * BM];
    __shared__ float step_s[2];
    for (int i = threadIdx.x; i < P; i += SC_BLOCK) {
        double r = 0;
        for (int t = 0; t < a.T; ++t) r += a.partials[((size_t)s * a.T + t) * P + i];
        tot[i] = r;
    }
    const int c0 = a.cur[s];
    const float *sed_in = a.sed[c0];
    float *sed_out = a.sed[1 - c0];
    for (int i = threadIdx.x; i < K * B; i += SC_BLOCK)
        sed_s[(i / B) * BM + (i % B)] = sed_in[(size_t)s * K * B + i];
    __syncthreads();
    __shared__ double Lc[2];
    __shared__ double eigbuf[2][2][64];
    const int it_new = a.it[s] + 1;                      // len(mse) after the append
    if (a.approximate_L) {
        if (threadIdx.x == 0) {
            // blend.py:189-202: traces of the two Gram matrices, doubled if the loss rose
            double LA = 0, LS = 0;
            int go = 0;
            for (int k = 0; k < K; ++k)
                for (int k2 = k; k2 < K; ++k2) { if (k2 == k) LA += tot[1 + K * B + go]; ++go; }
            for (int k = 0; k < K; ++k)
                for (int b = 0; b < B; ++b) LS += (double)sed_s[k * BM + b] * sed_s[k * BM + b];
            if (it_new > 1 && tot[0] > a.mse[(size_t)s * a.mse_capacity + it_new - 2]) { LA *= 2; LS *= 2; }
            Lc[0] = LA; Lc[1] = LS;
        }
        __syncthreads();
    } else {
        // blend.py:205-218: L_sed = lambda_max(S S^T), L_morph = lambda_max(A^T A)
        double *G = mat, *ATA = mat + KM * KM;
        for (int i = threadIdx.x; i < K * K; i += SC_BLOCK) {
            const int k = i / K, k2 = i - k * K, lo = k < k2 ? k : k2, hi = k < k2 ? k2 : k;
            G[k * KM + k2] = tot[1 + K * B + lo * K - (lo * (lo - 1)) / 2 + (hi - lo)];
        }
        for (int i = threadIdx.x; i < B * B; i += SC_BLOCK) {
            const int b = i / B, b2 = i - b * B;
            double r = 0;
            for (int k = 0; k < K; ++k) r += (double)sed_s[k * BM + b] * sed_s[k * BM + b2];
            ATA[b * BM + b2] = r;
        }
        __syncthreads();
        block_lipschitz(G, K, KM, ATA, B, BM, eigbuf, Lc);
    }
    if (threadIdx.x == 0) {
        // frame dtype is float32: the reference's L and 1/L are float32 scalars
        step_s[0] = 1.0f / (float)Lc[0];
        step_s[1] = 1.0f / (float)Lc[1];
        if (tile == 0) {
            if (it_new <= a.mse_capacity) a.mse[(size_t)s * a.mse_capacity + it_new - 1] = tot[0];
            a.lipschitz[2 * s] = Lc[0];
            a.lipschitz[2 * s + 1] = Lc[1];
        }
    }
    __syncthreads();
    const float step_sed = step_s[0], step_morph = step_s[1];
    if (tile == 0)
        for (int i = threadIdx.x; i < K * B; i += SC_BLOCK) {
            const int k = i / B;
            const float curv = sed_s[k * BM + (i % B)];
            const bool fixed = a.fix_sed && a.fix_sed[(size_t)s * K + k];
            sed_out[(size_t)s * K * B + i] = a.raw_gradient ? (float)tot[1 + i] : (fixed ? curv : curv - step_sed * (float)tot[1 + i]);
        }
    const float *mor = a.morph[c0] + (size_t)s * K * HW;
    float *mout = a.morph[1 - c0] + (size_t)s * K * HW;
    const float *G = a.real + (size_t)s * B * g.Fy * g.Fx;
    const int p_end = min(HW, (tile + 1) * SC_TILE_PIX);
    for (int p = tile * SC_TILE_PIX + threadIdx.x; p < p_end; p += SC_BLOCK) {
        const int y = p / g.W, x = p - y * g.W;
        const int gi0 = pos_mod(y + g.oy, g.Fy) * g.Fx + pos_mod(x + g.ox, g.Fx);
        float gb[BM];
#pragma unroll
        for (int b = 0; b < BM; ++b) gb[b] = b < B ? G[(size_t)b * g.Fy * g.Fx + gi0] : 0.f;
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (k < K) {
                const float m = mor[(size_t)k * HW + p];
                float gm = 0.f;
#pragma unroll
                for (int b = 0; b < BM; ++b)
                    if (b < B) gm += sed_s[k * BM + b] * gb[b];      // (entries b >= B of sed_s are never written)
                const bool fixed = a.fix_morph && a.fix_morph[(size_t)s * K + k];
                mout[(size_t)k * HW + p] = a.raw_gradient ? gm : (fixed ? m : m - step_morph * gm);
            }
    }
}

// ---- the same two kernels for COMPACT gradient planes G [S][B][H*W] (the LDS-resident convolution writes
// them unpadded): 16 B per lane on every stream (K morphologies, B planes of G, K morphologies out) instead of
// 4 B -- the scalar forms move 3.3 TB/s on BASELINE config 3, these are bound by HBM.  H * W % 4 == 0.
template <int KM, int BM>
__global__ __launch_bounds__(SC_BLOCK) void k_grad_psf4(PsfArgs a)
{
    const int s = blockIdx.y, tile = blockIdx.x;
    if (!a.active[s]) return;
    const int K = a.K, B = a.B, HW = a.g.H * a.g.W;
    __shared__ double red[SC_NWAVES][KM * BM + KM * (KM + 1) / 2];
    const int c0 = a.cur[s];
    float dsed[KM][BM], gram[KM * (KM + 1) / 2];
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int b = 0; b < BM; ++b) dsed[k][b] = 0.f;
#pragma unroll
    for (int i = 0; i < KM * (KM + 1) / 2; ++i) gram[i] = 0.f;
    const float4 *mor = reinterpret_cast<const float4 *>(a.morph[c0] + (size_t)s * K * HW);
    const float4 *G = reinterpret_cast<const float4 *>(a.real + (size_t)s * B * HW);
    const int HW4 = HW >> 2;
    const int g_end = min(HW4, (tile + 1) * (SC_TILE_PIX >> 2));
    for (int g = tile * (SC_TILE_PIX >> 2) + threadIdx.x; g < g_end; g += SC_BLOCK) {
        float4 m[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) m[k] = k < K ? mor[(size_t)k * HW4 + g] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int b = 0; b < BM; ++b)
            if (b < B) {
                const float4 gg = G[(size_t)b * HW4 + g];
                // pixel order inside the group as in the scalar kernel: x, y, z, w
#pragma unroll
                for (int k = 0; k < KM; ++k) {
                    dsed[k][b] += gg.x * m[k].x; dsed[k][b] += gg.y * m[k].y;
                    dsed[k][b] += gg.z * m[k].z; dsed[k][b] += gg.w * m[k].w;
                }
            }
        int gi = 0;
#pragma unroll
        for (int k = 0; k < KM; ++k)
#pragma unroll
            for (int k2 = k; k2 < KM; ++k2) {
                gram[gi] += m[k].x * m[k2].x; gram[gi] += m[k].y * m[k2].y;
                gram[gi] += m[k].z * m[k2].z; gram[gi] += m[k].w * m[k2].w;
                ++gi;
            }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int b = 0; b < BM; ++b)
            if (k < K && b < B) {
                const double v = wave_sum((double)dsed[k][b]);
                if (lane == 0) red[wid][k * B + b] = v;
            }
    {
        int gi = 0, go = 0;
#pragma unroll
        for (int k = 0; k < KM; ++k)
#pragma unroll
            for (int k2 = k; k2 < KM; ++k2) {
                if (k < K && k2 < K) {
                    const double v = wave_sum((double)gram[gi]);
                    if (lane == 0) red[wid][K * B + go] = v;
                    ++go;
                }
                ++gi;
            }
    }
    __syncthreads();
    const int P = n_partials(K, B);
    double *out = a.partials + ((size_t)s * a.T + tile) * P;
    for (int i = threadIdx.x; i < P - 1; i += SC_BLOCK) {
        double r = 0;
#pragma unroll
        for (int w = 0; w < SC_NWAVES; ++w) r += red[w][i];
        out[1 + i] = r;
    }
    if (threadIdx.x == 0) {
        double l = 0;
        if (tile == 0) for (int b = 0; b < B; ++b) l += a.loss_part[s * B + b];
        out[0] = l;
    }
}

// the morphology step of k_step_psf on compact planes, 16 B per lane; the scalar head (Lipschitz constants,
// SED step) is shared through a device function
template <int KM, int BM>
__device__ __forceinline__ void step_psf_head(const PsfArgs &a, int s, int tile, double *tot, double *mat, float *sed_s,
                                              float *step_s, double *Lc, double (*eigbuf)[2][64])
{
    const int K = a.K, B = a.B, P = n_partials(K, B);
    for (int i = threadIdx.x; i < P; i += SC_BLOCK) {
        double r = 0;
        for (int t = 0; t < a.T; ++t) r += a.partials[((size_t)s * a.T + t) * P + i];
        tot[i] = r;
    }
    const int c0 = a.cur[s];
    const float *sed_in = a.sed[c0];
    float *sed_out = a.sed[1 - c0];
    for (int i = threadIdx.x; i < K * B; i += SC_BLOCK)
        sed_s[(i / B) * BM + (i % B)] = sed_in[(size_t)s * K * B + i];
    __syncthreads();
    const int it_new = a.it[s] + 1;
    if (a.approximate_L) {
        if (threadIdx.x == 0) {
            double LA = 0, LS = 0;
            int go = 0;
            for (int k = 0; k < K; ++k)
                for (int k2 = k; k2 < K; ++k2) { if (k2 == k) LA += tot[1 + K * B + go]; ++go; }
            for (int k = 0; k < K; ++k)
                for (int b = 0; b < B; ++b) LS += (double)sed_s[k * BM + b] * sed_s[k * BM + b];
            if (it_new > 1 && tot[0] > a.mse[(size_t)s * a.mse_capacity + it_new - 2]) { LA *= 2; LS *= 2; }
            Lc[0] = LA; Lc[1] = LS;
        }
        __syncthreads();
    } else {
        double *G = mat, *ATA = mat + KM * KM;
        for (int i = threadIdx.x; i < K * K; i += SC_BLOCK) {
            const int k = i / K, k2 = i - k * K, lo = k < k2 ? k : k2, hi = k < k2 ? k2 : k;
            G[k * KM + k2] = tot[1 + K * B + lo * K - (lo * (lo - 1)) / 2 + (hi - lo)];
        }
        for (int i = threadIdx.x; i < B * B; i += SC_BLOCK) {
            const int b = i / B, b2 = i - b * B;
            double r = 0;
            for (int k = 0; k < K; ++k) r += (double)sed_s[k * BM + b] * sed_s[k * BM + b2];
            ATA[b * BM + b2] = r;
        }
        __syncthreads();
        block_lipschitz(G, K, KM, ATA, B, BM, eigbuf, Lc);
    }
    if (threadIdx.x == 0) {
        step_s[0] = 1.0f / (float)Lc[0];
        step_s[1] = 1.0f / (float)Lc[1];
        if (tile == 0) {
            if (it_new <= a.mse_capacity) a.mse[(size_t)s * a.mse_capacity + it_new - 1] = tot[0];
            a.lipschitz[2 * s] = Lc[0];
            a.lipschitz[2 * s + 1] = Lc[1];
        }
    }
    __syncthreads();
    if (tile == 0)
        for (int i = threadIdx.x; i < K * B; i += SC_BLOCK) {
            const int k = i / B;
            const float curv = sed_s[k * BM + (i % B)];
            const bool fixed = a.fix_sed && a.fix_sed[(size_t)s * K + k];
            sed_out[(size_t)s * K * B + i] = a.raw_gradient ? (float)tot[1 + i] : (fixed ? curv : curv - step_s[0] * (float)tot[1 + i]);
        }
}

template <int KM, int BM>
__global__ __launch_bounds__(SC_BLOCK) void k_step_psf4(PsfArgs a)
{
    const int s = blockIdx.y, tile = blockIdx.x;
    if (!a.active[s]) return;
    const int K = a.K, B = a.B, HW = a.g.H * a.g.W;
    __shared__ double tot[1 + KM * BM + KM * (KM + 1) / 2];
    __shared__ double mat[KM * KM + BM * BM];
    __shared__ float sed_s[KM * BM];
    __shared__ float step_s[2];
    __shared__ double Lc[2];
    __shared__ double eigbuf[2][2][64];
    step_psf_head<KM, BM>(a, s, tile, tot, mat, sed_s, step_s, Lc, eigbuf);
    const float step_morph = step_s[1];
    const int c0 = a.cur[s], HW4 = HW >> 2;
    const float4 *mor = reinterpret_cast<const float4 *>(a.morph[c0] + (size_t)s * K * HW);
    float4 *mout = reinterpret_cast<float4 *>(a.morph[1 - c0] + (size_t)s * K * HW);
    const float4 *G = reinterpret_cast<const float4 *>(a.real + (size_t)s * B * HW);
    const int g_end = min(HW4, (tile + 1) * (SC_TILE_PIX >> 2));
    for (int g = tile * (SC_TILE_PIX >> 2) + threadIdx.x; g < g_end; g += SC_BLOCK) {
        float4 gb[BM];
#pragma unroll
        for (int b = 0; b < BM; ++b) gb[b] = b < B ? G[(size_t)b * HW4 + g] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (k < K) {
                const float4 m = mor[(size_t)k * HW4 + g];
                float4 gm = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int b = 0; b < BM; ++b)
                    if (b < B) {
                        const float sk = sed_s[k * BM + b];
                        gm.x += sk * gb[b].x; gm.y += sk * gb[b].y; gm.z += sk * gb[b].z; gm.w += sk * gb[b].w;
                    }
                const bool fixed = a.fix_morph && a.fix_morph[(size_t)s * K + k];
                float4 o;
                if (a.raw_gradient) o = gm;
                else if (fixed) o = m;
                else o = make_float4(m.x - step_morph * gm.x, m.y - step_morph * gm.y, m.z - step_morph * gm.z, m.w - step_morph * gm.w);
                mout[(size_t)k * HW4 + g] = o;
            }
    }
}

// model planes, compact [S][B][H*W], 16 B per lane (k_psf_model with the geometry of an unpadded plane)
__global__ __launch_bounds__(SC_BLOCK) void k_psf_model4(PsfArgs a)
{
    const int s = blockIdx.y;
    if (!a.active[s]) return;
    const int c0 = a.cur[s], HW4 = (a.g.H * a.g.W) >> 2, K = a.K, B = a.B;
    __shared__ float sed_s[SC_KBIG * SC_BMAX];
    for (int i = threadIdx.x; i < K * B; i += SC_BLOCK)
        sed_s[(i / B) * SC_BMAX + (i % B)] = a.sed[c0][(size_t)s * K * B + i];
    __syncthreads();
    const int g = blockIdx.x * SC_BLOCK + threadIdx.x;
    if (g >= HW4) return;
    float4 v[SC_BMAX];
#pragma unroll
    for (int b = 0; b < SC_BMAX; ++b) v[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 *mor = reinterpret_cast<const float4 *>(a.morph[c0] + (size_t)s * K * HW4 * 4) + g;
    for (int k = 0; k < K; ++k) {
        const float4 m = mor[(size_t)k * HW4];
#pragma unroll
        for (int b = 0; b < SC_BMAX; ++b)
            if (b < B) {
                const float sk = sed_s[k * SC_BMAX + b];
                v[b].x += sk * m.x; v[b].y += sk * m.y; v[b].z += sk * m.z; v[b].w += sk * m.w;
            }
    }
    float4 *out = reinterpret_cast<float4 *>(a.real + (size_t)s * B * HW4 * 4) + g;
#pragma unroll
    for (int b = 0; b < SC_BMAX; ++b)
        if (b < B) out[(size_t)b * HW4] = v[b];
}

// ---- three-pass form of an iteration on compact planes (K <= SC_KMAX): the morphology step needs only
// lambda_max(A^T A) of the SEDs (blend.py:205-218), so the separate reduction pass over G and the morphologies
// (k_grad_psf4) is not needed before it:
//   k_psf_model4g : model planes + the Gram partials S S^T (the pass reads every morphology anyway)
//   k_psf_conv    : model -> G
//   k_step_psf4f  : morphology step + the SED-gradient partials sum_p G_b m_k of the same pass
//   k_sed_step    : per scene: partial sums, both Lipschitz constants, SED step, loss record
// Every sum keeps the per-thread order and the reduction tree of k_grad_psf4 / k_step_psf4: the results are
// bit-identical to the four-pass form (SCARLET_NO_PSF3PASS=1 runs that one).
template <int KM>
__global__ __launch_bounds__(SC_BLOCK) void k_psf_model4g(PsfArgs a)
{
    const int s = blockIdx.y, tile = blockIdx.x;
    if (!a.active[s]) return;
    const int c0 = a.cur[s], HW4 = (a.g.H * a.g.W) >> 2, K = a.K, B = a.B;
    __shared__ float sed_s[KM * SC_BMAX];
    __shared__ double red[SC_NWAVES][KM * (KM + 1) / 2];
    for (int i = threadIdx.x; i < K * B; i += SC_BLOCK)
        sed_s[(i / B) * SC_BMAX + (i % B)] = a.sed[c0][(size_t)s * K * B + i];
    __syncthreads();
    float sk[KM][SC_BMAX];               // scalars (SGPRs), as in k_step_psf4f
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int b = 0; b < SC_BMAX; ++b)
            sk[k][b] = (k < K && b < B) ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sed_s[k * SC_BMAX + b]))) : 0.f;
    float gram[KM * (KM + 1) / 2];
#pragma unroll
    for (int i = 0; i < KM * (KM + 1) / 2; ++i) gram[i] = 0.f;
    const float4 *mor = reinterpret_cast<const float4 *>(a.morph[c0] + (size_t)s * K * HW4 * 4);
    float4 *out = reinterpret_cast<float4 *>(a.real + (size_t)s * B * HW4 * 4);
    const int g_end = min(HW4, (tile + 1) * (SC_TILE_PIX >> 2));
    // (not unrolled: ~100 VGPRs keep four waves per SIMD in flight, which is what a streaming pass needs)
#pragma unroll 1
    for (int g = tile * (SC_TILE_PIX >> 2) + threadIdx.x; g < g_end; g += SC_BLOCK) {
        float4 m[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) m[k] = k < K ? mor[(size_t)k * HW4 + g] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int b = 0; b < SC_BMAX; ++b)
            if (b < B) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (k < K) {
                        v.x += sk[k][b] * m[k].x; v.y += sk[k][b] * m[k].y; v.z += sk[k][b] * m[k].z; v.w += sk[k][b] * m[k].w;
                    }
                out[(size_t)b * HW4 + g] = v;
            }
        int gi = 0;
#pragma unroll
        for (int k = 0; k < KM; ++k)
#pragma unroll
            for (int k2 = k; k2 < KM; ++k2) {
                gram[gi] += m[k].x * m[k2].x; gram[gi] += m[k].y * m[k2].y;
                gram[gi] += m[k].z * m[k2].z; gram[gi] += m[k].w * m[k2].w;
                ++gi;
            }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
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
    const int P = n_partials(K, B), NG = K * (K + 1) / 2;
    double *po = a.partials + ((size_t)s * a.T + tile) * P + 1 + K * B;
    for (int i = threadIdx.x; i < NG; i += SC_BLOCK) {
        double r = 0;
#pragma unroll
        for (int w = 0; w < SC_NWAVES; ++w) r += red[w][i];
        po[i] = r;
    }
}

template <int KM, int BM>
__global__ __launch_bounds__(SC_BLOCK, (KM * BM <= 32 ? 4 : 2)) void k_step_psf4f(PsfArgs a)
{
    const int s = blockIdx.y, tile = blockIdx.x;
    if (!a.active[s]) return;
    const int K = a.K, B = a.B, HW = a.g.H * a.g.W;
    __shared__ double red[SC_NWAVES][KM * BM];
    __shared__ double ATA[BM * BM];
    __shared__ float sed_s[KM * BM];
    __shared__ double Lm;
    __shared__ double eigbuf[2][64];
    const int c0 = a.cur[s];
    for (int i = threadIdx.x; i < K * B; i += SC_BLOCK)
        sed_s[(i / B) * BM + (i % B)] = a.sed[c0][(size_t)s * K * B + i];
    __syncthreads();
    // the morphology step's Lipschitz constant, as step_psf_head computes it
    if (a.approximate_L) {
        if (threadIdx.x == 0) {
            double LS = 0, l = 0;
            for (int k = 0; k < K; ++k)
                for (int b = 0; b < B; ++b) LS += (double)sed_s[k * BM + b] * sed_s[k * BM + b];
            for (int b = 0; b < B; ++b) l += a.loss_part[s * B + b];
            const int it_new = a.it[s] + 1;
            if (it_new > 1 && l > a.mse[(size_t)s * a.mse_capacity + it_new - 2]) LS *= 2;
            Lm = LS;
        }
    } else {
        for (int i = threadIdx.x; i < B * B; i += SC_BLOCK) {
            const int b = i / B, b2 = i - b * B;
            double r = 0;
            for (int k = 0; k < K; ++k) r += (double)sed_s[k * BM + b] * sed_s[k * BM + b2];
            ATA[b * BM + b2] = r;
        }
        __syncthreads();
        if (threadIdx.x < SC_WAVE) {
            double l = 0;
            if (B <= 4) { if (threadIdx.x == 0) l = lambda_max_charpoly4(ATA, B, BM); }
            else l = wave_lambda_max8(ATA, B, BM, eigbuf);
            if (threadIdx.x == 0) Lm = l;
        }
    }
    __syncthreads();
    const float step_morph = 1.0f / (float)Lm;
    // the SEDs as scalars (SGPRs): hoisted LDS reads would hold KM * BM vector registers across the loop
    float sk[KM][BM];
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int b = 0; b < BM; ++b)
            sk[k][b] = (k < K && b < B) ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sed_s[k * BM + b]))) : 0.f;
    float dsed[KM][BM];
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int b = 0; b < BM; ++b) dsed[k][b] = 0.f;
    const int HW4 = HW >> 2;
    const float4 *mor = reinterpret_cast<const float4 *>(a.morph[c0] + (size_t)s * K * HW);
    float4 *mout = reinterpret_cast<float4 *>(a.morph[1 - c0] + (size_t)s * K * HW);
    const float4 *G = reinterpret_cast<const float4 *>(a.real + (size_t)s * B * HW);
    const int g_end = min(HW4, (tile + 1) * (SC_TILE_PIX >> 2));
#pragma unroll 1
    for (int g = tile * (SC_TILE_PIX >> 2) + threadIdx.x; g < g_end; g += SC_BLOCK) {
        float4 gb[BM];
#pragma unroll
        for (int b = 0; b < BM; ++b) gb[b] = b < B ? G[(size_t)b * HW4 + g] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (k < K) {
                const float4 m = mor[(size_t)k * HW4 + g];
                float4 gm = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int b = 0; b < BM; ++b)
                    if (b < B) {
                        gm.x += sk[k][b] * gb[b].x; gm.y += sk[k][b] * gb[b].y; gm.z += sk[k][b] * gb[b].z; gm.w += sk[k][b] * gb[b].w;
                        dsed[k][b] += gb[b].x * m.x; dsed[k][b] += gb[b].y * m.y;
                        dsed[k][b] += gb[b].z * m.z; dsed[k][b] += gb[b].w * m.w;
                    }
                const bool fixed = a.fix_morph && a.fix_morph[(size_t)s * K + k];
                float4 o;
                if (a.raw_gradient) o = gm;
                else if (fixed) o = m;
                else o = make_float4(m.x - step_morph * gm.x, m.y - step_morph * gm.y, m.z - step_morph * gm.z, m.w - step_morph * gm.w);
                mout[(size_t)k * HW4 + g] = o;
            }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int b = 0; b < BM; ++b)
            if (k < K && b < B) {
                const double v = wave_sum((double)dsed[k][b]);
                if (lane == 0) red[wid][k * B + b] = v;
            }
    __syncthreads();
    const int P = n_partials(K, B);
    double *out = a.partials + ((size_t)s * a.T + tile) * P;
    for (int i = threadIdx.x; i < K * B; i += SC_BLOCK) {
        double r = 0;
#pragma unroll
        for (int w = 0; w < SC_NWAVES; ++w) r += red[w][i];
        out[1 + i] = r;
    }
    if (threadIdx.x == 0) {
        double l = 0;
        if (tile == 0) for (int b = 0; b < B; ++b) l += a.loss_part[s * B + b];
        out[0] = l;
    }
}

// one workgroup per scene: the scalar head of the step (partial sums, Lipschitz constants, SED step, loss record)
template <int KM, int BM>
__global__ __launch_bounds__(SC_BLOCK) void k_sed_step(PsfArgs a)
{
    const int s = blockIdx.x;
    if (!a.active[s]) return;
    __shared__ double tot[1 + KM * BM + KM * (KM + 1) / 2];
    __shared__ double mat[KM * KM + BM * BM];
    __shared__ float sed_s[KM * BM];
    __shared__ float step_s[2];
    __shared__ double Lc[2];
    __shared__ double eigbuf[2][2][64];
    step_psf_head<KM, BM>(a, s, 0, tot, mat, sed_s, step_s, Lc, eigbuf);
}

// kernel image [B][Py][Px] -> padded, shifted FFT input planes [B][Fy][Fx]
__global__ void k_psf_pad_kernel(const float *ker, int B, int Py, int Px, int Fy, int Fx, int oky, int okx,
                                 float *out)
{
    const int64_t total = (int64_t)B * Fy * Fx;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / ((int64_t)Fy * Fx)), e = (int)(i - (int64_t)b * Fy * Fx);
        const int iy = e / Fx, ix = e - iy * Fx;
        const int q = pos_mod(iy - oky, Fy), r = pos_mod(ix - okx, Fx);
        out[i] = (q < Py && r < Px) ? ker[((size_t)b * Py + q) * Px + r] : 0.f;
    }
}

// generic plane pad / crop for the standalone render (Observation.render)
__global__ void k_plane_pad(const float *in, int n, int H, int W, int Fy, int Fx, int oy, int ox, float *out)
{
    const int64_t total = (int64_t)n * Fy * Fx;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i / ((int64_t)Fy * Fx)), e = (int)(i - (int64_t)p * Fy * Fx);
        const int iy = e / Fx, ix = e - iy * Fx;
        const int y = pos_mod(iy - oy, Fy), x = pos_mod(ix - ox, Fx);
        out[i] = (y < H && x < W) ? in[((size_t)p * H + y) * W + x] : 0.f;
    }
}
__global__ void k_plane_crop(const float *in, int n, int H, int W, int Fy, int Fx, int oy, int ox, float *out)
{
    const int64_t total = (int64_t)n * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i / ((int64_t)H * W)), e = (int)(i - (int64_t)p * H * W);
        const int y = e / W, x = e - y * W;
        out[i] = in[((size_t)p * Fy + pos_mod(y + oy, Fy)) * Fx + pos_mod(x + ox, Fx)];
    }
}

// spectrum ratio a / b (fft.match_psfs): b has nb planes (1: the same for every plane of a)
__global__ void k_spec_div(float2 *a, const float2 *b, int nb, int plane_elems, int64_t total, float scale)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int plane = (int)(i / plane_elems), e = (int)(i - (int64_t)plane * plane_elems);
        const float2 d = b[(size_t)(plane % nb) * plane_elems + e];
        const float2 v = a[i];
        const float den = d.x * d.x + d.y * d.y;
        a[i] = make_float2((v.x * d.x + v.y * d.y) / den * scale, (v.y * d.x - v.x * d.y) / den * scale);
    }
}
