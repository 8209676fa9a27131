// engine.h -- kernels of the batched Blend.fit() iteration.
//
// HBM layout (DESIGN.md "Data layout"): scene-major, images [S][B][H][W], morph
// [S][K][H][W], sed [S][K][B]; factors are ping-pong buffered so that the buffer not
// being written always holds the previous iteration (the reference's _last_* copies).
//
// One iteration =
//   k_grad         grid (T, S)  : stream images + morphs once, per-tile partial sums of
//                                 loss, d loss/d sed, morph Gram (f64)           [a1-a5]
//   k_step         grid (T, S)  : reduce partials, Lipschitz constants (Jacobi eigen-
//                                 solve), SED step, recompute G and step the morphs [a6, a7]
//   k_source_update grid (S*K)  : constraint pipeline in LDS (prox_ops.h)        [a8-a17]
//   k_converge     grid (S/256) : flags / active / it bookkeeping                [a18]
#pragma once
#include "common.h"
#include "prox_ops.h"
#include "wave_ops.h"

#define SC_TILE_PIX 4096          // pixels per (scene, tile) workgroup in k_grad / k_step
#define SC_KMAX 8               // register-tiled gradient kernels
#define SC_KBIG 32              // chunked gradient kernels (bigk.h)
#define SC_BMAX 8

__host__ __device__ inline int n_partials(int K, int B) { return 1 + K * B + K * (K + 1) / 2; }

struct GradArgs {
    int S, K, B, HW, T;
    const float *images, *weights;
    float weight_scalar;
    float *sed[2], *morph[2];     // ping-pong factor buffers
    const int *cur;               // [S] index of the current buffer of each scene
    const uint8_t *fix_sed, *fix_morph;
    double *partials;             // [S][T][P]
    double *lipschitz, *mse;
    int mse_capacity;
    int *it;
    const int *active;
    int approximate_L;
    int raw_gradient;             // 1: write d loss/d sed and d loss/d morph instead of the stepped factors
};

// ------------------------------------------------------------------------------------
// k_grad: per pixel p of the tile
//   model_b = sum_k sed[k][b] morph[k][p]           (a1, a2; render is the identity, a3)
//   d_b = w (model_b - image_b), loss += d_b^2 / 2   (a4)
//   G_b = w d_b ; dsed[k][b] += G_b morph[k][p]      (a5)
//   gram[k][k'] += morph[k][p] morph[k'][p]          (a6, S S^T)
// Loads: lane-contiguous float per (band | component) plane -> fully coalesced.
template <int KM, int BM>
__global__ __launch_bounds__(SC_BLOCK) void k_grad(GradArgs a)
{
    const int s = blockIdx.y, tile = blockIdx.x;
    if (!a.active[s]) return;
    const int K = a.K, B = a.B, HW = a.HW;
    __shared__ float sed_s[KM * BM];
    __shared__ double red[SC_NWAVES][1 + KM * BM + KM * (KM + 1) / 2];
    const int c0 = a.cur[s];
    const float *sed_in = a.sed[c0];
    for (int i = threadIdx.x; i < K * B; i += SC_BLOCK)
        sed_s[(i / B) * BM + (i % B)] = sed_in[(size_t)s * K * B + i];
    __syncthreads();
    float sed[KM][BM];
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int b = 0; b < BM; ++b) sed[k][b] = (k < K && b < B) ? sed_s[k * BM + b] : 0.f;

    double loss = 0;
    float dsed[KM][BM];
    float gram[KM * (KM + 1) / 2];
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int b = 0; b < BM; ++b) dsed[k][b] = 0.f;
#pragma unroll
    for (int i = 0; i < KM * (KM + 1) / 2; ++i) gram[i] = 0.f;

    const float *img = a.images + (size_t)s * B * HW;
    const float *wgt = a.weights ? a.weights + (size_t)s * B * HW : nullptr;
    const float *mor = a.morph[c0] + (size_t)s * K * HW;
    const int p_end = min(HW, (tile + 1) * SC_TILE_PIX);
    for (int p = tile * SC_TILE_PIX + threadIdx.x; p < p_end; p += SC_BLOCK) {
        float m[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) m[k] = k < K ? mor[(size_t)k * HW + p] : 0.f;
#pragma unroll
        for (int b = 0; b < BM; ++b) {
            if (b < B) {
                float model = 0.f;
#pragma unroll
                for (int k = 0; k < KM; ++k) model += sed[k][b] * m[k];
                const float w = wgt ? wgt[(size_t)b * HW + p] : a.weight_scalar;
                const float d = w * (model - img[(size_t)b * HW + p]);
                loss += (double)d * (double)d;
                const float g = w * d;
#pragma unroll
                for (int k = 0; k < KM; ++k) dsed[k][b] += g * m[k];
            }
        }
        int gi = 0;
#pragma unroll
        for (int k = 0; k < KM; ++k)
#pragma unroll
            for (int k2 = k; k2 < KM; ++k2) gram[gi++] += m[k] * m[k2];
    }
    // block reduction (f64) -> partials[s][tile][:]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double v = wave_sum(0.5 * loss);
    if (lane == 0) red[wid][0] = v;
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int b = 0; b < BM; ++b) {
            if (k < K && b < B) {
                v = wave_sum((double)dsed[k][b]);
                if (lane == 0) red[wid][1 + k * B + b] = v;
            }
        }
    {
        int gi = 0, go = 0;
#pragma unroll
        for (int k = 0; k < KM; ++k)
#pragma unroll
            for (int k2 = k; k2 < KM; ++k2) {
                if (k < K && k2 < K) {
                    v = wave_sum((double)gram[gi]);
                    if (lane == 0) red[wid][1 + K * B + go] = v;
                    ++go;
                }
                ++gi;
            }
    }
    __syncthreads();
    const int P = n_partials(K, B);
    double *out = a.partials + ((size_t)s * a.T + tile) * P;
    for (int i = threadIdx.x; i < P; i += SC_BLOCK) {
        double r = 0;
#pragma unroll
        for (int w = 0; w < SC_NWAVES; ++w) r += red[w][i];
        out[i] = r;
    }
}

// ------------------------------------------------------------------------------------
// Largest eigenvalue of a symmetric n x n matrix (n <= 8), cyclic Jacobi in float64.
// Replaces np.linalg.eigvals(...).max() of blend.py:216-218 (the matrices are Gram
// matrices: real symmetric PSD, so the general solver's result is real).
__device__ inline double jacobi_lambda_max(double *A, int n, int ld)
{
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; ++i) {
            diag += A[i * ld + i] * A[i * ld + i];
            for (int j = i + 1; j < n; ++j) off += A[i * ld + j] * A[i * ld + j];
        }
        if (off <= 1e-30 * diag || off == 0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * ld + q];
                if (apq == 0) continue;
                const double theta = (A[q * ld + q] - A[p * ld + p]) / (2 * apq);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(tt * tt + 1), sn = tt * c;
                for (int r = 0; r < n; ++r) {
                    const double arp = A[r * ld + p], arq = A[r * ld + q];
                    A[r * ld + p] = c * arp - sn * arq;
                    A[r * ld + q] = sn * arp + c * arq;
                }
                for (int r = 0; r < n; ++r) {
                    const double apr = A[p * ld + r], aqr = A[q * ld + r];
                    A[p * ld + r] = c * apr - sn * aqr;
                    A[q * ld + r] = sn * apr + c * aqr;
                }
            }
    }
    double l = A[0];
    for (int i = 1; i < n; ++i) l = fmax(l, A[i * ld + i]);
    return l;
}

// Largest eigenvalue of a symmetric PSD n x n matrix, n <= 4, from its characteristic
// polynomial in float64:  p(x) = x^4 - c1 x^3 + c2 x^2 - c3 x + c4  with ck = the sum of the
// k x k principal minors (rows/columns >= n count as zero).  All roots are real and
// p, p', p'' > 0 to the right of the largest one, so Newton's iteration started at the
// trace (>= lambda_max) decreases monotonically onto it.  ~10 x fewer dependent
// instructions than a Jacobi sweep; the root's conditioning is that of the eigenvalue
// (error ~1e-16 separated, ~1e-8 relative for a double top eigenvalue).
// Replaces np.linalg.eigvals(...).max() of blend.py:216-218.
__device__ inline double lambda_max_charpoly4(const double *A, int n, int ld)
{
    double m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) m[i][j] = (i < n && j < n) ? A[i * ld + j] : 0.0;
    const double a = m[0][0], b = m[1][1], c = m[2][2], d = m[3][3];
    const double e = m[0][1], f = m[0][2], g = m[0][3], h = m[1][2], k = m[1][3], l = m[2][3];
    const double c1 = (a + b) + (c + d);
    // 2 x 2 minors of rows (0,1) and of rows (2,3), column pairs 01 02 03 12 13 23
    const double p01 = a * b - e * e, p02 = a * h - e * f, p03 = a * k - e * g;
    const double p12 = e * h - b * f, p13 = e * k - b * g, p23 = f * k - h * g;
    const double q01 = f * k - g * h, q02 = f * l - g * c, q03 = f * d - g * l;
    const double q12 = h * l - k * c, q13 = h * d - k * l, q23 = c * d - l * l;
    const double c2 = p01 + (a * c - f * f) + (a * d - g * g) + (b * c - h * h) + (b * d - k * k) + q23;
    // 3 x 3 principal minors
    const double d012 = c * p01 - h * p02 + f * p12;          // rows/cols 0,1,2 (expansion along row 2)
    const double d013 = d * p01 - k * p03 + g * p13;          // rows/cols 0,1,3 (along row 3)
    const double d023 = a * q23 - f * (f * d - l * g) + g * (f * l - c * g);
    const double d123 = b * q23 - h * (h * d - l * k) + k * (h * l - c * k);
    const double c3 = (d012 + d013) + (d023 + d123);
    // Laplace expansion over rows (0,1) x rows (2,3)
    const double c4 = p01 * q23 - p02 * q13 + p03 * q12 + p12 * q03 - p13 * q02 + p23 * q01;
    if (!(c1 > 0.0)) return c1;                                // zero / NaN matrix: 1/L is inf / NaN as in the reference
    // start: the smaller of two upper bounds of lambda_max, the trace (tight when one eigenvalue
    // dominates: SEDs of similar colour) and the largest absolute row sum (Gershgorin; tight when the
    // matrix is nearly diagonal: well separated morphologies).  Newton descends monotonically from any
    // point above the largest root.
    const double r0 = a + (fabs(e) + fabs(f) + fabs(g)), r1 = b + (fabs(e) + fabs(h) + fabs(k));
    const double r2 = c + (fabs(f) + fabs(h) + fabs(l)), r3 = d + (fabs(g) + fabs(k) + fabs(l));
    double x = fmin(c1, fmax(fmax(r0, r1), fmax(r2, r3)) * (1.0 + 0x1p-50));
    for (int it = 0; it < 64; ++it) {
        const double pv = (((x - c1) * x + c2) * x - c3) * x + c4;
        const double dv = ((4.0 * x - 3.0 * c1) * x + 2.0 * c2) * x - c3;
        if (!(pv > 0.0) || !(dv > 0.0)) break;                 // at (or, by rounding, just past) the root
        // pv / dv by one refinement of the hardware reciprocal (relative error ~1e-16: far inside the
        // step's own tolerance, and without the IEEE division's dozen dependent instructions)
        double rc = __builtin_amdgcn_rcp(dv);
        rc = fma(fma(-dv, rc, 1.0), rc, rc);
        const double dx = pv * rc;
        x -= dx;
        if (dx <= 1e-15 * x) break;
    }
    return x;
}

// Largest eigenvalue of an n x n PSD matrix, n <= 8, by ONE wave (all 64 lanes must call): lane (i, j)
// owns one element.  M = A / tr A is squared 30 times (renormalised by its trace each time), which
// leaves u1 u1^T up to terms (lambda_i / lambda_1)^(2^30); the Rayleigh quotient of its heaviest
// column with the ORIGINAL matrix misses lambda_1 by at most n / (e 2^31) relative whatever the
// spectral gaps (same scheme as k_bigk_lipschitz).  ~10k cycles instead of the ~200k of a single-lane
// Jacobi on an LDS-resident 8 x 8 matrix.  `buf`: 2 x 64 doubles of LDS owned by the calling wave.
__device__ inline double wave_lambda_max8(const double *A, int n, int ld, double (*buf)[64])
{
    const int lane = threadIdx.x & (SC_WAVE - 1), i = lane >> 3, j = lane & 7;
    const double a = (i < n && j < n) ? A[i * ld + j] : 0.0;
    const double tr0 = wave_sum(i == j ? a : 0.0);
    buf[0][lane] = a / tr0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    int cur = 0;
    for (int q = 0; q < 30; ++q) {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += buf[cur][(i << 3) + k] * buf[cur][(k << 3) + j];
        const double t = wave_sum(i == j ? acc : 0.0);
        buf[1 - cur][lane] = acc * (1.0 / t);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        cur ^= 1;
        // tr(M^2) of the trace-normalised M is sum lambda_i^2 / (sum lambda_i)^2: 1 - t ~ 2 (lambda_2 / lambda_1)^(2^q).
        // Once that is below float64 resolution M is u1 u1^T to round-off and further squarings change nothing
        // (typically after 6 - 8 of the 30; with a (near-)degenerate top eigenvalue t never gets there and all 30 run)
        if (1.0 - t < 1e-15) break;
    }
    int best = 0;
    double bv = buf[cur][0];
#pragma unroll
    for (int k = 1; k < 8; ++k) { const double d = buf[cur][k * 9]; if (d > bv) { bv = d; best = k; } }
    const double vi = buf[cur][(i << 3) + best], vj = buf[cur][(j << 3) + best];
    const double num = wave_sum(vi * a * vj), den = wave_sum(j == 0 ? vi * vi : 0.0);
    return num / den;
}

// The two Lipschitz constants of a scene from LDS-resident Gram matrices (blend.py:205-218): wave 0
// takes S S^T (n = K), wave 1 takes A^T A (n = B); n <= 4 uses the characteristic polynomial on one
// lane.  Every thread of the workgroup must call; results in L[0], L[1] after the trailing barrier.
__device__ inline void block_lipschitz(double *G, int K, int ldK, double *ATA, int B, int ldB,
                                       double (*buf)[2][64], double *L)
{
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wid < 2) {
        double *M = wid == 0 ? G : ATA;
        const int n = wid == 0 ? K : B, ld = wid == 0 ? ldK : ldB;
        double l = 0;
        if (n <= 4) { if (lane == 0) l = lambda_max_charpoly4(M, n, ld); }
        else l = wave_lambda_max8(M, n, ld, buf[wid]);
        if (lane == 0) L[wid] = l;
    }
    __syncthreads();
}

// Register-resident variant for the fused kernel: the whole (padded) N x N matrix lives in
// VGPRs of ONE lane, loops fully unrolled (no LDS round trips on the rotation chain).
// Rotation angles are computed in float32 (one v_rcp/v_sqrt each), then (c, s) is
// re-orthonormalised in float64 with one Newton step, so the similarity transforms stay
// orthogonal to ~1e-14 while the scalar part costs a handful of instructions; an angle
// that is only float32-accurate merely leaves a 1e-7-relative off-diagonal residue that the
// next sweep removes.  Eigenvalues are accurate to float64 round-off.
template <int N>
__device__ inline double jacobi_lambda_max_reg(const double *Ain, int n, int ld)
{
    double A[N][N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) A[i][j] = (i < n && j < n) ? Ain[i * ld + j] : 0.0;
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0, diag = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            diag += A[i][i] * A[i][i];
#pragma unroll
            for (int j = i + 1; j < N; ++j) off += A[i][j] * A[i][j];
        }
        if (off <= 1e-30 * diag) break;
#pragma unroll
        for (int p = 0; p < N - 1; ++p)
#pragma unroll
            for (int q = p + 1; q < N; ++q) {
                const double apq = A[p][q];
                if (apq != 0.0) {
                    const float th = (float)((A[q][q] - A[p][p]) / (2.0 * apq));
                    const float tf = (th >= 0.f ? 1.f : -1.f) / (fabsf(th) + sqrtf(th * th + 1.f));
                    const float cf = 1.0f / sqrtf(tf * tf + 1.f);
                    double c = (double)cf, sn = (double)(tf * cf);
                    const double fix = 1.5 - 0.5 * (c * c + sn * sn);      // Newton step of rsqrt
                    c *= fix; sn *= fix;
#pragma unroll
                    for (int r = 0; r < N; ++r) {
                        const double arp = A[r][p], arq = A[r][q];
                        A[r][p] = c * arp - sn * arq;
                        A[r][q] = sn * arp + c * arq;
                    }
#pragma unroll
                    for (int r = 0; r < N; ++r) {
                        const double apr = A[p][r], aqr = A[q][r];
                        A[p][r] = c * apr - sn * aqr;
                        A[q][r] = sn * apr + c * aqr;
                    }
                }
            }
    }
    double l = A[0][0];
#pragma unroll
    for (int i = 1; i < N; ++i) l = fmax(l, A[i][i]);
    return l;
}

// ------------------------------------------------------------------------------------
// k_step: finish _backward (mse), _set_lipschitz, and the gradient step of Blend.fit.
template <int KM, int BM>
__global__ __launch_bounds__(SC_BLOCK) void k_step(GradArgs a)
{
    const int s = blockIdx.y, tile = blockIdx.x;
    if (!a.active[s]) return;
    const int K = a.K, B = a.B, HW = a.HW, P = n_partials(K, B);
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
    // SED step (blend.py:91-93); tile 0 writes the new SEDs into the other buffer
    if (tile == 0) {
        for (int i = threadIdx.x; i < K * B; i += SC_BLOCK) {
            const int k = i / B;
            const float cur = sed_s[k * BM + (i % B)];
            const bool fixed = a.fix_sed && a.fix_sed[(size_t)s * K + k];
            sed_out[(size_t)s * K * B + i] = a.raw_gradient ? (float)tot[1 + i] : (fixed ? cur : cur - step_sed * (float)tot[1 + i]);
        }
    }
    // morphology step (blend.py:94-96): G recomputed from the streamed tiles
    float sed[KM][BM];
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int b = 0; b < BM; ++b) sed[k][b] = (k < K && b < B) ? sed_s[k * BM + b] : 0.f;
    bool fixm[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) fixm[k] = (k < K) && a.fix_morph && a.fix_morph[(size_t)s * K + k];
    const float *img = a.images + (size_t)s * B * HW;
    const float *wgt = a.weights ? a.weights + (size_t)s * B * HW : nullptr;
    const float *mor = a.morph[c0] + (size_t)s * K * HW;
    float *mout = a.morph[1 - c0] + (size_t)s * K * HW;
    const int p_end = min(HW, (tile + 1) * SC_TILE_PIX);
    for (int p = tile * SC_TILE_PIX + threadIdx.x; p < p_end; p += SC_BLOCK) {
        float m[KM], gm[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) { m[k] = k < K ? mor[(size_t)k * HW + p] : 0.f; gm[k] = 0.f; }
#pragma unroll
        for (int b = 0; b < BM; ++b) {
            if (b < B) {
                float model = 0.f;
#pragma unroll
                for (int k = 0; k < KM; ++k) model += sed[k][b] * m[k];
                const float w = wgt ? wgt[(size_t)b * HW + p] : a.weight_scalar;
                const float g = w * (w * (model - img[(size_t)b * HW + p]));
#pragma unroll
                for (int k = 0; k < KM; ++k) gm[k] += sed[k][b] * g;
            }
        }
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (k < K) mout[(size_t)k * HW + p] = a.raw_gradient ? gm[k] : (fixm[k] ? m[k] : m[k] - step_morph * gm[k]);
    }
    // a.it[s] is advanced by k_converge (every tile of this launch reads the old value)
}

// ------------------------------------------------------------------------------------
// k_source_update: PointSource.update / ExtendedSource.update (source.py:402-440) for one
// component per workgroup, entirely in LDS.
struct UpdateArgs {
    int S, K, B, H, W;
    float *sed[2], *morph[2];         // ping-pong factor buffers
    const int *cur;                   // [S]
    int in_iteration;                 // 1: called between k_step and k_converge -> operate on
                                      //    buffer 1-cur (just written), previous = buffer cur
                                      // 0: standalone -> operate on buffer cur, no a18 sums
    int *centers; double *shifts;
    const double *lipschitz;
    const int *it; const int *active;
    int *status;
    int symmetric, monotonic;
    float l0_thresh, l1_thresh;
    const double *centroid_psf; int centroid_P;
    double *conv;                     // [S][K][4]: d2_sed, n2_sed, d2_morph, n2_morph
    int force_it0;                    // 1: constructor call (it = 0, ignore `active`)
    float *gscratch;                  // GT kernels: [S*K][round16(H) * scratch_stride(round16(W))] (T = X B)
    int hybrid_sweep;                 // LDS tiles: sweep levels 1 .. 46 on one wave (wave_monotonic), the rest on the workgroup
    const int *only_flagged;          // [S*K] or NULL: run only for the components k_source_update_box left to the full path
    const int *group;                 // [S*K] or NULL: >= 0 = layer of a multi-component source: centre given (k_group_centers),
                                      // shift = None (soft symmetry); -1 = a source of its own
};

// MODE 0/1: the morphology tile lives in LDS (tiles up to ~128 x 128).
// MODE 2   : frames whose tile does not fit (up to 256 x 256, BASELINE config 5): the same
//             operators run IN PLACE on the morphology plane in HBM / L2 with the scratch in a
//             global workspace; only the Hankel vectors are in LDS.  Threads of the workgroup
//             exchange pixels through global memory across __syncthreads(), exactly as they do
//             through LDS in the other variant.
//   MODE 0 tile + GEMM scratch in LDS;  MODE 1 tile in LDS, scratch in HBM (halves the LDS
//   footprint of a 128 x 128 frame: two workgroups per CU);  MODE 2 both in HBM (the text above)
#define SC_BOX_R (SC_COMPACT_LAST / 2)           // levels <= 46 stay within 23 pixels of the peak
#define SC_BOX_LW (2 * SC_BOX_R + 3)
#define SC_BOX_FLOATS ((2 * SC_BOX_R + 1) * SC_BOX_LW)
template <int MODE>
__global__ __launch_bounds__(SC_BLOCK) void k_source_update(UpdateArgs a)
{
    constexpr bool GT = MODE == 2;
    extern __shared__ __align__(16) float lds[];
    const int c = blockIdx.x, s = c / a.K;
    if (!a.force_it0 && !a.active[s]) return;
    if (a.only_flagged && !a.only_flagged[c]) return;
    const int H = a.H, W = a.W, HW = H * W, B = a.B;
    const int hp = round16(H), wp = round16(W);
    Tile t; t.H = H; t.W = W;
    float *scr, *av;
    if (GT) {
        t.LW = W; t.m = nullptr;                    // set below: the plane itself
        scr = a.gscratch + (size_t)c * hp * scratch_stride(wp);
        av = lds;
    } else if (MODE == 1) {
        t.LW = tile_stride(W); t.m = lds;
        scr = a.gscratch + (size_t)c * hp * scratch_stride(wp);
        av = lds + H * t.LW;
    } else {
        t.LW = tile_stride(W); t.m = lds;
        scr = lds + H * t.LW;
        av = scr + hp * scratch_stride(wp);
    }
    float *bv = av + 2 * hp, *cv = bv + 2 * wp, *zv = cv + 2 * wp;
    float *stage = MODE == 0 ? nullptr : zv + wp;   // LDS staging of the operands that live in HBM
    __shared__ double red[SC_NWAVES];
    __shared__ float redf[SC_NWAVES];
    __shared__ int ctr[2];
    __shared__ double shf[2];
    __shared__ int stat;
    __shared__ int lastpos;
    __shared__ int hyb[2];

    const int c0 = a.cur[s];
    const int wbuf = a.in_iteration ? 1 - c0 : c0;
    float *gm = a.morph[wbuf] + (size_t)c * HW;
    // float4 groups walked without divisions: +256 groups per step
    const bool vec4 = (W & 3) == 0;
    const int gpr = W >> 2, ngroups = HW >> 2;
    const int dyq = vec4 ? SC_BLOCK / gpr : 0, dxq = vec4 ? SC_BLOCK - dyq * gpr : 0;
    const int y0 = vec4 ? (int)threadIdx.x / gpr : 0, x0 = vec4 ? (int)threadIdx.x - y0 * gpr : 0;
    if (GT) t.m = gm;
    else if (vec4) {
        int y = y0, xq = x0;
        for (int g = threadIdx.x; g < ngroups; g += SC_BLOCK) {
            lds_store4(t.m + y * t.LW + (xq << 2), reinterpret_cast<const float4 *>(gm)[g]);
            y += dyq; xq += dxq;
            if (xq >= gpr) { xq -= gpr; ++y; }
        }
    } else
        for (int i = threadIdx.x; i < HW; i += SC_BLOCK) {
            const int y = i / W, x = i - y * W;
            t.m[y * t.LW + x] = gm[i];
        }
    if (threadIdx.x == 0) stat = 0;
    __syncthreads();
    const int it = a.force_it0 ? 0 : a.it[s] + (a.in_iteration ? 1 : 0);   // len(mse)
    int cy = a.centers[2 * c], cx = a.centers[2 * c + 1];
    const bool grouped = a.group && a.group[c] >= 0;               // layer of a MultiComponentSource: centre from k_group_centers

    if (!grouped) {
        max_pixel_tile(t, cy, cx, ctr, &stat);                     // source.py:414
        cy = ctr[0]; cx = ctr[1];
    }
    if (a.symmetric) {
        double dy = grouped ? (double)__builtin_nanf("") : a.shifts[2 * c], dx = grouped ? dy : a.shifts[2 * c + 1];
        if (!grouped && it % 5 == 0) {                              // source.py:428-429
            __syncthreads();
            centroid_tile(t, a.centroid_psf, a.centroid_P, cy, cx, red, ctr, shf, &stat);
            cy = ctr[0]; cx = ctr[1]; dy = shf[0]; dx = shf[1];
            if (threadIdx.x == 0) { a.shifts[2 * c] = dy; a.shifts[2 * c + 1] = dx; }
        }
        // update.symmetric(algorithm="kspace") (source.py:432): shift None -> soft, s=1
        const bool none = (dy != dy);
        symmetry_tile(t, cy, cx, none ? SCARLET_SYM_SOFT : SCARLET_SYM_KSPACE, 1.0f, dy, dx,
                      false, 0.f, scr, av, bv, cv, zv, stage, GT);
    }
    int lstop = 1 << 30;          // last sweep level computed (early exit); pixels beyond are <= 0 -> 0
    if (a.monotonic) {                                                                   // source.py:436
        if (!GT && a.hybrid_sweep) {
            // tile in LDS: levels 1 .. 46 on one wave without barriers (wave_ops.h; a sweep usually stops
            // before), the rest level-synchronously on the whole workgroup
            if (threadIdx.x < SC_WAVE) {
                int done, quiet;
                wave_monotonic<float>(t, cy, cx, 0.f, &done, &quiet);
                if (threadIdx.x == 0) { hyb[0] = done; hyb[1] = quiet; }
            }
            __syncthreads();
            lstop = hyb[0];
            if (lstop == (1 << 30))
                lstop = monotonic_tile<false, float>(t, cy, cx, 0.f, &lastpos, 0.f, SC_COMPACT_LAST + 1, SC_COMPACT_LAST - hyb[1]);
        } else if (GT && a.hybrid_sweep && stage_floats(hp, wp) >= SC_BOX_FLOATS) {
            // plane in HBM: the pixels of levels 1 .. 46 (and their closer neighbours) lie within 23 pixels of the
            // peak -- that box goes to LDS, one wave sweeps it without barriers, the box goes back, and the
            // workgroup-level sweep on the plane continues from level 47 only if the wave did not stop.  Inside
            // the box every bounds decision of the walkers coincides with the frame's (b <= 15, a <= 23).
            float *box = stage;       // free during the sweep; stage_floats >= 16 * 144 >= SC_BOX_FLOATS for every frame that takes this path
            const int by0 = max(0, cy - SC_BOX_R), bx0 = max(0, cx - SC_BOX_R);
            const int bh = min(H, cy + SC_BOX_R + 1) - by0, bw = min(W, cx + SC_BOX_R + 1) - bx0;
            Tile bt; bt.H = bh; bt.W = bw; bt.LW = SC_BOX_LW; bt.m = box;
            for (int i = threadIdx.x; i < bh * bw; i += SC_BLOCK) {
                const int y = i / bw, x = i - y * bw;
                box[y * SC_BOX_LW + x] = t.m[(by0 + y) * t.LW + bx0 + x];
            }
            __syncthreads();
            if (threadIdx.x < SC_WAVE) {
                int done, quiet;
                wave_monotonic<float>(bt, cy - by0, cx - bx0, 0.f, &done, &quiet);
                if (threadIdx.x == 0) { hyb[0] = done; hyb[1] = quiet; }
            }
            __syncthreads();
            for (int i = threadIdx.x; i < bh * bw; i += SC_BLOCK) {
                const int y = i / bw, x = i - y * bw;
                t.m[(by0 + y) * t.LW + bx0 + x] = box[y * SC_BOX_LW + x];
            }
            __syncthreads();
            lstop = hyb[0];
            if (lstop == (1 << 30))
                lstop = monotonic_tile<false, float>(t, cy, cx, 0.f, &lastpos, 0.f, SC_COMPACT_LAST + 1, SC_COMPACT_LAST - hyb[1]);
        } else
            lstop = monotonic_tile<false, float>(t, cy, cx, 0.f, &lastpos);
    }
    if (threadIdx.x == 0) { a.centers[2 * c] = cy; a.centers[2 * c + 1] = cx; }

    // sparse_l0 / sparse_l1 (update.py:71-82; config 5), positive (update.py:27-32),
    // normalized('morph_max') (update.py:62-65)
    const float step_morph = 1.0f / (float)a.lipschitz[2 * s + 1];
    const float *gl = a.in_iteration ? a.morph[c0] + (size_t)c * HW : nullptr;
    const float l0 = a.l0_thresh >= 0.f ? a.l0_thresh * step_morph : -1.f;
    const float l1 = a.l1_thresh >= 0.f ? a.l1_thresh * step_morph : -1.f;
    auto sparse_plus = [&](float v, int y, int x) {      // update.py:71-82, 27-32 + the sweep's cut
        if (l0 >= 0.f && fabsf(v) < l0) v = 0.f;
        if (l1 >= 0.f) {
            const float mag = fabsf(v) - l1;
            v = (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f)) * (mag < 0.f ? 0.f : mag);
        }
        if (v < 0.f || sweep_level(y, x, cy, cx) > lstop) v = 0.f;
        return v;
    };
    float norm;
    double d2 = 0, n2 = 0;
    if (vec4 && a.monotonic) {
        // one pass in float4 groups; morph.max() is the processed peak pixel after a sweep (see
        // wave_pipeline below); a NaN elsewhere shows up in the sums
        __syncthreads();
        norm = sparse_plus(t.m[cy * t.LW + cx], cy, cx);
        __syncthreads();                                 // (in place when GT: read the peak before it is normalised)
        const bool regular = norm > 0.f && !isinf(norm);
        const float rnorm = 1.0f / norm;
        float d2f = 0.f, n2f = 0.f;
        int y = y0, xq = x0;
        for (int g = threadIdx.x; g < ngroups; g += SC_BLOCK) {
            const float *p = t.m + y * t.LW + (xq << 2);
            const float4 v4 = GT ? *reinterpret_cast<const float4 *>(p) : lds_load4(p);
            float v[4] = {v4.x, v4.y, v4.z, v4.w}, o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = sparse_plus(v[e], y, (xq << 2) + e);
                if (regular) { const float q = v[e] * rnorm; o[e] = fmaf(fmaf(-q, norm, v[e]), rnorm, q); }   // v / norm, see fused2.h
                else o[e] = v[e] / norm;
            }
            reinterpret_cast<float4 *>(gm)[g] = make_float4(o[0], o[1], o[2], o[3]);
            if (gl) {
                const float4 l = reinterpret_cast<const float4 *>(gl)[g];
                const float e0 = l.x - o[0], e1 = l.y - o[1], e2 = l.z - o[2], e3 = l.w - o[3];
                d2f += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
            }
            n2f += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
            y += dyq; xq += dxq;
            if (xq >= gpr) { xq -= gpr; ++y; }
        }
        d2 = (double)d2f; n2 = (double)n2f;
        d2 = block_sum(d2, red);
        n2 = block_sum(n2, red);
        if (n2 != n2 && norm == norm) {
            // a NaN pixel away from the peak: np.max is NaN and the reference's morph becomes NaN everywhere
            norm = __builtin_nanf("");
            for (int g = threadIdx.x; g < ngroups; g += SC_BLOCK) reinterpret_cast<float4 *>(gm)[g] = make_float4(norm, norm, norm, norm);
            d2 = norm;
        }
    } else {
        float vmax = -INFINITY;
        bool anynan = false;
        for (int i = threadIdx.x; i < HW; i += SC_BLOCK) {
            const int y = i / W, x = i - y * W;
            const float v = sparse_plus(t.m[y * t.LW + x], y, x);
            t.m[y * t.LW + x] = v;
            anynan |= (v != v);
            vmax = fmaxf(vmax, v);
        }
        norm = block_max_nan(vmax, anynan, redf);
        for (int i = threadIdx.x; i < HW; i += SC_BLOCK) {
            const int y = i / W, x = i - y * W;
            const float v = t.m[y * t.LW + x] / norm;
            gm[i] = v;
            if (gl) { const float d = gl[i] - v; d2 += (double)(d * d); }
            n2 += (double)(v * v);
        }
        d2 = block_sum(d2, red);
        n2 = block_sum(n2, red);
    }
    if (threadIdx.x == 0 && (!(norm > 0.f) || isinf(norm))) stat |= SCARLET_STATUS_NONFINITE;
    if (threadIdx.x == 0) {
        float *gs = a.sed[wbuf] + (size_t)c * B;
        const float *gsl = a.in_iteration ? a.sed[c0] + (size_t)c * B : nullptr;
        double d2s = 0, n2s = 0;
        for (int b = 0; b < B; ++b) {
            float v = gs[b];
            if (v < 0.f) v = 0.f;
            v = v * norm;
            gs[b] = v;
            if (gsl) { const float d = gsl[b] - v; d2s += (double)(d * d); }
            n2s += (double)(v * v);
        }
        a.conv[4 * c + 0] = d2s; a.conv[4 * c + 1] = n2s;
        a.conv[4 * c + 2] = d2;  a.conv[4 * c + 3] = n2;
        if (stat) atomicOr(&a.status[s], stat);
    }
}

// ------------------------------------------------------------------------------------
// k_group_centers: the shared centre of every MultiComponentSource (source.py:613-630), one wave per scene.
// For each source g: _morph = sum over its layers of morph_c * sed_c.sum() (float32, layer by layer, as
// np.sum over the list does), max_pixel around the previous centre, and on every fifth iteration the
// PSF-weighted centroid.  _morph is materialised only on the patch both windows can touch (the centroid's radius
// + 2 around the previous centre, clipped to the frame: inside it every bounds decision equals the frame's).
// The new centre (and shift) is written to EVERY layer of the source; the update kernels take it as given.
__global__ __launch_bounds__(SC_WAVE) void k_group_centers(UpdateArgs a)
{
    extern __shared__ __align__(16) float lds[];
    const int s = blockIdx.x;
    if (!a.force_it0 && !a.active[s]) return;
    const int K = a.K, B = a.B, H = a.H, W = a.W, HW = H * W, lane = threadIdx.x;
    const int c0 = a.cur[s], wbuf = a.in_iteration ? 1 - c0 : c0;
    const int it = a.force_it0 ? 0 : a.it[s] + (a.in_iteration ? 1 : 0);
    const int R = a.centroid_P / 2 + 2, PW = 2 * R + 1 + 1;                      // patch radius, LDS row stride
    for (int k0 = 0; k0 < K; ++k0) {
        const int g = a.group[s * K + k0];
        if (g < 0 || (k0 > 0 && a.group[s * K + k0 - 1] == g)) continue;          // not a layer / not the first layer of its source
        int k1 = k0 + 1;
        while (k1 < K && a.group[s * K + k1] == g) ++k1;
        int cy = a.centers[2 * (s * K + k0)], cx = a.centers[2 * (s * K + k0) + 1];
        const int py0 = max(0, cy - R), px0 = max(0, cx - R);
        const int ph = min(H, cy + R + 1) - py0, pw = min(W, cx + R + 1) - px0;
        for (int i = lane; i < ph * pw; i += SC_WAVE) {
            const int y = i / pw, x = i - y * pw;
            float acc = 0.f;
            for (int k = k0; k < k1; ++k) {
                const float *sed = a.sed[wbuf] + (size_t)(s * K + k) * B;
                float ssum = 0.f;
                for (int b = 0; b < B; ++b) ssum += sed[b];
                acc += a.morph[wbuf][(size_t)(s * K + k) * HW + (py0 + y) * W + px0 + x] * ssum;
            }
            lds[y * PW + x] = acc;
        }
        wave_sync();
        Tile t; t.H = ph; t.W = pw; t.LW = PW; t.m = lds;
        int stat = 0, ly = cy - py0, lx = cx - px0;
        wave_max_pixel(t, ly, lx, stat);
        double dy = a.shifts[2 * (s * K + k0)], dx = a.shifts[2 * (s * K + k0) + 1];
        bool new_shift = false;
        if (a.symmetric && it % 5 == 0) {
            wave_centroid(t, a.centroid_psf, a.centroid_P, ly, lx, dy, dx, stat);
            new_shift = true;
        }
        cy = ly + py0; cx = lx + px0;
        if (lane == 0) {
            for (int k = k0; k < k1; ++k) {
                a.centers[2 * (s * K + k)] = cy; a.centers[2 * (s * K + k) + 1] = cx;
                if (new_shift) { a.shifts[2 * (s * K + k)] = dy; a.shifts[2 * (s * K + k) + 1] = dx; }
            }
            if (stat) atomicOr(&a.status[s], stat);
        }
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------
// k_converge: Blend._check_convergence (blend.py:141-184), one thread per scene.
// Also closes the iteration for the scene: it += 1 (len(mse)), cur flips.
__global__ void k_converge(int S, int K, const double *conv, int *flags, int *active,
                           int *it, int *cur, double e_rel2)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S || !active[s]) return;
    const int it_new = it[s] + 1;
    it[s] = it_new;
    cur[s] = 1 - cur[s];
    if (it_new > 1) {
        bool done = true;
        for (int k = 0; k < K; ++k) {
            const double *c = conv + 4 * ((size_t)s * K + k);
            int f = flags[s * K + k];
            if (c[0] <= e_rel2 * c[1]) f &= ~SCARLET_FLAG_SED_NOT_CONVERGED;
            else { f |= SCARLET_FLAG_SED_NOT_CONVERGED; done = false; }
            if (c[2] <= e_rel2 * c[3]) f &= ~SCARLET_FLAG_MORPH_NOT_CONVERGED;
            else { f |= SCARLET_FLAG_MORPH_NOT_CONVERGED; done = false; }
            flags[s * K + k] = f;
        }
        if (done) active[s] = 0;
    }
}

// ------------------------------------------------------------------------------------
// k_source_update_w: the same pipeline with one WAVE per component (wave_ops.h), four
// components per 256-thread workgroup, no workgroup barriers.  H, W <= 64.
__device__ inline void wave_pipeline(const UpdateArgs &a, int c, float *lds_wave)
{
    const int s = c / a.K;
    const int H = a.H, W = a.W, HW = H * W, B = a.B;
    const int lane = lane_id();
    Tile t; t.H = H; t.W = W; t.LW = tile_stride(W); t.m = lds_wave;
    float *vec = lds_wave + H * t.LW;
    const int c0 = a.cur[s];
    const int wbuf = a.in_iteration ? 1 - c0 : c0;
    float *gm = a.morph[wbuf] + (size_t)c * HW;
    const bool vec4 = (W & 3) == 0;                     // float4 groups, rows 8-byte aligned in LDS
    const int gpr = W >> 2, ngroups = HW >> 2;
    const int dyq = vec4 ? SC_WAVE / gpr : 0, dxq = vec4 ? SC_WAVE - dyq * gpr : 0;    // +64 groups per step
    const int y0 = vec4 ? lane / gpr : 0, x0 = vec4 ? lane - y0 * gpr : 0;
    if (vec4) {
        int y = y0, xq = x0;
        for (int g = lane; g < ngroups; g += SC_WAVE) {
            lds_store4(t.m + y * t.LW + (xq << 2), reinterpret_cast<const float4 *>(gm)[g]);
            y += dyq; xq += dxq;
            if (xq >= gpr) { xq -= gpr; ++y; }
        }
    } else
        for (int i = lane; i < HW; i += SC_WAVE) t.m[(i / W) * t.LW + (i % W)] = gm[i];
    wave_sync();
    const int it = a.force_it0 ? 0 : a.it[s] + (a.in_iteration ? 1 : 0);
    int cy = a.centers[2 * c], cx = a.centers[2 * c + 1];
    int stat = 0;
    const bool grouped = a.group && a.group[c] >= 0;               // layer of a MultiComponentSource: centre from k_group_centers
    if (!grouped) wave_max_pixel(t, cy, cx, stat);
    if (a.symmetric) {
        double dy = grouped ? (double)__builtin_nanf("") : a.shifts[2 * c], dx = grouped ? dy : a.shifts[2 * c + 1];
        if (!grouped && it % 5 == 0) {
            wave_centroid(t, a.centroid_psf, a.centroid_P, cy, cx, dy, dx, stat);
            if (lane == 0) { a.shifts[2 * c] = dy; a.shifts[2 * c + 1] = dx; }
        }
        const bool none = (dy != dy);
        wave_symmetry(t, cy, cx, none ? SCARLET_SYM_SOFT : SCARLET_SYM_KSPACE, 1.0f, dy, dx, false, 0.f, vec);
    }
    int lstop = 1 << 30;                // last sweep level computed; pixels beyond are <= 0 -> 0
    if (a.monotonic) wave_monotonic<float>(t, cy, cx, 0.f, &lstop);
    if (lane == 0) { a.centers[2 * c] = cy; a.centers[2 * c + 1] = cx; }
    const float step_morph = 1.0f / (float)a.lipschitz[2 * s + 1];
    const float *gl = a.in_iteration ? a.morph[c0] + (size_t)c * HW : nullptr;
    const float l0 = a.l0_thresh >= 0.f ? a.l0_thresh * step_morph : -1.f;
    const float l1 = a.l1_thresh >= 0.f ? a.l1_thresh * step_morph : -1.f;
    auto sparse_plus = [&](float v, int y, int x) {      // update.py:71-82, 27-32 + the sweep's cut
        if (l0 >= 0.f && fabsf(v) < l0) v = 0.f;
        if (l1 >= 0.f) {
            const float mag = fabsf(v) - l1;
            v = (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f)) * (mag < 0.f ? 0.f : mag);
        }
        if (v < 0.f || sweep_level(y, x, cy, cx) > lstop) v = 0.f;
        return v;
    };
    float norm;
    double d2 = 0, n2 = 0;
    if (vec4 && a.monotonic) {
        // One pass in float4 groups.  After the sweep no pixel exceeds the peak pixel (each is capped
        // by a convex combination of pixels closer to the peak) and the maps above are monotone, so
        // morph.max() is the processed peak value; a NaN elsewhere shows up in the sums (see below).
        norm = sparse_plus(t.m[cy * t.LW + cx], cy, cx);
        const bool regular = norm > 0.f && !isinf(norm);
        const float rnorm = 1.0f / norm;
        float d2f = 0.f, n2f = 0.f;
        int y = y0, xq = x0;
        for (int g = lane; g < ngroups; g += SC_WAVE) {
            const float4 v4 = lds_load4(t.m + y * t.LW + (xq << 2));
            float v[4] = {v4.x, v4.y, v4.z, v4.w}, o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = sparse_plus(v[e], y, (xq << 2) + e);
                if (regular) { const float q = v[e] * rnorm; o[e] = fmaf(fmaf(-q, norm, v[e]), rnorm, q); }   // v / norm, see fused2.h
                else o[e] = v[e] / norm;
            }
            reinterpret_cast<float4 *>(gm)[g] = make_float4(o[0], o[1], o[2], o[3]);
            if (gl) {
                const float4 l = reinterpret_cast<const float4 *>(gl)[g];
                const float e0 = l.x - o[0], e1 = l.y - o[1], e2 = l.z - o[2], e3 = l.w - o[3];
                d2f += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
            }
            n2f += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
            y += dyq; xq += dxq;
            if (xq >= gpr) { xq -= gpr; ++y; }
        }
        if (__any(n2f != n2f) && norm == norm) {
            // a NaN pixel away from the peak: np.max is NaN and the reference's morph becomes NaN everywhere
            norm = __builtin_nanf("");
            for (int g = lane; g < ngroups; g += SC_WAVE) reinterpret_cast<float4 *>(gm)[g] = make_float4(norm, norm, norm, norm);
            d2f = norm; n2f = norm;
        }
        d2 = (double)d2f; n2 = (double)n2f;
    } else {
        float vmax = -INFINITY;
        bool anynan = false;
        for (int i = lane; i < HW; i += SC_WAVE) {
            float *p = &t.m[(i / W) * t.LW + (i % W)];
            const float v = sparse_plus(*p, i / W, i % W);
            *p = v;
            anynan |= (v != v);
            vmax = fmaxf(vmax, v);
        }
        norm = wave_max(vmax);
        if (__any(anynan)) norm = __builtin_nanf("");
        for (int i = lane; i < HW; i += SC_WAVE) {
            const float v = t.m[(i / W) * t.LW + (i % W)] / norm;
            gm[i] = v;
            if (gl) { const float d = gl[i] - v; d2 += (double)(d * d); }
            n2 += (double)(v * v);
        }
    }
    if (!(norm > 0.f) || isinf(norm)) stat |= SCARLET_STATUS_NONFINITE;
    d2 = wave_sum(d2); n2 = wave_sum(n2);
    // SED: positive, * norm (update.py:27-32,62-65) and its convergence sums
    float *gs = a.sed[wbuf] + (size_t)c * B;
    const float *gsl = a.in_iteration ? a.sed[c0] + (size_t)c * B : nullptr;
    double d2s = 0, n2s = 0;
    if (lane < B) {
        float v = gs[lane];
        if (v < 0.f) v = 0.f;
        v = v * norm;
        gs[lane] = v;
        if (gsl) { const float d = gsl[lane] - v; d2s = (double)(d * d); }
        n2s = (double)(v * v);
    }
    d2s = wave_sum(d2s); n2s = wave_sum(n2s);
    if (lane == 0) {
        a.conv[4 * c + 0] = d2s; a.conv[4 * c + 1] = n2s;
        a.conv[4 * c + 2] = d2;  a.conv[4 * c + 3] = n2;
        if (stat) atomicOr(&a.status[s], stat);
    }
}

__global__ __launch_bounds__(SC_BLOCK) void k_source_update_w(UpdateArgs a)
{
    extern __shared__ __align__(16) float lds[];
    const int wid = threadIdx.x / SC_WAVE;
    const int c = blockIdx.x * SC_NWAVES + wid;
    if (c >= a.S * a.K) return;
    if (!a.force_it0 && !a.active[c / a.K]) return;
    const int per_wave = a.H * tile_stride(a.W) + SC_WAVE_VEC_FLOATS;
    wave_pipeline(a, c, lds + (size_t)wid * per_wave);
}
