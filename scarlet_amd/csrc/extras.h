// extras.h -- operators off the default pipeline: the log-histogram noise
// cut of measurement.threshold / update.threshold, bbox.trim, and update.translation's Lanczos
// resampling.  One 256-thread workgroup per array; arrays are small (a morphology plane).
#pragma once
#include "common.h"

#define SC_HIST_MAX_BINS 50

__device__ __forceinline__ double block_min_d(double v, double *red)
{
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    const int lane = threadIdx.x & (SC_WAVE - 1), wid = threadIdx.x / SC_WAVE;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    double r = red[0];
#pragma unroll
    for (int w = 1; w < SC_NWAVES; ++w) r = fmin(r, red[w]);
    return r;
}

// measurement.py:97-105: number of positive pixels and the range of their log10.
// out [n][3] = {count, min log10, max log10} (float64)
__global__ __launch_bounds__(SC_BLOCK) void k_log_range(const float *x, int64_t count, double *out)
{
    __shared__ double red[SC_NWAVES];
    const float *p = x + (size_t)blockIdx.x * count;
    double cnt = 0, lo = INFINITY, hi = -INFINITY;
    for (int64_t i = threadIdx.x; i < count; i += SC_BLOCK) {
        const float v = p[i];
        if (v > 0.f) { const double l = log10((double)v); cnt += 1; lo = fmin(lo, l); hi = fmax(hi, l); }
    }
    cnt = block_sum(cnt, red);
    lo = block_min_d(lo, red);
    hi = -block_min_d(-hi, red);
    if (threadIdx.x == 0) { out[blockIdx.x * 3 + 0] = cnt; out[blockIdx.x * 3 + 1] = lo; out[blockIdx.x * 3 + 2] = hi; }
}

// np.histogram with equal bins (numpy/lib/_histograms_impl.py, "fast algorithm for equal bins"):
// index = trunc((v - first) / (last - first) * nbins), the right edge goes to the last bin, then
// one correction step against the tabulated edges in each direction.
// edges [n][SC_HIST_MAX_BINS + 1], nbins [n], hist [n][SC_HIST_MAX_BINS] (zeroed here)
__global__ __launch_bounds__(SC_BLOCK) void k_log_hist(const float *x, int64_t count, const double *edges,
                                                         const int32_t *nbins, int32_t *hist)
{
    __shared__ int h[SC_HIST_MAX_BINS];
    __shared__ double e[SC_HIST_MAX_BINS + 1];
    const int nb = nbins[blockIdx.x];
    const float *p = x + (size_t)blockIdx.x * count;
    for (int i = threadIdx.x; i < SC_HIST_MAX_BINS; i += SC_BLOCK) h[i] = 0;
    for (int i = threadIdx.x; i <= nb; i += SC_BLOCK) e[i] = edges[(size_t)blockIdx.x * (SC_HIST_MAX_BINS + 1) + i];
    __syncthreads();
    const double first = e[0], last = e[nb], denom = last - first;
    for (int64_t i = threadIdx.x; i < count; i += SC_BLOCK) {
        const float v = p[i];
        if (!(v > 0.f)) continue;
        const double l = log10((double)v);
        if (!(l >= first && l <= last)) continue;
        int idx = (int)(((l - first) / denom) * (double)nb);
        if (idx == nb) idx -= 1;
        if (l < e[idx]) idx -= 1;
        if (l >= e[idx + 1] && idx != nb - 1) idx += 1;
        atomicAdd(&h[idx], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SC_HIST_MAX_BINS; i += SC_BLOCK)
        hist[(size_t)blockIdx.x * SC_HIST_MAX_BINS + i] = h[i];
}

// update.py:98: morph[morph < thresh] = 0, compared in float64 like numpy does for a float64 scalar
__global__ void k_cut_below(float *x, int64_t count, double thresh)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x)
        if ((double)x[i] < thresh) x[i] = 0.f;
}

// bbox.trim (bbox.py:174-193): tight bounds of X > min_value.  box [n][4] = {bottom, top, left, right};
// an array without such a pixel gives {H, -1, W, -1} (the reference raises on the empty min()).
__global__ __launch_bounds__(SC_BLOCK) void k_trim(const float *x, int H, int W, float min_value, int32_t *box)
{
    __shared__ int b[4];
    if (threadIdx.x == 0) { b[0] = H; b[1] = -1; b[2] = W; b[3] = -1; }
    __syncthreads();
    const float *p = x + (size_t)blockIdx.x * H * W;
    int y0 = H, y1 = -1, x0 = W, x1 = -1;
    for (int i = threadIdx.x; i < H * W; i += SC_BLOCK)
        if (p[i] > min_value) {
            const int y = i / W, xx = i - y * W;
            y0 = min(y0, y); y1 = max(y1, y); x0 = min(x0, xx); x1 = max(x1, xx);
        }
    atomicMin(&b[0], y0); atomicMax(&b[1], y1); atomicMin(&b[2], x0); atomicMax(&b[3], x1);
    __syncthreads();
    if (threadIdx.x < 4) box[blockIdx.x * 4 + threadIdx.x] = b[threadIdx.x];
}

// interpolation.fft_resample (interpolation.py:408-448) as the linear convolution it equals: the
// reference pads by kernel size + 3 before its circular FFT product, more than the taps reach, so
//   out[y][x] = sum_i sum_j ky[i] kx[j] in[y - (y0 + i)][x - (x0 + j)]   (zero outside the image)
// with y0/x0 the first tap positions (window[0]).  Accumulated in float64.
// taps [n][2][SC_TAPS_MAX] float64 (ky then kx; Lanczos-5 has ten), win0 [n][2] int32
#define SC_TAPS_MAX 12
__global__ __launch_bounds__(SC_BLOCK) void k_resample(const float *in, float *out, int H, int W,
                                                         const double *taps, const int32_t *win0, int ny, int nx)
{
    __shared__ double ky[SC_TAPS_MAX], kx[SC_TAPS_MAX];
    const int a = blockIdx.y;
    if (threadIdx.x < ny) ky[threadIdx.x] = taps[((size_t)a * 2 + 0) * SC_TAPS_MAX + threadIdx.x];
    if (threadIdx.x < nx) kx[threadIdx.x] = taps[((size_t)a * 2 + 1) * SC_TAPS_MAX + threadIdx.x];
    __syncthreads();
    const int y0 = win0[a * 2 + 0], x0 = win0[a * 2 + 1];
    const float *p = in + (size_t)a * H * W;
    float *q = out + (size_t)a * H * W;
    for (int i = blockIdx.x * SC_BLOCK + threadIdx.x; i < H * W; i += gridDim.x * SC_BLOCK) {
        const int y = i / W, x = i - y * W;
        double acc = 0;
        for (int r = 0; r < ny; ++r) {
            const int sy = y - (y0 + r);
            if (sy < 0 || sy >= H) continue;
            double row = 0;
            for (int c = 0; c < nx; ++c) {
                const int sx = x - (x0 + c);
                if (sx >= 0 && sx < W) row += kx[c] * (double)p[sy * W + sx];
            }
            acc += ky[r] * row;
        }
        q[i] = (float)acc;
    }
}

// ------------------------------------------------------------------------------------------------
// Several observations per blend (blend.py:24-43, 120-139, 219-220) on the device.  The factors live in a
// STATE batch over the model frame's C channels; observation i is a batch over the channels band0[i] ..
// band0[i] + B_i - 1 with its own images / weights / PSF kernel, used for gradients only.
struct MultiObs {
    float *sed[2], *morph[2];       // the observation batch's buffers: [0] receives the factors, [1] returns the gradients
    int *cur, *it, *active;
    const double *mse;              // [S][capacity]: slot 0 = the loss of this observation
    int mse_capacity;
    int B, band0;
};
#define SC_MULTI_MAX 8
struct MultiArgs {
    int S, K, C, HW, n_obs;
    float *sed[2], *morph[2];       // state
    const int *cur, *active;
    int *it;
    double *lipschitz, *mse;
    int mse_capacity;
    const uint8_t *fix_sed, *fix_morph;
    int approximate_L;
    MultiObs obs[SC_MULTI_MAX];
};
// factors of the current buffer -> every observation's buffer 0 (its band slice of the SEDs, the morphologies)
__global__ __launch_bounds__(SC_BLOCK) void k_multi_scatter(MultiArgs a)
{
    const int c = blockIdx.y, s = c / a.K;                  // component
    if (!a.active[s]) return;
    const int c0 = a.cur[s];
    const float *m = a.morph[c0] + (size_t)c * a.HW;
    for (int o = 0; o < a.n_obs; ++o) {
        float *dst = a.obs[o].morph[0] + (size_t)c * a.HW;
        for (int i = blockIdx.x * SC_BLOCK + threadIdx.x; i < a.HW; i += gridDim.x * SC_BLOCK) dst[i] = m[i];
        if (blockIdx.x == 0) {
            if (threadIdx.x < a.obs[o].B)
                a.obs[o].sed[0][(size_t)c * a.obs[o].B + threadIdx.x] = a.sed[c0][(size_t)c * a.C + a.obs[o].band0 + threadIdx.x];
            if (threadIdx.x == 0 && c == s * a.K) { a.obs[o].cur[s] = 0; a.obs[o].it[s] = 0; a.obs[o].active[s] = 1; }
        }
    }
}
// loss = sum of the observations' losses; Lipschitz constants * n_obs (blend.py:219-220; approximate form:
// blend.py:189-201 on the summed loss); one thread per scene
__global__ void k_multi_loss(MultiArgs a, const double *approx_sums)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.S || !a.active[s]) return;
    double loss = 0;
    for (int o = 0; o < a.n_obs; ++o) loss += a.obs[o].mse[(size_t)s * a.obs[o].mse_capacity];
    const int it_new = a.it[s] + 1;
    if (it_new <= a.mse_capacity) a.mse[(size_t)s * a.mse_capacity + it_new - 1] = loss;
    double Ls = a.lipschitz[2 * s], Lm = a.lipschitz[2 * s + 1];
    if (a.approximate_L) {
        Ls = approx_sums[2 * s]; Lm = approx_sums[2 * s + 1];       // sum |morph|^2, sum |sed|^2
        if (it_new > 1 && loss > a.mse[(size_t)s * a.mse_capacity + it_new - 2]) { Ls *= 2; Lm *= 2; }
    }
    a.lipschitz[2 * s] = Ls * a.n_obs;
    a.lipschitz[2 * s + 1] = Lm * a.n_obs;
}
// sum |morph|^2 and sum |sed|^2 of the current factors of a scene (approximate_L), one workgroup per scene
__global__ __launch_bounds__(SC_BLOCK) void k_multi_approx(MultiArgs a, double *out)
{
    __shared__ double red[SC_NWAVES];
    const int s = blockIdx.x;
    if (!a.active[s]) return;
    const int c0 = a.cur[s];
    const float *m = a.morph[c0] + (size_t)s * a.K * a.HW, *sd = a.sed[c0] + (size_t)s * a.K * a.C;
    double am = 0, as = 0;
    for (int i = threadIdx.x; i < a.K * a.HW; i += SC_BLOCK) am += (double)m[i] * m[i];
    for (int i = threadIdx.x; i < a.K * a.C; i += SC_BLOCK) as += (double)sd[i] * sd[i];
    am = block_sum(am, red); as = block_sum(as, red);
    if (threadIdx.x == 0) { out[2 * s] = am; out[2 * s + 1] = as; }
}
// gradients summed over the observations, step x - g / L into buffer 1 - cur (blend.py:87-96)
__global__ __launch_bounds__(SC_BLOCK) void k_multi_step(MultiArgs a)
{
    const int c = blockIdx.y, s = c / a.K;
    if (!a.active[s]) return;
    const int c0 = a.cur[s];
    const float step_sed = 1.0f / (float)a.lipschitz[2 * s], step_morph = 1.0f / (float)a.lipschitz[2 * s + 1];
    const bool fixm = a.fix_morph && a.fix_morph[c], fixs = a.fix_sed && a.fix_sed[c];
    const float *m = a.morph[c0] + (size_t)c * a.HW;
    float *mo = a.morph[1 - c0] + (size_t)c * a.HW;
    for (int i = blockIdx.x * SC_BLOCK + threadIdx.x; i < a.HW; i += gridDim.x * SC_BLOCK) {
        float g = 0.f;
        for (int o = 0; o < a.n_obs; ++o) g += a.obs[o].morph[1][(size_t)c * a.HW + i];
        mo[i] = fixm ? m[i] : m[i] - step_morph * g;
    }
    if (blockIdx.x == 0 && threadIdx.x < a.C) {
        const int ch = threadIdx.x;
        float g = 0.f;
        for (int o = 0; o < a.n_obs; ++o) {
            const int b = ch - a.obs[o].band0;
            if (b >= 0 && b < a.obs[o].B) g += a.obs[o].sed[1][(size_t)c * a.obs[o].B + b];
        }
        const float x = a.sed[c0][(size_t)c * a.C + ch];
        a.sed[1 - c0][(size_t)c * a.C + ch] = fixs ? x : x - step_sed * g;
    }
}
