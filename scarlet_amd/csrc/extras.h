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
