// prox_ops.h -- per-component proximal operators on an LDS-resident morphology tile.
//
// One 256-thread workgroup owns one component.  The H x W float32 morphology lives in
// LDS (`m`, row stride LW) for the whole constraint pipeline, so every operator below is
// LDS/VALU/MFMA work with no HBM traffic; HBM sees one coalesced read and one write per
// component and iteration.  All functions are called by every thread of the block and
// end with the tile consistent (they synchronise internally).
//
// Reference rows: a8 max_pixel, a9 psf_weighted_centroid, a10/a11 radial
// monotonic sweep, a12 positivity, a13 normalisation, a14 L0/L1, a15-a17 symmetry.
#pragma once
#include "common.h"

template <typename T>
struct TileT {
    T *m;          // LDS, [H][LW]
    int H, W, LW;
};
typedef TileT<float> Tile;

__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }

// ------------------------------------------------------------------------------------
// a8  measurement.max_pixel (measurement.py:3-29).  First maximum, row-major, of the
// window rows [cy-2, cy+3) x cols [cx-2, cx+3) clipped at the HIGH edges only (numpy
// slice semantics); a negative start makes the reference fail -> status bit, centre kept.
// NaN: np.argmax returns the first NaN; reproduced.  Executed by thread 0; result
// broadcast through `out` (LDS int[2]).
__device__ inline void max_pixel_tile(const Tile &t, int cy, int cx, int *out, int *status_bits)
{
    if (threadIdx.x == 0) {
        int by = cy, bx = cx;
        if (cy >= t.H || cx >= t.W || cy < 0 || cx < 0) {
            // a centre outside the frame (never produced by the engine; a caller's bad input): flagged and
            // pulled onto the frame so that nothing downstream indexes outside the tile
            *status_bits |= SCARLET_STATUS_CENTER_AT_EDGE;
            by = min(max(cy, 0), t.H - 1); bx = min(max(cx, 0), t.W - 1);
        } else if (cy - 2 < 0 || cx - 2 < 0) {
            *status_bits |= SCARLET_STATUS_CENTER_AT_EDGE;
        } else {
            const int y1 = min(cy + 3, t.H), x1 = min(cx + 3, t.W);
            float best = 0.f;
            bool have = false, isnan_best = false;
            for (int y = cy - 2; y < y1; ++y)
                for (int x = cx - 2; x < x1; ++x) {
                    float v = t.m[y * t.LW + x];
                    if (!have) { best = v; by = y; bx = x; have = true; isnan_best = (v != v); }
                    else if (!isnan_best && (v > best || v != v)) {
                        best = v; by = y; bx = x; isnan_best = (v != v);
                    }
                }
        }
        out[0] = by; out[1] = bx;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------
// a9  measurement.psf_weighted_centroid (measurement.py:32-94).  psf: global float64
// [P][P].  Window = +-min(cy, H-1-cy, P/2) rows, same for columns.  First moments of
// morph*psf in window index units, float64 accumulation; new centre = rint(moment)
// (half-to-even, like np.round) + window origin, shift = rint(moment) - moment.
// `red` = SC_NWAVES doubles of LDS, `out` LDS int[2], `shift_out` LDS double[2].
__device__ inline void centroid_tile(const Tile &t, const double *__restrict__ psf, int P,
                                     int cy, int cx, double *red, int *out, double *shift_out,
                                     int *status_bits)
{
    const int rad = P / 2;
    const int ry = min(min(cy, t.H - 1 - cy), rad);
    const int rx = min(min(cx, t.W - 1 - cx), rad);
    const int hh = 2 * ry + 1, ww = 2 * rx + 1;
    double s0 = 0, sy = 0, sx = 0;
    for (int i = threadIdx.x; i < hh * ww; i += SC_BLOCK) {
        const int iy = i / ww, ix = i - iy * ww;
        const double w = (double)t.m[(cy - ry + iy) * t.LW + (cx - rx + ix)] *
                         psf[(rad - ry + iy) * P + (rad - rx + ix)];
        s0 += w; sy += iy * w; sx += ix * w;
    }
    s0 = block_sum(s0, red);
    sy = block_sum(sy, red);
    sx = block_sum(sx, red);
    if (threadIdx.x == 0) {
        const double my = sy / s0, mx = sx / s0;
        if (!(my == my) || !(mx == mx) || isinf(my) || isinf(mx)) {
            *status_bits |= SCARLET_STATUS_NONFINITE;
            out[0] = cy; out[1] = cx; shift_out[0] = 0; shift_out[1] = 0;
        } else {
            const double wy = rint(my), wx = rint(mx);
            out[0] = (int)wy + (cy - ry);
            out[1] = (int)wx + (cx - rx);
            shift_out[0] = wy - my;
            shift_out[1] = wx - mx;
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------
// a10/a11  radial monotonicity.  The reference sorts pixels by distance from the peak and
// sweeps them sequentially (operators_pybind11.cc:27-50) with an 8 x N table of
// normalised cos-weights (operator.py:540-621).  Here:
//  * weights are generated on the fly: for pixel offset (X, Y) from the peak and neighbour
//    offset (ox, oy), the neighbour is used iff it is in bounds and strictly closer, i.e.
//    2(X ox + Y oy) + |o|^2 < 0, with weight proportional to cos = -(X ox + Y oy)/(r |o|);
//    the 1/r factor cancels in the normalisation -- integer arithmetic plus one 1/sqrt(2);
//  * the order is replaced by a closed-form level function
//        level(X, Y) = 2 max(|X|,|Y|) + min(|X|,|Y|)
//    every strictly-closer 8-neighbour of a pixel has a strictly smaller level, so all
//    pixels of one level are independent: a level-synchronous sweep gives bit-identical
//    values to ANY sequential order the reference may use (tests/test_schedule.py).
//  * accumulation order i = 0..7 and separate multiply / add (no FMA) as in the C++ loop.
// neighbour i = 0..7 -> (oy, ox) in the reference's order (operator.py:117):
// (-1,-1) (-1,0) (-1,1) (0,-1) (0,1) (1,-1) (1,0) (1,1); folds when i is a constant
__device__ __forceinline__ int sc_oy(int i) { return i < 3 ? -1 : (i < 5 ? 0 : 1); }
__device__ __forceinline__ int sc_ox(int i) { return i < 3 ? i - 1 : (i == 3 ? -1 : (i == 4 ? 1 : i - 6)); }

// level of pixel (y, x) for a peak at (cy, cx): the sweep's topological level
__device__ __forceinline__ int sweep_level(int y, int x, int cy, int cx)
{
    const int ay = y < cy ? cy - y : y - cy, ax = x < cx ? cx - x : x - cx;
    return ay > ax ? 2 * ay + ax : 2 * ax + ay;
}

// `lastpos` (one int of LDS, or NULL): early exit as in wave_monotonic -- only for callers that
// zero everything <= `floor` afterwards (positivity: floor = 0; the detection cut of the source
// initialisation: floor = bg_cutoff >= 0), with 0 <= thresh <= 1.  Every thread that leaves a value
// above the floor at level l stores l there (all writers of a level store the same value); once three
// consecutive levels stored nothing, all later pixels end <= floor (a cap is a convex combination of
// closer pixels times 1 - thresh) and the sweep stops.  Returns the last level swept (1 << 30: all
// of them); the caller zeroes the pixels beyond it.
template <bool NEAREST, typename T>
__device__ inline int monotonic_tile(const TileT<T> &t, int cy, int cx, T thresh, int *lastpos = nullptr, T floor = (T)0,
                                     int ell0 = 1, int lastpos0 = 0)
{
    // (ell0, lastpos0): continue a sweep whose levels < ell0 are done, the last one with a value above the
    // floor being lastpos0
    if (lastpos) {
        if (threadIdx.x == 0) *lastpos = lastpos0;
        __syncthreads();
    }
    const int H = t.H, W = t.W, LW = t.LW;
    T *m = t.m;
    const int mxr = max(cx, W - 1 - cx), myr = max(cy, H - 1 - cy);
    const int Lmax = 2 * max(mxr, myr) + min(mxr, myr);
    const T one_minus = (T)1 - thresh;
    const int oct = threadIdx.x & 7;
    const bool sw = oct & 1, fx = oct & 2, fy = oct & 4;
    for (int ell = ell0; ell <= Lmax; ++ell) {
        const int a0 = (ell + 2) / 3, a1 = ell >> 1;
        for (int a = a0 + (threadIdx.x >> 3); a <= a1; a += (SC_BLOCK >> 3)) {
            const int b = ell - 2 * a;                 // 0 <= b <= a
            // the 8 images of (a, b); drop the duplicates on the axes and diagonals
            if (sw && a == b) continue;
            if (b == 0 && (sw ? fx : fy)) continue;
            const int u = sw ? b : a, v = sw ? a : b;
            const int X = fx ? -u : u, Y = fy ? -v : v;
            const int px = cx + X, py = cy + Y;
            if (px < 0 || px >= W || py < 0 || py >= H) continue;
            T c[8];
            T csum = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int oy = sc_oy(i), ox = sc_ox(i);
                const int qy = py + oy, qx = px + ox;
                const int dot = X * ox + Y * oy;
                const int n2 = ox * ox + oy * oy;
                const bool ok = qy >= 0 && qy < H && qx >= 0 && qx < W && (2 * dot + n2 < 0);
                c[i] = ok ? (T)(-dot) * (n2 == 2 ? (T)0.70710678118654752440 : (T)1) : (T)0;
                csum += c[i];
            }
            T cap;
            if (NEAREST) {
                // operator.py:591-600: the single neighbour best aligned with the peak
                int best = 0; T cb = c[0];
#pragma unroll
                for (int i = 1; i < 8; ++i) if (c[i] > cb) { cb = c[i]; best = i; }
                cap = m[(py + sc_oy(best)) * LW + px + sc_ox(best)] * one_minus;
            } else {
                const T inv = csum > 0 ? (T)1 / csum : (T)1;
                T ref = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (c[i] > 0)
                        ref = add_rn(ref, mul_rn(m[(py + sc_oy(i)) * LW + px + sc_ox(i)],
                                                 c[i] * inv));
                cap = ref * one_minus;
            }
            const T cur = m[py * LW + px];
            if (cap < cur) m[py * LW + px] = cap;
            if (lastpos && (cap < cur ? cap : cur) > floor) *lastpos = ell;
        }
        __syncthreads();
        // (a thread that already runs level ell + 1 may have stored ell + 1: the decision is the same)
        if (lastpos && ell - *lastpos >= 3) return ell;
    }
    return 1 << 30;
}

// ------------------------------------------------------------------------------------
// a15  window selection of operator.uncentered_operator (operator.py:197-219).
// Returns false when `center` is the array middle (H/2, W/2): the reference then applies
// the operator to the WHOLE array and discards a returned copy (Appendix A.1).
struct SymWindow { int y0, x0, h, w; bool centered; };
__device__ inline SymWindow sym_window(int H, int W, int cy, int cx)
{
    SymWindow s;
    s.centered = (cy == H / 2 && cx == W / 2);
    if (s.centered) { s.y0 = 0; s.x0 = 0; s.h = H; s.w = W; return s; }
    const int ry = min(cy, H - 1 - cy), rx = min(cx, W - 1 - cx);
    s.y0 = cy - ry; s.h = 2 * ry + 1;
    s.x0 = cx - rx; s.w = 2 * rx + 1;
    return s;
}

// a16  prox_soft_symmetry / prox_sdss_symmetry on the window (operator.py:231-251).
template <typename T>
__device__ inline void flip_symmetry_tile(const TileT<T> &t, const SymWindow &s, bool sdss,
                                          T strength)
{
    T *m = t.m;
    const int n = s.h * s.w;
    const T a = (T)(0.5 * (double)strength), bq = (T)1 - strength;
    // every pixel pairs with its point reflection; a pair is handled by the thread owning
    // the lower flat index, so reads and writes of a pair stay in one thread (no barrier)
    for (int i = threadIdx.x; i < n; i += SC_BLOCK) {
        const int j = n - 1 - i;
        if (j < i) break;
        const int iy = i / s.w, ix = i - iy * s.w;
        const int jy = j / s.w, jx = j - jy * s.w;
        T *pi = &m[(s.y0 + iy) * t.LW + s.x0 + ix];
        T *pj = &m[(s.y0 + jy) * t.LW + s.x0 + jx];
        const T xi = *pi, xj = *pj;
        if (sdss) {
            const T r = xj < xi ? xj : xi;          // np.min of the pair (NaN-agnostic here)
            *pi = r; *pj = r;
        } else {
            const T si = xi + xj, sj = xj + xi;
            *pi = a * si + bq * xi;
            *pj = a * sj + bq * xj;
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------
// a17  prox_kspace_symmetry (operator.py:253-288) as a real-space operator.
//
// The reference pads the (odd) h x w window to F = next_fast_len(2h+10) x (2w+10, even),
// rfftn's it, multiplies by exp(+2 pi i (f_y dy + f_x dx)), keeps the real part,
// multiplies by the conjugate phase, irfftn's, crops, and zeroes pixels where X <= 0.
// Re(.) of the half spectrum averages the window with its point reflection, and the phase
// pair turns the reflection into a fractional translation by 2(dy, dx).  Written out with
// numpy's rfftn/irfftn conventions (C2R drops Im of the DC/Nyquist bins; fftfreq uses
// f=-1/2 on even axes) this is EXACTLY
//     out = 1/2 X + 1/2 [ A X B  +  s (sigma sigma^T X) C ]          (then the X<=0 mask)
// with Hankel matrices over window-centred indices i', j' in [-r, r], t = i'+j' - 2 d:
//     A[i',k'] = sin(pi t) cot(pi t/F)/F        (F even)   or   sin(pi t)/(F sin(pi t/F))  (F odd)
//     B        = the same with (Fx, dx)  (Fx is always even)
//     C[j',k'] = -(1 - cos(pi t)) cot(pi t/Fx)/Fx,
//     sigma_i' = (-1)^i',   s = sin(2 pi dy)/Fy  for even Fy, 0 for odd Fy.
// (derivation + numpy check in DESIGN.md; parity vs the FFT form is ~1e-15 in float64).
// The two products are dense (h x w x w, h x h x w): genuine GEMMs, run on the MFMA
// pipe with v_mfma_f32_16x16x4_f32 (exact f32 FMA chain), operands read from LDS.
//
// LDS use: `scr` [hp][LS] floats (T = X B), vectors av[2hp], bv[2wp], cv[2wp], zv[wp].
__device__ inline void kspace_vectors(float *av, float *bv, float *cv, int hp, int wp,
                                      int ry, int rx, int h, int w, int Fy, int Fx,
                                      double dy, double dx)
{
    const double s2y = sinpi(2.0 * dy), s2x = sinpi(2.0 * dx), c2x = cospi(2.0 * dx);
    for (int q = threadIdx.x; q < 2 * hp; q += SC_BLOCK) {
        float val = 0.f;
        if (q <= 2 * (h - 1)) {
            const int n = q - 2 * ry;
            const double tt = (double)n - 2.0 * dy;
            const double sn = sinpi(tt / Fy), cs = cospi(tt / Fy);
            const double spt = (n & 1) ? s2y : -s2y;           // sin(pi t) = -(-1)^n sin(2 pi d)
            double a;
            if (sn == 0.0) a = 1.0;
            else if (Fy & 1) a = spt / (Fy * sn);
            else a = spt * cs / (Fy * sn);
            val = (float)a;
        }
        av[q] = val;
    }
    for (int q = threadIdx.x; q < 2 * wp; q += SC_BLOCK) {
        float vb = 0.f, vc = 0.f;
        if (q <= 2 * (w - 1)) {
            const int n = q - 2 * rx;
            const double tt = (double)n - 2.0 * dx;
            const double sn = sinpi(tt / Fx), cs = cospi(tt / Fx);
            const double spt = (n & 1) ? s2x : -s2x;
            const double cpt = (n & 1) ? -c2x : c2x;           // cos(pi t) = (-1)^n cos(2 pi d)
            if (sn == 0.0) { vb = 1.f; vc = 0.f; }
            else {
                vb = (float)(spt * cs / (Fx * sn));
                vc = (float)(-(1.0 - cpt) * cs / (Fx * sn));
            }
        }
        bv[q] = vb; cv[q] = vc;
    }
}

// `stage` (LDS, stage_floats(hp, wp) floats, or NULL): used when the tile and/or the scratch live in
// HBM.  GEMM 1 then reads each 16-row band of X into LDS once (stage_x) instead of once per column
// tile, GEMM 2 each 16-column band of T once instead of once per row tile.
__host__ __device__ inline int stage_floats(int hp, int wp) { return max(16 * tile_stride(wp), hp * 16); }
__device__ inline void kspace_symmetry_tile(const Tile &t, const SymWindow &s, double dy,
                                            double dx, float *scr, float *av, float *bv,
                                            float *cv, float *zv, float *stage = nullptr, bool stage_x = false)
{
    float *m = t.m;
    const int LW = t.LW;
    const int h = s.h, w = s.w, ry = h / 2, rx = w / 2;
    const int hp = round16(h), wp = round16(w), LS = scratch_stride(wp);
    const int Fy = dev_next_fast_len(2 * h + 10);
    int Fx = dev_next_fast_len(2 * w + 10);
    while (Fx & 1) Fx = dev_next_fast_len(Fx + 1);
    kspace_vectors(av, bv, cv, hp, wp, ry, rx, h, w, Fy, Fx, dy, dx);
    const float sy = (Fy & 1) ? 0.f : (float)(sinpi(2.0 * dy) / Fy);

    // rank-1 term, part 1: v[j] = sum_i (-1)^(i-ry) X[i][j]   (stored in zv, then z = C v)
    const bool need_rank1 = (sy != 0.f);
    float vloc = 0.f;
    if (need_rank1 && threadIdx.x < w) {
        for (int i = 0; i < h; ++i) {
            const float x = m[(s.y0 + i) * LW + s.x0 + threadIdx.x];
            vloc += ((i - ry) & 1) ? -x : x;
        }
    }
    __syncthreads();                       // vectors ready
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int tiles_x = wp >> 4, tiles = (hp >> 4) * tiles_x;

    // GEMM 1: T = Xw (hp x wp, zero outside the window) . Hankel(bv)
    if (stage && stage_x) {
        const int SW = tile_stride(wp);
        for (int i0 = 0; i0 < hp; i0 += 16) {
            for (int e = threadIdx.x; e < 16 * wp; e += SC_BLOCK) {
                const int r = e / wp, c = e - r * wp;
                stage[r * SW + c] = (i0 + r < h && c < w) ? m[(s.y0 + i0 + r) * LW + s.x0 + c] : 0.f;
            }
            __syncthreads();
            for (int tx = wid; tx < tiles_x; tx += SC_NWAVES) {
                const int j0 = tx << 4;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int k0 = 0; k0 < wp; k0 += 4) {
                    const int k = k0 + lq;
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(stage[lr * SW + k], bv[k + j0 + lr], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) scr[(i0 + lq * 4 + r) * LS + j0 + lr] = acc[r];
            }
            __syncthreads();
        }
    } else
    for (int tile = wid; tile < tiles; tile += SC_NWAVES) {
        const int i0 = (tile / tiles_x) << 4, j0 = (tile % tiles_x) << 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int ai = i0 + lr;
        const float *arow = &m[(s.y0 + ai) * LW + s.x0];
        const bool arow_ok = ai < h;
        for (int k0 = 0; k0 < wp; k0 += 4) {
            const int k = k0 + lq;
            const float a = (arow_ok && k < w) ? arow[k] : 0.f;
            const float b = bv[k + j0 + lr];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) scr[(i0 + lq * 4 + r) * LS + j0 + lr] = acc[r];
    }
    if (need_rank1) {
        if (threadIdx.x < wp) zv[threadIdx.x] = threadIdx.x < w ? vloc : 0.f;
    }
    __syncthreads();                       // T and v complete
    float zloc = 0.f;
    if (need_rank1 && threadIdx.x < w) {
        for (int j2 = 0; j2 < w; ++j2) zloc += cv[threadIdx.x + j2] * zv[j2];
    }
    __syncthreads();
    if (need_rank1 && threadIdx.x < w) zv[threadIdx.x] = zloc;
    __syncthreads();

    // GEMM 2: Y = Hankel(av) . T ; epilogue combines with X in place
    const int n_outer = stage ? tiles_x : 1;
    for (int outer = 0; outer < n_outer; ++outer) {
    if (stage) {
        for (int e = threadIdx.x; e < hp * 16; e += SC_BLOCK)
            stage[e] = scr[(e >> 4) * LS + (outer << 4) + (e & 15)];
        __syncthreads();
    }
    for (int tile = stage ? wid * tiles_x + outer : wid; tile < tiles; tile += (stage ? SC_NWAVES * tiles_x : SC_NWAVES)) {
        const int i0 = (tile / tiles_x) << 4, j0 = (tile % tiles_x) << 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < hp; k0 += 4) {
            const int k = k0 + lq;
            const float a = av[i0 + lr + k];
            const float b = stage ? stage[(k << 4) + lr] : scr[k * LS + j0 + lr];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
        const int j = j0 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + lq * 4 + r;
            if (i < h && j < w) {
                float *p = &m[(s.y0 + i) * LW + s.x0 + j];
                const float x = *p;
                float y2 = acc[r];
                if (need_rank1) y2 += (((i - ry) & 1) ? -sy : sy) * zv[j];
                *p = (x <= 0.f) ? 0.f : 0.5f * x + 0.5f * y2;
            }
        }
    }
    if (stage) __syncthreads();            // before the next column band overwrites the stage
    }
    __syncthreads();
}

// a15  operator.prox_uncentered_symmetry dispatch on an LDS tile.
__device__ inline void symmetry_tile(const Tile &t, int cy, int cx, int algorithm,
                                     float strength, double dy, double dx, bool use_fill,
                                     float fill, float *scr, float *av, float *bv, float *cv,
                                     float *zv, float *stage = nullptr, bool stage_x = false)
{
    SymWindow s = sym_window(t.H, t.W, cy, cx);
    if (algorithm & SCARLET_SYM_FULL_WINDOW) {   // bare operator on the whole array (operator.py:231-288)
        algorithm &= ~SCARLET_SYM_FULL_WINDOW;
        s.y0 = 0; s.x0 = 0; s.h = t.H; s.w = t.W; s.centered = false;
    }
    if (algorithm == SCARLET_SYM_KSPACE) {
        if (s.centered) return;          // Appendix A.1: result discarded by the reference
        kspace_symmetry_tile(t, s, dy, dx, scr, av, bv, cv, zv, stage, stage_x);
    } else {
        flip_symmetry_tile(t, s, algorithm == SCARLET_SYM_SDSS, strength);
    }
    if (use_fill && !s.centered) {
        for (int i = threadIdx.x; i < t.H * t.W; i += SC_BLOCK) {
            const int y = i / t.W, x = i - y * t.W;
            if (y < s.y0 || y >= s.y0 + s.h || x < s.x0 || x >= s.x0 + s.w)
                t.m[y * t.LW + x] = fill;
        }
        __syncthreads();
    }
}

// LDS floats needed by symmetry_tile for an H x W tile (worst-case window)
__host__ __device__ inline int symmetry_lds_floats(int H, int W)
{
    const int hp = round16(H), wp = round16(W);
    return hp * scratch_stride(wp) + 2 * hp + 2 * wp + 2 * wp + wp;
}
