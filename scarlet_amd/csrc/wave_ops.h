// wave_ops.h -- the constraint pipeline with ONE 64-lane wavefront per component.
//
// Why wave-level: the radial sweep is a chain of up to ~190 dependent levels (~40 with the early exit).  With a
// 256-thread workgroup per component (prox_ops.h) three of the four waves idle through the
// small levels and every level pays an s_barrier; with one wave per component there are
// no barriers at all (LDS operations of one wave complete in order), four components
// progress concurrently on the four SIMDs of a CU, and other workgroups hide the chain
// latency.  Requires H, W <= 64 (the k-space GEMM keeps T = X.B in 64 accumulator VGPRs);
// larger images use the workgroup-level code in prox_ops.h.
//
// Same reference rows as prox_ops.h (a8-a17); see the comments there for the maths.
#pragma once
#include "common.h"
#include "prox_ops.h"

// order LDS traffic of this wave (compiler + lgkmcnt), no s_barrier
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (SC_WAVE - 1); }

// ---------------------------------------------------------------- a8 max_pixel
// np.argmax order: larger value wins, NaN beats everything, ties -> smaller index.  Each of
// the 25 window lanes builds one 64-bit key whose unsigned order is exactly that order
// (monotone map of the float bits in the high word, 31 - rank in the low word); the maximum
// over lanes 0..31 takes four DPP steps inside the 16-lane rows and two v_readlane.
__device__ __forceinline__ unsigned argmax_key_hi(float v)
{
    unsigned u = __float_as_uint(v);
    if (u == 0x80000000u) u = 0;                                  // -0 == +0 for argmax
    return v != v ? 0xFFFFFFFFu : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));
}
template <int CTRL>
__device__ __forceinline__ void key_max_step(unsigned &hi, unsigned &lo)
{
    const unsigned h2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, 0xf, 0xf, true);
    const unsigned l2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, 0xf, 0xf, true);
    const bool take = h2 > hi || (h2 == hi && l2 > lo);
    hi = take ? h2 : hi; lo = take ? l2 : lo;
}
__device__ inline void wave_max_pixel(const Tile &t, int &cy, int &cx, int &status_bits)
{
    if (cy - 2 < 0 || cx - 2 < 0 || cy >= t.H || cx >= t.W) {
        // low edge: the reference fails (centre kept); a centre outside the frame (a caller's bad input) is
        // pulled onto the frame so that nothing downstream indexes outside the tile
        status_bits |= SCARLET_STATUS_CENTER_AT_EDGE;
        cy = min(max(cy, 0), t.H - 1); cx = min(max(cx, 0), t.W - 1);
        return;
    }
    const int lane = lane_id();
    const int wy = lane / 5, wx = lane - wy * 5;
    const int y = cy - 2 + wy, x = cx - 2 + wx;
    const bool in = lane < 25 && y < t.H && x < t.W;
    // lanes outside the window: key 0 (below every real key, whose low word is >= 7)
    unsigned hi = 0, lo = 0;
    if (in) { hi = argmax_key_hi(t.m[y * t.LW + x]); lo = 31u - (unsigned)lane; }   // row-major rank = lane
    key_max_step<SC_DPP_XOR1>(hi, lo);
    key_max_step<SC_DPP_XOR2>(hi, lo);
    key_max_step<SC_DPP_HALF_MIRROR>(hi, lo);
    key_max_step<SC_DPP_MIRROR>(hi, lo);
    const unsigned h0 = __builtin_amdgcn_readlane((int)hi, 0), l0 = __builtin_amdgcn_readlane((int)lo, 0);
    const unsigned h1 = __builtin_amdgcn_readlane((int)hi, 16), l1 = __builtin_amdgcn_readlane((int)lo, 16);
    const unsigned best = (h1 > h0 || (h1 == h0 && l1 > l0)) ? l1 : l0;
    const int idx = 31 - (int)best;
    cy = cy - 2 + idx / 5;
    cx = cx - 2 + idx % 5;
}

// ---------------------------------------------------------------- a9 centroid
// PSF-weighted moments of the window rows row0, row0 + rstep, ... (all rows: 0, 1): every lane gets the
// wave's sums.  One lane per window column (ww <= P <= 64 lanes), rows in the loop: no index division, the
// x moment of a lane is its column index times its column sum, four rows in flight (the weights come from L2)
__device__ inline void wave_centroid_sums(const Tile &t, const double *__restrict__ psf, int P, int cy, int cx,
                                          int row0, int rstep, double &s0, double &sy, double &sx)
{
    const int rad = P / 2;
    const int ry = min(min(cy, t.H - 1 - cy), rad), rx = min(min(cx, t.W - 1 - cx), rad);
    const int hh = 2 * ry + 1, ww = 2 * rx + 1;
    s0 = 0; sy = 0; sx = 0;
    if (ww <= SC_WAVE) {
        const int ix = lane_id();
        if (ix < ww) {
            const float *mp = t.m + (cy - ry) * t.LW + (cx - rx + ix);
            const double *pp = psf + (rad - ry) * P + (rad - rx + ix);
#pragma unroll 4
            for (int iy = row0; iy < hh; iy += rstep) {
                const double w = (double)mp[iy * t.LW] * pp[iy * P];
                s0 += w; sy += (double)iy * w;
            }
            sx = (double)ix * s0;
        }
    } else {
        for (int i = lane_id(); i < hh * ww; i += SC_WAVE) {
            const int iy = i / ww, ix = i - iy * ww;
            if (iy < row0 || (iy - row0) % rstep) continue;
            const double w = (double)t.m[(cy - ry + iy) * t.LW + (cx - rx + ix)] *
                             psf[(rad - ry + iy) * P + (rad - rx + ix)];
            s0 += w; sy += iy * w; sx += ix * w;
        }
    }
    s0 = wave_sum(s0); sy = wave_sum(sy); sx = wave_sum(sx);
}
// first moments -> new integer centre and sub-pixel shift (measurement.py:80-94)
__device__ inline void wave_centroid_finish(const Tile &t, int P, double s0, double sy, double sx,
                                            int &cy, int &cx, double &dy, double &dx, int &status_bits)
{
    const int rad = P / 2;
    const int ry = min(min(cy, t.H - 1 - cy), rad), rx = min(min(cx, t.W - 1 - cx), rad);
    const double my = sy / s0, mx = sx / s0;
    if (!(my == my) || !(mx == mx) || isinf(my) || isinf(mx)) {
        status_bits |= SCARLET_STATUS_NONFINITE; dy = 0; dx = 0;
        return;
    }
    const double wy = rint(my), wx = rint(mx);
    const int ncy = (int)wy + (cy - ry), ncx = (int)wx + (cx - rx);
    dy = wy - my; dx = wx - mx; cy = ncy; cx = ncx;
}
__device__ inline void wave_centroid(const Tile &t, const double *__restrict__ psf, int P,
                                     int &cy, int &cx, double &dy, double &dx, int &status_bits)
{
    double s0, sy, sx;
    wave_centroid_sums(t, psf, P, cy, cx, 0, 1, s0, sy, sx);
    wave_centroid_finish(t, P, s0, sy, sx, cy, cx, dy, dx, status_bits);
}

// ---------------------------------------------------------------- a10/a11 sweep
// Wedge/row-walk form of the weighted sweep.  A pixel at offset (X, Y) from the peak has
// a = max(|X|,|Y|) >= 1, b = min(|X|,|Y|), level = 2a + b; its strictly closer neighbours
// are, in (major, minor) coordinates,
//     n1 = (a-1, b-1)  cos ~ (a+b)/sqrt2   valid iff a+b > 1 (and in bounds when b = 0)
//     n2 = (a-1, b  )  cos ~ a             always
//     n3 = (a-1, b+1)  cos ~ (a-b)/sqrt2   valid iff b < a-1 and in bounds
//     n4 = (a,   b-1)  cos ~ b             valid iff b > 0
// (the 1/r factor of the cosines cancels in the normalisation).  The image is cut into
// four wedges: right/left (|X| >= |Y|, "x-major", one walker per ROW) and down/up
// (|X| < |Y|, "y-major", one walker per COLUMN).  A walker moves outwards one pixel every
// two levels, so at level ell only minor indices of one parity are active: lane k of a
// 32-lane half handles minor index 2k + ((ell + c_minor) & 1).  One wave64 covers the
// right+left wedges in one trip and the down+up wedges in a second; the four neighbour
// addresses are constant offsets from the walker's own address.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }   // <= 1 ulp
// x * c with 0 * anything == 0 (v_mul_legacy_f32): an unused neighbour has weight 0 and was read from a
// dummy address (possibly NaN / inf), so the legacy product replaces a compare + select per neighbour;
// for c > 0 it is the IEEE product
__device__ __forceinline__ float mul_or_zero(float x, float c)
{
    float r;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(c));
    return r;
}
__device__ __forceinline__ double mul_or_zero(double x, double c) { return c > 0 ? x * c : 0.0; }
__device__ __forceinline__ unsigned long long positive_lanes(float v) { return __builtin_amdgcn_fcmpf(v, 0.f, 2 /* FCMP_OGT */); }
__device__ __forceinline__ unsigned long long positive_lanes(double v) { return __builtin_amdgcn_fcmp(v, 0.0, 2 /* FCMP_OGT */); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ double fast_rcp(double x) { return 1.0 / x; }

// Per-lane walker: everything that does not change along a walk (one per wedge and level
// parity).  A walker is used every second level and moves one pixel outwards per use, so the
// per-level state is a running address and a running major index; validity of the two diagonal
// neighbours is folded into the constants (fb1 = -inf / fb3 = +inf when the neighbour row or
// column does not exist: the weight test below then fails by itself).
template <typename T>
struct Walker {
    int step;                          // address step per use (elements)
    int sa, off1, off3, off4;          // neighbour offsets from p (0 when never valid)
    T c4;                              // weight of n4 = b
    T fb1, fb3;                        // b (or -inf / +inf): n1 weight (a + b) R2 iff a + fb1 >= 2, n3 weight (a - b) R2 iff a - fb3 >= 2
    T famin, flim;                     // active iff famin <= a <= flim (famin = +inf: never)
    int base, b;                       // p = base + a * step ; a = (level - b) >> 1
};
template <typename T>
struct WalkState { int p; T fa; };     // LDS byte address and major index of the NEXT use

// walker of minor index mi (a row for the x-major wedges, a column for the y-major ones)
template <typename T>
__device__ __forceinline__ Walker<T> walker_at(bool xmajor, int mi, int H, int W, int LW, int cy, int cx, int half,
                                               bool enabled = true)
{
    Walker<T> w;
    const int cmin = xmajor ? cy : cx, cmaj = xmajor ? cx : cy;
    const int dimmin = xmajor ? H : W, dimmaj = xmajor ? W : H;
    const int strmin = xmajor ? LW : 1, strmaj = xmajor ? 1 : LW;
    const int dir = half ? -1 : 1;
    const int Yb = mi - cmin;
    const int s = Yb > 0 ? -1 : 1;                              // minor step towards the axis
    w.b = Yb < 0 ? -Yb : Yb;
    const bool valid = enabled && (unsigned)mi < (unsigned)dimmin;
    const bool ok1 = (unsigned)(mi + s) < (unsigned)dimmin;
    const bool ok3 = (unsigned)(mi - s) < (unsigned)dimmin;
    const int amin = xmajor ? max(w.b, 1) : w.b + 1;
    const int lim = half ? cmaj : dimmaj - 1 - cmaj;
    w.base = mi * strmin + cmaj * strmaj;
    w.step = dir * strmaj;
    w.sa = -w.step;
    w.off1 = ok1 ? w.sa + s * strmin : 0;
    w.off3 = ok3 ? w.sa - s * strmin : 0;
    w.off4 = w.b > 0 ? s * strmin : 0;
    w.c4 = (T)w.b;
    w.fb1 = ok1 ? (T)w.b : -(T)INFINITY;
    w.fb3 = ok3 ? (T)w.b : (T)INFINITY;
    w.famin = valid ? (T)amin : (T)INFINITY;
    w.flim = (T)lim;
    return w;
}
// state of walker w for its use at `level` (level = b mod 2)
template <typename T>
__device__ __forceinline__ WalkState<T> walk_from(const Walker<T> &w, int level, unsigned mb)
{
    const int a = (level - w.b) >> 1;
    WalkState<T> st;
    st.p = (int)mb + (int)sizeof(T) * (w.base + a * w.step);         // LDS byte address
    st.fa = (T)a;
    return st;
}
// two-trip layout: lane k of a 32-lane half owns minor index 2k + parity
template <typename T>
__device__ __forceinline__ Walker<T> make_walker(bool xmajor, int e, int H, int W, int LW, int cy, int cx,
                                                 int half, int k2)
{
    const int cmin = xmajor ? cy : cx;
    return walker_at<T>(xmajor, k2 + ((e + cmin) & 1), H, W, LW, cy, cx, half);   // active at levels = e (mod 2)
}
// Compact layout for the levels l <= SC_COMPACT_LAST: a level-l pixel has b <= l / 3 < 16, so
// 16 lanes per wedge (two sides of the axis x eight values b = 2 j + parity) cover all four
// wedges in ONE trip.  (Even, so that the two-trip loop resumes on an odd level.)
#define SC_COMPACT_LAST 46
#ifndef SC_SWEEP_MASKED
#define SC_SWEEP_MASKED 0            // measured (tools/ab_variants.sh): -1.5 % at tile stride 66, +0.8 % at stride 68
#endif
template <typename T>
__device__ __forceinline__ Walker<T> make_compact_walker(int e, int H, int W, int LW, int cy, int cx, int lane)
{
    const int wedge = lane >> 4, side = (lane >> 3) & 1, j = lane & 7;
    const bool xmajor = wedge < 2;
    const int b = 2 * j + e;
    const int mi = (xmajor ? cy : cx) + (side ? -b : b);
    return walker_at<T>(xmajor, mi, H, W, LW, cy, cx, wedge & 1, !(side && b == 0));   // the axis belongs to side 0
}

// `quiet_out` != NULL: compact levels only (1 .. 46, any tile size); *quiet_out receives the number of
// trailing levels without a positive value, for a caller that continues the sweep from level 47 itself.
// `floor`: the early exit counts values above it (0: the caller applies positivity; the detection cut of the
// source initialisation, which zeroes everything <= cut).
// `level_cap` > SC_COMPACT_LAST (with quiet_out): the sweep continues through the two-trip levels up to
// level_cap and reports there -- for a caller whose tile is a box around the peak that is complete only up to
// that level (boxupdate.h).
template <typename T>
__device__ inline void wave_monotonic(const TileT<T> &t, int cy, int cx, T thresh, int *last_level = nullptr,
                                      int *quiet_out = nullptr, T floor = (T)0, int level_cap = 0)
{
    const int H = t.H, W = t.W, LW = t.LW;
    T *m = t.m;
    const int lane = lane_id();
    const int mxr = max(cx, W - 1 - cx), myr = max(cy, H - 1 - cy);
    const int Lall = 2 * max(mxr, myr) + min(mxr, myr);
    const T one_minus = (T)1 - thresh;
    const T R2 = (T)0.70710678118654752440;
    // One level = prepare (addresses and weights: lane arithmetic only) -> load (five LDS reads)
    // -> finish (cap, conditional store).  The level's critical path is store -> load -> finish,
    // so the NEXT level is prepared while this level's reads are in flight.
    typedef __attribute__((address_space(3))) T LdsT;
    const unsigned mb = (unsigned)(uintptr_t)m;          // the tile lives in LDS (every caller's does): its LDS byte
                                                         // address is the low word of the generic pointer
    struct Prep { unsigned p, p1, p2, p3, p4; bool act; T c1, c2, c3; };     // LDS byte addresses of the pixel and its neighbours
    struct Vals { T x0, x1, x2, x3, x4; };
    // weights of this use from the running state, then one step outwards
    auto prepare = [&](WalkState<T> &st, const Walker<T> &w) {
        Prep q;
        const T fa = st.fa;
        q.act = fa >= w.famin && fa <= w.flim;
        // st.p is a running LDS byte address.  An idle lane's addresses may leave the tile or the
        // workgroup's LDS: it only reads there (out-of-range LDS reads return 0, nothing faults) and its
        // values are discarded -- no redirection to a safe address needed
        q.p = (unsigned)st.p;
        q.p2 = q.p + (unsigned)sizeof(T) * (unsigned)w.sa; q.p1 = q.p + (unsigned)sizeof(T) * (unsigned)w.off1;
        q.p3 = q.p + (unsigned)sizeof(T) * (unsigned)w.off3; q.p4 = q.p + (unsigned)sizeof(T) * (unsigned)w.off4;
        const T t1 = fa + w.fb1, t3 = fa - w.fb3;                 // a + b, a - b (exact small integers)
        q.c1 = t1 > (T)1.5 ? t1 * R2 : (T)0;
        q.c2 = fa;
        q.c3 = t3 > (T)1.5 ? t3 * R2 : (T)0;
        st.p += (int)sizeof(T) * w.step; st.fa = fa + (T)1;
        return q;
    };
    // keeps a prepared level where it was written: without it the compiler sinks the preparation
    // below the early-exit branch, i.e. behind the finish of the level before, and its ~15 lane
    // instructions land on the store -> load critical path instead of under the LDS reads
    auto pin = [&](const Prep &q) {
        asm volatile("" ::"v"(q.p), "v"(q.p1), "v"(q.p2), "v"(q.p3), "v"(q.p4), "v"(q.c1), "v"(q.c2), "v"(q.c3));
    };
    // Bank conflicts of these gathers (tools/sweep_banks.py simulates them over random centres): every lane issues its
    // five reads, idle ones at whatever address their walker state gives (values discarded).  At a tile stride == 2
    // (mod 32) that is 11.6 LDS cycles per 64-lane read on average instead of 2 -- the down / up wedges advance by
    // (row - 1, column + 2) per lane, stride - 2 == 0 (mod 32): eight walkers on ONE bank -- and the sweep's reads were
    // more than half of the kernel's LDS-active cycles.  Two remedies, measured on the headline kernel: a stride == 4
    // (mod 32) (common.h SC_XS_STRIDE: 5.9 cycles per read, -2.6 % kernel time) and masking the idle lanes
    // (SC_SWEEP_MASKED: 5.5 / 3.9 cycles at stride 66 / 68; -1.5 % at 66, but +0.8 % at 68: the masking's own
    // instructions).  The value registers persist across levels, so a masked lane keeps (and ignores) what it last read.
    Vals vals_reg = {};
    auto load = [&](const Prep &q, const Walker<T> &w) {
#if SC_SWEEP_MASKED
        if (q.act) {
            vals_reg.x0 = *(LdsT *)q.p; vals_reg.x2 = *(LdsT *)q.p2; vals_reg.x1 = *(LdsT *)q.p1;
            vals_reg.x3 = *(LdsT *)q.p3; vals_reg.x4 = *(LdsT *)q.p4;
        }
        return vals_reg;
#else
        Vals v;
        v.x0 = *(LdsT *)q.p; v.x2 = *(LdsT *)q.p2; v.x1 = *(LdsT *)q.p1; v.x3 = *(LdsT *)q.p3; v.x4 = *(LdsT *)q.p4;
        return v;
#endif
    };
    auto finish = [&](const Prep &q, const Vals &v, const Walker<T> &w) {
        const T inv = fast_rcp(q.c1 + q.c2 + q.c3 + w.c4);
        // unused neighbours were read from a dummy address: mask them (0 * NaN != 0)
        const T t1 = mul_or_zero(v.x1, q.c1), t3 = mul_or_zero(v.x3, q.c3), t4 = mul_or_zero(v.x4, w.c4);
        const T cap = (fma_t(v.x2, q.c2, t1) + (t3 + t4)) * inv * one_minus;
        const bool lower = q.act && cap < v.x0;
        if (lower) *(LdsT *)q.p = cap;
        // lanes whose pixel ends up positive, as a lane mask straight from the compare (idle lanes count as -1)
        return positive_lanes(q.act ? (lower ? cap : v.x0) - floor : (T)-1);
    };
    // Early exit (only when the caller applies positivity afterwards, as the source pipeline
    // does, and 0 <= thresh <= 1): every closer neighbour of a level-l pixel lies on levels
    // l-1, l-2 or l-3, and the cap is a non-negative combination of them.  Once three
    // consecutive levels hold no positive value, every later pixel ends up <= 0 and is
    // zeroed by prox_plus anyway -- the remaining levels need not be swept.  *last_level
    // receives the last level that was computed; the caller zeroes everything beyond it.
    const bool early = last_level != nullptr && thresh >= (T)0 && thresh <= (T)1;
    int quiet = 0;                               // consecutive levels without a positive value
    int done = 1 << 30;                          // 1 << 30: swept to the end
    int ell = 1;
    bool stop = false;
    {   // ---- levels 1 .. 46: one trip per level
        const Walker<T> c0 = make_compact_walker<T>(0, H, W, LW, cy, cx, lane);
        const Walker<T> c1 = make_compact_walker<T>(1, H, W, LW, cy, cx, lane);
        WalkState<T> s0 = walk_from(c0, 2, mb), s1 = walk_from(c1, 1, mb);
        const int Lc = min(Lall, SC_COMPACT_LAST);
        Prep pa = prepare(s1, c1);
        while (ell <= Lc) {
            Prep na;
            {   // odd level
                const Vals va = load(pa, c1);
                na = prepare(s0, c0);                           // (no active pixel beyond Lall)
                pin(na);
                const unsigned long long pm = finish(pa, va, c1);
                wave_sync();
                quiet = pm != 0 ? 0 : quiet + 1;
                if (early && quiet >= 3) { stop = true; break; }
                ++ell;
            }
            if (ell > Lc) break;
            {   // even level
                const Vals va = load(na, c0);
                pa = prepare(s1, c1);
                pin(pa);
                const unsigned long long pm = finish(na, va, c0);
                wave_sync();
                quiet = pm != 0 ? 0 : quiet + 1;
                if (early && quiet >= 3) { stop = true; break; }
                ++ell;
            }
        }
    }
    if (quiet_out) *quiet_out = quiet;
    const int Lend = level_cap > SC_COMPACT_LAST ? min(Lall, level_cap) : Lall;
    if (!stop && ell <= Lend && (!quiet_out || level_cap > SC_COMPACT_LAST)) {
        // ---- levels 47 ..: two trips per level (right/left wedges, then down/up), one walker
        // per 32-lane half and level parity.  (ell == 47 here.)
        const int half = lane >> 5, k2 = (lane & 31) << 1;
        const Walker<T> wx0 = make_walker<T>(true, 0, H, W, LW, cy, cx, half, k2);
        const Walker<T> wx1 = make_walker<T>(true, 1, H, W, LW, cy, cx, half, k2);
        const Walker<T> wy0 = make_walker<T>(false, 0, H, W, LW, cy, cx, half, k2);
        const Walker<T> wy1 = make_walker<T>(false, 1, H, W, LW, cy, cx, half, k2);
        WalkState<T> sx0 = walk_from(wx0, ell + 1, mb), sx1 = walk_from(wx1, ell, mb);
        WalkState<T> sy0 = walk_from(wy0, ell + 1, mb), sy1 = walk_from(wy1, ell, mb);
        Prep pa = prepare(sx1, wx1), pb = prepare(sy1, wy1);
        while (ell <= Lend) {
            Prep na, nb;
            {   // odd level
                const Vals va = load(pa, wx1), vb = load(pb, wy1);
                na = prepare(sx0, wx0); nb = prepare(sy0, wy0);
                pin(na); pin(nb);
                const unsigned long long pm = finish(pa, va, wx1) | finish(pb, vb, wy1);
                wave_sync();
                quiet = pm != 0 ? 0 : quiet + 1;
                if (early && quiet >= 3) { stop = true; break; }
                ++ell;
            }
            if (ell > Lend) break;
            {   // even level
                const Vals va = load(na, wx0), vb = load(nb, wy0);
                pa = prepare(sx1, wx1); pb = prepare(sy1, wy1);
                pin(pa); pin(pb);
                const unsigned long long pm = finish(na, va, wx0) | finish(nb, vb, wy0);
                wave_sync();
                quiet = pm != 0 ? 0 : quiet + 1;
                if (early && quiet >= 3) { stop = true; break; }
                ++ell;
            }
        }
    }
    if (quiet_out && level_cap > SC_COMPACT_LAST) *quiet_out = quiet;
    if (stop) done = ell;                        // the last level that was computed
    if (last_level) *last_level = done;
}

// ---------------------------------------------------------------- a16 flip symmetry
template <typename T>
__device__ inline void wave_flip_symmetry(const TileT<T> &t, const SymWindow &s, bool sdss, T strength)
{
    T *m = t.m;
    const int n = s.h * s.w;
    const T a = (T)(0.5 * (double)strength), bq = (T)1 - strength;
    for (int i = lane_id(); 2 * i <= n - 1; i += SC_WAVE) {
        const int jx = n - 1 - i;
        const int iy = i / s.w, ix = i - iy * s.w;
        const int jy = jx / s.w, jxx = jx - jy * s.w;
        T *pi = &m[(s.y0 + iy) * t.LW + s.x0 + ix];
        T *pj = &m[(s.y0 + jy) * t.LW + s.x0 + jxx];
        const T xi = *pi, xj = *pj;
        if (sdss) { const T r = xj < xi ? xj : xi; *pi = r; *pj = r; }
        else { *pi = a * (xi + xj) + bq * xi; *pj = a * (xj + xi) + bq * xj; }
    }
    wave_sync();
}

// ---- k-space symmetry, split over a pair of waves (see wave_kspace_symmetry for the maths)
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct KsGeom { int h, w, ry, rx, ntr, ntc, wp, Fy, Fx; };
__device__ __forceinline__ KsGeom ks_geom(const SymWindow &s)
{
    KsGeom g;
    g.h = s.h; g.w = s.w; g.ry = s.h / 2; g.rx = s.w / 2;
    g.ntr = round16(s.h) >> 4; g.wp = round16(s.w); g.ntc = g.wp >> 4;
    g.Fy = uniform(dev_next_fast_len(2 * s.h + 10));
    int Fx = dev_next_fast_len(2 * s.w + 10);
    while (Fx & 1) Fx = dev_next_fast_len(Fx + 1);
    g.Fx = uniform(Fx);
    return g;
}

// Hankel vectors: this wave fills entries q = lane + 64 * half.  Returns s = sin(2 pi dy) / Fy
// (0 for odd Fy), the coefficient of the rank-1 term.
__device__ inline float pair_ks_vectors(const KsGeom &g, double dy, double dx, float *vec, int half,
                                        float *made = nullptr)
{
    float *av = vec, *bv = vec + 128, *cv = vec + 256;
    // float64 only where the cancellation lives (the arguments and sin/cos); the final quotients
    // are float32: the kernel entries feed a float32 GEMM
    double s2x, c2x;
    const double s2y = sinpi(2.0 * dy);
    sincospi(2.0 * dx, &s2x, &c2x);
    const double iFy = 1.0 / g.Fy, iFx = 1.0 / g.Fx;
    const int q = lane_id() + SC_WAVE * half;
    float va = 0.f, vb = 0.f, vc = 0.f;
    if (q <= 2 * (g.h - 1)) {
        const int n = q - 2 * g.ry;
        const double tt = (double)n - 2.0 * dy;
        double sn, cs;
        sincospi(tt * iFy, &sn, &cs);
        const double spt = (n & 1) ? s2y : -s2y;
        va = sn == 0.0 ? 1.f : (float)((g.Fy & 1) ? spt : spt * cs) / (float)(g.Fy * sn);
    }
    if (q <= 2 * (g.w - 1)) {
        const int n = q - 2 * g.rx;
        const double tt = (double)n - 2.0 * dx;
        double sn, cs;
        sincospi(tt * iFx, &sn, &cs);
        const double spt = (n & 1) ? s2x : -s2x;
        const double cpt = (n & 1) ? -c2x : c2x;
        if (sn == 0.0) { vb = 1.f; vc = 0.f; }
        else {
            const float r = 1.0f / (float)(g.Fx * sn);
            vb = (float)(spt * cs) * r; vc = (float)(-(1.0 - cpt) * cs) * r;
        }
    }
    av[q] = va; bv[q] = vb; cv[q] = vc;
    if (made) { made[0] = va; made[1] = vb; made[2] = vc; }
    return (g.Fy & 1) ? 0.f : (float)(s2y * iFy);
}

// rank-1 term, part 1 (tile reads only): zv[j] = sum_i (-1)^(i-ry) X[i][j]
__device__ inline void pair_ks_colsums(const Tile &t, const SymWindow &s, const KsGeom &g, float *zv)
{
    const int lane = lane_id();
    float v = 0.f;
    if (lane < g.w)
        for (int i = 0; i < g.h; ++i) {
            const float x = t.m[(s.y0 + i) * t.LW + s.x0 + lane];
            v += ((i - g.ry) & 1) ? -x : x;
        }
    zv[lane] = lane < g.w ? v : 0.f;
}
// part 2 (needs the partner's half of cv): zv <- C zv
__device__ inline void pair_ks_z(const KsGeom &g, const float *cv, float *zv)
{
    const int lane = lane_id();
    wave_sync();
    // four terms per trip: zv[j2 .. j2 + 3] in two 8-byte broadcast reads, cv in two paired reads.  Same order of
    // the FMA chain; the up to three extra terms multiply zv = 0 (zv is zero from w on, cv runs into finite
    // neighbours), which leaves the sum unchanged
    float z = 0.f;
    if (lane < g.w)
        for (int j2 = 0; j2 < g.w; j2 += 4) {
            const float4 z4 = lds_load4(zv + j2);              // (8-byte aligned for every tile shape, not always 16)
            const float *cp = cv + lane + j2;
            z = fma_t(cp[0], z4.x, z); z = fma_t(cp[1], z4.y, z);
            z = fma_t(cp[2], z4.z, z); z = fma_t(cp[3], z4.w, z);
        }
    wave_sync();
    zv[lane] = z;
    wave_sync();
}

// GEMM 1: T[:, tc] = Xw . Hankel(bv)[:, tc] for this wave's column tiles tc = half, half + 2.
// NTR row tiles x NI column tiles are compile-time (the caller switches on the wave-uniform
// window size): straight-line MFMA chains, no control flow between them.
// PEEL: the operand masking below is compiled only into the fetch of the last k-group (a uniform branch picks
// the variant); the second code path costs registers in callers that inline both halves of the pair
// decomposition (wave_kspace_symmetry), so only the pair kernel asks for it.
template <int NTR, int NI, bool PEEL>
__device__ __forceinline__ void pair_ks_gemm1_impl(const Tile &t, const SymWindow &s, const KsGeom &g, const float *vec,
                                                   int half, f32x4 (&T)[4][2])
{
    // The sum over k is taken in groups of 16: in group k0 the j-th MFMA (j = 0..3) lets lane
    // (lr, lq) supply k = k0 + lq + 4 j -- every lane reads four floats of its X row and of the Hankel
    // vector per group through one address and four immediate offsets (0, 4, 8, 12), and a whole group
    // is in flight while the previous one multiplies.  The lane-group offset is lq (not 4 lq): with the
    // tile stride == 2 (mod 32) the 32 lanes of a read's half-wave then hit banks 2 lr + lq, all
    // different (4 lq collided with lr + 2: two-way conflicts on every operand read); and the k of one
    // MFMA are consecutive, so the products are summed in ascending k.
    const float *bv = vec + 128;
    const int LW = t.LW, lane = lane_id(), lr = lane & 15, lq = lane >> 4;
    const float *rowp[NTR];                     // rows beyond the window read row h - 1 (zeroed at the end)
#pragma unroll
    for (int tr = 0; tr < NTR; ++tr)
        rowp[tr] = t.m + (s.y0 + min((tr << 4) + lr, g.h - 1)) * LW + s.x0 + lq;
    const float *bp = bv + (half << 4) + lr + lq;
    struct Ops { float a[NTR][4], b[NI][4]; };
    // columns beyond the window (only in the last group: wp - w < 16) must count as zero
    auto fetch_as = [&](int k0, auto masked) {
        constexpr int MASKED = decltype(masked)::value;     // 0: never, 1: always, 2: decided per group at run time
        Ops o;
        const bool last = MASKED == 1 || (MASKED == 2 && k0 + 16 > g.w);
#pragma unroll
        for (int tr = 0; tr < NTR; ++tr)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = rowp[tr][k0 + 4 * j];
                o.a[tr][j] = (last && k0 + lq + 4 * j >= g.w) ? 0.f : x;
            }
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) o.b[i][j] = bp[k0 + (i << 5) + 4 * j];
        return o;
    };
    auto fetch = [&](int k0) {
        if (!PEEL) return fetch_as(k0, std::integral_constant<int, 2>{});
        if (k0 + 16 > g.w) return fetch_as(k0, std::integral_constant<int, 1>{});
        return fetch_as(k0, std::integral_constant<int, 0>{});
    };
    auto multiply = [&](const Ops &o) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tr = 0; tr < NTR; ++tr)
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    T[tr][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[tr][j], o.b[i][j], T[tr][i], 0, 0, 0);
    };
    // two groups per trip, ping-pong operand sets (no register copies)
    Ops o0 = fetch(0);
    for (int k0 = 0; k0 < g.wp; k0 += 32) {
        const Ops o1 = fetch(min(k0 + 16, g.wp - 16));
        multiply(o0);
        if (k0 + 16 < g.wp) {
            o0 = fetch(min(k0 + 32, g.wp - 16));
            multiply(o1);
        }
    }
    // rows of T beyond the window (last row tile only) came from the clamped row pointer: zero them
    const int rbase = ((NTR - 1) << 4) + 4 * lq;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const f32x4 v = T[NTR - 1][i];
        T[NTR - 1][i] = (f32x4){rbase < g.h ? v[0] : 0.f, rbase + 1 < g.h ? v[1] : 0.f,
                                rbase + 2 < g.h ? v[2] : 0.f, rbase + 3 < g.h ? v[3] : 0.f};
    }
}

template <bool PEEL = false>
__device__ inline void pair_ks_gemm1(const Tile &t, const SymWindow &s, const KsGeom &g, const float *vec,
                                     int half, f32x4 (&T)[4][2])
{
#pragma unroll
    for (int tr = 0; tr < 4; ++tr)
#pragma unroll
        for (int i = 0; i < 2; ++i) T[tr][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int ni = half < g.ntc ? (half + 2 < g.ntc ? 2 : 1) : 0;
    if (ni == 0) return;
#define SC_G1(NTR_) do { if (ni == 2) pair_ks_gemm1_impl<NTR_, 2, PEEL>(t, s, g, vec, half, T);    \
                         else pair_ks_gemm1_impl<NTR_, 1, PEEL>(t, s, g, vec, half, T); } while (0)
    switch (g.ntr) { case 1: SC_G1(1); break; case 2: SC_G1(2); break; case 3: SC_G1(3); break; default: SC_G1(4); }
#undef SC_G1
}

// GEMM 2 + epilogue in place for this wave's column tiles
// PAIR: address arithmetic of the epilogue for the pair kernel (one address per column tile, the last row
// tile alone checks the bottom edge); the other callers keep the form that costs them fewer registers
template <int NTR, int NI, bool PAIR>
__device__ __forceinline__ void pair_ks_gemm2_impl(const Tile &t, const SymWindow &s, const KsGeom &g, const float *vec,
                                                   const float *zv, int half, const f32x4 (&T)[4][2], float sy, bool rank1)
{
    float *m = t.m;
    const float *av = vec;
    const int LW = t.LW, lane = lane_id(), lr = lane & 15, lq = lane >> 4;
    // rank-1 term: row parity of (row - ry) depends on r only (16 tr and 4 lq are even)
    float syr[4], zc[NI];
#pragma unroll
    for (int r = 0; r < 4; ++r) syr[r] = rank1 ? (((r - g.ry) & 1) ? -sy : sy) : 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) zc[i] = rank1 ? zv[((half + 2 * i) << 4) + lr] : 0.f;
    // Hankel operand: A[i][k] = av[i + k] depends on tr + tk only -- the 2 NTR - 1 distinct 16 x 4
    // slices are read once, up front
    float *dummy = const_cast<float *>(vec) + 128 + lane;
    float ah[2 * NTR - 1][4];
#pragma unroll
    for (int d = 0; d < 2 * NTR - 1; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) ah[d][r] = av[(d << 4) + lr + 4 * lq + r];
#pragma unroll
    for (int tr = 0; tr < NTR; ++tr) {
        // the pixels this lane will combine with: requested before the MFMA chain, used after it
        float xin[NI][4];
        float *pp[NI][4];
        bool ok[NI][4];
        const int row0 = (tr << 4) + lq * 4;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int jcol = ((half + 2 * i) << 4) + lr;
            float *p0 = &m[(s.y0 + row0) * LW + s.x0 + jcol];     // one address per column tile, rows at + r LW
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + r;
                // lanes outside the window combine a dummy slot (the B vector is dead after GEMM 1):
                // no branches between the MFMA chains
                if (PAIR) {
                    // only the last row tile can leave the window at the bottom (h > 16 (NTR - 1))
                    ok[i][r] = (tr < NTR - 1 || row < g.h) && jcol < g.w;
                    pp[i][r] = ok[i][r] ? p0 + r * LW : dummy;
                } else {
                    ok[i][r] = row < g.h && jcol < g.w;
                    pp[i][r] = ok[i][r] ? &m[(s.y0 + row) * LW + s.x0 + jcol] : dummy;
                }
                xin[i][r] = *pp[i][r];
            }
        }
        f32x4 acc[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tk = 0; tk < NTR; ++tk)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[tr + tk][r], T[tk][i][r], acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = xin[i][r];
                const float y2 = acc[i][r] + syr[r] * zc[i];
                *pp[i][r] = (x <= 0.f) ? 0.f : 0.5f * x + 0.5f * y2;
            }
    }
}

template <bool PAIR = false>
__device__ inline void pair_ks_gemm2(const Tile &t, const SymWindow &s, const KsGeom &g, const float *vec,
                                     const float *zv, int half, const f32x4 (&T)[4][2], float sy, bool rank1)
{
    const int ni = half < g.ntc ? (half + 2 < g.ntc ? 2 : 1) : 0;
    if (ni == 0) return;
#define SC_G2(NTR_) do { if (ni == 2) pair_ks_gemm2_impl<NTR_, 2, PAIR>(t, s, g, vec, zv, half, T, sy, rank1);   \
                         else pair_ks_gemm2_impl<NTR_, 1, PAIR>(t, s, g, vec, zv, half, T, sy, rank1); } while (0)
    switch (g.ntr) { case 1: SC_G2(1); break; case 2: SC_G2(2); break; case 3: SC_G2(3); break; default: SC_G2(4); }
#undef SC_G2
}

// ---------------------------------------------------------------- a17 k-space symmetry
// out = 1/2 X + 1/2 [A (X B) + s (sigma sigma^T X) C], X<=0 -> 0   (see prox_ops.h).
// T = X B (up to 64 x 64) stays in 16 accumulator tiles (64 VGPRs); the second product
// consumes it straight from the accumulators: accumulator register r of lane (q = lane>>4)
// holds row 4q + r of a 16-row tile, so k-step r of the second MFMA sums over rows
// {r, 4+r, 8+r, 12+r} and the Hankel operand is simply read at the matching index.
// vec: LDS floats, 2*64 (av) + 2*64 (bv) + 2*64 (cv) + 64 (zv) per wave.
__device__ inline void wave_kspace_symmetry(const Tile &t, const SymWindow &s, double dy, double dx,
                                            float *vec, long long *dbg = nullptr)
{
    // One wave does both halves of the pair decomposition above (column tiles {0, 2} and {1, 3});
    // the window geometry must be wave-uniform (wave_symmetry moves it to SGPRs) so that the tile
    // counts select straight-line MFMA chains.
    (void)dbg;
    const KsGeom g = ks_geom(s);
    float *zv = vec + 384;
    const float sy = pair_ks_vectors(g, dy, dx, vec, 0);
    pair_ks_vectors(g, dy, dx, vec, 1);
    const bool rank1 = sy != 0.f;
    if (rank1) { pair_ks_colsums(t, s, g, zv); pair_ks_z(g, vec + 256, zv); }
    wave_sync();
    f32x4 T0[4][2], T1[4][2];
    pair_ks_gemm1(t, s, g, vec, 0, T0);
    pair_ks_gemm1(t, s, g, vec, 1, T1);
    wave_sync();
    pair_ks_gemm2(t, s, g, vec, zv, 0, T0, sy, rank1);
    pair_ks_gemm2(t, s, g, vec, zv, 1, T1, sy, rank1);
    wave_sync();
}

__device__ inline void wave_symmetry(const Tile &t, int cy, int cx, int algorithm, float strength,
                                     double dy, double dx, bool use_fill, float fill, float *vec,
                                     long long *dbg = nullptr)
{
    cy = uniform(cy); cx = uniform(cx); dy = uniform(dy); dx = uniform(dx);   // wave-uniform by construction
    SymWindow s = sym_window(t.H, t.W, cy, cx);
    if (algorithm & SCARLET_SYM_FULL_WINDOW) {   // bare operator on the whole array (operator.py:231-288)
        algorithm &= ~SCARLET_SYM_FULL_WINDOW;
        s.y0 = 0; s.x0 = 0; s.h = t.H; s.w = t.W; s.centered = false;
    }
    if (algorithm == SCARLET_SYM_KSPACE) {
        if (s.centered) return;
        wave_kspace_symmetry(t, s, dy, dx, vec, dbg);
    } else {
        wave_flip_symmetry<float>(t, s, algorithm == SCARLET_SYM_SDSS, strength);
    }
    if (use_fill && !s.centered) {
        for (int i = lane_id(); i < t.H * t.W; i += SC_WAVE) {
            const int y = i / t.W, x = i - y * t.W;
            if (y < s.y0 || y >= s.y0 + s.h || x < s.x0 || x >= s.x0 + s.w) t.m[y * t.LW + x] = fill;
        }
        wave_sync();
    }
}

#define SC_WAVE_VEC_FLOATS 448       // av, bv, cv (128 each) + zv (64)
