// fused2.h -- k_iterate2: the fused iteration of fused.h with EIGHT waves per scene.
//
// The constraint chain of one component is a long dependent sequence (k-space symmetry
// GEMMs -> radial sweep -> normalisation); with four waves per scene and LDS limiting a CU
// to two scenes, each SIMD holds two waves and sits idle ~45 % of the time waiting on LDS /
// MFMA / HBM latencies (profiles/r01_notes.md).  LDS cannot hold a third scene, so the
// second lever is used: the same two scenes per CU run as 2 x 8 waves (four per SIMD, 128
// VGPRs each) and every component is served by a PAIR of waves (waves 2k and 2k + 1, on two
// different SIMDs):
//
//   phases 0/1   pixel-parallel over 512 threads (two float4 groups per thread)
//   k-space symmetry  vectors: 64 entries per wave; T = X B and Y = A T split by column
//                tiles (wave h owns tiles h, h + 2) -- T never leaves the accumulators
//   sweep        one wave of the pair (its partner waits at the barrier and costs no issue slots)
//   tail         float4 groups split between the pair
//
// The two waves of a pair synchronise with each other only (phase counters in LDS, release /
// acquire fences, s_sleep while waiting), so the four components of a scene run their chains
// independently and the workgroup meets again at the end of the constraint phase.  Used when
// K <= 4 (four pairs) and B <= 5; other shapes run k_iterate.  Same reference rows as fused.h.
#pragma once
#include "fused.h"

#define SC_FB2 512
#define SC_NW2 (SC_FB2 / SC_WAVE)
#define SC_PAIR_VEC_FLOATS 512       // av, bv, cv (128 each) + one zv (64) per wave of the pair

// FFT lengths of a window of half-size r (h = 2 r + 1): fl[0][r] = next_fast_len(2 h + 10) and
// fl[1][r] = the first EVEN fast length from there (fft.py:95-115 via operator.py:253-288).  The
// 64 entries are compile-time constants (r < 32: windows of up to 63 pixels); every workgroup copies
// them into LDS at kernel start -- requested before the tile loads, written after they are issued --
// so that the constraint phase does not wait on constant loads.
struct KsLengths {
    unsigned short v[2][32];
    static constexpr int fast_len(int n)
    {   // smallest 2^a 3^b 5^c >= n (scipy.fftpack.next_fast_len)
        for (int m = n;; ++m) {
            int q = m;
            while (q % 2 == 0) q /= 2;
            while (q % 3 == 0) q /= 3;
            while (q % 5 == 0) q /= 5;
            if (q == 1) return m;
        }
    }
    constexpr KsLengths() : v{}
    {
        for (int r = 0; r < 32; ++r) {
            const int F = fast_len(2 * (2 * r + 1) + 10);
            int Fe = F;
            while (Fe & 1) Fe = fast_len(Fe + 1);
            v[0][r] = (unsigned short)F; v[1][r] = (unsigned short)Fe;
        }
    }
};
__device__ const KsLengths sc_ks_lengths = KsLengths();
__device__ __forceinline__ unsigned ks_request_lengths(int tid)
{
    return tid < 64 ? (unsigned)(&sc_ks_lengths.v[0][0])[tid] : 0u;
}
__device__ __forceinline__ void ks_fill_lengths(unsigned short (*fl)[32], int tid, unsigned requested)
{
    if (tid < 64) (&fl[0][0])[tid] = (unsigned short)requested;
}
__device__ __forceinline__ KsGeom ks_geom(const SymWindow &s, const unsigned short (*fl)[32])
{
    KsGeom g;
    g.h = s.h; g.w = s.w; g.ry = s.h / 2; g.rx = s.w / 2;
    g.ntr = round16(s.h) >> 4; g.wp = round16(s.w); g.ntc = g.wp >> 4;
    g.Fy = uniform((int)fl[0][g.ry]);
    g.Fx = uniform((int)fl[1][g.rx]);
    return g;
}

// XS > 0: the exact-shape instance -- K == KM, B == BM, H == W == XS, scalar weight 1 and the default
// constraint pipeline (symmetric, monotonic, no sparsity) are compile-time facts, so the index
// arithmetic, the bounds predicates and the pipeline switches fold away (same arithmetic on the
// pixels, bit-identical results).  XS == 0: everything is read from the arguments.
//
// P (persistent): the body is one iteration of k_fit2x's loop (below).  `reentered`: not the launch's first
// iteration of this scene; `resident`: the LDS tiles already hold this scene's morphologies (the final pass of
// the previous iteration left them there), so phase 0 reads LDS instead of HBM.  Returns bit 0 = scene still
// active, bit 1 = tiles resident for the next iteration.
//
// What one iteration hands to the next inside a launch travels through LDS (tiles, SEDs, centres, shifts), never
// through a global store followed by a cached global load: the CU's vector L1 can still hold the line from the
// PREVIOUS iteration's load of the same address (centres, shifts: re-read every iteration; measured: ~1 % of the
// scenes of a 10 000-scene launch then started an iteration from a stale centre or SED), and nothing short of an
// L1 invalidate per iteration (buffer_inv, ~2 us) repairs that.  The two streams that must come back from memory
// -- the previous morphology for the convergence sums, the cached Hankel vectors -- are loaded past the L1
// (non-temporal loads, L2-served).
template <int KM, int BM, int XS, bool P>
__device__ __forceinline__ int iterate2_body(const FusedArgs &a, const int s, const int c0, const int it_old, const bool resident,
                                             const bool reentered = false)
{
    static_assert(KM <= 4, "one pair of waves per component");
    static_assert(1 + KM * BM <= 32 && 4 * SC_NW2 == 32, "the partial sums are combined by 32 rows of 16 lanes");
    constexpr bool X = XS > 0;
    extern __shared__ __align__(16) float lds[];
    const int K = X ? KM : a.K, B = X ? BM : a.B, H = X ? XS : a.H, W = X ? XS : a.W, HW = H * W, LW = X ? SC_XS_STRIDE : tile_stride(W);
    const int tile_floats = H * LW;
    // float4 access to the tiles: one 16-byte LDS instruction where the rows are 16-byte aligned (the exact-shape
    // stride 68), two 8-byte ones otherwise
#ifndef SC_TILE_B128
#define SC_TILE_B128 1
#endif
    auto tl4 = [](const float *p) {
        if constexpr (X && SC_TILE_B128 && (SC_XS_STRIDE % 4) == 0) return *reinterpret_cast<const float4 *>(p);
        else return lds_load4(p);
    };
    auto ts4 = [](float *p, float4 v) {
        if constexpr (X && SC_TILE_B128 && (SC_XS_STRIDE % 4) == 0) *reinterpret_cast<float4 *>(p) = v;
        else lds_store4(p, v);
    };
    const bool symmetric = X ? true : a.symmetric != 0, monotonic = X ? true : a.monotonic != 0;
    float *tiles = lds;
    float *vecs = lds + (size_t)K * tile_floats;
    constexpr int NG = KM * (KM + 1) / 2;
    constexpr int NP = 1 + KM * BM;
    constexpr int GPT = 2;                       // float4 groups per thread in phases 0/1 (H, W <= 64)
    constexpr int GPW = 8;                       // float4 groups per lane in the pair's final pass
    __shared__ float red[4 * SC_NW2][NP > NG ? NP : NG];  // partial sums per 16-lane row of every wave
    __shared__ double mat[2][KM * KM > BM * BM ? KM * KM : BM * BM];
    __shared__ float sed_s[KM * BM], sed_new[KM * BM];
    __shared__ float step_s[2];
    __shared__ double conv_s[KM][2], conv_m[KM][2][2];
    __shared__ int lstop_s[KM];
    __shared__ float nmax_s[KM][2];
    __shared__ int pair_flag[KM][2];           // phase counters of the pair-local synchronisation
    __shared__ double cent_s[KM][2][3];        // centroid moments of the two row halves
    __shared__ unsigned short fl_s[2][32];
    __shared__ int ctl_s[2][2];                // persistent form, by iteration parity: {scene still active, tiles resident}
    __shared__ int carry_cen[KM][2];           // persistent form: centre / shift / SED of every component as the iteration left them
    __shared__ double carry_sh[KM][2];
    __shared__ float carry_sed[KM * BM];
    // persistent form: rows [lo, hi] outside which factor buffer 0 / 1 of component k holds exact zeros -- what this
    // launch's final passes wrote there (everything beyond the sweep's cut); [0, H - 1] until a buffer has been written
    __shared__ short rowrange_s[KM][2][2];
    const int tid = threadIdx.x, lane = tid & 63, wid = uniform(tid >> 6);    // wid in an SGPR: scalar branches
    float *const morph0 = a.morph[0], *const morph1 = a.morph[1], *const sed0 = a.sed[0], *const sed1 = a.sed[1];
    const float *min_g = (c0 ? morph1 : morph0) + (size_t)s * K * HW;
    float *mout_g = (c0 ? morph0 : morph1) + (size_t)s * K * HW;
    const float *sed_in = (c0 ? sed1 : sed0) + (size_t)s * K * B;
    float *sed_out = (c0 ? sed0 : sed1) + (size_t)s * K * B;
    const int it_new = it_old + 1;
    const int ngroups = HW >> 2, gpr = W >> 2;           // float4 groups, groups per row
    const float *img = a.images + (size_t)s * B * HW;
    const float *wgt = (!X && a.weights) ? a.weights + (size_t)s * B * HW : nullptr;
#define STAMP(i) do { if (a.stamps && tid == 0) a.stamps[(size_t)s * 16 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    STAMP(0);
    // (diagnostics: the constant 100 MHz counter beside the shader clock of stamps 0 / 6 gives the clock the chip holds)
    if (a.stamps && tid == 0) a.stamps[(size_t)s * 16 + 7] = (long long)__builtin_amdgcn_s_memrealtime();
#ifdef SC_STAMP_PERIOD      // (diagnostic build: the start of the last two iterations, by parity, in slots 8 / 9 of the second half)
    if (P && a.stamps && tid == 0) a.stamps[(size_t)a.S * 16 + (size_t)s * 2 + (it_new & 1)] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    if (P && a.stamps && tid == 0 && !reentered) {     // launch-level diagnostics: when and where this workgroup started
        a.stamps[(size_t)s * 16 + 14] = (long long)__builtin_amdgcn_s_memrealtime();
        a.stamps[(size_t)s * 16 + 15] = (long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11))        // HW_ID
                                        | ((long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 32);   // XCC_ID
    }
    const unsigned fl_req = ks_request_lengths(tid);
    // The Hankel vectors of the k-space symmetry depend on (H, W, centre, shift) only; the shift moves
    // every fifth iteration and the centre rarely, so each wave keeps its half of the vectors of the
    // previous iteration in the workspace and recomputes them (float64 sincospi, ~4k cycles on the
    // constraint chain) only when that key changed.  Requested here, under everything else.
    // centre and shift of this pair's component: requested here as well (they are read again only
    // after the gradient step; a load at that point would put an HBM round trip on the constraint chain)
    int pre_cy = 0, pre_cx = 0;
    double pre_dy = 0, pre_dx = 0;
    if (X && (wid >> 1) < K) {                  // (the generic instance has no registers to spare: it loads late)
        const int cpre = s * K + (wid >> 1);
        if (!(P && reentered)) {
            pre_cy = a.centers[2 * cpre]; pre_cx = a.centers[2 * cpre + 1];
            pre_dy = a.shifts[2 * cpre]; pre_dx = a.shifts[2 * cpre + 1];
        }
        if (P) {       // (read unconditionally, selected by value: a select between an LDS and a global POINTER would
                       // turn both loads into flat loads)
            const int lcy = carry_cen[wid >> 1][0], lcx = carry_cen[wid >> 1][1];
            const double ldy = carry_sh[wid >> 1][0], ldx = carry_sh[wid >> 1][1];
            if (reentered) { pre_cy = lcy; pre_cx = lcx; pre_dy = ldy; pre_dx = ldx; }
        }
    }
    float *kcache = nullptr;
    float kc_v[3] = {0.f, 0.f, 0.f};
    unsigned kc_hdr = 0;
    if (X && a.kscache && (wid >> 1) < K) {
        kcache = a.kscache + ((size_t)(s * K + (wid >> 1)) * 2 + (wid & 1)) * SC_KSC_FLOATS;
        if (P) {                               // past the L1: this wave stored them an iteration ago
            kc_v[0] = __builtin_nontemporal_load(kcache + lane); kc_v[1] = __builtin_nontemporal_load(kcache + 64 + lane);
            kc_v[2] = __builtin_nontemporal_load(kcache + 128 + lane);
            kc_hdr = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(kcache) + 192 + (lane & 7));
        } else {
            kc_v[0] = kcache[lane]; kc_v[1] = kcache[64 + lane]; kc_v[2] = kcache[128 + lane];
            kc_hdr = reinterpret_cast<const unsigned *>(kcache)[192 + (lane & 7)];
        }
    }

    // ---------------- phase 0: issue every global load, tiles -> LDS, Gram
    float4 mreg[GPT][KM];
    auto load4 = [&](const float *p, bool past_l1) {
        if (P && past_l1) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
            return make_float4(v[0], v[1], v[2], v[3]);
        }
        return *reinterpret_cast<const float4 *>(p);
    };
    if (!P || !resident) {
        // (re-entered without resident tiles -- after a NaN result, or the diagnostic switch: the planes were stored
        // by this workgroup an iteration ago, read them past the L1)
#pragma unroll
        for (int j = 0; j < GPT; ++j) {
            const int g = tid + j * SC_FB2;
#pragma unroll
            for (int k = 0; k < KM; ++k)
                mreg[j][k] = (g < ngroups && k < K) ? load4(min_g + (size_t)k * HW + 4 * (size_t)g, reentered)
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // images: the first group's are requested now, the second group's when phase 1 starts
    // (under the first group's arithmetic) -- 128 VGPRs do not hold both next to the accumulators
    float4 ireg[GPT][BM];
    auto load_images = [&](int j) {
        const int g = tid + j * SC_FB2;
#pragma unroll
        for (int b = 0; b < BM; ++b)
            ireg[j][b] = (g < ngroups && b < B) ? reinterpret_cast<const float4 *>(img + (size_t)b * HW)[g]
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    load_images(0);
    for (int i = tid; i < K * B; i += SC_FB2) {
        float v = 0.f;
        if (!(P && reentered)) v = sed_in[i];
        if (P) { const float lv = carry_sed[(i / B) * BM + (i % B)]; if (reentered) v = lv; }
        sed_s[(i / B) * BM + (i % B)] = v;
    }
    ks_fill_lengths(fl_s, tid, fl_req);
    if (tid < 2 * KM) (&pair_flag[0][0])[tid] = 0;
    if (P && tid == 0) { ctl_s[it_old & 1][0] = 1; ctl_s[it_old & 1][1] = 1; }
    if (P && !reentered && tid < KM * 4) (&rowrange_s[0][0][0])[tid] = (short)((tid & 1) ? H - 1 : 0);
    if (P && resident) {
        // (the previous iteration's last barrier ordered its tile writes before these reads)
#pragma unroll
        for (int j = 0; j < GPT; ++j) {
            const int g = tid + j * SC_FB2;
            const int y = g / gpr, x = (g - y * gpr) << 2;
#pragma unroll
            for (int k = 0; k < KM; ++k)
                mreg[j][k] = (g < ngroups && k < K) ? tl4(tiles + k * tile_floats + y * LW + x) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const bool small_side = (K <= B);          // nonzero spectrum of A^T A == that of A A^T
    __syncthreads();                           // sed_s visible
    {
        // packed-f32 products (v_pk_fma_f32): the .x lanes of gram2 sum pixels 0 and 2 of the groups,
        // the .y lanes pixels 1 and 3
        f32x2 gram2[NG];
#pragma unroll
        for (int i = 0; i < NG; ++i) gram2[i] = (f32x2){0.f, 0.f};
#pragma unroll
        for (int j = 0; j < GPT; ++j) {
            const int g = tid + j * SC_FB2;
            if (g < ngroups) {
                const int y = g / gpr, x = (g - y * gpr) << 2;
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (k < K && !(P && resident)) ts4(tiles + k * tile_floats + y * LW + x, mreg[j][k]);
                int gi = 0;
#pragma unroll
                for (int k = 0; k < KM; ++k)
#pragma unroll
                    for (int k2 = k; k2 < KM; ++k2) {
                        const float4 p = mreg[j][k], q = mreg[j][k2];
                        gram2[gi] += (f32x2){p.x, p.y} * (f32x2){q.x, q.y};
                        gram2[gi] += (f32x2){p.z, p.w} * (f32x2){q.z, q.w};
                        ++gi;
                    }
            }
        }
        float gram[NG];
#pragma unroll
        for (int i = 0; i < NG; ++i) gram[i] = gram2[i].x + gram2[i].y;
        // float partials per 16-lane row (float64 across the 4 x 8 rows): red[4 wid + row][gi], gi =
        // packed upper-triangle index over KM x KM; lane & 3 = q holds the row sums of the entries 4 m + q
        float gq[(NG + 3) / 4];
        wave_rowsum_quads(gram, gq);
        if ((lane & 12) == 12) {
#pragma unroll
            for (int m = 0; m < (NG + 3) / 4; ++m)
                if (4 * m + 3 < NG || 4 * m + (lane & 3) < NG) red[4 * wid + (lane >> 4)][4 * m + (lane & 3)] = gq[m];
        }
    }
#pragma unroll
    for (int j = 1; j < GPT; ++j) load_images(j);       // the tile registers are free now
    __syncthreads();
    STAMP(1);
    // Lipschitz constants (blend.py:205-218): L_sed = lambda_max(S S^T) on lane 0,
    // L_morph = lambda_max(A^T A) on lane 1 of wave 0 -- one instruction stream for both
    if (wid == 0) {
        {   // the 4 x 8 row partials of every Gram entry: lane = 4 entry + q adds eight of them, the four
            // lanes of a quad combine (K * K <= 16 entries on the 64 lanes)
            const int e = lane >> 2, q = lane & 3;
            const int k = e / K, k2 = e - k * K;
            const int lo = k < k2 ? k : k2, hi = k < k2 ? k2 : k;
            const int go = lo * KM - (lo * (lo - 1)) / 2 + (hi - lo);    // packed upper-triangle index (KM x KM)
            double r = 0;
            if (e < K * K) {
#pragma unroll
                for (int w = 0; w < SC_NW2; ++w) r += (double)red[q * SC_NW2 + w][go];
            }
            r += dpp_mov<SC_DPP_XOR1>(r);
            r += dpp_mov<SC_DPP_XOR2>(r);
            if (e < K * K && q == 0) mat[0][k * KM + k2] = r;
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
    {
        float sed[KM][BM];                     // wave-uniform: lives in SGPRs
#pragma unroll
        for (int k = 0; k < KM; ++k)
#pragma unroll
            for (int b = 0; b < BM; ++b)
                sed[k][b] = (k < K && b < B) ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(
                                                   __builtin_bit_cast(int, sed_s[k * BM + b]))) : 0.f;
        bool fixm[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) fixm[k] = (k < K) && a.fix_morph && a.fix_morph[(size_t)s * K + k];
        // two pixels per instruction (v_pk_fma_f32): p = 0 holds pixels (0, 1) of the float4
        // group, p = 1 pixels (2, 3); the SED gradient and loss accumulators stay paired
        f32x2 dsed2[KM][BM];
#pragma unroll
        for (int k = 0; k < KM; ++k)
#pragma unroll
            for (int b = 0; b < BM; ++b) dsed2[k][b] = (f32x2){0.f, 0.f};
        f32x2 loss2 = {0.f, 0.f};
        const f32x2 ws2 = {a.weight_scalar, a.weight_scalar};      // (exact instance: 1, the multiplications fold away)
#pragma unroll
        for (int j = 0; j < GPT; ++j) {
            const int g = tid + j * SC_FB2;
            if (g < ngroups) {
                const int y = g / gpr, x = (g - y * gpr) << 2;
                f32x2 m2[KM][2], gm2[KM][2];
#pragma unroll
                for (int k = 0; k < KM; ++k) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (k < K) v = tl4(tiles + k * tile_floats + y * LW + x);
                    m2[k][0] = (f32x2){v.x, v.y}; m2[k][1] = (f32x2){v.z, v.w};
                    gm2[k][0] = (f32x2){0.f, 0.f}; gm2[k][1] = gm2[k][0];
                }
#pragma unroll
                for (int b = 0; b < BM; ++b) {
                    if (b < B) {
                        const float4 iv = ireg[j][b];
                        f32x2 im2[2] = {{iv.x, iv.y}, {iv.z, iv.w}}, ww2[2] = {ws2, ws2};
                        if (wgt) {
                            const float4 wv = reinterpret_cast<const float4 *>(wgt + (size_t)b * HW)[g];
                            ww2[0] = (f32x2){wv.x, wv.y}; ww2[1] = (f32x2){wv.z, wv.w};
                        }
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            f32x2 model = sed[0][b] * m2[0][p];
#pragma unroll
                            for (int k = 1; k < KM; ++k) model += sed[k][b] * m2[k][p];
                            const f32x2 d = X ? model - im2[p] : ww2[p] * (model - im2[p]);
                            // explicit FMAs: the loss is reported, not fed back, so nothing else would expose a
                            // contraction choice that differs between two instances of this kernel
                            loss2 = (f32x2){__builtin_fmaf(d.x, d.x, loss2.x), __builtin_fmaf(d.y, d.y, loss2.y)};
                            const f32x2 gg = X ? d : ww2[p] * d;
#pragma unroll
                            for (int k = 0; k < KM; ++k) { dsed2[k][b] += gg * m2[k][p]; gm2[k][p] += sed[k][b] * gg; }
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (k < K && !fixm[k]) {
                        const f32x2 o0 = m2[k][0] - step_morph * gm2[k][0], o1 = m2[k][1] - step_morph * gm2[k][1];
                        ts4(tiles + k * tile_floats + y * LW + x, make_float4(o0.x, o0.y, o1.x, o1.y));
                    }
            }
        }
        float part[NP];
        part[0] = 0.5f * (loss2.x + loss2.y);
#pragma unroll
        for (int k = 0; k < KM; ++k)
#pragma unroll
            for (int b = 0; b < BM; ++b) part[1 + k * BM + b] = dsed2[k][b].x + dsed2[k][b].y;
        // float partials per 16-lane row (float64 across the 4 x 8 rows): red[4 wid + row][i], i = 0
        // (loss) or 1 + k * BM + b; lane & 3 = q holds the row sums of the entries 4 m + q
        float pq[(NP + 3) / 4];
        wave_rowsum_quads(part, pq);
        if ((lane & 12) == 12) {
#pragma unroll
            for (int m = 0; m < (NP + 3) / 4; ++m)
                if (4 * m + 3 < NP || 4 * m + (lane & 3) < NP) red[4 * wid + (lane >> 4)][4 * m + (lane & 3)] = pq[m];
        }
    }
    __syncthreads();
    STAMP(3);
    {   // the 4 x 8 row partials of the loss and of every SED-gradient entry: one 16-lane row per entry, two
        // partials per lane, the row combines (1 + K B <= 32 entries on the 512 threads)
        const int i = tid >> 4, p = tid & 15;
        const bool live = i < 1 + K * B;
        const int src = (!live || i == 0) ? 0 : 1 + ((i - 1) / B) * BM + (i - 1) % B;   // (k, b) of the K x B list in the KM x BM layout
        double r = live ? (double)red[p][src] + (double)red[p + 16][src] : 0.0;
        r += dpp_mov<SC_DPP_XOR1>(r);
        r += dpp_mov<SC_DPP_XOR2>(r);
        r += dpp_mov<SC_DPP_HALF_MIRROR>(r);
        r += dpp_mov<SC_DPP_MIRROR>(r);
        // the lane that holds an entry's total takes the SED step (blend.py:91-93) / stores the loss (blend.py:138)
        if (live && p == 0) {
            if (i == 0) {
                if (it_new <= a.mse_capacity) a.mse[(size_t)s * a.mse_capacity + it_new - 1] = r;
            } else {
                const int k = (i - 1) / B, b = (i - 1) - k * B;
                const float curv = sed_s[k * BM + b];
                const bool fixed = a.fix_sed && a.fix_sed[(size_t)s * K + k];
                sed_new[k * BM + b] = fixed ? curv : curv - step_sed * (float)r;
            }
        }
    }
    __syncthreads();
    STAMP(4);

    // ---------------- phase 2: constraints, one PAIR of waves (2k, 2k + 1) per component.  The two
    // waves sit on different SIMDs, so the MFMA work of a big window spreads over two matrix
    // pipes; the wave that also runs the sweep and the bookkeeping (`lead`) alternates so that the
    // four sweeps of a scene land on four different SIMDs.
    const int k = wid >> 1, half = wid & 1;
    const bool lead = half == ((k >> 1) & 1);
    // Synchronisation of the two waves of a pair WITHOUT the other components of the scene: each wave
    // publishes the phase it has completed (after a release fence on its LDS writes) and sleeps until
    // its partner has published the same phase.  The four components then run their chains
    // independently; the workgroup meets again at the end of phase 2.
    // LONG: the wait for the partner's sweep (~15k cycles): poll every 512 cycles instead of every 128 -- a poll
    // is four instructions taken from the SIMD's other waves
    auto pair_sync = [&](int phase, auto long_wait) {
        constexpr bool LONG = decltype(long_wait)::value;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&pair_flag[k & (KM - 1)][half], phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(&pair_flag[k & (KM - 1)][1 - half], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < phase) {
            if (LONG) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    const bool mine = k < K;
    const int c = s * K + (mine ? k : 0);
    Tile t; t.H = H; t.W = W; t.LW = LW; t.m = tiles + (mine ? k : 0) * tile_floats;
    float *vec = vecs + (mine ? k : 0) * SC_PAIR_VEC_FLOATS;
    float *zv = vec + 384 + SC_WAVE * half;
    int cy = 0, cx = 0, stat = 0;
    int mode = 0;                                   // 0: nothing, 1: k-space, 2: flip (no shift yet)
    SymWindow sw = {0, 0, 1, 1, true};
    KsGeom kg = {};
    float sy = 0.f;
    bool rank1 = false;
    if (mine) {
        cy = X ? pre_cy : a.centers[2 * c]; cx = X ? pre_cx : a.centers[2 * c + 1];
        wave_max_pixel(t, cy, cx, stat);            // both waves of the pair, identically
        cy = uniform(cy); cx = uniform(cx);
        if (symmetric) {
            double dy = X ? pre_dy : a.shifts[2 * c], dx = X ? pre_dx : a.shifts[2 * c + 1];
            if (it_new % 5 == 0) {
                // the window rows are split between the two waves (even / odd): half as many round trips to the
                // weight table in L2 on each wave's chain; both then add the two partial sums in the same
                // order and finish identically
                double p0, p1, p2;
                wave_centroid_sums(t, a.centroid_psf, a.centroid_P, cy, cx, half, 2, p0, p1, p2);
                if (lane == 0) { cent_s[k][half][0] = p0; cent_s[k][half][1] = p1; cent_s[k][half][2] = p2; }
                pair_sync(1, std::false_type{});                    // B0 (centroid iterations only)
                wave_centroid_finish(t, a.centroid_P, cent_s[k][0][0] + cent_s[k][1][0], cent_s[k][0][1] + cent_s[k][1][1],
                                     cent_s[k][0][2] + cent_s[k][1][2], cy, cx, dy, dx, stat);
                if (lead && lane == 0) { a.shifts[2 * c] = dy; a.shifts[2 * c + 1] = dx; }
            }
            cy = uniform(cy); cx = uniform(cx); dy = uniform(dy); dx = uniform(dx);
            if (P && lead && lane == 0) { carry_cen[k][0] = cy; carry_cen[k][1] = cx; carry_sh[k][0] = dy; carry_sh[k][1] = dx; }
            sw = sym_window(H, W, cy, cx);
            mode = (dy != dy) ? 2 : (sw.centered ? 0 : 1);
            if (mode == 1) {
                kg = ks_geom(sw, fl_s);
                const unsigned key[6] = {(unsigned)H << 24 | (unsigned)W << 16 | (unsigned)cy << 8 | (unsigned)cx, SC_KSC_MAGIC,
                                         (unsigned)__double2loint(dy), (unsigned)__double2hiint(dy),
                                         (unsigned)__double2loint(dx), (unsigned)__double2hiint(dx)};
                bool hit = kcache != nullptr;
#pragma unroll
                for (int i = 0; i < 6; ++i) hit = hit && (unsigned)__builtin_amdgcn_readlane((int)kc_hdr, i) == key[i];
                if (hit) {
                    const int q = lane + SC_WAVE * half;
                    vec[q] = kc_v[0]; vec[128 + q] = kc_v[1]; vec[256 + q] = kc_v[2];
                    sy = __int_as_float(__builtin_amdgcn_readlane((int)kc_hdr, 6));
                } else {
                    sy = pair_ks_vectors(kg, dy, dx, vec, half, kc_v);
                    if (kcache) {
                        kcache[lane] = kc_v[0]; kcache[64 + lane] = kc_v[1]; kcache[128 + lane] = kc_v[2];
                        unsigned word = __float_as_uint(sy);           // header word of this lane (selects: no private array)
#pragma unroll
                        for (int i = 0; i < 6; ++i) word = lane == i ? key[i] : word;
                        if (lane < 7) reinterpret_cast<unsigned *>(kcache)[192 + lane] = word;
                    }
                }
                rank1 = sy != 0.f;
                if (rank1) pair_ks_colsums(t, sw, kg, zv);
            }
        }
    }
    STAMP(8);
    if (mine) pair_sync(2, std::false_type{});                         // B1: Hankel vectors complete
    STAMP(12);
    f32x4 T[4][2];
    if (mine && mode == 1) {
        if (rank1) pair_ks_z(kg, vec + 256, zv);
        pair_ks_gemm1<true>(t, sw, kg, vec, half, T);
    }
    if (mine) pair_sync(3, std::false_type{});                         // B2: every read of X is done
    STAMP(13);
    if (mine && mode == 1) pair_ks_gemm2<true>(t, sw, kg, vec, zv, half, T, sy, rank1);
    if (mine && mode == 2 && lead) wave_flip_symmetry<float>(t, sw, false, 1.0f);
    if (mine) pair_sync(4, std::false_type{});                         // B3
    STAMP(9);
    // lane -> (row, float4 group) walk of the final pass without divisions: +128 groups per step
    const int dyq = (2 * SC_WAVE) / gpr, dxq = 2 * SC_WAVE - dyq * gpr;
    const int g0 = lane + SC_WAVE * half;
    const int y0 = g0 / gpr, x0 = g0 - y0 * gpr;
    // the previous morphology (buffer c0) for the convergence sums: requested now, all 16 B/lane
    // loads in flight together, so that the HBM latency is paid under the sweep
    // persistent form: rows of the previous morphology (buffer c0) and of the plane this iteration overwrites (buffer
    // 1 - c0) that can hold anything but zeros; read here, before the pair's last synchronisation (the lead rewrites
    // the second range after its final pass)
    int rl_lo = 0, rl_hi = H - 1, ro_lo = 0, ro_hi = H - 1;
    if (P && mine) {
        rl_lo = uniform((int)rowrange_s[k][c0][0]); rl_hi = uniform((int)rowrange_s[k][c0][1]);
        ro_lo = uniform((int)rowrange_s[k][1 - c0][0]); ro_hi = uniform((int)rowrange_s[k][1 - c0][1]);
    }
    float4 lastv[GPW];
    auto load_last = [&]() {
        const float4 *last4 = reinterpret_cast<const float4 *>(min_g + (size_t)k * HW);
#pragma unroll
        for (int j = 0; j < GPW; ++j) {
            const int g = g0 + j * 2 * SC_WAVE;
            // (persistent form: the previous morphology is zero outside its recorded rows -- not loaded)
            const int row = y0 + j * dyq;              // (dxq == 0 for the exact shape: a lane keeps its column group)
            const bool want = g < ngroups && (!P || (row >= rl_lo && row <= rl_hi));
            lastv[j] = want ? load4(reinterpret_cast<const float *>(last4 + g), true) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (mine && !lead) load_last();                 // the idle wave of the pair: before the barrier
    if (mine && lead) {
        int lstop = 1 << 30;                        // last sweep level computed (early exit)
        if (monotonic) wave_monotonic<float>(t, cy, cx, 0.f, &lstop);
        if (lane == 0) {
            lstop_s[k] = lstop; a.centers[2 * c] = cy; a.centers[2 * c + 1] = cx;
            // persistent form: the final pass writes the NORMALISED values back into the tile, the peak pixel among
            // them, while the partner wave may not have read the peak yet -- the lead hands it over with lstop
            if (P) nmax_s[k][0] = t.m[cy * LW + cx];
        }
        load_last();
    }
    if (mine) pair_sync(5, std::true_type{});                         // B4: sweep done, lstop published
    STAMP(10);
    // ---- sparsity, positivity (update.py:71-82, 27-32), normalisation (update.py:62-65),
    // store, convergence sums: one pass over the LDS tile, float4 groups split between the pair
    float norm = 0.f;
    float l0 = (!X && a.l0_thresh >= 0.f) ? a.l0_thresh * step_morph : -1.f;
    float l1 = (!X && a.l1_thresh >= 0.f) ? a.l1_thresh * step_morph : -1.f;
    auto sparse = [&](float v) {
        if (l0 >= 0.f && fabsf(v) < l0) v = 0.f;
        if (l1 >= 0.f) {
            const float mag = fabsf(v) - l1;
            v = (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f)) * (mag < 0.f ? 0.f : mag);
        }
        return v;
    };
    if (!monotonic) {
        float vmax = -INFINITY;
        bool anynan = false;
        if (mine) {
            int y = y0, xq = x0;
            for (int g = g0; g < ngroups; g += 2 * SC_WAVE) {
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
            vmax = wave_max(vmax);
            if (__any(anynan)) vmax = __builtin_nanf("");
            if (lane == 0) nmax_s[k][half] = vmax;
        }
        if (mine) pair_sync(6, std::false_type{});
        if (mine) {
            const float m0 = nmax_s[k][0], m1 = nmax_s[k][1];
            norm = (m0 != m0 || m1 != m1) ? __builtin_nanf("") : fmaxf(m0, m1);
        }
        l0 = -1.f; l1 = -1.f;                       // applied
    } else if (mine) {
        // after the sweep no pixel exceeds the peak pixel (each is capped by a convex
        // combination of pixels closer to the peak) and the maps above are monotone:
        // morph.max() is the processed peak value (a NaN elsewhere: see below)
        norm = sparse(P ? nmax_s[k][0] : t.m[cy * LW + cx]);
        if (norm < 0.f) norm = 0.f;
    }
    if (mine) {
        const int lstop = lstop_s[k];
        const bool cut = lstop < (1 << 30);
        const bool regular = norm > 0.f && !isinf(norm);             // else: the reference's 0/0, x/inf, NaN results
        const float rnorm = 1.0f / norm;
        float4 *out4 = reinterpret_cast<float4 *>(mout_g + (size_t)k * HW);
        f32x2 d2p = {0.f, 0.f}, n2p = {0.f, 0.f};        // <= 32 float terms per lane, then f64 across lanes
        const f32x2 rn2 = {rnorm, rnorm}, nm2 = {norm, norm};
        // CUT: zero beyond the sweep's last level; GEN: thresholds and/or an irregular norm
        // persistent form: only the rows where the new morphology (inside the sweep's cut), the previous one or the
        // old content of the output plane can be non-zero; everywhere else all three are exact zeros: nothing to
        // store, nothing to add to the sums
        int u_lo = 0, u_hi = H - 1;
        if (P && cut && regular && l0 < 0.f && l1 < 0.f) {
            const int rt = lstop >> 1;                                 // rows beyond cy +- lstop / 2 are cut whole
            u_lo = min(min(rl_lo, ro_lo), max(cy - rt, 0)); u_hi = max(max(rl_hi, ro_hi), min(cy + rt, H - 1));
        }
        auto final_pass = [&](auto cut_c, auto gen_c) {
            constexpr bool CUT = decltype(cut_c)::value, GEN = decltype(gen_c)::value;
            int y = y0, xq = x0;
#pragma unroll
            for (int j = 0; j < GPW; ++j) {
                const int g = g0 + j * 2 * SC_WAVE;
                // (the wave's four rows of this trip: wave-uniform test, exact shape only)
                const int yw = uniform(y0 + j * dyq) & ~3;
                const bool skip = P && X && (yw + 3 < u_lo || yw > u_hi);
                // (the tile itself must read zero there for the next iteration's phase 0: it holds stepped values)
                if (g < ngroups && skip) ts4(t.m + y * LW + (xq << 2), make_float4(0.f, 0.f, 0.f, 0.f));
                if (g < ngroups && !skip) {
                    const float4 v4 = tl4(t.m + y * LW + (xq << 2));
                    const float4 l = lastv[j];
                    float v[4] = {v4.x, v4.y, v4.z, v4.w};
                    // level(x, y) = 2 max(ax, ay) + min(ax, ay) <= lstop  <=>  ax <= axmax(ay): one
                    // bound per row instead of a level per pixel
                    int axmax = 0, xb = 0;
                    if (CUT) {
                        const int ay = y < cy ? cy - y : y - cy;
                        const int h1 = (lstop - ay) >> 1;                      // from 2 ax + ay <= lstop (ax >= ay)
                        axmax = h1 >= ay ? h1 : lstop - 2 * ay;                // else from 2 ay + ax <= lstop
                        xb = (xq << 2) - cx + axmax;                           // x - cx + axmax in [0, 2 axmax] <=> |x - cx| <= axmax
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (GEN) v[e] = sparse(v[e]);
                        // positivity and the sweep's cut in one select (NaN inside the cut stays NaN)
                        const bool beyond = CUT && (axmax < 0 || (unsigned)(xb + e) > (unsigned)(2 * axmax));
                        v[e] = (v[e] < 0.f || beyond) ? 0.f : v[e];
                    }
                    f32x2 o01, o23;
                    if (GEN && !regular) {
                        o01 = (f32x2){v[0] / norm, v[1] / norm}; o23 = (f32x2){v[2] / norm, v[3] / norm};
                    } else {
                        // v / norm, correctly rounded (but for rare double roundings): one Newton step
                        // on v * (1 / norm); two pixels per instruction
                        const f32x2 a01 = {v[0], v[1]}, a23 = {v[2], v[3]};
                        const f32x2 q01 = a01 * rn2, q23 = a23 * rn2;
                        o01 = (a01 - q01 * nm2) * rn2 + q01;
                        o23 = (a23 - q23 * nm2) * rn2 + q23;
                    }
                    const float4 o4 = make_float4(o01.x, o01.y, o23.x, o23.y);
                    out4[g] = o4;
                    if (P) ts4(t.m + y * LW + (xq << 2), o4);     // the next iteration's phase 0 reads the tile
                    const f32x2 e01 = (f32x2){l.x, l.y} - o01, e23 = (f32x2){l.z, l.w} - o23;
                    d2p += e01 * e01; d2p += e23 * e23;
                    n2p += o01 * o01; n2p += o23 * o23;
                }
                y += dyq; xq += dxq;
                if (xq >= gpr) { xq -= gpr; ++y; }
                if (j & 1) __builtin_amdgcn_sched_barrier(0);    // two groups in flight (VGPRs)
            }
        };
        using std::true_type; using std::false_type;
        if (regular && l0 < 0.f && l1 < 0.f) {
            if (cut) final_pass(true_type{}, false_type{}); else final_pass(false_type{}, false_type{});
        } else {
            final_pass(true_type{}, true_type{});
        }
        const double d2 = wave_sum((double)(d2p.x + d2p.y)), n2 = wave_sum((double)(n2p.x + n2p.y));
        if (lane == 0) { conv_m[k][half][0] = d2; conv_m[k][half][1] = n2; }
        if (P && lead && lane == 0) {                   // the rows of the plane just written that are not all zero
            const bool known = cut && regular && l0 < 0.f && l1 < 0.f;
            rowrange_s[k][1 - c0][0] = (short)(known ? max(cy - (lstop >> 1), 0) : 0);
            rowrange_s[k][1 - c0][1] = (short)(known ? min(cy + (lstop >> 1), H - 1) : H - 1);
        }
        if (lead) {
            double d2s = 0, n2s = 0;
            if (lane < B) {
                float v = sed_new[k * BM + lane];
                if (v < 0.f) v = 0.f;
                v = v * norm;
                sed_out[k * B + lane] = v;
                if (P) carry_sed[k * BM + lane] = v;
                const float d = sed_s[k * BM + lane] - v;
                d2s = (double)(d * d);
                n2s = (double)(v * v);
            }
            d2s = wave_sum(d2s); n2s = wave_sum(n2s);
            if (lane == 0) { conv_s[k][0] = d2s; conv_s[k][1] = n2s; }
        }
    }
    __syncthreads();                                // B5
    if (mine && norm == norm) {
        // a NaN pixel away from the peak (NaN sums with a non-NaN norm): np.max is NaN and the
        // reference's morph becomes NaN everywhere
        const double n2 = conv_m[k][0][1] + conv_m[k][1][1];
        if (n2 != n2) {
            norm = __builtin_nanf("");
            float4 *out4 = reinterpret_cast<float4 *>(mout_g + (size_t)k * HW);
            const float4 nan4 = make_float4(norm, norm, norm, norm);
            for (int g = g0; g < ngroups; g += 2 * SC_WAVE) out4[g] = nan4;
            if (lead && lane < B) { sed_out[k * B + lane] = norm; if (P) carry_sed[k * BM + lane] = norm; }
            if (P && lane == 0) {
                ctl_s[it_old & 1][1] = 0;                        // the tile does not hold this result: reload from HBM
                rowrange_s[k][1 - c0][0] = 0; rowrange_s[k][1 - c0][1] = (short)(H - 1);
            }
        }
    }
    if (mine && lead && lane == 0) {
        if (!(norm > 0.f) || isinf(norm)) stat |= SCARLET_STATUS_NONFINITE;
        if (stat) atomicOr(&a.status[s], stat);
    }
    STAMP(5);

    // ---------------- phase 3: Blend._check_convergence + bookkeeping (blend.py:141-184): lane k of wave 0 tests
    // component k (one lane for all of them was ~2.6k cycles at the end of every scene's chain)
    if (wid == 0) {
        bool pending = false;
        if (lane < K && it_new > 1) {
            // the two convergence bits are rewritten whatever they were, the other bits stay: posted
            // atomics instead of load - modify - store (a load here would put an HBM round trip at the
            // very end of the workgroup, with its LDS and registers still held)
            const int kk = lane;
            int set = 0;
            const double d2 = conv_m[kk][0][0] + conv_m[kk][1][0], n2 = conv_m[kk][0][1] + conv_m[kk][1][1];
            if (!(n2 == n2 && conv_s[kk][0] <= a.e_rel2 * conv_s[kk][1])) set |= SCARLET_FLAG_SED_NOT_CONVERGED;
            if (!(d2 <= a.e_rel2 * n2)) set |= SCARLET_FLAG_MORPH_NOT_CONVERGED;
            const int clear = (SCARLET_FLAG_SED_NOT_CONVERGED | SCARLET_FLAG_MORPH_NOT_CONVERGED) & ~set;
            if (clear) atomicAnd(&a.flags[s * K + kk], ~clear);
            if (set) atomicOr(&a.flags[s * K + kk], set);
            pending = set != 0;
        }
        const bool done = it_new > 1 && __builtin_amdgcn_ballot_w64(pending) == 0;
        if (lane == 0) {
            a.it[s] = it_new;
            a.cur[s] = 1 - c0;
            if (done) { a.active[s] = 0; if (P) ctl_s[it_old & 1][0] = 0; }
        }
    }
    STAMP(6);
    if (a.stamps && tid == 0) a.stamps[(size_t)s * 16 + 11] = (long long)__builtin_amdgcn_s_memrealtime();
#undef STAMP
    if (P) {
        // (no wait for this iteration's global stores here: what the next iteration needs of them is in LDS, the
        // previous morphology is re-read tens of thousands of cycles from now and the cached Hankel vectors by the
        // wave that stored them.  The wait cost ~14k cycles per iteration -- the drain of the final pass's 64 KB.)
        __syncthreads();
        // (by parity: a wave that is already in the next iteration re-arms the OTHER pair of words)
        return uniform(ctl_s[it_old & 1][0] | (ctl_s[it_old & 1][1] << 1));
    }
    return 0;
}

template <int KM, int BM, int XS = 0>
__global__ __launch_bounds__(SC_FB2, 4) void k_iterate2(FusedArgs a)
{
    const int s = blockIdx.x;
    // the three per-scene words are requested together (one scalar round trip, not three in a row), and both
    // buffer pointers come from fixed kernel-argument offsets (an index into a.morph[] would be a fourth)
    const int active_s = a.active[s], c0 = a.cur[s], it_old = a.it[s];
    asm volatile("" ::"s"(c0), "s"(it_old));     // (keeps the two loads above the branch: the compiler sinks them otherwise)
    if (!active_s) return;
    iterate2_body<KM, BM, XS, false>(a, s, c0, it_old, false);
}

// ---- k_fit2x: SEVERAL iterations of a scene in one launch (the loop of Blend.fit, blend.py:79-102, per workgroup),
// for the exact-shape instance <4, 5, 64>.
//
// A workgroup keeps its scene for up to n_iter iterations: the morphologies stay in the LDS tiles from one
// iteration's final pass to the next iteration's gradient step (HBM sees them written once per iteration -- the
// other buffer is the reference's _last_morph -- and read once, for the convergence sums), converged scenes leave
// the loop (the ragged stop of _check_convergence), and there is no launch, no workgroup start-up and no grid tail
// per iteration.
//
// The loop is NOT a loop the compiler sees.  Written as one (round 2, profiles/r02_notes.md) it hoists the
// iteration's per-thread invariants out of the loop and spills them (0.85 - 1.02 ms against 0.72); with the
// iteration as a separately allocated function (`noinline`) the calling convention saves and restores 32
// callee-saved VGPRs per wave and call through scratch memory -- 128 KB of extra traffic per scene-iteration.
// Instead the kernel is the straight-line iteration (register-allocated exactly like k_iterate2) and, while
// iterations remain, its last instruction is a jump back to its own first instruction with the registers a fresh
// wave starts with re-created: kernel-argument pointer, workgroup id and work-item id, plus the loop state in the
// two registers a 1-D grid leaves at zero (workgroup id y = {re-entered, tiles resident, buffer index, iterations
// left}, workgroup id z = iteration count).  The same jump, with those two registers at zero and another workgroup
// id x, starts the NEXT SCENE of the launch's queue on this workgroup.  LDS is untouched by the jump.  The register assignment this relies
// on (user SGPRs = kernel-argument pointer only, workgroup ids x / y / z in s2 / s3 / s4, packed work-item id in
// v0, no private segment) is the HSA ABI for the kernel descriptor the compiler emits; tools/check_reentry_abi.py
// verifies those descriptor fields on every build and the library refuses to launch this kernel otherwise.
#define SC_FIT2X_KERNEL k_fit2x
// re-creates the registers a fresh wave of this kernel starts with and jumps to the kernel's first instruction
// (never returns); `scene` / `state` / `it` arrive as workgroup ids x / y / z
#define SC_FIT2X_REENTER(scene, state, it)                                                                     \
    asm volatile("s_mov_b64 exec, -1\n\t"                                                                      \
                 "s_mov_b64 s[0:1], %0\n\t"                                                                    \
                 "s_mov_b32 s2, %1\n\t"                                                                        \
                 "s_mov_b32 s3, %2\n\t"                                                                        \
                 "s_mov_b32 s4, %3\n\t"                                                                        \
                 "v_mov_b32 v0, %4\n\t"                                                                        \
                 "s_getpc_b64 s[6:7]\n\t"                                                                      \
                 "s_add_u32 s6, s6, k_fit2x@rel32@lo+4\n\t"                                                    \
                 "s_addc_u32 s7, s7, k_fit2x@rel32@hi+12\n\t"                                                  \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                                    \
                 "s_setpc_b64 s[6:7]"                                                                          \
                 :                                                                                             \
                 : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "s"(scene), "s"(state), "s"(it), "v"((int)threadIdx.x) \
                 : "s0", "s1", "s2", "s3", "s4", "s6", "s7", "v0", "scc", "memory")
extern "C" __global__ __launch_bounds__(SC_FB2, 4) void SC_FIT2X_KERNEL(FusedArgs a, int n_iter, int *queue, int n_workgroups)
{
    const int s = blockIdx.x;                                            // the scene (workgroup id x, rewritten on re-entry)
    const unsigned st = (unsigned)__builtin_amdgcn_workgroup_id_y();     // 0 at launch (1-D grid) and for a new scene
    int c0 = 0, it = 0, left = 0;
    bool resident = false, run = true;
    if (st >> 31) {
        resident = ((st >> 30) & 1) && !(n_iter & (1 << 30));       // (bit 30 of n_iter: diagnostic, tiles reloaded from HBM)
        c0 = (int)((st >> 29) & 1); left = (int)(st & 0x1fffffffu);
        it = (int)__builtin_amdgcn_workgroup_id_z();
    } else {
        const int active_s = a.active[s];
        c0 = a.cur[s]; it = a.it[s];
        asm volatile("" ::"s"(c0), "s"(it));
        left = n_iter & 0xffffff;
        run = active_s && left > 0;
    }
    if (run) {
        const int r = iterate2_body<4, 5, 64, true>(a, s, c0, it, resident, (st >> 31) != 0);
        if ((r & 1) && left > 1) {
            const unsigned nst = 0x80000000u | ((unsigned)(r >> 1) << 30) | ((unsigned)(c0 ^ 1) << 29) | (unsigned)(left - 1);
            if (!(r & 2) || (n_iter & (1 << 30))) {
                // the next iteration reloads its tiles from the planes this one has just stored (a NaN result, or the
                // diagnostic switch): every wave's stores first
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            // (nothing the next iteration loads from memory through a cache was stored by this one: see iterate2_body)
            SC_FIT2X_REENTER(s, nst, it + 1);
        }
    }
    // This scene is finished for this launch (iterations done, converged, or inactive from the start): take the next
    // one from the launch's queue.  The grid holds only as many workgroups as the chip keeps resident (n_workgroups);
    // scenes beyond the first n_workgroups are handed out by a counter.  Leaving that to the hardware dispatcher
    // -- one workgroup per scene -- left a CU's second slot empty for ~70 us (median 34, p90 206) after every workgroup
    // of 1.6 ms: the dispatcher deals workgroups to CUs in a fixed order and waits for THAT CU (measured with
    // tools/occupancy.py: 1.84 resident workgroups per CU instead of 2).
    __shared__ int next_s;
    if (threadIdx.x == 0) next_s = queue ? atomicAdd(queue, 1) + n_workgroups : a.S;
    __syncthreads();
    const int nx = uniform(next_s);
    __syncthreads();                                                    // (next_s is rewritten by whoever gets here first next time)
    if (nx < a.S) SC_FIT2X_REENTER(nx, 0u, 0);
}
