// boxupdate.h -- k_source_update_box: the constraint pipeline of PointSource/ExtendedSource.update
// (source.py:402-440) for frames beyond the wave-level tile (H or W > 64, up to 256 x 256), computed ONLY
// WHERE ITS RESULT CAN BE NON-ZERO.
//
// With the default pipeline (symmetry -> radial monotonicity -> positivity) the weighted sweep ends as soon
// as three consecutive levels hold no positive value (wave_ops.h): every pixel beyond that level is capped
// by non-positive neighbours and positivity zeroes it.  A sweep that ends by level 2 R = 62, i.e. within
// SC_UB_R = 31 pixels of the peak -- the usual case: a source's footprint plus the few pixels the noise needs
// to pull the envelope below zero -- therefore needs the symmetrised morphology only inside the 63 x 63 box
// around the peak (a box of half-size R holds every pixel of the levels <= 2 R).
// The k-space symmetry out = X/2 + (A X B + s (sigma sigma^T X) C)/2 (prox_ops.h) restricted to output
// rows / columns of the box is two much smaller GEMMs,
//     T = X B[:, box]   (h x w x 64)        Y = A[box, :] T   (64 x h x 64),
// ~2.7 x fewer MFMAs than the full h x w x (h + w) products on a 128 x 128 window and ~6 x on 256 x 256, and
// neither the frame-sized tile nor the frame-sized scratch has to live in LDS / HBM: X streams through LDS
// once in bands of rows.  (The k sums run in a different order than in kspace_symmetry_tile: equal to float32
// rounding, not bit for bit.)
//
// A component whose sweep does NOT end within the box (or the rare centred soft-symmetry window) is left
// untouched and flagged in `fallback`; k_source_update<MODE> then runs for the flagged components only.
#pragma once
#include "engine.h"

// Two box sizes: R = 31 (63 x 63, complete for the sweep levels <= 62, swept by ONE wave without barriers) for
// every component, and R = 63 (127 x 127, levels <= 126, compact levels on one wave and the rest level-synchronously
// on the workgroup) for the components the first size flagged; only what the second one flags too goes to the
// full-frame kernel.
#define SC_UB_BR 16                                  // rows of X per band in LDS
__host__ __device__ constexpr int ub_lw(int R) { return 2 * R + 3; }
__host__ __device__ constexpr int ub_floats(int R) { return (2 * R + 1) * (2 * R + 3); }
__host__ __device__ constexpr int ub_n(int R) { return 2 * R + 2; }       // padded box side: 64 / 128 = 4 / 8 MFMA tiles
// T = X B[:, box] never leaves the registers: wave w owns the 16 box columns 16 w .. 16 w + 15 of T for ALL
// window rows (one MFMA accumulator per band of 16 rows), and the accumulator layout -- lane (lr, lq), element
// r holds T[16 band + 4 lq + r][16 w + lr] -- is exactly a B operand of the second product when its k steps
// enumerate (band, r): k(lq) = 16 band + 4 lq + r.  The Hankel operand A[i][k] = av[i + k] is read at that k.
// (A permuted summation order over k in both products: T is accumulated in two interleaved chains.)
// Measured alternatives (profiles/r02_notes.md): T in LDS (41 KB: two or three workgroups per CU instead of
// four) 1.1 - 1.6 x slower; bands of 32 rows with two accumulation chains per wave spill at 128 VGPRs, slower.

__host__ __device__ inline size_t ub_lds_floats(int H, int W, int R)
{
    const int hp = round16(H), wp = round16(W);
    const size_t stage = (size_t)SC_UB_BR * tile_stride(wp);          // the band of X and the box share their space
    return (stage > (size_t)ub_floats(R) ? stage : (size_t)ub_floats(R)) + 2 * hp + 4 * wp + wp + ub_n(R);
}

// NB: bands of 16 window rows the kernel is built for (8: frames up to 128 rows, 16: up to 256); R: box half-size.
// ub_component: the pipeline for component `c` by the calling workgroup (every return is uniform over it).
// `list` / `count` != NULL (the small box): a component whose sweep leaves the box is appended to the list for the
// large-box kernel; without them it is only flagged in `fallback` (the full-frame kernel runs for those).
// XS: 0 = frame shape from the arguments; 128 / 256 = a square frame of that size as compile-time constants (BASELINE
// configs 3 and 5: the address arithmetic folds, as in the headline kernel's exact-shape instance)
template <int NB, int R, int XS>
__device__ __forceinline__ void ub_component(const UpdateArgs &a, const int c, int *fallback, int *list, int *count, long long *stamps_all)
{
    constexpr int SC_UB_R = R, SC_UB_LW = ub_lw(R), SC_UB_FLOATS = ub_floats(R), SC_UB_N = ub_n(R), SC_UB_NT = SC_UB_N / 16;
    constexpr int CTW = SC_UB_NT / SC_NWAVES;             // column tiles of T per wave (1 or 2)
    extern __shared__ __align__(16) float lds[];
    const int s = c / a.K;
    if (!a.force_it0 && !a.active[s]) return;
    long long *stamps = stamps_all ? stamps_all + (size_t)c * 16 : nullptr;
#define UB_STAMP(i) do { if (stamps && threadIdx.x == 0) stamps[(i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    UB_STAMP(0);
    const int H = XS ? XS : a.H, W = XS ? XS : a.W, HW = H * W, B = a.B;
    const int hpF = round16(H), wpF = round16(W);
    const int BR = SC_UB_BR, SW = tile_stride(wpF);
    float *stage = lds;                                   // [BR][SW]   band of X rows (GEMM 1) ...
    float *box = lds;                                     // [63][65]   ... then the box around the peak
    const size_t stage_floats_ = (size_t)BR * SW;
    float *av = lds + (stage_floats_ > SC_UB_FLOATS ? stage_floats_ : (size_t)SC_UB_FLOATS), *bv = av + 2 * hpF, *cv = bv + 2 * wpF, *zv = cv + 2 * wpF, *vsum = zv + wpF;
    __shared__ double red[SC_NWAVES];
    __shared__ int ctr[2];
    __shared__ double shf[2];
    __shared__ int stat;
    __shared__ int hyb[2];
    __shared__ int lastpos;
    const int c0 = a.cur[s];
    const int wbuf = a.in_iteration ? 1 - c0 : c0;
    float *gm = a.morph[wbuf] + (size_t)c * HW;
    Tile tg; tg.H = H; tg.W = W; tg.LW = W; tg.m = gm;    // the stepped morphology, in place in HBM / L2
    if (threadIdx.x == 0) stat = 0;
    __syncthreads();
    const int it = a.force_it0 ? 0 : a.it[s] + (a.in_iteration ? 1 : 0);
    int cy = a.centers[2 * c], cx = a.centers[2 * c + 1];
    const bool grouped = a.group && a.group[c] >= 0;      // layer of a MultiComponentSource: centre from k_group_centers, shift = None
    if (!grouped) {
        if (threadIdx.x < SC_WAVE) {                                         // source.py:414 (25 lanes, one load each)
            int st = 0;
            wave_max_pixel(tg, cy, cx, st);
            if (threadIdx.x == 0) { ctr[0] = cy; ctr[1] = cx; if (st) stat |= st; }
        }
        __syncthreads();
        cy = ctr[0]; cx = ctr[1];
    }
    double dy = grouped ? (double)__builtin_nanf("") : a.shifts[2 * c], dx = grouped ? dy : a.shifts[2 * c + 1];
    bool new_shift = false;
    if (!grouped && a.symmetric && it % 5 == 0) {                            // source.py:428-429
        __syncthreads();
        centroid_tile(tg, a.centroid_psf, a.centroid_P, cy, cx, red, ctr, shf, &stat);
        cy = ctr[0]; cx = ctr[1]; dy = shf[0]; dx = shf[1];
        new_shift = true;
    }
    UB_STAMP(1);
    const SymWindow sw = sym_window(H, W, cy, cx);
    // 0: no symmetry, 1: k-space, 2: soft flip (no shift yet)
    const int mode = !a.symmetric ? 0 : ((dy != dy) ? 2 : (sw.centered ? 0 : 1));
    if (mode == 2 && sw.centered) {                       // the flip about the array middle leaves the box: full path
        if (threadIdx.x == 0) fallback[c] = 1;
        return;
    }
    // the box, clipped to the frame (frame coordinates) ...
    const int by0 = max(0, cy - SC_UB_R), bx0 = max(0, cx - SC_UB_R);
    const int bh = min(H, cy + SC_UB_R + 1) - by0, bw = min(W, cx + SC_UB_R + 1) - bx0;
    // ... and its part inside the symmetry window (window coordinates): rows [ia, ia + nbh), columns [ja, ja + nbw)
    const int ia = max(by0 - sw.y0, 0), ja = max(bx0 - sw.x0, 0);
    const int nbh = min(sw.h, by0 + bh - sw.y0) - ia, nbw = min(sw.w, bx0 + bw - sw.x0) - ja;
    // global loads are requested as early as their addresses are known and consumed late: the box (X inside it,
    // 16 values per thread) and the first band of window rows now, under the float64 trigonometry of the vectors
    // (the large box is read when it is stored instead: 64 registers per thread are not free)
    constexpr bool BOX_REGS = R <= 31;
    // box pixel of a thread's value j: column threadIdx.x % BXC, row j * BXR + threadIdx.x / BXC (no index division)
    constexpr int BXC = R <= 31 ? 64 : 128, BXR = SC_BLOCK / BXC, NBXJ = (2 * R + 1 + BXR - 1) / BXR;    // 64 x 4 x 16, 128 x 2 x 64
    constexpr int NBX = BOX_REGS ? NBXJ : 1;
    const int bxx = threadIdx.x & (BXC - 1), bxy = threadIdx.x / BXC;
    float bxr[NBX];
    if (BOX_REGS) {
#pragma unroll
        for (int j = 0; j < NBX; ++j) {
            const int y = j * BXR + bxy;
            bxr[j] = (y < bh && bxx < bw) ? gm[(by0 + y) * W + bx0 + bxx] : 0.f;
        }
    }
    Tile bt; bt.H = bh; bt.W = bw; bt.LW = SC_UB_LW; bt.m = box;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lr = lane & 15, lq = lane >> 4;
    const int hW = sw.h, wW = sw.w, wpW = round16(wW);
    // band of BR window rows x wp columns: value j of a thread is row j (NB = 16: the 256 threads span the columns) or
    // row 2 j + threadIdx.x / 128 (NB = 8: two rows of up to 128 columns per j) -- no index division, a row's lanes
    // read consecutive addresses (the first form, element e = threadIdx.x + j * 256 -> (e / wp, e % wp), spent ~20
    // instructions per value on the division, twice: load and LDS store).
    // values per thread and band: the NB = 8 instance serves frames up to 128 x 128 (wp <= 128: 8 values) and keeps
    // TWO bands in flight -- with one, each band of the first product waited a full HBM round trip (~6 k cycles x 8
    // bands, measured) --, the NB = 16 instance (wp <= 256: 16 values) one
    constexpr int NBAND = NB == 8 ? 8 : 16, NPF = NB == 8 ? 2 : 1;
    float xr[NPF][NBAND];
    constexpr int BCOLS = NB == 8 ? 128 : 256, BRPJ = SC_BLOCK / BCOLS;            // columns and rows covered per value index
    static_assert(NBAND * BRPJ == SC_UB_BR, "a band is NBAND values per thread");
    const int bcc = threadIdx.x & (BCOLS - 1), brr = threadIdx.x / BCOLS;
    auto load_band = [&](int i0, int slot) {
#pragma unroll
        for (int j = 0; j < NBAND; ++j) {
            const int r = j * BRPJ + brr, cc = bcc;
            const float v = (i0 + r < hW && cc < wW) ? gm[(sw.y0 + i0 + r) * W + sw.x0 + cc] : 0.f;
#pragma unroll
            for (int q = 0; q < NPF; ++q) if (q == slot) xr[q][j] = v;
        }
    };
    if (mode == 1) {
        load_band(0, 0);
        if (NPF > 1) load_band(SC_UB_BR, 1);
    }
    auto store_box = [&]() {
        if (BOX_REGS) {
#pragma unroll
            for (int j = 0; j < NBX; ++j) {
                const int y = j * BXR + bxy;
                if (y < bh && bxx < bw) box[y * SC_UB_LW + bxx] = bxr[j];
            }
        } else {
#pragma unroll 8
            for (int j = 0; j < NBXJ; ++j) {
                const int y = j * BXR + bxy;
                if (y < bh && bxx < bw) box[y * SC_UB_LW + bxx] = gm[(by0 + y) * W + bx0 + bxx];
            }
        }
    };
    if (mode == 1) {
        const int h = sw.h, w = sw.w, ry = h / 2, rx = w / 2;
        const int hp = round16(h), wp = round16(w);
        const int Fy = dev_next_fast_len(2 * h + 10);
        int Fx = dev_next_fast_len(2 * w + 10);
        while (Fx & 1) Fx = dev_next_fast_len(Fx + 1);
        kspace_vectors(av, bv, cv, hp, wp, ry, rx, h, w, Fy, Fx, dy, dx);
        const float sy = (Fy & 1) ? 0.f : (float)(sinpi(2.0 * dy) / Fy);
        const bool need_rank1 = (sy != 0.f);
        float vloc = 0.f;
        UB_STAMP(2);
        // GEMM 1: T[:, box columns] = X (h x w, zero outside the window) . Hankel(bv); X streams through `stage`,
        // wave `wid` accumulates the box columns 16 wid .. 16 wid + 15 for every band (registers)
        f32x4 Tacc[NB][CTW];
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
            const int i0 = bi * SC_UB_BR;
#pragma unroll
            for (int q = 0; q < CTW; ++q) Tacc[bi][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (i0 < hp) {
                __syncthreads();                                   // vectors ready / previous band consumed
#pragma unroll
                for (int j = 0; j < NBAND; ++j) {
                    const int r = j * BRPJ + brr;
                    if (bcc < wp) stage[r * SW + bcc] = xr[bi % NPF][j];
                }
                __syncthreads();
                if (i0 + NPF * BR < hp) load_band(i0 + NPF * BR, bi % NPF);     // NPF bands ahead: in flight under the MFMAs
                if (need_rank1 && threadIdx.x < w) {               // v[j] = sum_i (-1)^(i - ry) X[i][j], i ascending
                    const int rows = min(BR, h - i0);
                    for (int r = 0; r < rows; ++r) {
                        const float x = stage[r * SW + threadIdx.x];
                        vloc += ((i0 + r - ry) & 1) ? -x : x;
                    }
                }
                // two accumulation chains per tile (k steps 0, 2, 4 .. and 1, 3, 5 ..), added at the end: a dependent
                // MFMA waits ~40 cycles for its accumulator, two interleaved chains hide half of it
                f32x4 acc[CTW], acc2[CTW];
#pragma unroll
                for (int q = 0; q < CTW; ++q) { acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[q] = acc[q]; }
                const float *arow = stage + lr * SW;
#pragma unroll 2
                for (int k0 = 0; k0 < wp; k0 += 8) {
                    const int k = k0 + lq;
                    const float xa = arow[k], xb = arow[k + 4];
#pragma unroll
                    for (int q = 0; q < CTW; ++q) {
                        const int jc = ja + (wid + SC_NWAVES * q) * 16 + lr;
                        acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, bv[min(k + jc, 2 * wp - 1)], acc[q], 0, 0, 0);
                        acc2[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb, bv[min(k + 4 + jc, 2 * wp - 1)], acc2[q], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int q = 0; q < CTW; ++q) Tacc[bi][q] = acc[q] + acc2[q];
            }
        }
        if (need_rank1 && threadIdx.x < wp) zv[threadIdx.x] = threadIdx.x < w ? vloc : 0.f;
        __syncthreads();                                           // T and v complete; the last band is consumed
        store_box();                                               // (the box takes the band's place; read from GEMM 2's epilogue on)
        UB_STAMP(3);
        if (need_rank1 && (int)threadIdx.x < SC_UB_N) {            // z[j] = sum_j2 C[j][j2] v[j2] for the box columns
            const int j = ja + threadIdx.x;
            float zloc = 0.f;
            if (j < w)
                for (int j2 = 0; j2 < w; ++j2) zloc += cv[j + j2] * zv[j2];
            vsum[threadIdx.x] = zloc;
        }
        __syncthreads();
        UB_STAMP(4);
        // GEMM 2: Y[box rows, box columns] = Hankel(av)[box rows, :] . T ; epilogue combines with X in the box.
        // Wave `wid`: its column tile of T from the accumulators, the four row tiles of the box.
        // (row tiles in groups of four: the accumulators of a group, 4 x CTW, stay in registers)
#pragma unroll
        for (int rg = 0; rg < SC_UB_NT; rg += 4) {
            f32x4 Y[4][CTW];
#pragma unroll
            for (int rt2 = 0; rt2 < 4; ++rt2)
#pragma unroll
                for (int q = 0; q < CTW; ++q) Y[rt2][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (rg * 16 < nbh) {
#pragma unroll
                for (int bi = 0; bi < NB; ++bi)
                    if (bi * 16 < hp) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int k = bi * 16 + 4 * lq + r;
#pragma unroll
                            for (int rt2 = 0; rt2 < 4; ++rt2) {
                                const float aa = av[min(ia + (rg + rt2) * 16 + lr + k, 2 * hp - 1)];
#pragma unroll
                                for (int q = 0; q < CTW; ++q)
                                    Y[rt2][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa, Tacc[bi][q][r], Y[rt2][q], 0, 0, 0);
                            }
                        }
                    }
            }
#pragma unroll
            for (int rt2 = 0; rt2 < 4; ++rt2)
#pragma unroll
                for (int q = 0; q < CTW; ++q) {
                    const int ct2 = wid + SC_NWAVES * q, rt = rg + rt2;
                    if (rt * 16 >= nbh || ct2 * 16 >= nbw) continue;
                    const f32x4 acc = Y[rt2][q];
                    const int jl = ct2 * 16 + lr, j = ja + jl;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int il = rt * 16 + lq * 4 + r, i = ia + il;
                        if (il < nbh && jl < nbw) {
                            float *p = &box[(sw.y0 + i - by0) * SC_UB_LW + (sw.x0 + j - bx0)];
                            const float x = *p;
                            float y2 = acc[r];
                            if (need_rank1) y2 += (((i - ry) & 1) ? -sy : sy) * vsum[jl];
                            *p = (x <= 0.f) ? 0.f : 0.5f * x + 0.5f * y2;
                        }
                    }
                }
        }
        __syncthreads();
    } else if (mode == 2) {
        store_box();
        // soft symmetry, strength 1 (operator.py:242-251) on the part of the window inside the box: a pixel and
        // its point reflection about the peak are both in it
        const int n = nbh * nbw;
        const float aa = 0.5f, bq = 0.f;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += SC_BLOCK) {
            const int j = n - 1 - i;
            if (j < i) break;
            const int iy = i / nbw, ix = i - iy * nbw, jy = j / nbw, jx = j - jy * nbw;
            float *pi = &box[(sw.y0 + ia + iy - by0) * SC_UB_LW + (sw.x0 + ja + ix - bx0)];
            float *pj = &box[(sw.y0 + ia + jy - by0) * SC_UB_LW + (sw.x0 + ja + jx - bx0)];
            const float xi = *pi, xj = *pj;
            const float si = xi + xj, sj = xj + xi;
            *pi = aa * si + bq * xi;
            *pj = aa * sj + bq * xj;
        }
        __syncthreads();
    } else {
        store_box();
        __syncthreads();
    }
    UB_STAMP(5);
    // the previous morphology for the convergence sums: the first groups are requested now, under the sweep
    const float *gl = a.in_iteration ? a.morph[c0] + (size_t)c * HW : nullptr;
#ifndef SC_UB_NLAST
#define SC_UB_NLAST 8       // float4 groups of the previous morphology a thread requests ahead of the sweep (16: 39 - 51 spilled VGPRs)
#endif
    constexpr int NLAST = SC_UB_NLAST;
    float4 lastv[NLAST];
    const bool vec4 = (W & 3) == 0;
    if (vec4) {
#pragma unroll
        for (int j = 0; j < NLAST; ++j) {
            const int g = threadIdx.x + j * SC_BLOCK;
            lastv[j] = (gl && g < (HW >> 2)) ? reinterpret_cast<const float4 *>(gl)[g] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // radial monotonicity on the box.  R = 31: levels 1 .. 62 on one wave, no barriers.  R = 63: the compact levels
    // (1 .. 46) on one wave, the rest level-synchronously on the workgroup; valid while it ends by level 2 R
    if (threadIdx.x < SC_WAVE) {
        int done, quiet;
        if (R <= 31) wave_monotonic<float>(bt, cy - by0, cx - bx0, 0.f, &done, &quiet, 0.f, 2 * SC_UB_R);
        else wave_monotonic<float>(bt, cy - by0, cx - bx0, 0.f, &done, &quiet);
        if (threadIdx.x == 0) { hyb[0] = done; hyb[1] = quiet; }
    }
    __syncthreads();
    int lstop = hyb[0];
    if (R > 31 && lstop == (1 << 30)) {
        lstop = monotonic_tile<false, float>(bt, cy - by0, cx - bx0, 0.f, &lastpos, 0.f, SC_COMPACT_LAST + 1, SC_COMPACT_LAST - hyb[1]);
        if (lstop > 2 * SC_UB_R) lstop = 1 << 30;          // the box is complete only up to level 2 R
    }
    UB_STAMP(6);
    if (stamps && threadIdx.x == 0) { stamps[8] = lstop; stamps[9] = R; }
    if (lstop == (1 << 30)) {                              // the footprint leaves the box: full path, nothing written yet
        if (threadIdx.x == 0) {
            fallback[c] = 1;
            if (list) list[atomicAdd(count, 1)] = c;       // (any order: the components are independent)
        }
        return;
    }
    if (threadIdx.x == 0) {
        fallback[c] = 0;
        a.centers[2 * c] = cy; a.centers[2 * c + 1] = cx;
        if (new_shift) { a.shifts[2 * c] = dy; a.shifts[2 * c + 1] = dx; }
    }
    // sparse_l0 / sparse_l1, positive, normalized('morph_max') (update.py:71-82, 27-32, 62-65): as k_source_update
    const float step_morph = 1.0f / (float)a.lipschitz[2 * s + 1];
    const float l0 = a.l0_thresh >= 0.f ? a.l0_thresh * step_morph : -1.f;
    const float l1 = a.l1_thresh >= 0.f ? a.l1_thresh * step_morph : -1.f;
    auto sparse_plus = [&](float v, int y, int x) {
        if (l0 >= 0.f && fabsf(v) < l0) v = 0.f;
        if (l1 >= 0.f) {
            const float mag = fabsf(v) - l1;
            v = (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f)) * (mag < 0.f ? 0.f : mag);
        }
        if (v < 0.f || sweep_level(y, x, cy, cx) > lstop) v = 0.f;
        return v;
    };
    float norm = sparse_plus(box[(cy - by0) * SC_UB_LW + (cx - bx0)], cy, cx);
    const bool regular = norm > 0.f && !isinf(norm);
    const float rnorm = 1.0f / norm;
    float d2f = 0.f, n2f = 0.f;
    auto out_pixel = [&](int y, int x) {
        // outside the box: level >= 2 (R + 1) > lstop -> 0 (0 / norm keeps the reference's NaN when norm is 0 or NaN)
        float v = 0.f;
        const int yb = y - by0, xb = x - bx0;
        if ((unsigned)yb < (unsigned)bh && (unsigned)xb < (unsigned)bw) v = sparse_plus(box[yb * SC_UB_LW + xb], y, x);
        if (regular) { const float q = v * rnorm; return fmaf(fmaf(-q, norm, v), rnorm, q); }
        return v / norm;
    };
    if (vec4) {
        const int gpr = W >> 2, ngroups = HW >> 2;
        const bool plain = regular && l0 < 0.f && l1 < 0.f;       // no thresholds, finite positive norm: the common case
        // Row and column of a thread's groups without a division per group: when the groups of a row divide the
        // workgroup (W = 16 .. 1024 in powers of two), group threadIdx.x + j * 256 sits in the thread's own column,
        // 256 / gpr rows further down per j (the division was ~20 of the ~70 instructions of a group, and the pass
        // costs its instruction count: 64 groups per thread on a 256 x 256 plane)
        const bool rowstep = (SC_BLOCK % gpr) == 0;
        const int ty = threadIdx.x / gpr, tx = (threadIdx.x - ty * gpr) << 2, dyj = rowstep ? SC_BLOCK / gpr : 0;
        auto do_group = [&](int g, int jj, const float4 &l) {
            int y, x;
            if (rowstep) { y = ty + jj * dyj; x = tx; }
            else { y = g / gpr; x = (g - y * gpr) << 2; }
            float o[4];
            // level(x, y) = 2 max(ax, ay) + min(ax, ay) <= lstop  <=>  ax <= axmax(ay): one bound per group's row
            // instead of a level per pixel (as the fused kernel's final pass does); lstop <= 2 R keeps it inside the box
            const int ay = y < cy ? cy - y : y - cy;
            const int h1 = (lstop - ay) >> 1;
            const int axmax = h1 >= ay ? h1 : lstop - 2 * ay;
            const int xb0 = x - cx + axmax;                        // pixel e is inside the cut iff 0 <= xb0 + e <= 2 axmax
            if (regular && (axmax < 0 || xb0 + 3 < 0 || xb0 > 2 * axmax)) { o[0] = o[1] = o[2] = o[3] = 0.f; }
            else if (plain) {
                const float *bp = box + (y - by0) * SC_UB_LW + (x - bx0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = (unsigned)(xb0 + e) <= (unsigned)(2 * axmax);
                    float v = in ? bp[e] : 0.f;                    // (inside the cut implies inside the box)
                    v = v < 0.f ? 0.f : v;
                    const float q = v * rnorm;
                    o[e] = fmaf(fmaf(-q, norm, v), rnorm, q);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = out_pixel(y, x + e);
            }
            reinterpret_cast<float4 *>(gm)[g] = make_float4(o[0], o[1], o[2], o[3]);
            if (gl) {
                const float e0 = l.x - o[0], e1 = l.y - o[1], e2 = l.z - o[2], e3 = l.w - o[3];
                d2f += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
            }
            n2f += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
        };
#pragma unroll
        for (int j = 0; j < NLAST; ++j) {
            const int g = threadIdx.x + j * SC_BLOCK;
            if (g < ngroups) do_group(g, j, lastv[j]);
        }
        // larger frames: the remaining groups in chunks of NLAST, loads of a chunk together before its stores
        for (int g0 = NLAST * SC_BLOCK; g0 < ngroups; g0 += NLAST * SC_BLOCK) {
#pragma unroll
            for (int j = 0; j < NLAST; ++j) {
                const int g = g0 + threadIdx.x + j * SC_BLOCK;
                lastv[j] = (gl && g < ngroups) ? reinterpret_cast<const float4 *>(gl)[g] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < NLAST; ++j) {
                const int g = g0 + threadIdx.x + j * SC_BLOCK;
                if (g < ngroups) do_group(g, g0 / SC_BLOCK + j, lastv[j]);
            }
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += SC_BLOCK) {
            const int y = i / W, x = i - y * W;
            const float o = out_pixel(y, x);
            gm[i] = o;
            if (gl) { const float d = gl[i] - o; d2f += d * d; }
            n2f += o * o;
        }
    }
    double d2 = block_sum((double)d2f, red), n2 = block_sum((double)n2f, red);
    if (n2 != n2 && norm == norm) {
        norm = __builtin_nanf("");
        for (int i = threadIdx.x; i < HW; i += SC_BLOCK) gm[i] = norm;
        d2 = norm;
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
    UB_STAMP(7);
#undef UB_STAMP
}

// NB: bands of 16 window rows the kernel is built for (8: frames up to 128 rows, 16: up to 256).
// The small box: one workgroup per component, four per CU.
#ifndef SC_UB_WAVES
#define SC_UB_WAVES 4            // workgroups of the small-box kernel per CU (tools/ab_box.sh builds variants with -DSC_UB_WAVES=n)
#endif
template <int NB, int XS>
__global__ __launch_bounds__(SC_BLOCK, SC_UB_WAVES) void k_source_update_box(UpdateArgs a, int *fallback, int *list, int *count, long long *stamps_all)
{
    ub_component<NB, 31, XS>(a, blockIdx.x, fallback, list, count, stamps_all);
}

// The large box for the listed components: workgroup i takes list[i]; workgroups past the end of the list exit at
// once -- after every listed one has been dispatched (~35 us for the 32768 components of config 3).  (The first form, one workgroup per
// component of the batch with an early exit for the unlisted ones BETWEEN the listed ones, kept ~0.6 of the two
// resident workgroups per CU busy: 0.7 ms for the ~4 % listed components of BASELINE config 3.  A loop over the
// list inside the workgroup costs the register budget: the body's per-thread invariants get hoisted and spill.)
template <int NB, int XS>
__global__ __launch_bounds__(SC_BLOCK, 2) void k_source_update_box_listed(UpdateArgs a, int *fallback, const int *list, const int *count,
                                                                           long long *stamps_all)
{
    if ((int)blockIdx.x >= *count) return;
    ub_component<NB, 63, XS>(a, list[blockIdx.x], fallback, nullptr, nullptr, stamps_all);
}
