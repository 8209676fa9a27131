// fftconv.h -- row a3b without a library FFT: Observation.render with a PSF difference kernel and its
// adjoint (observation.py:198-239; fft.py:304-317, 264-279, 193-211, 138-181) as ONE kernel per
// iteration.  A workgroup owns one (scene, band) plane and keeps its half-spectrum in LDS from the
// model build to the gradient plane:
//
//   model_b = sum_k sed[k][b] morph[k]            (global -> LDS, rows 0..H-1, real pairs as complex)
//   2-D real FFT, * K-hat_b, inverse              -> render_b                 (in LDS)
//   d = w (render_b - image_b), loss_b; w d       (image streamed once from HBM)
//   2-D real FFT, * conj K-hat_b, inverse         -> G_b = render^T(w d)      (LDS -> global, cropped)
//
// replacing k_psf_model + R2C + k_spec_mul + C2R + k_psf_resid + R2C + k_spec_mul + C2R of the
// hipFFT chain (13 streaming passes over padded planes) by one read of the K morphologies and the
// image and one write of G.
//
// The transform.  The reference pads to next_fast_len(N + P + 3) and crops (fft.py:68-106); the cropped
// result is the LINEAR convolution, which any circular length F >= max(P + N - 1 + o, N - o) reproduces
// (o = the kernel's offset (Fr - P + 1)//2 - Fr//2 <= 0 inside the reference's padded array -- it
// decides which pixel of an even-sized kernel is its centre; scarlet_hip.hip fft_len_min()).  The image
// sits at rows/columns 0..N-1 of the F-periodic plane, the kernel at (q + o) mod F, and the output is
// read back at 0..N-1: pad, ifftshift, fftshift and crop of the reference collapse into that placement.
//
// 1-D engine: L = R1 * R2 with both radices <= 16, two passes IN PLACE (a thread writes only where it
// read, so one barrier per pass and no ping-pong buffer):
//   A  : for n2 < R2: radix-R1 DFT over x[R2 n1 + n2], times w_L^(n2 k1), stored at R2 k1 + n2
//   B  : for k1 < R1: radix-R2 DFT over the R2 contiguous values, stored at R2 k1 + k2  = X[k1 + R1 k2]
// The spectrum stays in that permuted order -- K-hat is produced by the same code, so the pointwise
// product needs no reordering -- and the inverse runs B^-1 (conj twiddle after it), A^-1.
// Real rows: z[n] = x[2n] + i x[2n+1] (a float2 load IS that packing), length-M = Fx/2 complex FFT,
// then the pairs (k, M-k) are untangled in place; X[M] goes to an extra slot M of the row.
// Radix codelets are straight-line register code generated from templates (constexpr twiddles).
// tools/fft_proto.py is the numpy prototype of exactly this index algebra.
#pragma once
#include <type_traits>
#include "common.h"
#include "psf_path.h"

// ------------------------------------------------------------------------------------------------
// compile-time trigonometry for the codelets' internal twiddles
constexpr double cx_pi = 3.141592653589793238462643383279502884;
constexpr double cx_sin_series(double x)
{   // |x| <= pi: 27 terms, error < 1e-15
    double term = x, sum = x;
    for (int n = 1; n < 27; ++n) { term *= -x * x / ((2 * n) * (2 * n + 1)); sum += term; }
    return sum;
}
constexpr double cx_cos_series(double x)
{
    double term = 1, sum = 1;
    for (int n = 1; n < 27; ++n) { term *= -x * x / ((2 * n - 1) * (2 * n)); sum += term; }
    return sum;
}
// cos / sin of 2 pi num / den, num reduced to (-den/2, den/2]
constexpr double cx_cos2pi(int num, int den)
{
    int r = ((num % den) + den) % den;
    if (2 * r > den) r -= den;
    return cx_cos_series(2 * cx_pi * r / den);
}
constexpr double cx_sin2pi(int num, int den)
{
    int r = ((num % den) + den) % den;
    if (2 * r > den) r -= den;
    return cx_sin_series(2 * cx_pi * r / den);
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Complex numbers are 2-vectors of the compiler's native vector type: (re, im) in one aligned register pair, so that
// additions, scalings and the two halves of a complex product each become ONE packed instruction (v_pk_add_f32,
// v_pk_mul_f32, v_pk_fma_f32; the swap of re and im rides in the instruction's op_sel bits).  What the packed forms
// cannot express is a sign on ONE lane, so every product is written as  a * c + swap(a) * (-s, s)  with the pair
// (-s, s) prepared once: compile-time for the codelets' own twiddles, a second pair of the table entry for the
// twiddles read from LDS (cf4 = (c, s, -s, s)); the conjugate product takes the same pair swapped.  (With
// HIP's float2 and scalar re / im expressions the vectoriser found the packed forms only partly: a product cost five
// instructions -- one of them a move that merged two half-used results -- and 29 % of the kernel's vector
// instructions were moves; measured on k_psf_conv_x128, profiles/r03_notes.md.)
typedef float cf __attribute__((ext_vector_type(2)));
typedef float cf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ cf swp(cf a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ cf vfma(cf a, cf b, cf c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ cf cf_zero() { return cf{0.f, 0.f}; }
// a * w, a * conj(w) for a table twiddle w4 = (c, s, -s, s)
__device__ __forceinline__ cf cmul_t(cf a, cf4 w) { return vfma(swp(a), w.zw, a * w.xx); }
__device__ __forceinline__ cf cmul_conj_t(cf a, cf4 w) { return vfma(swp(a), w.wz, a * w.xx); }
// a * k, a * conj(k) for a plain complex k (K-hat from global memory): the pair (-k.y, k.y) costs one instruction
__device__ __forceinline__ cf cmul(cf a, cf k) { return vfma(swp(a), k.yy * cf{-1.f, 1.f}, a * k.xx); }
__device__ __forceinline__ cf cmul_conj(cf a, cf k) { return vfma(swp(a), k.yy * cf{1.f, -1.f}, a * k.xx); }
__device__ __forceinline__ cf cconj(cf a) { return a * cf{1.f, -1.f}; }
// t - i d (forward quarter turn of d) / t + i d
template <bool INV> __device__ __forceinline__ cf add_rot(cf t, cf d)
{
    return vfma(swp(d), INV ? cf{-1.f, 1.f} : cf{1.f, -1.f}, t);
}

// v * w_N^J (forward: e^{-2 pi i J / N}; INV: conjugate), J and N compile-time
template <int N, int J, bool INV>
__device__ __forceinline__ cf twiddle_const(cf v)
{
    constexpr int j = ((J % N) + N) % N;
    if constexpr (j == 0) return v;
    else if constexpr (4 * j == N) return swp(v) * (INV ? cf{-1.f, 1.f} : cf{1.f, -1.f});
    else if constexpr (2 * j == N) return -v;
    else if constexpr (4 * j == 3 * N) return swp(v) * (INV ? cf{1.f, -1.f} : cf{-1.f, 1.f});
    else {
        constexpr float c = (float)cx_cos2pi(j, N);
        constexpr float s = (float)(INV ? cx_sin2pi(j, N) : -cx_sin2pi(j, N));
        return vfma(swp(v), cf{-s, s}, v * c);
    }
}

constexpr int first_factor(int n)
{
    if (n % 4 == 0 && n > 4) return 4;
    if (n % 2 == 0) return 2;
    if (n % 3 == 0) return 3;
    if (n % 5 == 0) return 5;
    if (n % 7 == 0) return 7;
    return n;
}

// ---- radix codelets: in-register DFT of v[0..N-1], natural order in and out
template <int N, bool INV> struct Dft;

template <bool INV> struct Dft<1, INV> { static __device__ __forceinline__ void run(cf (&)[1]) {} };
template <bool INV> struct Dft<2, INV> {
    static __device__ __forceinline__ void run(cf (&v)[2])
    {
        const cf a = v[0], b = v[1];
        v[0] = a + b; v[1] = a - b;
    }
};
template <bool INV> struct Dft<4, INV> {
    static __device__ __forceinline__ void run(cf (&v)[4])
    {
        const cf t0 = v[0] + v[2], t1 = v[0] - v[2], t2 = v[1] + v[3], d = v[1] - v[3];
        v[0] = t0 + t2; v[2] = t0 - t2; v[1] = add_rot<INV>(t1, d); v[3] = add_rot<!INV>(t1, d);
    }
};
// odd primes 3, 5, 7: pairs x_j +- x_{P-j}
template <int P, bool INV> struct DftOddPrime {
    static __device__ __forceinline__ void run(cf (&v)[P])
    {
        constexpr int Hf = (P - 1) / 2;
        cf t[Hf], d[Hf];
        static_for<0, Hf>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            t[j] = v[j + 1] + v[P - 1 - j];
            d[j] = v[j + 1] - v[P - 1 - j];
        });
        const cf x0 = v[0];
        cf sum = x0;
        static_for<0, Hf>([&](auto jc) { sum = sum + t[decltype(jc)::value]; });
        v[0] = sum;
        static_for<1, Hf + 1>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            cf m = x0, n = cf_zero();
            static_for<0, Hf>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr float c = (float)cx_cos2pi((j + 1) * k, P);
                constexpr float s = (float)cx_sin2pi((j + 1) * k, P);
                m = vfma(t[j], cf{c, c}, m);
                n = j == 0 ? d[j] * s : vfma(d[j], cf{s, s}, n);
            });
            // forward: X_k = m - i n, X_{P-k} = m + i n
            v[k] = add_rot<INV>(m, n);
            v[P - k] = add_rot<!INV>(m, n);
        });
    }
};
template <bool INV> struct Dft<3, INV> : DftOddPrime<3, INV> {};
template <bool INV> struct Dft<5, INV> : DftOddPrime<5, INV> {};
template <bool INV> struct Dft<7, INV> : DftOddPrime<7, INV> {};

// composite N = A B (Cooley-Tukey in registers): n = B a + b, k = ka + A kb
template <int N, bool INV> struct Dft {
    static constexpr int A = first_factor(N), B = N / A;
    static_assert(A > 1 && A < N, "unsupported radix");
    static __device__ __forceinline__ void run(cf (&v)[N])
    {
        cf u[N];
        static_for<0, B>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            cf t[A];
            static_for<0, A>([&](auto ac) { constexpr int a = decltype(ac)::value; t[a] = v[B * a + b]; });
            Dft<A, INV>::run(t);
            static_for<0, A>([&](auto kc) {
                constexpr int ka = decltype(kc)::value;
                u[ka * B + b] = twiddle_const<N, b * ka, INV>(t[ka]);
            });
        });
        static_for<0, A>([&](auto kc) {
            constexpr int ka = decltype(kc)::value;
            cf w[B];
            static_for<0, B>([&](auto bc) { constexpr int b = decltype(bc)::value; w[b] = u[ka * B + b]; });
            Dft<B, INV>::run(w);
            static_for<0, B>([&](auto bc) { constexpr int kb = decltype(bc)::value; v[ka + A * kb] = w[kb]; });
        });
    }
};

// ------------------------------------------------------------------------------------------------
// the plan: shapes, radices, placement; tables in device memory (built on the host in float64)
#ifndef SC_FFT_NT
#define SC_FFT_NT 512                   // threads of the convolution workgroup
#endif
#define SC_FFT_MAXF 256
struct FftPlan {
    int H, W;                           // image
    int Fy, Fx, M, RS;                  // transform lengths, M = Fx / 2, RS = row stride of the LDS plane in float2 (>= M + 1)
    int R1y, R2y, R1x, R2x;             // Fy = R1y R2y, M = R1x R2x
    int Py, Px, oky, okx;               // kernel shape and its offsets inside the periodic plane
    float scale;                        // 1 / (M Fy): the two unnormalised inverse transforms
    int stagger_wgs;                    // k_psf_conv: number of workgroups of the first generation (CUs of the device)
    int tab_off;                        // LDS offset (float2) of the tables: behind the plane and the image staging area
    int dma_image;                      // k_psf_conv: the image is staged in LDS by LDS-DMA (rows >= H + the space behind the plane)
    const float2 *tables;               // device: twiddles as (c, s, -s, s) quadruples (cf4) -- twy[Fy] = w_Fy^j, twm[M] = w_M^j,
                                        // twx[NP] = w_Fx^k, twp[NP] = twx in the order of the column pairs -- then posx[M]
                                        // (uint16) and the column pairs in position order: pair[NP] = {ra, rb, cb, 0} (uint16 x 4)
};
// NP = M/2 + 1 pairs (k, M - k) of spectrum columns.  The fused column passes visit them in the order of their
// POSITION in the permuted row (ra ascending; then rb = const - ra descends): consecutive lanes read consecutive
// float2, where the order of k walked the row at stride R2x (two-way bank conflicts on every paired read).
__host__ __device__ inline int fft_table_float2s(int Fy, int M) { return 2 * (Fy + M + 2 * (M / 2 + 1)) + (M + 3) / 4 + (M / 2 + 1); }
// LDS of the transform kernels: the plane [Fy][RS], then the tables.  k_psf_conv additionally stages the image
// plane (H x W floats) from row H on -- rows >= H are idle between the render's column stage and the adjoint's,
// which is when the residual needs the image -- so its tables sit behind max(plane, H rows + image).
__host__ __device__ inline int fft_tab_off(int Fy, int RS, int H, int W, bool stage_image)
{
    const int plane = Fy * RS, staged = H * RS + (H * W + 1) / 2;
    return (((stage_image && staged > plane) ? staged : plane) + 1) & ~1;       // the tables are read 16 bytes at a time
}
__host__ __device__ inline size_t fft_lds_bytes(int Fy, int M, int RS, int H = 0, int W = 0, bool stage_image = false)
{
    return sizeof(float2) * ((size_t)fft_tab_off(Fy, RS, H, W, stage_image) + fft_table_float2s(Fy, M));
}

// exact u / d for 0 <= u < 2^20, 1 <= d < 2^12 (float reciprocal)
__device__ __forceinline__ int fast_div(int u, float rcp_d) { return (int)(((float)u + 0.5f) * rcp_d); }

// one pass over `nlines` independent 1-D transforms: item (line, j) loads v[q] = A[line*ls + j*joff + q*qs],
// q < R, transforms, multiplies output k by (conj) tw[j k] when use_tw, and stores back in place.
template <int R, bool INV, int NT = SC_FFT_NT>
__device__ __forceinline__ void fft_items(cf *A, int nlines, int ls, int J, int joff, int qs, const cf4 *tw,
                                          bool use_tw, bool line_fastest)
{
    const int total = nlines * J;
    const float rcp = 1.0f / (float)(line_fastest ? nlines : J);
    for (int u = threadIdx.x; u < total; u += NT) {
        int line, j;
        if (line_fastest) { j = fast_div(u, rcp); line = u - j * nlines; }
        else { line = fast_div(u, rcp); j = u - line * J; }
        cf *p = A + line * ls + j * joff;
        cf v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = p[q * qs];
        Dft<R, INV>::run(v);
        if (use_tw) {
#pragma unroll
            for (int k = 1; k < R; ++k) {
                const cf4 w = tw[j * k];
                v[k] = INV ? cmul_conj_t(v[k], w) : cmul_t(v[k], w);
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) p[q * qs] = v[q];
    }
}

template <bool INV, int NT = SC_FFT_NT>
__device__ __forceinline__ void fft_pass(int R, cf *A, int nlines, int ls, int J, int joff, int qs, const cf4 *tw,
                                         bool use_tw, bool line_fastest)
{
    switch (R) {
#define SC_FFT_CASE(R_) case R_: fft_items<R_, INV, NT>(A, nlines, ls, J, joff, qs, tw, use_tw, line_fastest); break;
    SC_FFT_CASE(4) SC_FFT_CASE(5) SC_FFT_CASE(6) SC_FFT_CASE(7) SC_FFT_CASE(8) SC_FFT_CASE(9) SC_FFT_CASE(10)
    SC_FFT_CASE(12) SC_FFT_CASE(14) SC_FFT_CASE(15) SC_FFT_CASE(16)
#undef SC_FFT_CASE
    default: break;
    }
    __syncthreads();
}
__host__ __device__ inline bool fft_radix_ok(int r)
{
    return r == 4 || r == 5 || r == 6 || r == 7 || r == 8 || r == 9 || r == 10 || r == 12 || r == 14 || r == 15 || r == 16;
}

// forward / inverse 1-D transforms of `nlines` lines (es = element stride, ls = line stride), in place
__device__ __forceinline__ void fft_lines_fwd(cf *A, int nlines, int ls, int es, int R1, int R2, const cf4 *tw, bool lf)
{
    fft_pass<false>(R1, A, nlines, ls, R2, es, R2 * es, tw, true, lf);          // A
    fft_pass<false>(R2, A, nlines, ls, R1, R2 * es, es, tw, false, lf);         // B
}
__device__ __forceinline__ void fft_lines_inv(cf *A, int nlines, int ls, int es, int R1, int R2, const cf4 *tw, bool lf)
{
    fft_pass<true>(R2, A, nlines, ls, R1, R2 * es, es, tw, true, lf);           // B^-1
    fft_pass<true>(R1, A, nlines, ls, R2, es, R2 * es, tw, false, lf);          // A^-1
}

struct FftPair { unsigned short ra, rb, cb, pad; };    // read positions of the pair, write position of its second member
struct FftLds {
    cf *A;                      // [Fy][RS]
    const cf4 *twy, *twm, *twx, *twp;
    const unsigned short *posx;
    const FftPair *pair;
};
template <int NT = SC_FFT_NT>
__device__ __forceinline__ FftLds fft_lds_setup(float2 *lds, const FftPlan &p)
{
    FftLds l;
    l.A = reinterpret_cast<cf *>(lds);
    float2 *t = lds + p.tab_off;
    const int nt = fft_table_float2s(p.Fy, p.M), NP = p.M / 2 + 1;
    for (int i = threadIdx.x; i < nt; i += NT) t[i] = p.tables[i];
    l.twy = reinterpret_cast<const cf4 *>(t); l.twm = l.twy + p.Fy; l.twx = l.twm + p.M; l.twp = l.twx + NP;
    l.posx = (const unsigned short *)(l.twp + NP);
    l.pair = (const FftPair *)(t + 2 * (p.Fy + p.M + 2 * NP) + (p.M + 3) / 4);
    return l;
}

// real rows: a = Z[k], b = Z[M-k] of the packed transform -> X[k], X[M-k] (w = w_Fx^k)
//   X[k] = (a + b*)/2 - i w (a - b*)/2 ;  X[M-k] = (b + a*)/2 + i w* (b - a*)/2 = conj(s) - i conj(w d)
__device__ __forceinline__ void untangle_pair(cf a, cf b, cf4 w, cf &xa, cf &xb)
{
    const cf h = a * 0.5f;
    const cf s = vfma(b, cf{0.5f, -0.5f}, h);         // (a + b*)/2
    const cf d = vfma(b, cf{-0.5f, 0.5f}, h);         // (a - b*)/2
    const cf wd = cmul_t(d, w);
    xa = vfma(swp(wd), cf{1.f, -1.f}, s);             // s - i wd
    xb = vfma(s, cf{1.f, -1.f}, -swp(wd));            // (s.x - wd.y, -s.y - wd.x)
}
// the inverse: xk = X[k], xm = X[M-k] -> Z[k] = E + i O, Z[M-k] = conj(E) + i conj(O), E = (xk + xm*)/2, O = (xk - xm*)/2 conj(w)
__device__ __forceinline__ void tangle_pair(cf xk, cf xm, cf4 w, cf &zk, cf &zm)
{
    const cf h = xk * 0.5f;
    const cf E = vfma(xm, cf{0.5f, -0.5f}, h);
    const cf D = vfma(xm, cf{-0.5f, 0.5f}, h);
    const cf O = cmul_conj_t(D, w);
    zk = vfma(swp(O), cf{-1.f, 1.f}, E);              // (E.x - O.y, E.y + O.x)
    zm = vfma(E, cf{1.f, -1.f}, swp(O));              // (E.x + O.y, -E.y + O.x)
}

// real rows after the length-M complex FFT: untangle the pairs (k, M - k) in place (tools/fft_proto.py rows_fwd)
__device__ __forceinline__ void fft_rows_untangle(const FftLds &l, const FftPlan &p, int nrows)
{
    const int M = p.M, half = M / 2 + 1;          // k = 0 .. M/2
    const int total = nrows * half;
    const float rcp = 1.0f / (float)half;
    for (int u = threadIdx.x; u < total; u += SC_FFT_NT) {
        const int y = fast_div(u, rcp), k = u - y * half;
        cf *r = l.A + y * p.RS;
        if (k == 0) {
            const cf z = r[l.posx[0]];
            r[l.posx[0]] = cf{z.x + z.y, 0.f};
            r[M] = cf{z.x - z.y, 0.f};
        } else if (2 * k == M) {
            const int q = l.posx[k];
            r[q] = cconj(r[q]);
        } else if (2 * k < M) {
            const int pa = l.posx[k], pb = l.posx[M - k];
            cf xa, xb;
            untangle_pair(r[pa], r[pb], l.twx[k], xa, xb);
            r[pa] = xa;
            r[pb] = xb;
        }
    }
    __syncthreads();
}
// inverse of fft_rows_untangle (tools/fft_proto.py rows_inv)
__device__ __forceinline__ void fft_rows_tangle(const FftLds &l, const FftPlan &p, int nrows)
{
    const int M = p.M, half = M / 2 + 1;
    const int total = nrows * half;
    const float rcp = 1.0f / (float)half;
    for (int u = threadIdx.x; u < total; u += SC_FFT_NT) {
        const int y = fast_div(u, rcp), k = u - y * half;
        cf *r = l.A + y * p.RS;
        if (k == 0) {
            const float x0 = r[l.posx[0]].x, xm = r[M].x;
            r[l.posx[0]] = cf{0.5f * (x0 + xm), 0.5f * (x0 - xm)};
        } else if (2 * k == M) {
            const int q = l.posx[k];
            r[q] = cconj(r[q]);
        } else if (2 * k < M) {
            const int pa = l.posx[k], pb = l.posx[M - k];
            cf zk, zm;
            tangle_pair(r[pa], r[pb], l.twx[k], zk, zm);
            r[pa] = zk;
            r[pb] = zm;
        }
    }
    __syncthreads();
}

// 2-D forward / inverse on the plane in LDS; rows < nrows_in hold data (the others are zero on input /
// not needed on output)
__device__ __forceinline__ void fft2d_fwd(const FftLds &l, const FftPlan &p, int nrows)
{
    fft_lines_fwd(l.A, nrows, p.RS, 1, p.R1x, p.R2x, l.twm, false);
    fft_rows_untangle(l, p, nrows);
    fft_lines_fwd(l.A, p.M + 1, 1, p.RS, p.R1y, p.R2y, l.twy, true);
}
__device__ __forceinline__ void fft2d_inv(const FftLds &l, const FftPlan &p, int nrows)
{
    fft_lines_inv(l.A, p.M + 1, 1, p.RS, p.R1y, p.R2y, l.twy, true);
    fft_rows_tangle(l, p, nrows);
    fft_lines_inv(l.A, nrows, p.RS, 1, p.R1x, p.R2x, l.twm, false);
}
// A *= (conj) K-hat (already scaled), all Fy x (M + 1) slots; K-hat [Fy][M + 1] in global memory
template <bool CONJ>
__device__ __forceinline__ void fft_spec_mul(const FftLds &l, const FftPlan &p, const float2 *khat)
{
    const int cols = p.M + 1, total = p.Fy * cols;
    const float rcp = 1.0f / (float)cols;
    for (int u = threadIdx.x; u < total; u += SC_FFT_NT) {
        const int y = fast_div(u, rcp), c = u - y * cols;
        const cf k = reinterpret_cast<const cf *>(khat)[u];
        cf *q = l.A + y * p.RS + c;
        *q = CONJ ? cmul_conj(*q, k) : cmul(*q, k);
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// K-hat: one workgroup per kernel plane.  kernel [nk][Py][Px] -> khat [nk][Fy][M + 1] (permuted, scaled)
__global__ __launch_bounds__(SC_FFT_NT) void k_fft_khat(const float *ker, FftPlan p, float2 *khat)
{
    extern __shared__ __align__(16) float2 fft_lds[];
    const FftLds l = fft_lds_setup(fft_lds, p);
    const int plane = blockIdx.x;
    const float *kp = ker + (size_t)plane * p.Py * p.Px;
    const int M = p.M, cols = M + 1;
    for (int u = threadIdx.x; u < p.Fy * p.RS; u += SC_FFT_NT) {
        const int iy = u / p.RS, n = u - iy * p.RS;
        cf v = cf_zero();
        if (n < M) {
            const int q = pos_mod(iy - p.oky, p.Fy);
            if (q < p.Py) {
                const int r0 = pos_mod(2 * n - p.okx, p.Fx), r1 = pos_mod(2 * n + 1 - p.okx, p.Fx);
                if (r0 < p.Px) v.x = kp[q * p.Px + r0];
                if (r1 < p.Px) v.y = kp[q * p.Px + r1];
            }
        }
        l.A[u] = v;
    }
    __syncthreads();
    fft2d_fwd(l, p, p.Fy);
    float2 *out = khat + (size_t)plane * p.Fy * cols;
    for (int u = threadIdx.x; u < p.Fy * cols; u += SC_FFT_NT) {
        const int y = u / cols, c = u - y * cols;
        const cf v = l.A[y * p.RS + c];
        out[u] = make_float2(v.x * p.scale, v.y * p.scale);
    }
}

// standalone render of n planes (Observation.render / scarlet_convolve_same): out = crop(in (*) kernel)
__global__ __launch_bounds__(SC_FFT_NT) void k_fft_convolve(const float *in, FftPlan p, const float2 *khat, int nk, float *out)
{
    extern __shared__ __align__(16) float2 fft_lds[];
    const FftLds l = fft_lds_setup(fft_lds, p);
    const int plane = blockIdx.x, H = p.H, W = p.W, M = p.M, Wh = (W + 1) / 2;
    const float *ip = in + (size_t)plane * H * W;
    for (int u = threadIdx.x; u < p.Fy * p.RS; u += SC_FFT_NT) {
        const int y = u / p.RS, n = u - y * p.RS;
        cf v = cf_zero();
        if (y < H && n < Wh) {
            v.x = ip[y * W + 2 * n];
            if (2 * n + 1 < W) v.y = ip[y * W + 2 * n + 1];
        }
        l.A[u] = v;
    }
    __syncthreads();
    fft2d_fwd(l, p, H);
    fft_spec_mul<false>(l, p, khat + (size_t)(nk == 1 ? 0 : plane) * p.Fy * (M + 1));
    fft2d_inv(l, p, H);
    float *op = out + (size_t)plane * H * W;
    for (int u = threadIdx.x; u < H * W; u += SC_FFT_NT) {
        const int y = u / W, x = u - y * W;
        const cf v = l.A[y * p.RS + (x >> 1)];
        op[u] = (x & 1) ? v.y : v.x;
    }
}

// ------------------------------------------------------------------------------------------------
// Fused passes of the iteration kernel.  Neighbouring passes that touch the SAME set of positions per
// thread are merged, so the data makes the LDS round trip once instead of two or three times:
//   cols_A_untangle     : untangle of the real-row pairs (k, M-k) + column pass A on both columns
//   cols_B_mul_Binv     : column pass B, * (conj) K-hat, inverse pass B^-1  (same R2y positions)
//   cols_Ainv_tangle    : inverse column pass A^-1 on both columns + re-tangling of the pair
//   rows_Ainv_resid_A   : last inverse row pass -> pixels in registers -> residual/loss -> first
//                         forward row pass of the adjoint convolution        (same R1x positions)
//   rows_Ainv_store     : last inverse row pass of the adjoint -> G in global memory
// 14 passes (= barriers) per plane and iteration instead of 25.  Rows >= H of the plane are never
// read from LDS (they are zero by construction: predicated), so nothing has to clear them.
template <int R, int NT = SC_FFT_NT>
__device__ __forceinline__ void cols_A_untangle(const FftLds &l, const FftPlan &p)
{
    const int M = p.M, NP = M / 2 + 1, R2 = p.R2y, RS = p.RS, H = p.H;
    const int total = NP * R2;
    const float rcp = 1.0f / (float)NP;
    for (int u = threadIdx.x; u < total; u += NT) {
        const int n2 = fast_div(u, rcp), i = u - n2 * NP;
        const FftPair pr = l.pair[i];
        const int ra = pr.ra, rb = pr.rb, cb = pr.cb;
        const cf4 w = l.twp[i];
        cf xa[R], xb[R];
#pragma unroll
        for (int n1 = 0; n1 < R; ++n1) {
            const int r = R2 * n1 + n2;
            cf a = cf_zero(), b = a;
            if (r < H) { a = l.A[r * RS + ra]; b = l.A[r * RS + rb]; }
            untangle_pair(a, b, w, xa[n1], xb[n1]);
        }
        Dft<R, false>::run(xa);
        Dft<R, false>::run(xb);
#pragma unroll
        for (int k1 = 0; k1 < R; ++k1) {
            cf va = xa[k1], vb = xb[k1];
            if (k1 > 0) { const cf4 t = l.twy[n2 * k1]; va = cmul_t(va, t); vb = cmul_t(vb, t); }
            cf *q = l.A + (R2 * k1 + n2) * RS;
            q[ra] = va;
            q[cb] = vb;               // k == M/2: the same value at the same place
        }
    }
    __syncthreads();
}

template <int R, bool CONJ, int NT = SC_FFT_NT>
__device__ __forceinline__ void cols_B_mul_Binv(const FftLds &l, const FftPlan &p, const float2 *khat)
{
    const int cols = p.M + 1, J = p.R1y, RS = p.RS;
    const int total = cols * J;
    const float rcp = 1.0f / (float)cols;
    for (int u = threadIdx.x; u < total; u += NT) {
        const int k1 = fast_div(u, rcp), c = u - k1 * cols;
        cf *q = l.A + (R * k1) * RS + c;
        const cf *kq = reinterpret_cast<const cf *>(khat) + (R * k1) * cols + c;
        cf v[R], kk[R];
#pragma unroll
        for (int r = 0; r < R; ++r) kk[r] = kq[r * cols];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = q[r * RS];
        Dft<R, false>::run(v);
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = CONJ ? cmul_conj(v[r], kk[r]) : cmul(v[r], kk[r]);
        Dft<R, true>::run(v);
#pragma unroll
        for (int r = 1; r < R; ++r) v[r] = cmul_conj_t(v[r], l.twy[r * k1]);
#pragma unroll
        for (int r = 0; r < R; ++r) q[r * RS] = v[r];
    }
    __syncthreads();
}

template <int R, int NT = SC_FFT_NT>
__device__ __forceinline__ void cols_Ainv_tangle(const FftLds &l, const FftPlan &p)
{
    const int M = p.M, NP = M / 2 + 1, R2 = p.R2y, RS = p.RS, H = p.H;
    const int total = NP * R2;
    const float rcp = 1.0f / (float)NP;
    for (int u = threadIdx.x; u < total; u += NT) {
        const int n2 = fast_div(u, rcp), i = u - n2 * NP;
        const FftPair pr = l.pair[i];
        const int ra = pr.ra, rb = pr.rb, cb = pr.cb;
        const bool first = cb == M;                              // (k = 0: its second member lives in the extra slot M)
        const cf4 w = l.twp[i];
        cf xa[R], xb[R];
#pragma unroll
        for (int k1 = 0; k1 < R; ++k1) {
            const cf *q = l.A + (R2 * k1 + n2) * RS;
            xa[k1] = q[ra];
            xb[k1] = q[cb];
        }
        Dft<R, true>::run(xa);
        Dft<R, true>::run(xb);
#pragma unroll
        for (int n1 = 0; n1 < R; ++n1) {
            const int r = R2 * n1 + n2;
            if (r < H) {
                cf xk = xa[n1], xm = xb[n1], zk, zm;
                if (first) { xk.y = 0.f; xm.y = 0.f; }             // X[0], X[M] of a real signal are real
                tangle_pair(xk, xm, w, zk, zm);
                l.A[r * RS + rb] = zm;                             // k == 0, M/2: same place as zk, written first
                l.A[r * RS + ra] = zk;
            }
        }
    }
    __syncthreads();
}

// The image plane of the residual pass, global -> LDS by LDS-DMA (no registers: a register prefetch across the
// render's passes is either sunk to its uses by the compiler -- sixteen exposed HBM round trips, measured --
// or spills).  16 B per lane, wave-uniform LDS base + lane * 16: the staged image is lane-linear [H][W] floats.
// Requested after the render's column stage, landed by the barrier that ends the next row pass.
#define SC_FFT_PF 16                     // model-plane pairs a thread holds in registers at kernel start
template <int NT = SC_FFT_NT>
__device__ __forceinline__ void fft_dma_image(const FftLds &l, const FftPlan &p, const float *img)
{
    const int nchunks = (p.H * p.W) >> 2;             // 16-byte pieces (H W % 4 == 0: checked on the host)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *stage = reinterpret_cast<float *>(l.A + p.H * p.RS);
    for (int c0 = wave * 64; c0 < nchunks; c0 += NT) {
        if (c0 + lane < nchunks)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(img + (size_t)(c0 + lane) * 4),
                                             (__attribute__((address_space(3))) void *)(stage + (size_t)c0 * 4), 16, 0, 0);
    }
}
// ---- row passes fused with what happens to the pixels (tools/fft_proto.py: the last inverse row pass A^-1 of item
// (row y, n2) leaves the R1x pixel pairs n = n2 + R2x n1 of its row in registers, and the first forward row pass A
// of the next transform starts from exactly those positions): the residual, the model-plane load and the store of G
// ride on the row passes instead of being LDS round trips of their own (3 passes = barriers less per plane).
// rows_Ainv_resid_A: render's last inverse row pass -> d = w (render - image), loss, w d -> adjoint's first row pass
// `pre` (NI > 0): the items' image pairs already in registers (ImagePairs below), [item of the thread][n1]; the thread's
// items are then walked by a compile-time count
template <int R, int NT, int NI = 0>
__device__ __forceinline__ void rows_Ainv_resid_A(const FftLds &l, const FftPlan &p, const float2 *img, const float2 *wgt,
                                                  float wscalar, double &loss, const cf (*pre)[R] = nullptr)
{
    const int R2 = p.R2x, RS = p.RS, Wh = p.W >> 1;
    const int total = p.H * R2;
    const float rcp = 1.0f / (float)R2;
    const cf *stage = l.A + p.H * p.RS;               // the image as the LDS-DMA left it: [H][W/2] pairs, lane-linear
    const cf *imgv = reinterpret_cast<const cf *>(img), *wgtv = reinterpret_cast<const cf *>(wgt);
    auto item = [&](int u, const cf *mine) {
        const int y = fast_div(u, rcp), n2 = u - y * R2;
        cf *q = l.A + y * RS + n2;
        cf v[R];
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = q[k * R2];
        Dft<R, true>::run(v);
#pragma unroll
        for (int n1 = 0; n1 < R; ++n1) {
            const int n = n2 + n1 * R2;
            if (n < Wh) {
                const int e = y * Wh + n;
                const cf im = mine ? mine[n1] : (p.dma_image ? stage[e] : imgv[e]);
                const cf w = wgt ? wgtv[e] : cf{wscalar, wscalar};
                const cf d = w * (v[n1] - im);
                loss += (double)d.x * (double)d.x + (double)d.y * (double)d.y;
                v[n1] = w * d;
            } else
                v[n1] = cf_zero();                   // columns W/2 .. M-1: the adjoint's input is zero there
        }
        Dft<R, false>::run(v);
#pragma unroll
        for (int k = 1; k < R; ++k) v[k] = cmul_t(v[k], l.twm[n2 * k]);
#pragma unroll
        for (int k = 0; k < R; ++k) q[k * R2] = v[k];
    };
    if constexpr (NI > 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int u = threadIdx.x + i * NT;
            if (u < total) item(u, pre[i]);
        }
    } else {
        for (int u = threadIdx.x; u < total; u += NT) item(u, nullptr);
    }
    __syncthreads();
}
// rows_Ainv_store: the adjoint's last inverse row pass -> G in global memory
template <int R, int NT>
__device__ __forceinline__ void rows_Ainv_store(const FftLds &l, const FftPlan &p, cf *gp)
{
    const int R2 = p.R2x, RS = p.RS, Wh = p.W >> 1;
    const int total = p.H * R2;
    const float rcp = 1.0f / (float)R2;
    for (int u = threadIdx.x; u < total; u += NT) {
        const int y = fast_div(u, rcp), n2 = u - y * R2;
        const cf *q = l.A + y * RS + n2;
        cf v[R];
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = q[k * R2];
        Dft<R, true>::run(v);
#pragma unroll
        for (int n1 = 0; n1 < R; ++n1) {
            const int n = n2 + n1 * R2;
            if (n < Wh) gp[y * Wh + n] = v[n1];
        }
    }
}
// rows_load_A: model plane (global, pixel pairs) -> first forward row pass.  The loads of ALL of a thread's items are
// issued before anything else the kernel does (tables, ...): NI = items per thread, compile-time (exact-shape instance)
template <int R, int NT, int NI>
struct RowsLoadA {
    cf v[NI][R];
    __device__ __forceinline__ void request(const FftPlan &p, const cf *gp)
    {
        const int R2 = p.R2x, Wh = p.W >> 1, total = p.H * R2;
        const float rcp = 1.0f / (float)R2;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int u = threadIdx.x + i * NT;
            const int y = fast_div(u, rcp), n2 = u - y * R2;
#pragma unroll
            for (int n1 = 0; n1 < R; ++n1) {
                const int n = n2 + n1 * R2;
                v[i][n1] = (u < total && n < Wh) ? gp[y * Wh + n] : cf_zero();
            }
        }
    }
    __device__ __forceinline__ void run(const FftLds &l, const FftPlan &p)
    {
        const int R2 = p.R2x, RS = p.RS, total = p.H * R2;
        const float rcp = 1.0f / (float)R2;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int u = threadIdx.x + i * NT;
            if (u < total) {
                const int y = fast_div(u, rcp), n2 = u - y * R2;
                cf *q = l.A + y * RS + n2;
                Dft<R, false>::run(v[i]);
#pragma unroll
                for (int k = 1; k < R; ++k) v[i][k] = cmul_t(v[i][k], l.twm[n2 * k]);
#pragma unroll
                for (int k = 0; k < R; ++k) q[k * R2] = v[i][k];
            }
        }
        __syncthreads();
    }
};

// the image pairs of a thread's items of rows_Ainv_resid_A, requested one pass ahead into registers (exact-shape instance:
// no staging area in LDS -- the 64 KB it took are what lets the other pipeline's workgroups share the CU)
template <int R, int NT, int NI>
struct ImagePairs {
    cf v[NI][R];
    __device__ __forceinline__ void request(const FftPlan &p, const cf *img)
    {
        const int R2 = p.R2x, Wh = p.W >> 1, total = p.H * R2;
        const float rcp = 1.0f / (float)R2;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int u = threadIdx.x + i * NT;
            const int y = fast_div(u, rcp), n2 = u - y * R2;
#pragma unroll
            for (int n1 = 0; n1 < R; ++n1) {
                const int n = n2 + n1 * R2;
                v[i][n1] = (u < total && n < Wh) ? img[y * Wh + n] : cf_zero();
            }
        }
    }
};

#define SC_FFT_DISPATCH(R_, CALL)                                                                                  \
    switch (R_) {                                                                                                  \
    case 4: { constexpr int RR = 4; CALL; } break;    case 5: { constexpr int RR = 5; CALL; } break;               \
    case 6: { constexpr int RR = 6; CALL; } break;    case 7: { constexpr int RR = 7; CALL; } break;               \
    case 8: { constexpr int RR = 8; CALL; } break;    case 9: { constexpr int RR = 9; CALL; } break;               \
    case 10: { constexpr int RR = 10; CALL; } break;  case 12: { constexpr int RR = 12; CALL; } break;             \
    case 14: { constexpr int RR = 14; CALL; } break;  case 15: { constexpr int RR = 15; CALL; } break;             \
    case 16: { constexpr int RR = 16; CALL; } break;  default: break;                                              \
    }

// one convolution between its first forward row pass and its last inverse one: rows B, columns (fused), rows B^-1
#define FFT_STAMP(i) do { if (stamps && threadIdx.x == 0) stamps[(i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
template <bool CONJ, int NT = SC_FFT_NT, typename AfterColumns>
__device__ __forceinline__ void fft_conv_middle(const FftLds &l, const FftPlan &p, const float2 *khat, long long *stamps,
                                                AfterColumns &&after_columns)
{
    fft_pass<false, NT>(p.R2x, l.A, p.H, p.RS, p.R1x, p.R2x, 1, l.twm, false, false);                    // rows B
    FFT_STAMP(1);
    SC_FFT_DISPATCH(p.R1y, (cols_A_untangle<RR, NT>(l, p)))
    FFT_STAMP(2);
    SC_FFT_DISPATCH(p.R2y, (cols_B_mul_Binv<RR, CONJ, NT>(l, p, khat)))
    FFT_STAMP(3);
    SC_FFT_DISPATCH(p.R1y, (cols_Ainv_tangle<RR, NT>(l, p)))
    FFT_STAMP(4);
    after_columns();                     // the low-register-pressure row passes follow: global loads go here
    fft_pass<true, NT>(p.R2x, l.A, p.H, p.RS, p.R1x, p.R2x, 1, l.twm, true, false);                      // rows B^-1
    FFT_STAMP(5);
}

#ifndef SC_X128_DMA
#define SC_X128_DMA 0                // exact-shape instance: 1 = image staged in LDS by LDS-DMA (158 KB), 0 = in registers (97 KB)
#endif
// the iteration's convolution pair: G_b = render^T( w^2 (render(model)_b - image_b) ), loss_b.
// `G` holds the model planes model_b = sum_k sed[k][b] morph[k] on entry ([S][B][H][W], written by k_psf_model:
// one streaming read of the K morphologies per SCENE -- building the plane here costs K plane reads per
// BAND through this CU's load path, a third of the kernel when measured) and the gradient planes on exit.
// grid: one workgroup per (scene, band); consecutive scenes go to consecutive XCDs.
// XP: the exact-shape instance for BASELINE config 3 (128 x 128 frames, 41 x 41 kernel whose offset in the reference's padded
// array is -20: F = 150 = 10 x 15 both ways, M = 75 = 5 x 15) -- the plan's shapes and radices are compile-time facts (index
// arithmetic and radix dispatch fold away: the kernel fits 128 VGPRs) and the workgroup has NT = 1024 threads: the passes
// of 570 - 760 items take ONE round of threads instead of two half-empty ones, at four waves per SIMD instead of two.
template <bool XP, int NT>
__device__ __forceinline__ void psf_conv_body(const PsfArgs &a, const FftPlan &p_in, float *G, long long *stamps_all)
{
    FftPlan p = p_in;
    if (XP) {
        p.H = 128; p.W = 128; p.Fy = 150; p.Fx = 150; p.M = 75; p.RS = 76;
        p.R1y = 10; p.R2y = 15; p.R1x = 5; p.R2x = 15; p.dma_image = SC_X128_DMA;
        p.tab_off = fft_tab_off(150, 76, 128, 128, SC_X128_DMA != 0);
    }
    extern __shared__ __align__(16) float2 fft_lds[];
    const int B = a.B;
    const int xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
    const int grp = t / B, b = t - grp * B, s = grp * 8 + xcd;
    if (s >= a.S || !a.active[s]) return;
    long long *stamps = stamps_all ? stamps_all + ((size_t)s * B + b) * 32 : nullptr;
    // Every workgroup does the same work in the same time, one per CU: left alone, all CUs stay in lockstep and
    // each global-memory phase (model plane, image, K-hat, G) hits HBM as one chip-wide burst, served at the
    // chip's rate with every wave stalled at issue (measured: 6k cycles to issue 16 loads).  The first
    // generation of workgroups is therefore spread over one plane time; the offsets persist because each CU
    // runs its workgroups back to back.
    if (blockIdx.x < (unsigned)p.stagger_wgs) {
        const int slots = (int)((blockIdx.x * 2654435761u) >> 26);          // 0 .. 63, scrambled
        for (int i = 0; i < slots; ++i) __builtin_amdgcn_s_sleep(32);       // 64 x ~2k cycles
    }
    FFT_STAMP(30);
    const int H = p.H, W = p.W, M = p.M, HW = H * W, Wh = W >> 1;           // W even (checked on the host)
    const size_t plane = (size_t)s * B + b;
    cf *gp = (cf *)(G + plane * HW);
    const float2 *img = (const float2 *)(a.images + plane * HW);
    const float2 *wgt = a.weights ? (const float2 *)(a.weights + plane * HW) : nullptr;
    // Exact-shape instance: the model plane goes from global memory straight into the first forward row pass (requested
    // here, before the tables are staged); otherwise it is written to rows < H as pixel pairs, zeros up to column M
    constexpr int NI1 = XP ? (128 * 15 + NT - 1) / NT : 1;
    RowsLoadA<XP ? 5 : 1, NT, NI1> first;
    if constexpr (XP) first.request(p, gp);
    constexpr int PF = XP ? 1 : SC_FFT_PF * SC_FFT_NT / NT;      // (generic: 8192 pairs in registers at kernel start)
    cf mreg[PF];
    if constexpr (!XP) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int u = threadIdx.x + j * NT;
            mreg[j] = u < H * Wh ? gp[u] : cf_zero();
        }
    }
    const FftLds l = fft_lds_setup<NT>(fft_lds, p);
    __shared__ double red[NT / SC_WAVE];
    if constexpr (!XP) {
        const float rcp = 1.0f / (float)M;
        for (int u = threadIdx.x; u < H * M; u += NT) {
            const int y = fast_div(u, rcp), n = u - y * M;
            if (n >= Wh) l.A[y * p.RS + n] = cf_zero();
        }
        const float rcpw = 1.0f / (float)Wh;
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int u = threadIdx.x + j * NT;
            if (u < H * Wh) { const int y = fast_div(u, rcpw), n = u - y * Wh; l.A[y * p.RS + n] = mreg[j]; }
        }
        for (int u = threadIdx.x + PF * NT; u < H * Wh; u += NT) {
            const int y = fast_div(u, rcpw), n = u - y * Wh;
            l.A[y * p.RS + n] = gp[u];
        }
    }
    __syncthreads();
    const float2 *khat = a.khat + (size_t)(a.khat_per_scene ? s * B + b : b) * p.Fy * (M + 1);
    double loss = 0;
    FFT_STAMP(31);
    if constexpr (XP) first.run(l, p);
    else fft_pass<false, NT>(p.R1x, l.A, p.H, p.RS, p.R2x, 1, p.R2x, l.twm, true, false);                // rows A
    FFT_STAMP(0);
    // the image is requested after the render's column stage: in flight under the next row pass
    ImagePairs<XP ? 5 : 1, NT, NI1> ipairs;
    constexpr bool IREG = XP && !SC_X128_DMA;
    fft_conv_middle<false, NT>(l, p, khat, stamps, [&]() {
        if (IREG) ipairs.request(p, reinterpret_cast<const cf *>(img));
        else if (p.dma_image) fft_dma_image<NT>(l, p, a.images + plane * HW);
    });
    if constexpr (IREG) rows_Ainv_resid_A<5, NT, NI1>(l, p, img, wgt, a.weight_scalar, loss, ipairs.v);
    else SC_FFT_DISPATCH(p.R1x, (rows_Ainv_resid_A<RR, NT>(l, p, img, wgt, a.weight_scalar, loss)))
    FFT_STAMP(8);
    fft_conv_middle<true, NT>(l, p, khat, stamps ? stamps + 8 : nullptr, []() {});
    SC_FFT_DISPATCH(p.R1x, (rows_Ainv_store<RR, NT>(l, p, gp)))
    FFT_STAMP(15);
    // loss of the plane
    loss = wave_sum(loss);
    if ((threadIdx.x & (SC_WAVE - 1)) == 0) red[threadIdx.x / SC_WAVE] = loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0;
        for (int w = 0; w < NT / SC_WAVE; ++w) r += red[w];
        a.loss_part[plane] = 0.5 * r;
    }
}

__global__ __launch_bounds__(SC_FFT_NT) void k_psf_conv(PsfArgs a, FftPlan p, float *G, long long *stamps_all)
{
    psf_conv_body<false, SC_FFT_NT>(a, p, G, stamps_all);
}
#ifndef SC_FFT_NT_X
#define SC_FFT_NT_X 1024
#endif
__global__ __launch_bounds__(SC_FFT_NT_X) void k_psf_conv_x128(PsfArgs a, FftPlan p, float *G, long long *stamps_all)
{
    psf_conv_body<true, SC_FFT_NT_X>(a, p, G, stamps_all);
}
