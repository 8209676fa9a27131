/*
 * scarlet_hip.h -- C ABI of the MI355X (gfx950) proximal-gradient deblending engine.
 *
 * Drop-in boundary for the hot path of scarlet's Blend.fit().
 * The reference's only native boundary is the pybind11 module built from
 * scarlet/operators_pybind11.cc; everything else on the path is numpy called from
 * Python.  This header therefore has three groups:
 *
 *   1. host-pointer drop-ins for the three pybind11 functions (same argument
 *      meaning, caller-owned host buffers mutated in place),
 *   2. batched device-pointer operators -- one call per (reference function x batch
 *      of arrays): these are what scarlet_amd/operator.py, update.py, measurement.py
 *      bind, i.e. what a maintainer would call from the reference's operator.py /
 *      update.py / measurement.py in place of the numpy code,
 *   3. the batched Blend.fit() engine (state struct + iteration driver).
 *
 * Conventions: plain C types only.  Device pointers are HIP device pointers into
 * memory owned by the caller (PyTorch-ROCm tensors in the Python host layer).  All
 * device arrays are C-contiguous float32 unless stated otherwise.  `stream` is a
 * hipStream_t passed as void*; every device entry point is asynchronous on it and
 * allocates nothing.  Return value: 0 = ok, negative = SCARLET_E_* (argument errors
 * are detected on the host before any launch).
 *
 * Threads and global state.  Every entry point may be called from several host threads at once
 * on different batches / streams; scarlet_last_error() is per thread.  The library keeps exactly
 * four process-wide objects, each behind its own mutex or atomic: the constant table of fast
 * FFT lengths (filled once per device), the hipFFT plan cache of the large-frame fallback of the
 * PSF path, the diagnostic switches of scarlet_set_option(), and the event recorder of
 * scarlet_profile_begin/end (one profiled region at a time, whichever batch launches inside it).
 * Per calling thread (and device) it keeps one further stream and three events: with more than eight
 * components per scene the Gram matrix and its eigenvalue run beside the morphology step, forked from and
 * joined back into the caller's stream by events -- to the caller the entry point stays asynchronous on
 * the stream it passed, and a stream capture records both branches.
 * Set-up and host-pointer entry points that need temporary device memory release it on every
 * exit path, error paths included.
 */
#ifndef SCARLET_HIP_H
#define SCARLET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCARLET_OK            0
#define SCARLET_E_ARG        -1   /* bad shape / null pointer / unsupported option  */
#define SCARLET_E_TOO_LARGE  -2   /* frame larger than 256 x 256                    */
#define SCARLET_E_HIP        -3   /* a HIP runtime call failed (see scarlet_last_error) */
#define SCARLET_E_NOTIMPL    -4   /* reference raises NotImplementedError here      */

/* BlendFlag bits -- scarlet/component.py:13-36 */
#define SCARLET_FLAG_SED_NOT_CONVERGED   1
#define SCARLET_FLAG_MORPH_NOT_CONVERGED 2
#define SCARLET_FLAG_EDGE_PIXELS         4
#define SCARLET_FLAG_NO_VALID_PIXELS     8

/* per-scene status bits written by the engine (0 = healthy) */
#define SCARLET_STATUS_CENTER_AT_EDGE    1  /* max_pixel window start < 0: the reference
                                               would raise (measurement.py:24-29)          */
#define SCARLET_STATUS_NONFINITE         2  /* NaN/Inf met in centroid or normalisation   */

/* symmetry algorithms -- scarlet/operator.py:291-350 */
#define SCARLET_SYM_KSPACE 0
#define SCARLET_SYM_SOFT   1
#define SCARLET_SYM_SDSS   2
/* or-ed into `algorithm`: apply the bare operator to the WHOLE array, symmetric about
 * index (H/2, W/2) -- operator.prox_soft/sdss/kspace_symmetry called directly -- instead of
 * going through uncentered_operator's window selection */
#define SCARLET_SYM_FULL_WINDOW 16

/* normalisation types -- scarlet/update.py:35-68 */
#define SCARLET_NORM_SED       0
#define SCARLET_NORM_MORPH     1
#define SCARLET_NORM_MORPH_MAX 2

const char *scarlet_version(void);
const char *scarlet_last_error(void);
/* Diagnostic switches: NO_EXACT, NO_KSCACHE, FUSED_V1, NO_FUSED, FORCE_BLOCK_UPDATE,
 * NO_HYBRID_SWEEP, PAD_LDS, STAMPS, PSF_HIPFFT, NO_BOX, NO_BOX2, NO_PSF3PASS, NO_SIDE_STREAM,
 * NO_GRAM_MFMA, NO_BIGK_FUSED, NO_PIPELINE, NO_PERSIST (one launch per iteration instead of k_fit2x),
 * PERSIST_DBG.  Each starts from the environment variable SCARLET_<NAME>, read once at first use;
 * afterwards only this call changes it.  Returns the previous value (0 / 1) or SCARLET_E_ARG for an
 * unknown name.  None changes results beyond float32 rounding.
 * PSF_HIPFFT and STAMPS decide the layout of a PSF batch's workspace: they are frozen by the first
 * scarlet_batch_workspace_bytes / scarlet_batch_prepare_psf call of the process on a batch with a
 * PSF; a later call that would change either returns SCARLET_E_ARG and changes nothing. */
int scarlet_set_option(const char *name, int value);
/* Diagnostics: with the STAMPS switch on, kernels that carry phase stamps (k_source_update_box: 16 shader-clock
 * values per component) write them into a buffer owned by the library; this copies up to `capacity` of
 * them to `out` (host) after a device synchronisation and returns the count (0 when the switch is off). */
int64_t scarlet_debug_stamps(int64_t *out, int64_t capacity);
/* 5-smooth fast FFT length (scipy.fftpack.next_fast_len as used by fft.py:99) */
int scarlet_next_fast_len(int n);

/* ------------------------------------------------------------------------------
 * 1. Host-pointer drop-ins for scarlet/operators_pybind11.cc
 * ---------------------------------------------------------------------------- */

/* replaces prox_monotonic (operators_pybind11.cc:11-25): x[d] = min(x[d],
 * x[ref_idx[d]]*(1-thresh)) for d in dist_idx order.  x: n doubles, mutated. */
int scarlet_host_prox_monotonic_f64(double *x, int n, const int *ref_idx,
                                    const int *dist_idx, int n_dist, double thresh);

/* replaces prox_weighted_monotonic<float> / <double> (operators_pybind11.cc:27-50,
 * 82-85).  weights: row-major [8][n] (numpy order), offsets: 8 flat neighbour
 * offsets, dist_idx: radius-sorted pixel indices without the peak.  x mutated. */
int scarlet_host_prox_weighted_monotonic_f32(float *x, int n, const float *weights,
                                             const int *offsets, const int *dist_idx,
                                             int n_dist, float thresh);
int scarlet_host_prox_weighted_monotonic_f64(double *x, int n, const double *weights,
                                             const int *offsets, const int *dist_idx,
                                             int n_dist, double thresh);

/* replaces apply_filter<float> / apply_filter<double> (operators_pybind11.cc:53-70; both overloads are
 * exported, :87-88): result = sum_n values[n] * shifted block of image. */
int scarlet_host_apply_filter_f32(const float *image, int H, int W, const float *values,
                                  const int *y_start, const int *y_end, const int *x_start,
                                  const int *x_end, int n, float *result);
int scarlet_host_apply_filter_f64(const double *image, int H, int W, const double *values,
                                  const int *y_start, const int *y_end, const int *x_start,
                                  const int *x_end, int n, double *result);

/* ------------------------------------------------------------------------------
 * 2. Batched device operators (n arrays of H x W, row stride W, array stride H*W)
 * ---------------------------------------------------------------------------- */

/* update.monotonic default path: operator.prox_strict_monotonic(use_nearest=False)
 * -> operators_pybind11.prox_weighted_monotonic (update.py:106-156, operator.py:81-122,
 * 540-621).  The radial cos-weights and the sweep order are generated on the fly from
 * `centers` (device int32 [n][2] = (y, x)); no weight table, no argsort.  In place. */
int scarlet_prox_weighted_monotonic(float *x, int n, int H, int W, const int32_t *centers,
                                    float thresh, void *stream);

/* operator.prox_strict_monotonic(use_nearest=True) (operator.py:104-113): reference pixel
 * = first neighbour with the largest cos-weight.  thresh must be 0 (ValueError in the
 * reference otherwise -> SCARLET_E_ARG). */
int scarlet_prox_nearest_monotonic(float *x, int n, int H, int W, const int32_t *centers,
                                   float thresh, void *stream);

/* operator.prox_uncentered_symmetry (operator.py:291-350) incl. uncentered_operator
 * window selection (:175-228), prox_soft_symmetry (:242-251), prox_sdss_symmetry
 * (:231-239), prox_kspace_symmetry (:253-288, evaluated as the equivalent real-space
 * Dirichlet-kernel operator).  shifts: device float64 [n][2] = (dy, dx)
 * or NULL; `algorithm` is applied as given (the caller resolves the reference's
 * "kspace -> soft when shift is None/0" rule, operator.py:337-339).  use_fill != 0
 * writes `fill` outside the symmetric window.  In place. */
int scarlet_prox_symmetry(float *x, int n, int H, int W, const int32_t *centers,
                          const double *shifts, int algorithm, float strength,
                          int use_fill, float fill, void *stream);

/* measurement.max_pixel (measurement.py:3-29): 5x5 window argmax, first hit row-major.
 * centers_io updated in place; status (device int32 [n] or NULL) gets
 * SCARLET_STATUS_CENTER_AT_EDGE or-ed in when the reference would have failed. */
int scarlet_max_pixel(const float *x, int n, int H, int W, int32_t *centers_io,
                      int32_t *status, void *stream);

/* measurement.psf_weighted_centroid (measurement.py:32-94).  psf: device float64 [P][P]
 * (P odd).  centers_io updated, shifts_out float64 [n][2]. */
int scarlet_psf_weighted_centroid(const float *x, int n, int H, int W, const double *psf,
                                  int P, int32_t *centers_io, double *shifts_out,
                                  int32_t *status, void *stream);

/* proxmin prox_plus / prox_hard / prox_soft as bound by update.py:13-32,71-82.
 * `count` contiguous floats; step per call (thresh*step is the cut). */
int scarlet_prox_plus(float *x, int64_t count, void *stream);
int scarlet_prox_hard(float *x, int64_t count, float thresh_times_step, void *stream);
int scarlet_prox_soft(float *x, int64_t count, float thresh_times_step, void *stream);

/* update.normalized (update.py:35-68) for n components: sed [n][B], morph [n][H*W]. */
int scarlet_normalize(float *sed, float *morph, int n, int B, int HW, int type, void *stream);

/* measurement.threshold (measurement.py:97-112), device part, for n arrays of `count` floats:
 * scarlet_log_range: out [n][3] float64 = {number of positive pixels, min, max of their log10};
 * scarlet_log_hist: np.histogram of log10(positive pixels) over nbins[i] <= 50 equal bins whose
 * edges [n][51] float64 the caller tabulates (np.linspace, as numpy does); hist [n][50] int32.
 * The caller picks the lower edge of the last empty bin (measurement.py:107-112). */
int scarlet_log_range(const float *x, int n, int64_t count, double *out, void *stream);
int scarlet_log_hist(const float *x, int n, int64_t count, const double *edges,
                     const int32_t *nbins, int32_t *hist, void *stream);
/* update.threshold (update.py:98): x[x < thresh] = 0. */
int scarlet_cut_below(float *x, int64_t count, double thresh, void *stream);
/* bbox.trim (bbox.py:174-193) for n planes [H][W]: box [n][4] int32 = {bottom, top, left, right} of
 * x > min_value (inclusive bounds); {H, -1, W, -1} when no pixel qualifies. */
int scarlet_trim(const float *x, int n, int H, int W, float min_value, int32_t *box, void *stream);
/* interpolation.fft_resample (interpolation.py:408-448) as used by update.translation
 * (update.py:159-167), n planes [H][W], out of place: separable taps [n][2][12] float64 (ky, kx:
 * ny / nx <= 12 of them used), first tap positions win0 [n][2] int32 (the kernels' window[0]). */
int scarlet_resample(const float *in, float *out, int n, int H, int W, const double *taps,
                     const int32_t *win0, int ny, int nx, void *stream);

/* apply_filter on device (operators_pybind11.cc:53-70), one image. */
int scarlet_apply_filter(const float *image, int H, int W, const float *values,
                         const int32_t *y_start, const int32_t *y_end, const int32_t *x_start,
                         const int32_t *x_end, int n, float *result, void *stream);

/* fft.match_psfs (fft.py:282-301) for n PSFs on the device: out[i] = the difference kernel that
   turns psf2[i] (or psf2[0] when n2 == 1) into psf1[i]: ratio of the spectra at the reference's
   FFT shape next_fast_len(P1 + P2 + 3) (last axis even), cropped to psf1's shape.
   psf1 [n][P1y][P1x], psf2 [n2][P2y][P2x], out [n][P1y][P1x], all device float32. */
int scarlet_match_psfs(const float *psf1, int n, int P1y, int P1x, const float *psf2, int n2,
                       int P2y, int P2x, float *out, void *stream);

/* ------------------------------------------------------------------------------
 * 3. Batched Blend.fit() engine (blend.py:65-223, source.py:402-440)
 *
 * Supported shapes: K <= 32 components per scene, B <= 8 bands, frames up to 256 x 256,
 * with or without a PSF difference kernel.  Which kernels run is an internal choice:
 * one fused launch per iteration when the K morphology tiles fit LDS (H, W <= 64), the
 * four-kernel general path otherwise, chunked gradient passes for K > 8, operators in
 * place in HBM for frames beyond the LDS tile.  Results do not depend on the choice
 * beyond float32 rounding.
 * ---------------------------------------------------------------------------- */

typedef struct scarlet_batch {
    /* shapes: S scenes, K components per scene, B bands, H x W pixels */
    int32_t S, K, B, H, W;
    /* data (read only) */
    const float *images;      /* [S][B][H][W]                                          */
    const float *weights;     /* [S][B][H][W] or NULL -> scalar `weight_scalar`
                                 (observation.py:148-151)                              */
    float weight_scalar;
    /* factors, ping-pong: buffer `cur` holds the current values, the other one the
       values of the previous iteration (_last_sed/_last_morph, blend.py:179-182)      */
    float *sed[2];            /* [S][K][B]                                             */
    float *morph[2];          /* [S][K][H][W]                                          */
    int32_t *cur;             /* [S] device: index of each scene's current buffer; a scene's
                                 index flips once per iteration it takes part in        */
    /* per component */
    int32_t *centers;         /* [S][K][2] pixel_center (y, x)                         */
    double *shifts;           /* [S][K][2] sub-pixel shift from the last centroid      */
    int32_t *flags;           /* [S][K]    BlendFlag bits                              */
    const uint8_t *fix_sed;   /* [S][K] or NULL (component.py:111-112)                 */
    const uint8_t *fix_morph; /* [S][K] or NULL                                        */
    /* per scene */
    double *lipschitz;        /* [S][2]  (L_sed, L_morph) of the last iteration        */
    double *mse;              /* [S][mse_capacity] loss before each step (blend.py:138) */
    int32_t mse_capacity;
    int32_t *it;              /* [S] = len(mse)                                        */
    int32_t *active;          /* [S] 1 while the scene has not met e_rel in this fit()  */
    int32_t *status;          /* [S] SCARLET_STATUS_* bits                             */
    /* constraint pipeline of PointSource/ExtendedSource.update (source.py:402-440)    */
    int32_t symmetric, monotonic;
    float l0_thresh, l1_thresh;  /* < 0 -> off; else update.sparse_l0/l1 before positive */
    const double *centroid_psf;  /* [P][P] float64 centroid weight (source.py:483-490) */
    int32_t centroid_P;
    /* PSF difference kernel of Observation.match (observation.py:191-194): NULL (render =
       identity), [B][psf_h][psf_w] shared by all scenes (diff_kernel_per_scene = 0), or one
       set per scene [S][B][psf_h][psf_w] (diff_kernel_per_scene = 1: every scene was observed
       with its own PSFs).  When set, call scarlet_batch_prepare_psf() once before fitting
       (and again if it changes).                                                         */
    const float *diff_kernel;
    int32_t psf_h, psf_w;
    int32_t diff_kernel_per_scene;
    /* workspace owned by the caller: scarlet_batch_workspace_bytes() bytes, ZEROED before
       the first call (it also holds a cache keyed by a magic word)                     */
    void *workspace;
    /* MultiComponentSource (source.py:538-641): [S][K] device int32 or NULL.  -1: the component is a
       source of its own (PointSource / ExtendedSource.update, source.py:402-440); g >= 0: it is a layer of
       multi-component source g of its scene -- the layers of a source share ONE centre, measured on their
       flux-weighted sum (max_pixel, and psf_weighted_centroid every fifth iteration, source.py:613-630),
       and are symmetrised about it with shift = None (update.symmetric falls back to soft symmetry,
       operator.py:337-339).  The members of a source must be adjacent components.             */
    const int32_t *group;
} scarlet_batch;

/* bytes of device workspace needed for `b` (depends on S, K, B, H, W only) */
int64_t scarlet_batch_workspace_bytes(const scarlet_batch *b);
/* Number of pipelines scarlet_fit() runs `b` as: 2 for a large batch with a PSF (>= 1024 scenes, K <= 8, the
 * LDS-resident transform) -- the two halves of the batch as views with their own workspace regions, the second on the
 * calling thread's second stream, joined back into the caller's stream before scarlet_fit() returns or synchronises,
 * so that one half's convolution (latency-bound) runs beside the other half's streaming passes; results are
 * bit-identical to one pipeline (scenes are independent).  1 otherwise, and with the NO_PIPELINE switch. */
int scarlet_batch_pipelines(const scarlet_batch *b);

/* Run up to `max_iter` proximal-gradient iterations on every active scene
 * (Blend.fit, blend.py:65-102).  Per iteration and scene: loss + analytic gradient
 * (_backward/_loss, :105-139), Lipschitz constants (_set_lipschitz, :186-223, exact or
 * approximate), gradient step (:87-96), per-component constraint pipeline
 * (source.py:402-440), convergence flags (_check_convergence, :141-184).  A scene that
 * converges stops iterating (its `active` becomes 0), exactly like the reference's
 * `break`.  `check_every` > 0: every that many iterations the active flags are copied
 * to the host (one stream sync) to stop early when every scene is done; 0: never sync.
 * Returns the number of iterations launched (>= 0) or an error. */
int scarlet_fit(scarlet_batch *b, int max_iter, double e_rel, int approximate_L,
                int check_every, void *stream);

/* Blend.fit with SEVERAL observations (blend.py:24-43, 120-139, 219-220), no host synchronisation per
 * iteration.  `state` holds the factors over the model frame's C = state->B channels (its `images` are not
 * read: pass any buffer of the right size); obs[i] (n_obs <= 8) is a batch over the channels band0[i] ..
 * band0[i] + obs[i]->B - 1 with its own images / weights / PSF kernel / workspace and the same S, K, H, W --
 * its factor buffers are scratch of this call.  Per iteration: every observation's loss gradient
 * (scarlet_backward_gradients), their sum, L * n_obs (exact or approximate), the step, the constraint
 * pipeline and the convergence test on `state`.  Returns the number of iterations launched. */
int scarlet_fit_multi(scarlet_batch *state, scarlet_batch *const *obs, const int32_t *band0, int n_obs,
                      int max_iter, double e_rel, int approximate_L, int check_every, void *stream);

/* Single phases, exposed for tests and for Python-overridden update() methods:        */
/* _backward + _set_lipschitz + gradient step (blend.py:81-96): reads buffer cur, writes
 * the stepped factors into buffer 1-cur; cur/it are NOT advanced yet                   */
int scarlet_backward_step(scarlet_batch *b, int approximate_L, void *stream);

/* As scarlet_backward_step, but buffer 1-cur receives the GRADIENTS of the loss -- d loss/d sed
   [S][K][B] and d loss/d morph [S][K][H*W] (Blend._backward, blend.py:105-118) -- instead of the
   stepped factors; `lipschitz` and `mse` are written as usual, fix_sed / fix_morph are not applied.
   For callers that combine gradients themselves: several observations per blend (blend.py:136-137,
   219-220) and Prior hooks. */
int scarlet_backward_gradients(scarlet_batch *b, int approximate_L, void *stream);
/* the built-in constraint pipeline (source.py:402-440).  in_iteration=1: on buffer 1-cur
 * (between backward_step and check_convergence); 0: on buffer cur with it=0 semantics,
 * as the source constructors do (source.py:400,492)                                    */
int scarlet_source_update(scarlet_batch *b, int in_iteration, void *stream);
/* _check_convergence (blend.py:141-184) on buffer 1-cur vs cur, then closes the
 * iteration: it += 1, cur flips, active cleared for converged scenes                   */
int scarlet_check_convergence(scarlet_batch *b, double e_rel, void *stream);

/* Per-kernel timing of scarlet_fit with hipEvents recorded on the launch stream (used by
 * bench.py for the roofline line).  begin: allocate events for up to max_iterations
 * iterations and start recording; end: synchronise, return per kernel class
 * {0 k_grad, 1 k_step, 2 k_source_update, 3 k_converge, 4 k_iterate (fused), 5-7 unused}
 * the summed milliseconds and launch counts, and stop recording. */
/* diagnostics (STAMPS switch on): byte offset inside b->workspace of the convolution kernel's phase stamps
 * ([S][B][32] int64 shader-clock values, written by every k_psf_conv launch), or -1 */
int64_t scarlet_debug_psf_stamps_offset(const scarlet_batch *b);
/* diagnostics: plan of the LDS-resident convolution (fftconv.h) for this batch: {H, W, Fy, Fx, M, RS, R1y, R2y, R1x, R2x,
 * oky, okx, image staged in LDS, exact-shape instance, LDS bytes, 0}; 0, or -1 when the batch takes another path.
 * Host-only: no device call. */
int scarlet_debug_psf_plan(const scarlet_batch *b, int32_t *out16);
int scarlet_profile_begin(int max_iterations);
int scarlet_profile_end(double total_ms[8], int64_t launches[8]);
/* the same with both counts: `iterations` = iterations covered by the class's launches (what
 * scarlet_profile_end reports as `launches`: a k_fit2x launch covers several), `launches` = kernel
 * launches actually recorded.  Either array may be NULL. */
int scarlet_profile_end_ex(double total_ms[8], int64_t iterations[8], int64_t launches[8]);

/* ExtendedSource initialisation on device (source.py:139-180, rank f1 of SURVEY 8f):
 * per component: pixel SED (optionally PSF-corrected by the caller through sed_scale
 * [B] or NULL), detection coadd, sdss symmetry, thresh=0.1 weighted monotone sweep,
 * cut at bg_cutoff, divide by the centre pixel.  Writes sed/morph of buffer b->cur,
 * flags (NO_VALID_PIXELS when nothing is above the cut), then runs the constraint
 * pipeline once with it=0 as the constructor does (source.py:492) when run_update != 0.
 * init_symmetric / init_monotonic: the `symmetric` / `monotonic` arguments of
 * init_extended_source (ExtendedSource always passes symmetric=True).  bg_rms: host [B]. */
int scarlet_init_extended(scarlet_batch *b, const float *bg_rms_host, float thresh,
                          const float *sed_scale_host, int init_symmetric, int init_monotonic,
                          int run_update, void *stream);

/* Row a3b set-up: FFT the difference kernel into the workspace (K-hat at the reference's FFT
 * shape next_fast_len(N + P + 3), fft.py:68-106) and create the batched hipFFT plans (cached
 * per shape inside the library; the only allocation the library makes). */
int scarlet_batch_prepare_psf(scarlet_batch *b, void *stream);

/* Observation.render / fft.convolve (observation.py:198-220, fft.py:304-317) for n planes:
 * out[p] = crop(model[p] (*) kernel[p or 0]) with the reference's pad / shift / crop
 * conventions.  model, out: device [n][H][W]; kernel: device [nk][Py][Px] with nk == n or 1.
 * Allocates temporary FFT buffers (set-up / test helper, not on the iteration path). */
int scarlet_convolve_same(const float *model, int n, int H, int W, const float *kernel, int nk,
                          int Py, int Px, float *out, void *stream);

/* Convergence sums (blend.py:159-171) of buffer 1-cur vs cur for every component, for
 * callers that ran their own update() between scarlet_backward_step and
 * scarlet_check_convergence (scarlet_source_update computes them itself). */
int scarlet_convergence_sums(scarlet_batch *b, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SCARLET_HIP_H */
