// scarlet_hip.hip -- C ABI (include/scarlet_hip.h) of the gfx950 deblending engine.
//
// Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared scarlet_hip.hip -o libscarlet_hip.so
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <atomic>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
#include <algorithm>
#include <hipfft/hipfft.h>

#include "common.h"
#include "prox_ops.h"
#include "engine.h"
#include "fused.h"
#include "fused2.h"
#include "bigk.h"
#include "boxupdate.h"
#include "psf_path.h"
#include "fftconv.h"
#include "extras.h"

__constant__ unsigned short sc_nfl_table[SC_NFL_MAX];

static thread_local char g_err[512] = "";
static int set_err(int code, const char *msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            snprintf(g_err, sizeof(g_err), "%s failed: %s (%s:%d)", #expr,               \
                     hipGetErrorString(e_), __FILE__, __LINE__);                         \
            return SCARLET_E_HIP;                                                        \
        }                                                                                \
    } while (0)

extern "C" const char *scarlet_version(void) { return "scarlet_amd-hip 0.2 (gfx950)"; }

// ---- diagnostic switches: process-wide integers, initialised ONCE from the
// environment (SCARLET_<NAME>) at first use and changed afterwards only through scarlet_set_option.
// None of them changes results beyond float32 rounding.
enum { OPT_NO_EXACT = 0, OPT_NO_KSCACHE, OPT_FUSED_V1, OPT_NO_FUSED, OPT_FORCE_BLOCK_UPDATE, OPT_NO_HYBRID_SWEEP,
       OPT_PAD_LDS, OPT_STAMPS, OPT_PSF_HIPFFT, OPT_NO_BOX, OPT_NO_BOX2, OPT_NO_PSF3PASS, OPT_NO_SIDE_STREAM, OPT_NO_GRAM_MFMA, OPT_NO_BIGK_FUSED, OPT_NO_PIPELINE, OPT_NO_PERSIST, OPT_PERSIST_DBG, OPT_COUNT };
static const char *const g_opt_names[OPT_COUNT] = {"NO_EXACT", "NO_KSCACHE", "FUSED_V1", "NO_FUSED", "FORCE_BLOCK_UPDATE",
                                                   "NO_HYBRID_SWEEP", "PAD_LDS", "STAMPS", "PSF_HIPFFT", "NO_BOX", "NO_BOX2",
                                                   "NO_PSF3PASS", "NO_SIDE_STREAM", "NO_GRAM_MFMA", "NO_BIGK_FUSED", "NO_PIPELINE", "NO_PERSIST", "PERSIST_DBG"};
static std::atomic<int> g_opt[OPT_COUNT];
static std::once_flag g_opt_once;
static void options_init(void)
{
    std::call_once(g_opt_once, [] {
        for (int i = 0; i < OPT_COUNT; ++i) {
            char name[64];
            snprintf(name, sizeof(name), "SCARLET_%s", g_opt_names[i]);
            const char *v = getenv(name);
            int val = 0;
            if (v) { val = atoi(v); if (val == 0 && v[0] != '0') val = 1; }    // "yes", "" ... count as on
            if (v && !v[0]) val = 1;
            g_opt[i].store(val, std::memory_order_relaxed);
        }
    });
}
static inline int opt(int which) { options_init(); return g_opt[which].load(std::memory_order_relaxed); }
// PSF_HIPFFT and STAMPS decide the LAYOUT of a PSF batch's workspace (psf_layout): they are read when a workspace is
// sized and again by every later call on that batch, so a change in between would move regions under a live batch
// (K-hat and tables read from the wrong offsets, writes past the allocation).  The first layout computed in the
// process therefore FREEZES the two switches: scarlet_set_option then refuses a different value (SCARLET_E_ARG).
static std::atomic<bool> g_layout_frozen{false};
extern "C" int scarlet_set_option(const char *name, int value)
{
    options_init();
    if (!name) return set_err(SCARLET_E_ARG, "null option name");
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcmp(name, g_opt_names[i])) {
            if ((i == OPT_PSF_HIPFFT || i == OPT_STAMPS) && g_layout_frozen.load() && (g_opt[i].load() != 0) != (value != 0))
                return set_err(SCARLET_E_ARG, "PSF_HIPFFT / STAMPS fix the workspace layout of PSF batches: they cannot change "
                                              "after the first PSF workspace of the process was sized");
            return g_opt[i].exchange(value) != 0 ? 1 : 0;
        }
    return set_err(SCARLET_E_ARG, "unknown option");
}

// device allocation released on every exit path of the set-up / host-pointer entry points
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    template <typename T> T *as() const { return (T *)p; }
};
#define DEV_ALLOC(buf, bytes) HIP_TRY(hipMalloc(&(buf).p, (bytes)))
extern "C" const char *scarlet_last_error(void) { return g_err; }

// scipy.fftpack.next_fast_len: smallest 2^a 3^b 5^c >= n
extern "C" int scarlet_next_fast_len(int n)
{
    if (n <= 1) return 1;
    long best = -1;
    for (long p5 = 1; p5 < 2L * n; p5 *= 5)
        for (long p35 = p5; p35 < 2L * n; p35 *= 3) {
            long v = p35;
            while (v < n) v *= 2;
            if (best < 0 || v < best) best = v;
        }
    return (int)best;
}

static int ensure_tables(void)
{
    static bool done[64] = {false};
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return set_err(SCARLET_E_HIP, "device index out of range");
    if (done[dev]) return SCARLET_OK;
    std::vector<unsigned short> t(SC_NFL_MAX);
    for (int i = 0; i < SC_NFL_MAX; ++i) t[i] = (unsigned short)scarlet_next_fast_len(i);
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(sc_nfl_table), t.data(), SC_NFL_MAX * sizeof(unsigned short)));
    done[dev] = true;
    return SCARLET_OK;
}

// diagnostics (STAMPS switch): a device buffer of shader-clock stamps owned by the library, read back with
// scarlet_debug_stamps; NULL when the switch is off
static long long *g_dbg_stamps = nullptr;
static size_t g_dbg_count = 0;
static std::mutex g_dbg_mu;
static long long *debug_stamps(size_t count)
{
    if (!opt(OPT_STAMPS)) return nullptr;
    std::lock_guard<std::mutex> lock(g_dbg_mu);
    if (count > g_dbg_count) {
        if (g_dbg_stamps) (void)hipFree(g_dbg_stamps);
        g_dbg_stamps = nullptr; g_dbg_count = 0;
        if (hipMalloc((void **)&g_dbg_stamps, count * sizeof(long long)) != hipSuccess) return nullptr;
        g_dbg_count = count;
    }
    (void)hipMemset(g_dbg_stamps, 0, g_dbg_count * sizeof(long long));
    return g_dbg_stamps;
}
extern "C" int64_t scarlet_debug_stamps(int64_t *out, int64_t capacity)
{
    std::lock_guard<std::mutex> lock(g_dbg_mu);
    if (!g_dbg_stamps || !out || capacity <= 0) return 0;
    const int64_t n = capacity < (int64_t)g_dbg_count ? capacity : (int64_t)g_dbg_count;
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(out, g_dbg_stamps, n * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}

// compute units of the current device (cached per device index)
static int device_cu_count(void)
{
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        hipDeviceProp_t prop;
        n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// LDS bytes of the per-component update kernel for an H x W image
static size_t update_lds_bytes(int H, int W)
{
    return sizeof(float) * ((size_t)H * tile_stride(W) + symmetry_lds_floats(H, W));
}
static const size_t LDS_LIMIT = 160 * 1024 - 1024;   // leave room for static __shared__

template <typename Kern>
static int allow_lds(Kern k, size_t bytes)
{
    if (bytes > LDS_LIMIT) return set_err(SCARLET_E_TOO_LARGE, "image tile does not fit in LDS");
    if (bytes > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return SCARLET_OK;
}

// =====================================================================================
// 2. batched device operators
// =====================================================================================
enum { OP_MONO_WEIGHTED = 0, OP_MONO_NEAREST = 1, OP_SYMMETRY = 2, OP_MAX_PIXEL = 3, OP_CENTROID = 4 };

struct OpArgs {
    float *x; int n, H, W;
    int *centers; double *shifts; int *status;
    int op, algorithm, use_fill;
    float thresh, strength, fill;
    const double *psf; int P;
};

// GT: arrays whose tile does not fit LDS (up to 256 x 256) are processed in place in HBM, the GEMM
// scratch of the k-space symmetry in `gscratch` (see k_source_update<2> in engine.h)
template <bool GT>
__global__ __launch_bounds__(SC_BLOCK) void k_operator(OpArgs a, float *gscratch)
{
    extern __shared__ __align__(16) float lds[];
    const int c = blockIdx.x, H = a.H, W = a.W, HW = H * W;
    const int hp = round16(H), wp = round16(W);
    float *g = a.x + (size_t)c * HW;
    Tile t; t.H = H; t.W = W;
    float *scr, *av;
    if (GT) { t.LW = W; t.m = g; scr = gscratch + (size_t)c * hp * scratch_stride(wp); av = lds; }
    else    { t.LW = tile_stride(W); t.m = lds; scr = lds + H * t.LW; av = scr + hp * scratch_stride(wp); }
    float *bv = av + 2 * hp, *cv = bv + 2 * wp, *zv = cv + 2 * wp;
    float *stage = GT ? zv + wp : nullptr;
    __shared__ double red[SC_NWAVES];
    __shared__ int ctr[2];
    __shared__ double shf[2];
    __shared__ int stat;
    if (!GT)
        for (int i = threadIdx.x; i < HW; i += SC_BLOCK) t.m[(i / W) * t.LW + (i % W)] = g[i];
    if (threadIdx.x == 0) stat = 0;
    __syncthreads();
    const int cy = a.centers[2 * c], cx = a.centers[2 * c + 1];
    if (cy < 0 || cy >= H || cx < 0 || cx >= W) {
        // a centre outside the array (the reference raises IndexError): leave the array alone, flag it
        if (threadIdx.x == 0 && a.status) atomicOr(&a.status[c], SCARLET_STATUS_CENTER_AT_EDGE);
        return;
    }
    bool writeback = !GT;
    switch (a.op) {
    case OP_MONO_WEIGHTED: monotonic_tile<false, float>(t, cy, cx, a.thresh); break;
    case OP_MONO_NEAREST:  monotonic_tile<true, float>(t, cy, cx, a.thresh); break;
    case OP_SYMMETRY: {
        const double dy = a.shifts ? a.shifts[2 * c] : 0.0, dx = a.shifts ? a.shifts[2 * c + 1] : 0.0;
        symmetry_tile(t, cy, cx, a.algorithm, a.strength, dy, dx, a.use_fill != 0, a.fill,
                      scr, av, bv, cv, zv, stage, GT);
        break;
    }
    case OP_MAX_PIXEL:
        max_pixel_tile(t, cy, cx, ctr, &stat);
        if (threadIdx.x == 0) { a.centers[2 * c] = ctr[0]; a.centers[2 * c + 1] = ctr[1]; }
        writeback = false;
        break;
    case OP_CENTROID:
        centroid_tile(t, a.psf, a.P, cy, cx, red, ctr, shf, &stat);
        if (threadIdx.x == 0) {
            a.centers[2 * c] = ctr[0]; a.centers[2 * c + 1] = ctr[1];
            a.shifts[2 * c] = shf[0]; a.shifts[2 * c + 1] = shf[1];
        }
        writeback = false;
        break;
    }
    __syncthreads();
    if (writeback)
        for (int i = threadIdx.x; i < HW; i += SC_BLOCK) g[i] = t.m[(i / W) * t.LW + (i % W)];
    if (threadIdx.x == 0 && stat && a.status) atomicOr(&a.status[c], stat);
}

// the same operators, one wave per array (wave_ops.h), H, W <= 64
__global__ __launch_bounds__(SC_BLOCK) void k_operator_w(OpArgs a)
{
    extern __shared__ __align__(16) float lds[];
    const int wid = threadIdx.x / SC_WAVE, lane = threadIdx.x & (SC_WAVE - 1);
    const int c = blockIdx.x * SC_NWAVES + wid;
    if (c >= a.n) return;
    const int H = a.H, W = a.W, HW = H * W;
    Tile t; t.H = H; t.W = W; t.LW = tile_stride(W);
    t.m = lds + (size_t)wid * (H * t.LW + SC_WAVE_VEC_FLOATS);
    float *vec = t.m + H * t.LW;
    float *g = a.x + (size_t)c * HW;
    for (int i = lane; i < HW; i += SC_WAVE) t.m[(i / W) * t.LW + (i % W)] = g[i];
    wave_sync();
    int cy = a.centers[2 * c], cx = a.centers[2 * c + 1];
    int stat = 0;
    bool writeback = true;
    if (cy < 0 || cy >= H || cx < 0 || cx >= W) {         // as in k_operator
        if (lane == 0 && a.status) atomicOr(&a.status[c], SCARLET_STATUS_CENTER_AT_EDGE);
        return;
    }
    switch (a.op) {
    case OP_MONO_WEIGHTED: wave_monotonic<float>(t, cy, cx, a.thresh); break;
    case OP_SYMMETRY: {
        const double dy = a.shifts ? a.shifts[2 * c] : 0.0, dx = a.shifts ? a.shifts[2 * c + 1] : 0.0;
        wave_symmetry(t, cy, cx, a.algorithm, a.strength, dy, dx, a.use_fill != 0, a.fill, vec);
        break;
    }
    case OP_MAX_PIXEL:
        wave_max_pixel(t, cy, cx, stat);
        if (lane == 0) { a.centers[2 * c] = cy; a.centers[2 * c + 1] = cx; }
        writeback = false;
        break;
    case OP_CENTROID: {
        double dy = 0, dx = 0;
        wave_centroid(t, a.psf, a.P, cy, cx, dy, dx, stat);
        if (lane == 0) {
            a.centers[2 * c] = cy; a.centers[2 * c + 1] = cx;
            a.shifts[2 * c] = dy; a.shifts[2 * c + 1] = dx;
        }
        writeback = false;
        break;
    }
    }
    wave_sync();
    if (writeback)
        for (int i = lane; i < HW; i += SC_WAVE) g[i] = t.m[(i / W) * t.LW + (i % W)];
    if (lane == 0 && stat && a.status) atomicOr(&a.status[c], stat);
}

static int launch_operator(OpArgs a, void *stream)
{
    if (!a.x || a.n < 0 || a.H <= 0 || a.W <= 0 || a.W > 256 || !a.centers)
        return set_err(SCARLET_E_ARG, "bad operator arguments");
    if (a.n == 0) return SCARLET_OK;
    int rc = ensure_tables();
    if (rc) return rc;
    if (a.H <= 64 && a.W <= 64 && a.op != OP_MONO_NEAREST && !opt(OPT_FORCE_BLOCK_UPDATE)) {
        const size_t lds = sizeof(float) * SC_NWAVES * ((size_t)a.H * tile_stride(a.W) + SC_WAVE_VEC_FLOATS);
        rc = allow_lds(k_operator_w, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k_operator_w, dim3((a.n + SC_NWAVES - 1) / SC_NWAVES), dim3(SC_BLOCK), lds,
                           (hipStream_t)stream, a);
    } else if (update_lds_bytes(a.H, a.W) <= LDS_LIMIT) {
        const size_t lds = update_lds_bytes(a.H, a.W);
        rc = allow_lds(k_operator<false>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k_operator<false>, dim3(a.n), dim3(SC_BLOCK), lds, (hipStream_t)stream, a, (float *)nullptr);
    } else {
        if (a.H > 256) return set_err(SCARLET_E_TOO_LARGE, "arrays larger than 256 x 256 are not supported");
        DevBuf gscratch;                      // released on every path, after the kernel has finished
        if (a.op == OP_SYMMETRY)
            DEV_ALLOC(gscratch, sizeof(float) * (size_t)a.n * round16(a.H) * scratch_stride(round16(a.W)));
        const size_t lds = sizeof(float) * (2 * round16(a.H) + 5 * round16(a.W) + stage_floats(round16(a.H), round16(a.W)));
        hipLaunchKernelGGL(k_operator<true>, dim3(a.n), dim3(SC_BLOCK), lds, (hipStream_t)stream, a, gscratch.as<float>());
        HIP_TRY(hipGetLastError());
        if (gscratch.p) HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    }
    HIP_TRY(hipGetLastError());
    return SCARLET_OK;
}

extern "C" int scarlet_prox_weighted_monotonic(float *x, int n, int H, int W, const int32_t *centers,
                                               float thresh, void *stream)
{
    OpArgs a = {};
    a.x = x; a.n = n; a.H = H; a.W = W; a.centers = (int *)centers; a.op = OP_MONO_WEIGHTED;
    a.thresh = thresh;
    return launch_operator(a, stream);
}

extern "C" int scarlet_prox_nearest_monotonic(float *x, int n, int H, int W, const int32_t *centers,
                                              float thresh, void *stream)
{
    if (thresh != 0.f)   // operator.py:107-110 raises ValueError
        return set_err(SCARLET_E_ARG, "Thresholding does not work with nearest neighbor monotonicity");
    OpArgs a = {};
    a.x = x; a.n = n; a.H = H; a.W = W; a.centers = (int *)centers; a.op = OP_MONO_NEAREST;
    return launch_operator(a, stream);
}

extern "C" int scarlet_prox_symmetry(float *x, int n, int H, int W, const int32_t *centers,
                                     const double *shifts, int algorithm, float strength,
                                     int use_fill, float fill, void *stream)
{
    const int base_alg = algorithm & ~SCARLET_SYM_FULL_WINDOW;
    if (base_alg < 0 || base_alg > 2) return set_err(SCARLET_E_ARG, "algorithm must be one of 'soft', 'sdss', 'kspace'");
    if (base_alg == SCARLET_SYM_KSPACE && !shifts) return set_err(SCARLET_E_ARG, "kspace symmetry needs shifts");
    OpArgs a = {};
    a.x = x; a.n = n; a.H = H; a.W = W; a.centers = (int *)centers; a.shifts = (double *)shifts;
    a.op = OP_SYMMETRY; a.algorithm = algorithm; a.strength = strength; a.use_fill = use_fill; a.fill = fill;
    return launch_operator(a, stream);
}

extern "C" int scarlet_max_pixel(const float *x, int n, int H, int W, int32_t *centers_io,
                                 int32_t *status, void *stream)
{
    OpArgs a = {};
    a.x = (float *)x; a.n = n; a.H = H; a.W = W; a.centers = centers_io; a.status = status; a.op = OP_MAX_PIXEL;
    return launch_operator(a, stream);
}

extern "C" int scarlet_psf_weighted_centroid(const float *x, int n, int H, int W, const double *psf,
                                             int P, int32_t *centers_io, double *shifts_out,
                                             int32_t *status, void *stream)
{
    if (!psf || P <= 0 || !(P & 1) || !shifts_out) return set_err(SCARLET_E_ARG, "bad centroid arguments");
    OpArgs a = {};
    a.x = (float *)x; a.n = n; a.H = H; a.W = W; a.centers = centers_io; a.shifts = shifts_out;
    a.status = status; a.op = OP_CENTROID; a.psf = psf; a.P = P;
    return launch_operator(a, stream);
}

// ---- elementwise prox (proxmin semantics pinned by the reference's tests/test_update.py)
enum { EW_PLUS = 0, EW_HARD = 1, EW_SOFT = 2 };
__global__ void k_elementwise(float *x, int64_t count, int op, float t)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x) {
        float v = x[i];
        if (op == EW_PLUS) { if (v < 0.f) v = 0.f; }
        else if (op == EW_HARD) { if (fabsf(v) < t) v = 0.f; }
        else {
            const float mag = fabsf(v) - t;
            v = (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f)) * (mag < 0.f ? 0.f : mag);
        }
        x[i] = v;
    }
}
static int launch_elementwise(float *x, int64_t count, int op, float t, void *stream)
{
    if (count < 0 || (count > 0 && !x)) return set_err(SCARLET_E_ARG, "bad elementwise arguments");
    if (count == 0) return SCARLET_OK;
    int64_t blocks = (count + SC_BLOCK - 1) / SC_BLOCK;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_elementwise, dim3((unsigned)blocks), dim3(SC_BLOCK), 0, (hipStream_t)stream, x, count, op, t);
    HIP_TRY(hipGetLastError());
    return SCARLET_OK;
}
extern "C" int scarlet_prox_plus(float *x, int64_t count, void *stream) { return launch_elementwise(x, count, EW_PLUS, 0.f, stream); }
extern "C" int scarlet_prox_hard(float *x, int64_t count, float t, void *stream) { return launch_elementwise(x, count, EW_HARD, t, stream); }
extern "C" int scarlet_prox_soft(float *x, int64_t count, float t, void *stream) { return launch_elementwise(x, count, EW_SOFT, t, stream); }

// ---- update.normalized (update.py:35-68): one workgroup per component
__global__ __launch_bounds__(SC_BLOCK) void k_normalize(float *sed, float *morph, int B, int HW, int type)
{
    __shared__ double red[SC_NWAVES];
    __shared__ float redf[SC_NWAVES];
    const int c = blockIdx.x;
    float *m = morph + (size_t)c * HW, *s = sed + (size_t)c * B;
    float norm;
    if (type == SCARLET_NORM_MORPH_MAX) {
        float vmax = -INFINITY; bool anynan = false;
        for (int i = threadIdx.x; i < HW; i += SC_BLOCK) { const float v = m[i]; anynan |= (v != v); vmax = fmaxf(vmax, v); }
        norm = block_max_nan(vmax, anynan, redf);
    } else if (type == SCARLET_NORM_MORPH) {
        double acc = 0;
        for (int i = threadIdx.x; i < HW; i += SC_BLOCK) acc += (double)m[i];
        norm = (float)block_sum(acc, red);
    } else {
        double acc = 0;
        for (int i = threadIdx.x; i < B; i += SC_BLOCK) acc += (double)s[i];
        norm = (float)block_sum(acc, red);
    }
    __syncthreads();
    const bool sed_type = (type == SCARLET_NORM_SED);
    for (int i = threadIdx.x; i < HW; i += SC_BLOCK) m[i] = sed_type ? m[i] * norm : m[i] / norm;
    for (int i = threadIdx.x; i < B; i += SC_BLOCK) s[i] = sed_type ? s[i] / norm : s[i] * norm;
}
extern "C" int scarlet_normalize(float *sed, float *morph, int n, int B, int HW, int type, void *stream)
{
    if (type < 0 || type > 2) return set_err(SCARLET_E_ARG, "Unrecognized normalization");
    if (!sed || !morph || n < 0 || B <= 0 || HW <= 0) return set_err(SCARLET_E_ARG, "bad normalize arguments");
    if (n == 0) return SCARLET_OK;
    hipLaunchKernelGGL(k_normalize, dim3(n), dim3(SC_BLOCK), 0, (hipStream_t)stream, sed, morph, B, HW, type);
    HIP_TRY(hipGetLastError());
    return SCARLET_OK;
}

// ---- measurement.threshold / update.threshold / bbox.trim / update.translation (extras.h)
extern "C" int scarlet_log_range(const float *x, int n, int64_t count, double *out, void *stream)
{
    if (n < 0 || count <= 0 || !x || !out) return set_err(SCARLET_E_ARG, "bad log_range arguments");
    if (n == 0) return SCARLET_OK;
    hipLaunchKernelGGL(k_log_range, dim3(n), dim3(SC_BLOCK), 0, (hipStream_t)stream, x, count, out);
    HIP_TRY(hipGetLastError());
    return SCARLET_OK;
}
extern "C" int scarlet_log_hist(const float *x, int n, int64_t count, const double *edges,
                                const int32_t *nbins, int32_t *hist, void *stream)
{
    if (n < 0 || count <= 0 || !x || !edges || !nbins || !hist) return set_err(SCARLET_E_ARG, "bad log_hist arguments");
    if (n == 0) return SCARLET_OK;
    hipLaunchKernelGGL(k_log_hist, dim3(n), dim3(SC_BLOCK), 0, (hipStream_t)stream, x, count, edges, nbins, hist);
    HIP_TRY(hipGetLastError());
    return SCARLET_OK;
}
extern "C" int scarlet_cut_below(float *x, int64_t count, double thresh, void *stream)
{
    if (count < 0 || (count > 0 && !x)) return set_err(SCARLET_E_ARG, "bad cut_below arguments");
    if (count == 0) return SCARLET_OK;
    int64_t blocks = (count + SC_BLOCK - 1) / SC_BLOCK;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_cut_below, dim3((unsigned)blocks), dim3(SC_BLOCK), 0, (hipStream_t)stream, x, count, thresh);
    HIP_TRY(hipGetLastError());
    return SCARLET_OK;
}
extern "C" int scarlet_trim(const float *x, int n, int H, int W, float min_value, int32_t *box, void *stream)
{
    if (n < 0 || H <= 0 || W <= 0 || !x || !box) return set_err(SCARLET_E_ARG, "bad trim arguments");
    if (n == 0) return SCARLET_OK;
    hipLaunchKernelGGL(k_trim, dim3(n), dim3(SC_BLOCK), 0, (hipStream_t)stream, x, H, W, min_value, box);
    HIP_TRY(hipGetLastError());
    return SCARLET_OK;
}
extern "C" int scarlet_resample(const float *in, float *out, int n, int H, int W, const double *taps,
                                const int32_t *win0, int ny, int nx, void *stream)
{
    if (n < 0 || H <= 0 || W <= 0 || !in || !out || in == out || !taps || !win0)
        return set_err(SCARLET_E_ARG, "bad resample arguments");
    if (ny < 1 || nx < 1 || ny > SC_TAPS_MAX || nx > SC_TAPS_MAX)
        return set_err(SCARLET_E_ARG, "resampling kernels have 1 to 12 taps");
    if (n == 0) return SCARLET_OK;
    int bx = (H * W + SC_BLOCK - 1) / SC_BLOCK;
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(k_resample, dim3(bx, n), dim3(SC_BLOCK), 0, (hipStream_t)stream, in, out, H, W, taps, win0, ny, nx);
    HIP_TRY(hipGetLastError());
    return SCARLET_OK;
}

// ---- apply_filter (operators_pybind11.cc:53-70): gather form, one thread per output pixel
template <typename T>
__global__ void k_apply_filter(const T *image, int H, int W, const T *values,
                               const int *y_start, const int *y_end, const int *x_start,
                               const int *x_end, int n, T *result)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i - y * W;
    T acc = (T)0;
    for (int k = 0; k < n; ++k) {
        const int rows = H - y_start[k] - y_end[k], cols = W - x_start[k] - x_end[k];
        const int r = y - y_start[k], cc = x - x_start[k];
        if (r >= 0 && r < rows && cc >= 0 && cc < cols)
            acc += values[k] * image[(y_end[k] + r) * W + x_end[k] + cc];
    }
    result[i] = acc;
}
extern "C" int scarlet_apply_filter(const float *image, int H, int W, const float *values,
                                    const int32_t *y_start, const int32_t *y_end, const int32_t *x_start,
                                    const int32_t *x_end, int n, float *result, void *stream)
{
    if (!image || !result || H <= 0 || W <= 0 || n < 0) return set_err(SCARLET_E_ARG, "bad apply_filter arguments");
    hipLaunchKernelGGL(k_apply_filter<float>, dim3((H * W + SC_BLOCK - 1) / SC_BLOCK), dim3(SC_BLOCK), 0,
                       (hipStream_t)stream, image, H, W, values, y_start, y_end, x_start, x_end, n, result);
    HIP_TRY(hipGetLastError());
    return SCARLET_OK;
}

// =====================================================================================
// 1. host-pointer drop-ins for operators_pybind11 (stage through device memory)
// =====================================================================================
// The reference's loops take an explicit order (dist_idx) and weight table; to honour
// those arguments exactly, one wavefront replays the given order sequentially (the 8
// neighbour terms of a pixel are fetched by 8 lanes, accumulated in order i = 0..7).
// These entry points exist for drop-in completeness and tests; the batched engine uses
// the table-free, level-parallel sweep of prox_ops.h instead.
template <typename T>
__global__ __launch_bounds__(SC_WAVE) void k_host_weighted(T *x, int n, const T *weights, const int *offsets,
                                                           const int *dist_idx, int n_dist, T thresh)
{
    // sequential semantics, parallelised over the 8 neighbours: lane i < 8 fetches one term
    const int lane = threadIdx.x;
    const T one_minus = (T)1 - thresh;
    for (int d = 0; d < n_dist; ++d) {
        const int p = dist_idx[d];
        T term = 0; bool use = false;
        if (lane < 8) {
            const T w = weights[(size_t)lane * n + p];
            if (w > 0) { term = mul_rn(x[p + offsets[lane]], w); use = true; }
        }
        T ref = 0;
        for (int i = 0; i < 8; ++i) {                 // ordered accumulation i = 0..7
            const T ti = __shfl(term, i, SC_WAVE);
            const int ui = __shfl((int)use, i, SC_WAVE);
            if (ui) ref = add_rn(ref, ti);
        }
        if (lane == 0) {
            const T cap = ref * one_minus;
            if (cap < x[p]) x[p] = cap;
        }
        __threadfence();                 // lane 0's store visible to the next step's loads
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ void k_host_nearest(double *x, const int *ref_idx, const int *dist_idx, int n_dist, double thresh)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int d = 0; d < n_dist; ++d) {
        const int p = dist_idx[d];
        const double r = x[ref_idx[p]] * (1 - thresh);
        if (r < x[p]) x[p] = r;
    }
}

template <typename T>
static int dev_alloc_copy(DevBuf &buf, const T *host, size_t count)
{
    DEV_ALLOC(buf, count * sizeof(T) + 16);
    if (host) HIP_TRY(hipMemcpy(buf.p, host, count * sizeof(T), hipMemcpyHostToDevice));
    return SCARLET_OK;
}

template <typename T>
static int host_weighted(T *x, int n, const T *weights, const int *offsets, const int *dist_idx,
                         int n_dist, T thresh)
{
    if (!x || !weights || !offsets || !dist_idx || n <= 0 || n_dist < 0) return set_err(SCARLET_E_ARG, "bad arguments");
    for (int d = 0; d < n_dist; ++d)                       // the reference does not bounds-check (mutable_unchecked)
        if (dist_idx[d] < 0 || dist_idx[d] >= n) return set_err(SCARLET_E_ARG, "dist_idx out of range");
    DevBuf dx, dw, doff, dd;
    int rc;
    if ((rc = dev_alloc_copy(dx, x, n))) return rc;
    if ((rc = dev_alloc_copy(dw, weights, (size_t)8 * n))) return rc;
    if ((rc = dev_alloc_copy(doff, offsets, 8))) return rc;
    if ((rc = dev_alloc_copy(dd, dist_idx, n_dist > 0 ? n_dist : 1))) return rc;
    hipLaunchKernelGGL(k_host_weighted<T>, dim3(1), dim3(SC_WAVE), 0, 0, dx.as<T>(), n, dw.as<T>(), doff.as<int>(),
                       dd.as<int>(), n_dist, thresh);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(x, dx.p, (size_t)n * sizeof(T), hipMemcpyDeviceToHost));
    return SCARLET_OK;
}

extern "C" int scarlet_host_prox_weighted_monotonic_f32(float *x, int n, const float *weights, const int *offsets,
                                                        const int *dist_idx, int n_dist, float thresh)
{ return host_weighted<float>(x, n, weights, offsets, dist_idx, n_dist, thresh); }
extern "C" int scarlet_host_prox_weighted_monotonic_f64(double *x, int n, const double *weights, const int *offsets,
                                                        const int *dist_idx, int n_dist, double thresh)
{ return host_weighted<double>(x, n, weights, offsets, dist_idx, n_dist, thresh); }

extern "C" int scarlet_host_prox_monotonic_f64(double *x, int n, const int *ref_idx, const int *dist_idx,
                                               int n_dist, double thresh)
{
    if (!x || !ref_idx || !dist_idx || n <= 0 || n_dist < 0) return set_err(SCARLET_E_ARG, "bad arguments");
    for (int d = 0; d < n_dist; ++d)
        if (dist_idx[d] < 0 || dist_idx[d] >= n || ref_idx[dist_idx[d]] < 0 || ref_idx[dist_idx[d]] >= n)
            return set_err(SCARLET_E_ARG, "dist_idx / ref_idx out of range");
    DevBuf dx, dr, dd;
    int rc;
    if ((rc = dev_alloc_copy(dx, x, n))) return rc;
    if ((rc = dev_alloc_copy(dr, ref_idx, n))) return rc;
    if ((rc = dev_alloc_copy(dd, dist_idx, n_dist > 0 ? n_dist : 1))) return rc;
    hipLaunchKernelGGL(k_host_nearest, dim3(1), dim3(SC_WAVE), 0, 0, dx.as<double>(), dr.as<int>(), dd.as<int>(), n_dist, thresh);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(x, dx.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return SCARLET_OK;
}

// both overloads of the reference (operators_pybind11.cc:87-88: apply_filter<float>, apply_filter<double>)
template <typename T>
static int host_apply_filter(const T *image, int H, int W, const T *values, const int *y_start, const int *y_end,
                             const int *x_start, const int *x_end, int n, T *result)
{
    if (!image || !result || H <= 0 || W <= 0 || n < 0) return set_err(SCARLET_E_ARG, "bad arguments");
    if (n > 0 && (!values || !y_start || !y_end || !x_start || !x_end)) return set_err(SCARLET_E_ARG, "bad arguments");
    DevBuf di, dv, dr, idx[4];
    const int *hidx[4] = {y_start, y_end, x_start, x_end};
    int rc;
    if ((rc = dev_alloc_copy(di, image, (size_t)H * W))) return rc;
    if ((rc = dev_alloc_copy(dv, values, n > 0 ? n : 1))) return rc;
    if ((rc = dev_alloc_copy(dr, (const T *)nullptr, (size_t)H * W))) return rc;
    for (int i = 0; i < 4; ++i) if ((rc = dev_alloc_copy(idx[i], hidx[i], n > 0 ? n : 1))) return rc;
    hipLaunchKernelGGL(k_apply_filter<T>, dim3((H * W + SC_BLOCK - 1) / SC_BLOCK), dim3(SC_BLOCK), 0, (hipStream_t)0,
                       di.as<T>(), H, W, dv.as<T>(), idx[0].as<int>(), idx[1].as<int>(), idx[2].as<int>(), idx[3].as<int>(),
                       n, dr.as<T>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(result, dr.p, (size_t)H * W * sizeof(T), hipMemcpyDeviceToHost));
    return SCARLET_OK;
}
extern "C" int scarlet_host_apply_filter_f32(const float *image, int H, int W, const float *values,
                                             const int *y_start, const int *y_end, const int *x_start,
                                             const int *x_end, int n, float *result)
{
    return host_apply_filter<float>(image, H, W, values, y_start, y_end, x_start, x_end, n, result);
}
extern "C" int scarlet_host_apply_filter_f64(const double *image, int H, int W, const double *values,
                                             const int *y_start, const int *y_end, const int *x_start,
                                             const int *x_end, int n, double *result)
{
    return host_apply_filter<double>(image, H, W, values, y_start, y_end, x_start, x_end, n, result);
}

// =====================================================================================
// 3. batched Blend.fit() engine
// =====================================================================================
// ---- optional per-kernel event timing (scarlet_profile_begin/end)
#define SC_NCLASS 8
struct Profiler {
    std::atomic<bool> on{false};
    std::vector<hipEvent_t> ev;     // pairs (start, stop)
    std::vector<int> cls, weight;   // weight: iterations one launch covers (k_fit2: several)
    int used = 0, cap = 0;
};
static Profiler g_prof;                // process-wide: one profiled fit at a time (documented in the header)
static std::mutex g_prof_mu;
static inline void prof_start(int cls, hipStream_t st, int weight = 1)
{
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (g_prof.on && g_prof.used < g_prof.cap) {
        g_prof.cls[g_prof.used] = cls;
        g_prof.weight[g_prof.used] = weight;
        (void)hipEventRecord(g_prof.ev[2 * g_prof.used], st);
    }
}
static inline void prof_stop(hipStream_t st)
{
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (g_prof.on && g_prof.used < g_prof.cap) {
        (void)hipEventRecord(g_prof.ev[2 * g_prof.used + 1], st);
        ++g_prof.used;
    }
}
extern "C" int scarlet_profile_begin(int max_iterations)
{
    if (max_iterations <= 0) return set_err(SCARLET_E_ARG, "max_iterations <= 0");
    std::lock_guard<std::mutex> lock(g_prof_mu);
    for (auto e : g_prof.ev) (void)hipEventDestroy(e);
    const int cap = max_iterations * 16;                 // (two half-batches per iteration when scarlet_fit pipelines them)
    g_prof.ev.assign((size_t)cap * 2, nullptr);
    g_prof.cls.assign(cap, 0);
    g_prof.weight.assign(cap, 1);
    for (auto &e : g_prof.ev) HIP_TRY(hipEventCreate(&e));
    g_prof.cap = cap; g_prof.used = 0; g_prof.on = true;
    return SCARLET_OK;
}
extern "C" int scarlet_profile_end_ex(double total_ms[SC_NCLASS], int64_t iterations[SC_NCLASS], int64_t launches[SC_NCLASS])
{
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (!g_prof.on) return set_err(SCARLET_E_ARG, "profiler not active");
    if (!total_ms) return set_err(SCARLET_E_ARG, "null total_ms");
    g_prof.on = false;
    for (int k = 0; k < SC_NCLASS; ++k) { total_ms[k] = 0; if (iterations) iterations[k] = 0; if (launches) launches[k] = 0; }
    for (int i = 0; i < g_prof.used; ++i) {
        HIP_TRY(hipEventSynchronize(g_prof.ev[2 * i + 1]));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
        total_ms[g_prof.cls[i]] += ms;
        if (iterations) iterations[g_prof.cls[i]] += g_prof.weight[i];
        if (launches) launches[g_prof.cls[i]] += 1;
    }
    for (auto e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear(); g_prof.cls.clear(); g_prof.weight.clear(); g_prof.cap = g_prof.used = 0;
    return SCARLET_OK;
}
extern "C" int scarlet_profile_end(double total_ms[SC_NCLASS], int64_t launches[SC_NCLASS])
{
    return scarlet_profile_end_ex(total_ms, launches, nullptr);
}

static int check_batch(const scarlet_batch *b)
{
    // Every entry point ends with a look at hipGetLastError(); that value is per thread and keeps whatever an EARLIER HIP
    // call of the thread left there.  Start from a clean slate: what is reported is ours.
    (void)hipGetLastError();
    if (!b) return set_err(SCARLET_E_ARG, "null batch");
    if (b->S <= 0 || b->K <= 0 || b->B <= 0 || b->H <= 0 || b->W <= 0) return set_err(SCARLET_E_ARG, "bad batch shape");
    if (b->K > SC_KBIG || b->B > SC_BMAX)
        return set_err(SCARLET_E_NOTIMPL, "K > 32 or B > 8 not supported by this build of the gradient kernels");
    if (b->W > 256) return set_err(SCARLET_E_TOO_LARGE, "W > 256 unsupported");
    if (!b->images || !b->sed[0] || !b->sed[1] || !b->morph[0] || !b->morph[1] || !b->cur || !b->centers ||
        !b->shifts || !b->flags || !b->lipschitz || !b->mse || !b->it || !b->active || !b->status || !b->workspace)
        return set_err(SCARLET_E_ARG, "null pointer in batch");
    if (b->symmetric && (!b->centroid_psf || b->centroid_P <= 0 || !(b->centroid_P & 1)))
        return set_err(SCARLET_E_ARG, "symmetric pipeline needs an odd-sized centroid psf");
    return SCARLET_OK;
}

static int n_tiles(const scarlet_batch *b) { return (b->H * b->W + SC_TILE_PIX - 1) / SC_TILE_PIX; }

// ---- PSF path geometry (fft.py:68-106: next_fast_len(N + P + 3) per axis, last axis even)
// FFT shape of the device convolution.  The reference pads to next_fast_len(N + P + 3)
// (fft.py:68-106: 5-smooth, 3 pixels of slack); the cropped result is the LINEAR convolution, which
// any length >= N + P - 1 reproduces exactly, so the device takes the smallest 7-smooth length from
// there (rocFFT has radix-7 kernels: 168 x 168 instead of 180 x 180 for a 128 x 128 frame with a
// 41 x 41 kernel is 1.56 x faster per transform pair, measured).  Last axis even for the R2C layout.
static int smooth7_len(int n, bool even)
{
    for (int m = n;; ++m) {
        if (even && (m & 1)) continue;
        int r = m;
        for (int p : {2, 3, 5, 7}) while (r % p == 0) r /= p;
        if (r == 1) return m;
    }
}
static PsfGeom psf_geom(int H, int W, int Py, int Px)
{
    PsfGeom g;
    g.H = H; g.W = W;
    g.Fy = smooth7_len(H + Py - 1, false);
    g.Fx = smooth7_len(W + Px - 1, true);
    // placement of image and kernel inside the reference's padded arrays (centred pad + ifftshift,
    // fft.py:27-66): only these parities decide which pixel of an even-sized kernel is its centre
    g.Fry = scarlet_next_fast_len(H + Py + 3);
    g.Frx = scarlet_next_fast_len(W + Px + 3);
    while (g.Frx & 1) g.Frx = scarlet_next_fast_len(g.Frx + 1);
    g.Fxh = g.Fx / 2 + 1;
    g.oy = (g.Fry - H + 1) / 2 - g.Fry / 2;
    g.ox = (g.Frx - W + 1) / 2 - g.Frx / 2;
    return g;
}
static int64_t align256(int64_t v) { return (v + 255) & ~(int64_t)255; }
// frames whose tile does not fit LDS: per-component GEMM scratch of the k-space symmetry in HBM
static int64_t gscratch_bytes(const scarlet_batch *b)
{
    if (b->H <= 64 && b->W <= 64) return 0;
    if (update_lds_bytes(b->H, b->W) <= 80 * 1024) return 0;      // two workgroups per CU already
    return align256(sizeof(float) * (int64_t)b->S * b->K * round16(b->H) * scratch_stride(round16(b->W)));
}
// cache of the k-space symmetry's Hankel vectors (k_iterate2, fused2.h); zero = empty (no magic word)
static int64_t kscache_bytes(const scarlet_batch *b)
{
    if (b->K > 4 || b->B > 5 || b->H > 64 || b->W > 64) return 0;
    return align256(sizeof(float) * (int64_t)b->S * b->K * 2 * SC_KSC_FLOATS);
}
static int64_t base_workspace_bytes(const scarlet_batch *b)
{
    const int64_t P = n_partials(b->K, b->B);
    // K > 8 (bigk.h): one scratch plane set [S][B][HW] for G = w^2 (model - image)
    const int64_t resid = b->K > SC_KMAX ? align256(sizeof(float) * (int64_t)b->S * b->B * b->H * b->W) : 0;
    return align256(sizeof(double) * ((int64_t)b->S * n_tiles(b) * P + (int64_t)b->S * b->K * 4) + sizeof(int) * (2 * (int64_t)b->S * b->K + 64)) +
           resid + gscratch_bytes(b) + kscache_bytes(b) + 256;
}
// ---- LDS-resident convolution (fftconv.h): plan = lengths, radices, kernel placement
// smallest circular length that reproduces the cropped linear convolution: image at 0..N-1, kernel at
// (q + o) mod F, output read at 0..N-1 (o <= 0 is the kernel's offset in the reference's padded array)
static int fft_len_min(int N, int P, int o)
{
    int m = P + N - 1 + o;
    if (N - o > m) m = N - o;
    if (N > m) m = N;
    if (P > m) m = P;
    return m;
}
// smallest L = r1 r2 >= Lmin with both radices on the menu; ties: odd r2 first when `prefer_odd_r2`
// (the contiguous run of pass B, bank conflicts), then the more balanced pair
static bool fft_choose(int Lmin, bool prefer_odd_r2, int *L, int *r1, int *r2)
{
    static const int menu[] = {4, 5, 6, 7, 8, 9, 10, 12, 14, 15, 16};
    long best = -1;
    for (int a : menu)
        for (int b : menu) {
            const int l = a * b;
            if (l < Lmin) continue;
            const int bal = a > b ? a - b : b - a;
            const long score = (long)l * 1000 + ((prefer_odd_r2 && !(b & 1)) ? 100 : 0) + bal;
            if (best < 0 || score < best) { best = score; *L = l; *r1 = a; *r2 = b; }
        }
    return best >= 0;
}
// plan for an H x W image and a Py x Px kernel; false when no menu length fits or the plane exceeds LDS
static bool fft_make_plan(int H, int W, int Py, int Px, FftPlan *p)
{
    const PsfGeom g = psf_geom(H, W, Py, Px);         // reference FFT shape -> kernel offsets
    const int oky = (g.Fry - Py + 1) / 2 - g.Fry / 2, okx = (g.Frx - Px + 1) / 2 - g.Frx / 2;
    const int Fy_min = fft_len_min(H, Py, oky), Fx_min = fft_len_min(W, Px, okx);
    int Fy, M, r1y, r2y, r1x, r2x;
    if (!fft_choose(Fy_min, false, &Fy, &r1y, &r2y)) return false;
    if (!fft_choose((Fx_min + 1) / 2, true, &M, &r1x, &r2x)) return false;
    p->H = H; p->W = W; p->Fy = Fy; p->Fx = 2 * M; p->M = M; p->RS = M + 1;
    // (A row stride == R2x (mod 32) -- 85 instead of 81 for BASELINE config 3 -- was tried so that the bank windows of
    // consecutive lines tile in the row passes: SQ_LDS_BANK_CONFLICT rose from 0.76e8 to 1.10e8 per launch and the
    // iteration took 8.19 ms instead of 8.00; the plane keeps its dense stride.  profiles/r03_notes.md)
    p->R1y = r1y; p->R2y = r2y; p->R1x = r1x; p->R2x = r2x;
    p->Py = Py; p->Px = Px; p->oky = oky; p->okx = okx;
    p->scale = (float)(1.0 / ((double)M * (double)Fy));
    p->tables = nullptr;
    p->stagger_wgs = 0;
    p->dma_image = 0;
    p->tab_off = fft_tab_off(Fy, p->RS, H, W, false);
    return fft_lds_bytes(Fy, M, p->RS) <= LDS_LIMIT - 4096;
}
// twiddle / permutation tables of a plan, float64 -> float32 (layout: fftconv.h FftPlan::tables)
static void fft_fill_tables(const FftPlan &p, std::vector<float2> &t)
{
    const double tau = 6.283185307179586476925286766559;
    const int NP = p.M / 2 + 1;
    t.assign(fft_table_float2s(p.Fy, p.M), make_float2(0.f, 0.f));
    // a twiddle w = c + i s is stored as (c, s, -s, s): fftconv.h cmul_t
    struct Tw4 { float c, s, ns, s2; };
    auto tw = [&](double num, double den) {
        const float c = (float)cos(tau * num / den), sn = (float)-sin(tau * num / den);
        return Tw4{c, sn, -sn, sn};
    };
    Tw4 *twy = (Tw4 *)t.data(), *twm = twy + p.Fy, *twx = twm + p.M, *twp = twx + NP;
    for (int j = 0; j < p.Fy; ++j) twy[j] = tw(j, p.Fy);
    for (int j = 0; j < p.M; ++j) twm[j] = tw(j, p.M);
    for (int k = 0; k < NP; ++k) twx[k] = tw(k, p.Fx);
    unsigned short *posx = (unsigned short *)(twp + NP);
    for (int k = 0; k < p.M; ++k) posx[k] = (unsigned short)(p.R2x * (k % p.R1x) + k / p.R1x);
    // the column pairs (k, M - k), k = 0 .. M/2, in the order of their first member's position (fftconv.h FftPair)
    FftPair *pair = (FftPair *)(t.data() + 2 * (p.Fy + p.M + 2 * NP) + (p.M + 3) / 4);
    std::vector<int> order(NP);
    for (int k = 0; k < NP; ++k) order[k] = k;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return posx[a] < posx[b]; });
    for (int i = 0; i < NP; ++i) {
        const int k = order[i];
        const bool first = k == 0, mid = 2 * k == p.M;
        const int ra = posx[k], rb = (first || mid) ? ra : posx[p.M - k];
        pair[i].ra = (unsigned short)ra; pair[i].rb = (unsigned short)rb;
        pair[i].cb = (unsigned short)(first ? p.M : rb); pair[i].pad = 0;
        twp[i] = twx[k];
    }
}
static bool psf_lds_possible(const scarlet_batch *b, FftPlan *p)
{
    FftPlan tmp;
    if (!p) p = &tmp;
    if (b->W & 1) return false;                       // k_psf_conv reads pixel pairs (float2)
    return fft_make_plan(b->H, b->W, b->psf_h, b->psf_w, p);
}

// k_psf_conv's view of the plan: where the image is staged; true when the exact-shape instance applies
static bool psf_conv_finish_plan(const scarlet_batch *b, FftPlan *fp)
{
    // the image plane is staged in LDS by LDS-DMA when it fits behind the H data rows (16-byte pieces)
    fp->dma_image = ((b->H * b->W) % 4 == 0 &&
                     fft_lds_bytes(fp->Fy, fp->M, fp->RS, b->H, b->W, true) <= LDS_LIMIT - 1024) ? 1 : 0;
    fp->tab_off = fft_tab_off(fp->Fy, fp->RS, b->H, b->W, fp->dma_image != 0);
    // BASELINE config 3's plan has an exact-shape instance with 1024 threads per workgroup (fftconv.h)
    return fp->H == 128 && fp->W == 128 && fp->Fy == 150 && fp->Fx == 150 && fp->M == 75 && fp->RS == 76 && fp->R1y == 10 &&
           fp->R2y == 15 && fp->R1x == 5 && fp->R2x == 15 && fp->dma_image == 1 && !opt(OPT_NO_EXACT);
}

// Workspace of a batch with a PSF.  Which convolution runs is a function of the shapes alone -- the
// LDS-resident transform whenever the half-spectrum plane fits LDS, batched hipFFT otherwise (frames
// beyond ~150 + P pixels) -- except for the diagnostic switch PSF_HIPFFT, which forces the library path
// and must not change during the life of a batch (it is read when the workspace is sized).
static bool psf_use_lds(const scarlet_batch *b, FftPlan *p) { return psf_lds_possible(b, p) && !opt(OPT_PSF_HIPFFT); }
struct PsfLayout { int64_t loss, real, spec, khat, lds_khat, lds_tables, stamps, total; };
static PsfLayout psf_layout(const scarlet_batch *b)
{
    g_layout_frozen.store(true);                      // (see scarlet_set_option)
    const PsfGeom g = psf_geom(b->H, b->W, b->psf_h, b->psf_w);
    const int64_t planes = (int64_t)b->S * b->B;
    const int64_t nk = b->diff_kernel_per_scene ? planes : (int64_t)b->B;
    FftPlan p;
    const bool lds = psf_use_lds(b, &p), want_hipfft = !lds;
    PsfLayout l;
    l.loss = base_workspace_bytes(b);
    l.real = l.loss + align256(planes * (int64_t)sizeof(double));
    // `real`: padded FFT planes (hipFFT path) or the compact gradient planes G [S][B][H][W] (LDS path)
    l.spec = l.real + align256(planes * (want_hipfft ? (int64_t)g.Fy * g.Fx : (int64_t)b->H * b->W) * (int64_t)sizeof(float));
    l.khat = l.spec + (want_hipfft ? align256(planes * g.Fy * g.Fxh * (int64_t)sizeof(float2)) : 0);
    l.lds_khat = l.khat + (want_hipfft ? align256(nk * g.Fy * g.Fxh * (int64_t)sizeof(float2)) : 0);
    l.lds_tables = l.lds_khat + (lds ? align256(nk * p.Fy * (p.M + 1) * (int64_t)sizeof(float2)) : 0);
    l.stamps = l.lds_tables + (lds ? align256(fft_table_float2s(p.Fy, p.M) * (int64_t)sizeof(float2)) : 0);
    // diagnostics (STAMPS switch, read when the workspace is sized): 32 shader-clock stamps per plane
    l.total = l.stamps + ((lds && opt(OPT_STAMPS)) ? align256(planes * 32 * (int64_t)sizeof(long long)) : 0) + 256;
    return l;
}

// ---- two half-batches on two streams (scarlet_fit with a PSF, LDS-resident transform, K <= 8).
// The convolution kernel is bound by the latency of its passes at one workgroup per CU and moves under 1 TB/s;
// the passes around it (model planes, gradient step, constraints) stream at HBM rate and hardly use the ALUs.
// Scenes are independent, so scarlet_fit runs the two halves of a large batch as two pipelines, the second on the
// calling thread's second stream: one half's convolution overlaps the other half's streaming passes.  Each half
// is a VIEW of the batch (every per-scene array advanced to its first scene) with its own workspace region behind
// the batch's (its own K-hat and tables, prepared with the batch's): the kernels do not know.
static bool split_possible(const scarlet_batch *b)
{
    return b->diff_kernel && b->psf_h > 0 && b->psf_w > 0 && b->K <= SC_KMAX && b->S >= 1024 && !b->group &&
           (b->H * b->W) % 4 == 0 && psf_lds_possible(b, nullptr);
}
static scarlet_batch batch_view(const scarlet_batch *b, int s0, int n, void *ws)
{
    scarlet_batch v = *b;
    const size_t HW = (size_t)b->H * b->W, K = b->K, B = b->B;
    v.S = n;
    v.images += s0 * B * HW;
    if (v.weights) v.weights += s0 * B * HW;
    for (int i = 0; i < 2; ++i) { v.sed[i] += s0 * K * B; v.morph[i] += s0 * K * HW; }
    v.cur += s0; v.centers += s0 * K * 2; v.shifts += s0 * K * 2; v.flags += s0 * K;
    if (v.fix_sed) v.fix_sed += s0 * K;
    if (v.fix_morph) v.fix_morph += s0 * K;
    v.lipschitz += 2 * (size_t)s0; v.mse += (size_t)s0 * b->mse_capacity; v.it += s0; v.active += s0; v.status += s0;
    if (v.diff_kernel_per_scene) v.diff_kernel += s0 * B * (size_t)b->psf_h * b->psf_w;
    if (v.group) v.group += s0 * K;
    v.workspace = ws;
    return v;
}
static void split_views(const scarlet_batch *b, scarlet_batch v[2])
{
    const int n0 = ((b->S / 2 + 7) / 8) * 8;            // (the convolution maps groups of eight scenes to the XCDs)
    char *ws = (char *)b->workspace + psf_layout(b).total;
    v[0] = batch_view(b, 0, n0, ws);
    v[1] = batch_view(b, n0, b->S - n0, ws + align256(psf_layout(&v[0]).total));
}

// diagnostics (STAMPS switch): byte offset, inside the batch's workspace, of k_psf_conv's phase stamps
// ([S][B][32] int64 shader-clock values), or -1 when the batch has none
// diagnostics: the plan of the LDS-resident convolution for this batch, 16 ints {H, W, Fy, Fx, M, RS, R1y, R2y, R1x, R2x,
// oky, okx, dma_image, exact-shape instance, LDS bytes, 0}; returns 0, or -1 when the batch takes another path
extern "C" int scarlet_debug_psf_plan(const scarlet_batch *b, int32_t *out16)
{
    if (!b || !out16 || !b->diff_kernel || b->psf_h <= 0 || b->psf_w <= 0) return -1;
    FftPlan p;
    if (!psf_use_lds(b, &p)) return -1;
    const bool x = psf_conv_finish_plan(b, &p);
    const int v[16] = {p.H, p.W, p.Fy, p.Fx, p.M, p.RS, p.R1y, p.R2y, p.R1x, p.R2x, p.oky, p.okx, p.dma_image, x ? 1 : 0,
                       (int)fft_lds_bytes(p.Fy, p.M, p.RS, b->H, b->W, p.dma_image != 0), 0};
    for (int i = 0; i < 16; ++i) out16[i] = v[i];
    return 0;
}

extern "C" int64_t scarlet_debug_psf_stamps_offset(const scarlet_batch *b)
{
    if (!b || !b->diff_kernel || b->psf_h <= 0 || b->psf_w <= 0 || !opt(OPT_STAMPS)) return -1;
    FftPlan p;
    if (!psf_use_lds(b, &p)) return -1;
    return psf_layout(b).stamps;
}

extern "C" int scarlet_batch_pipelines(const scarlet_batch *b)
{
    if (!b) return 0;
    return (split_possible(b) && !opt(OPT_PSF_HIPFFT) && !opt(OPT_NO_PIPELINE) && !opt(OPT_NO_SIDE_STREAM)) ? 2 : 1;
}

extern "C" int64_t scarlet_batch_workspace_bytes(const scarlet_batch *b)
{
    if (!b) return 0;
    if (b->diff_kernel && b->psf_h > 0 && b->psf_w > 0) {
        int64_t total = psf_layout(b).total;
        if (split_possible(b)) {
            scarlet_batch v[2];
            scarlet_batch tmp = *b;
            tmp.workspace = nullptr;
            const int n0 = ((b->S / 2 + 7) / 8) * 8;
            v[0] = tmp; v[0].S = n0; v[1] = tmp; v[1].S = b->S - n0;
            total += align256(psf_layout(&v[0]).total) + psf_layout(&v[1]).total;
        }
        return total;
    }
    return base_workspace_bytes(b);
}

static double *ws_partials(const scarlet_batch *b) { return (double *)b->workspace; }
static double *ws_conv(const scarlet_batch *b)
{
    return (double *)b->workspace + (size_t)b->S * n_tiles(b) * n_partials(b->K, b->B);
}

static float *ws_resid(const scarlet_batch *b)
{
    const int64_t P = n_partials(b->K, b->B);
    return (float *)((char *)b->workspace +
                     align256(sizeof(double) * ((int64_t)b->S * n_tiles(b) * P + (int64_t)b->S * b->K * 4) + sizeof(int) * (2 * (int64_t)b->S * b->K + 64)));
}
// per component: 1 = k_source_update_box left it to the full-frame kernel (behind the convergence sums)
static int *ws_box_fallback(const scarlet_batch *b) { return (int *)(ws_conv(b) + (size_t)b->S * b->K * 4); }
// the components the small box left to the large one: list [S * K] + its length (behind the fallback flags)
static int *ws_box_list(const scarlet_batch *b) { return ws_box_fallback(b) + (size_t)b->S * b->K; }

static float *ws_gscratch(const scarlet_batch *b)
{
    const int64_t resid = b->K > SC_KMAX ? align256(sizeof(float) * (int64_t)b->S * b->B * b->H * b->W) : 0;
    return (float *)((char *)ws_resid(b) + resid);
}

static float *ws_kscache(const scarlet_batch *b)
{
    return kscache_bytes(b) ? (float *)((char *)ws_gscratch(b) + gscratch_bytes(b)) : nullptr;
}

static GradArgs grad_args(const scarlet_batch *b, int approximate_L, int raw_gradient = 0)
{
    GradArgs a;
    a.S = b->S; a.K = b->K; a.B = b->B; a.HW = b->H * b->W; a.T = n_tiles(b);
    a.images = b->images; a.weights = b->weights; a.weight_scalar = b->weight_scalar;
    a.sed[0] = b->sed[0]; a.sed[1] = b->sed[1]; a.morph[0] = b->morph[0]; a.morph[1] = b->morph[1];
    a.cur = b->cur; a.fix_sed = b->fix_sed; a.fix_morph = b->fix_morph;
    a.partials = ws_partials(b); a.lipschitz = b->lipschitz; a.mse = b->mse; a.mse_capacity = b->mse_capacity;
    a.it = b->it; a.active = b->active; a.approximate_L = approximate_L; a.raw_gradient = raw_gradient;
    return a;
}

// ---- hipFFT plan cache: one (R2C, C2R) pair per (Fy, Fx, batch)
struct FftPlans { hipfftHandle r2c, c2r; };
static std::map<std::tuple<int, int, int>, FftPlans> g_plans;
static std::mutex g_plans_mu;          // guards the map; a plan's stream is set and used under g_fft_exec_mu
static std::mutex g_fft_exec_mu;
static int get_plans(int Fy, int Fx, int batch, FftPlans *out)
{
    std::lock_guard<std::mutex> lock(g_plans_mu);
    auto key = std::make_tuple(Fy, Fx, batch);
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
        FftPlans p;
        int n[2] = {Fy, Fx};
        if (hipfftPlanMany(&p.r2c, 2, n, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_R2C, batch) != HIPFFT_SUCCESS ||
            hipfftPlanMany(&p.c2r, 2, n, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_C2R, batch) != HIPFFT_SUCCESS)
            return set_err(SCARLET_E_HIP, "hipfftPlanMany failed");
        it = g_plans.emplace(key, p).first;
    }
    *out = it->second;
    return SCARLET_OK;
}
static int fft_r2c(const FftPlans &p, float *in, float2 *out, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(g_fft_exec_mu);
    if (hipfftSetStream(p.r2c, st) != HIPFFT_SUCCESS ||
        hipfftExecR2C(p.r2c, (hipfftReal *)in, (hipfftComplex *)out) != HIPFFT_SUCCESS)
        return set_err(SCARLET_E_HIP, "hipfftExecR2C failed");
    return SCARLET_OK;
}
static int fft_c2r(const FftPlans &p, float2 *in, float *out, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(g_fft_exec_mu);
    if (hipfftSetStream(p.c2r, st) != HIPFFT_SUCCESS ||
        hipfftExecC2R(p.c2r, (hipfftComplex *)in, (hipfftReal *)out) != HIPFFT_SUCCESS)
        return set_err(SCARLET_E_HIP, "hipfftExecC2R failed");
    return SCARLET_OK;
}
static unsigned grid_for(int64_t n) { int64_t g = (n + SC_BLOCK - 1) / SC_BLOCK; return (unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

static int prepare_psf_impl(scarlet_batch *b, void *stream);
extern "C" int scarlet_batch_prepare_psf(scarlet_batch *b, void *stream)
{
    int rc = prepare_psf_impl(b, stream);
    if (rc == SCARLET_OK && split_possible(b) && !opt(OPT_PSF_HIPFFT)) {
        scarlet_batch v[2];
        split_views(b, v);
        for (int h = 0; h < 2 && rc == SCARLET_OK; ++h) rc = prepare_psf_impl(&v[h], stream);
    }
    return rc;
}
static int prepare_psf_impl(scarlet_batch *b, void *stream)
{
    int rc = check_batch(b);
    if (rc) return rc;
    if (!b->diff_kernel || b->psf_h <= 0 || b->psf_w <= 0) return set_err(SCARLET_E_ARG, "no diff_kernel in batch");
    const PsfGeom g = psf_geom(b->H, b->W, b->psf_h, b->psf_w);
    const PsfLayout l = psf_layout(b);
    hipStream_t st = (hipStream_t)stream;
    const int nk = b->diff_kernel_per_scene ? b->S * b->B : b->B;
    FftPlan fp;
    if (psf_use_lds(b, &fp)) {
        // LDS-resident transform (fftconv.h): tables, then K-hat by the same forward code as the iteration
        std::vector<float2> tab;
        fft_fill_tables(fp, tab);
        float2 *dtab = (float2 *)((char *)b->workspace + l.lds_tables);
        HIP_TRY(hipMemcpyAsync(dtab, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));            // `tab` is host memory of this call
        fp.tables = dtab;
        const size_t lds = fft_lds_bytes(fp.Fy, fp.M, fp.RS);
        if ((rc = allow_lds(k_fft_khat, lds))) return rc;
        hipLaunchKernelGGL(k_fft_khat, dim3(nk), dim3(SC_FFT_NT), lds, st, b->diff_kernel, fp,
                           (float2 *)((char *)b->workspace + l.lds_khat));
        HIP_TRY(hipGetLastError());
        return SCARLET_OK;
    }
    float *real = (float *)((char *)b->workspace + l.real);
    float2 *khat = (float2 *)((char *)b->workspace + l.khat);
    const int oky = (g.Fry - b->psf_h + 1) / 2 - g.Fry / 2, okx = (g.Frx - b->psf_w + 1) / 2 - g.Frx / 2;
    hipLaunchKernelGGL(k_psf_pad_kernel, dim3(grid_for((int64_t)nk * g.Fy * g.Fx)), dim3(SC_BLOCK), 0, st,
                       b->diff_kernel, nk, b->psf_h, b->psf_w, g.Fy, g.Fx, oky, okx, real);
    FftPlans pk;
    if ((rc = get_plans(g.Fy, g.Fx, nk, &pk))) return rc;
    if ((rc = fft_r2c(pk, real, khat, st))) return rc;
    FftPlans pb;
    if ((rc = get_plans(g.Fy, g.Fx, b->S * b->B, &pb))) return rc;     // create the big plans now
    HIP_TRY(hipGetLastError());
    return SCARLET_OK;
}

// A second stream per calling thread for work that is off an iteration's critical path (fork / join by events, so a
// caller capturing its stream into a hipGraph captures both branches).
struct SideStream { hipStream_t st; hipEvent_t ev[3]; int device; };
static int side_stream(SideStream **out)
{
    static thread_local SideStream t_side[16];          // one per device this thread has used
    static thread_local int t_n = 0;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    for (int i = 0; i < t_n; ++i)
        if (t_side[i].device == dev) { *out = &t_side[i]; return SCARLET_OK; }
    if (t_n == 16) { *out = nullptr; return SCARLET_OK; }   // (more devices than that in one thread: no second stream)
    SideStream &n = t_side[t_n];
    HIP_TRY(hipStreamCreateWithFlags(&n.st, hipStreamNonBlocking));
    for (int i = 0; i < 3; ++i) {
        const hipError_t e = hipEventCreateWithFlags(&n.ev[i], hipEventDisableTiming);
        if (e != hipSuccess) {                                  // nothing of a half-built entry survives
            for (int j = 0; j < i; ++j) (void)hipEventDestroy(n.ev[j]);
            (void)hipStreamDestroy(n.st);
            n.st = nullptr;
            snprintf(g_err, sizeof(g_err), "HIP error: %s (side stream events)", hipGetErrorString(e));
            return SCARLET_E_HIP;
        }
    }
    n.device = dev;
    ++t_n;
    *out = &n;
    return SCARLET_OK;
}

// the Gram partials of the many-component path: one MFMA pass when the planes allow 16-byte loads
static void launch_bigk_gram(const GradArgs &a, int nch, hipStream_t st)
{
    if ((a.HW & 3) == 0 && !opt(OPT_NO_GRAM_MFMA))
        hipLaunchKernelGGL(k_bigk_gram_mfma, dim3(a.T, a.S), dim3(SC_BLOCK), 0, st, a);
    else
        hipLaunchKernelGGL(k_bigk_gram, dim3(a.T, nch * (nch + 1) / 2, a.S), dim3(SC_BLOCK), 0, st, a);
}

// k_bigk_step by band count (the accumulators of absent bands would cost occupancy)
static void launch_bigk_step(const GradArgs &a, int nch, const float *resid, hipStream_t st)
{
    const dim3 grid(a.T, nch, a.S);
    if (a.B <= 4) hipLaunchKernelGGL((k_bigk_step<4>), grid, dim3(SC_BLOCK), 0, st, a, resid);
    else if (a.B <= 6) hipLaunchKernelGGL((k_bigk_step<6>), grid, dim3(SC_BLOCK), 0, st, a, resid);
    else hipLaunchKernelGGL((k_bigk_step<SC_BMAX>), grid, dim3(SC_BLOCK), 0, st, a, resid);
}

static int backward_step_psf(scarlet_batch *b, int approximate_L, int raw_gradient, void *stream)
{
    const PsfGeom g = psf_geom(b->H, b->W, b->psf_h, b->psf_w);
    const PsfLayout l = psf_layout(b);
    hipStream_t st = (hipStream_t)stream;
    PsfArgs a;
    a.S = b->S; a.K = b->K; a.B = b->B; a.T = n_tiles(b); a.g = g;
    a.images = b->images; a.weights = b->weights; a.weight_scalar = b->weight_scalar;
    a.sed[0] = b->sed[0]; a.sed[1] = b->sed[1]; a.morph[0] = b->morph[0]; a.morph[1] = b->morph[1];
    a.cur = b->cur; a.fix_sed = b->fix_sed; a.fix_morph = b->fix_morph;
    a.real = (float *)((char *)b->workspace + l.real);
    a.spec = (float2 *)((char *)b->workspace + l.spec);
    a.khat = (const float2 *)((char *)b->workspace + l.khat);
    a.partials = ws_partials(b); a.loss_part = (double *)((char *)b->workspace + l.loss);
    a.lipschitz = b->lipschitz; a.mse = b->mse; a.mse_capacity = b->mse_capacity;
    a.it = b->it; a.active = b->active; a.approximate_L = approximate_L; a.raw_gradient = raw_gradient;
    const int planes = b->S * b->B;
    const int plane_elems = g.Fy * g.Fxh;
    const float scale = 1.0f / ((float)g.Fy * (float)g.Fx);
    const int nkh = b->diff_kernel_per_scene ? planes : b->B;
    a.khat_per_scene = b->diff_kernel_per_scene;
    int rc;
    FftPlan fp;
    const bool lds_path = psf_use_lds(b, &fp);
    const bool three_pass = lds_path && (b->H * b->W) % 4 == 0 && b->K <= SC_KMAX && !opt(OPT_NO_PSF3PASS);
    if (lds_path) {
        // one kernel: model, render, residual + loss, adjoint -> compact gradient planes G [S][B][H][W] in `real`
        fp.tables = (const float2 *)((char *)b->workspace + l.lds_tables);
        a.khat = (const float2 *)((char *)b->workspace + l.lds_khat);
        const bool x128 = psf_conv_finish_plan(b, &fp);
        const size_t lds = fft_lds_bytes(fp.Fy, fp.M, fp.RS, b->H, b->W, fp.dma_image != 0);
        if ((rc = allow_lds(k_psf_conv, lds))) return rc;
        const int groups = (b->S + 7) / 8;
        prof_start(5, st);
        // model planes, compact [S][B][H][W], into `real` (k_psf_model with the geometry of an unpadded plane)
        a.g.Fy = b->H; a.g.Fx = b->W; a.g.Fxh = b->W / 2 + 1; a.g.oy = 0; a.g.ox = 0;
        if (three_pass) {
            // model planes + Gram partials in one pass (psf_path.h, "three-pass form")
            if (b->K <= 4) hipLaunchKernelGGL((k_psf_model4g<4>), dim3(a.T, a.S), dim3(SC_BLOCK), 0, st, a);
            else hipLaunchKernelGGL((k_psf_model4g<SC_KMAX>), dim3(a.T, a.S), dim3(SC_BLOCK), 0, st, a);
        } else if ((b->H * b->W) % 4 == 0)
            hipLaunchKernelGGL(k_psf_model4, dim3(((b->H * b->W) / 4 + SC_BLOCK - 1) / SC_BLOCK, b->S), dim3(SC_BLOCK), 0, st, a);
        else
            hipLaunchKernelGGL(k_psf_model, dim3((b->H * b->W + SC_BLOCK - 1) / SC_BLOCK, b->S), dim3(SC_BLOCK), 0, st, a);
        long long *stamps = opt(OPT_STAMPS) ? (long long *)((char *)b->workspace + l.stamps) : nullptr;
        fp.stagger_wgs = 0;
        if (x128) {
            // (the instance keeps the image in registers unless built with SC_X128_DMA: plane + tables only)
            const size_t lds_x = fft_lds_bytes(fp.Fy, fp.M, fp.RS, b->H, b->W, SC_X128_DMA != 0);
            if ((rc = allow_lds(k_psf_conv_x128, lds_x))) return rc;
            hipLaunchKernelGGL(k_psf_conv_x128, dim3(groups * 8 * b->B), dim3(SC_FFT_NT_X), lds_x, st, a, fp, a.real, stamps);
        } else
        hipLaunchKernelGGL(k_psf_conv, dim3(groups * 8 * b->B), dim3(SC_FFT_NT), lds, st, a, fp, a.real, stamps);
        prof_stop(st);
        // (the gradient kernels below read G through the same geometry struct: compact planes, no offset)
    } else {
    FftPlans p;
    rc = get_plans(g.Fy, g.Fx, planes, &p);
    if (rc) return rc;
    prof_start(5, st);
    hipLaunchKernelGGL(k_psf_model, dim3((g.Fy * g.Fx + SC_BLOCK - 1) / SC_BLOCK, b->S), dim3(SC_BLOCK), 0, st, a);
    if ((rc = fft_r2c(p, a.real, a.spec, st))) return rc;
    hipLaunchKernelGGL(k_spec_mul, dim3(grid_for((int64_t)planes * plane_elems)), dim3(SC_BLOCK), 0, st,
                       a.spec, a.khat, nkh, plane_elems, (int64_t)planes * plane_elems, 0, scale);
    if ((rc = fft_c2r(p, a.spec, a.real, st))) return rc;
    hipLaunchKernelGGL(k_psf_resid, dim3(planes), dim3(SC_BLOCK), 0, st, a);
    if ((rc = fft_r2c(p, a.real, a.spec, st))) return rc;
    hipLaunchKernelGGL(k_spec_mul, dim3(grid_for((int64_t)planes * plane_elems)), dim3(SC_BLOCK), 0, st,
                       a.spec, a.khat, nkh, plane_elems, (int64_t)planes * plane_elems, 1, scale);
    if ((rc = fft_c2r(p, a.spec, a.real, st))) return rc;
    prof_stop(st);
    }
    dim3 grid(a.T, a.S);
    if (b->K > SC_KMAX) {
        // many components: G is cropped out of the FFT buffers once (the LDS path's planes are compact already),
        // then the chunked passes of bigk.h
        GradArgs ga = grad_args(b, approximate_L, raw_gradient);
        float *resid = lds_path ? a.real : ws_resid(b);
        const int nch = (b->K + SC_CHUNK - 1) / SC_CHUNK;
        prof_start(0, st);
        if (!lds_path)
            hipLaunchKernelGGL(k_plane_crop, dim3(grid_for((int64_t)planes * b->H * b->W)), dim3(SC_BLOCK), 0, st,
                               (const float *)a.real, planes, b->H, b->W, g.Fy, g.Fx, g.oy, g.ox, resid);
        hipLaunchKernelGGL(k_bigk_loss_from_planes, dim3((b->S + SC_BLOCK - 1) / SC_BLOCK), dim3(SC_BLOCK), 0, st, ga,
                           (const double *)a.loss_part);
        launch_bigk_gram(ga, nch, st);
        hipLaunchKernelGGL(k_bigk_lipschitz, dim3(ga.S), dim3(SC_BLOCK), 0, st, ga, 0);
        prof_stop(st); prof_start(1, st);
        launch_bigk_step(ga, nch, resid, st);
        hipLaunchKernelGGL(k_bigk_sed, dim3(ga.S), dim3(SC_BLOCK), 0, st, ga, 0);
        prof_stop(st);
        HIP_TRY(hipGetLastError());
        return SCARLET_OK;
    }
    if (three_pass) {
        // morphology step + SED-gradient partials in one pass over G, then the per-scene scalar head
        prof_start(1, st);
        // (instances by band count: the accumulators of absent bands would cost occupancy)
        if (b->K <= 4) {
            if (b->B <= 4) hipLaunchKernelGGL((k_step_psf4f<4, 4>), grid, dim3(SC_BLOCK), 0, st, a);
            else if (b->B <= 6) hipLaunchKernelGGL((k_step_psf4f<4, 6>), grid, dim3(SC_BLOCK), 0, st, a);
            else hipLaunchKernelGGL((k_step_psf4f<4, SC_BMAX>), grid, dim3(SC_BLOCK), 0, st, a);
            hipLaunchKernelGGL((k_sed_step<4, SC_BMAX>), dim3(a.S), dim3(SC_BLOCK), 0, st, a);
        } else {
            if (b->B <= 4) hipLaunchKernelGGL((k_step_psf4f<SC_KMAX, 4>), grid, dim3(SC_BLOCK), 0, st, a);
            else if (b->B <= 6) hipLaunchKernelGGL((k_step_psf4f<SC_KMAX, 6>), grid, dim3(SC_BLOCK), 0, st, a);
            else hipLaunchKernelGGL((k_step_psf4f<SC_KMAX, SC_BMAX>), grid, dim3(SC_BLOCK), 0, st, a);
            hipLaunchKernelGGL((k_sed_step<SC_KMAX, SC_BMAX>), dim3(a.S), dim3(SC_BLOCK), 0, st, a);
        }
        prof_stop(st);
    } else if (lds_path && (b->H * b->W) % 4 == 0) {
        // compact gradient planes: 16 B per lane
        prof_start(0, st);
        if (b->K <= 4) hipLaunchKernelGGL((k_grad_psf4<4, SC_BMAX>), grid, dim3(SC_BLOCK), 0, st, a);
        else hipLaunchKernelGGL((k_grad_psf4<SC_KMAX, SC_BMAX>), grid, dim3(SC_BLOCK), 0, st, a);
        prof_stop(st); prof_start(1, st);
        if (b->K <= 4) hipLaunchKernelGGL((k_step_psf4<4, SC_BMAX>), grid, dim3(SC_BLOCK), 0, st, a);
        else hipLaunchKernelGGL((k_step_psf4<SC_KMAX, SC_BMAX>), grid, dim3(SC_BLOCK), 0, st, a);
        prof_stop(st);
    } else if (b->K <= 4) {
        prof_start(0, st);
        hipLaunchKernelGGL((k_grad_psf<4, SC_BMAX>), grid, dim3(SC_BLOCK), 0, st, a);
        prof_stop(st); prof_start(1, st);
        hipLaunchKernelGGL((k_step_psf<4, SC_BMAX>), grid, dim3(SC_BLOCK), 0, st, a);
        prof_stop(st);
    } else {
        prof_start(0, st);
        hipLaunchKernelGGL((k_grad_psf<SC_KMAX, SC_BMAX>), grid, dim3(SC_BLOCK), 0, st, a);
        prof_stop(st); prof_start(1, st);
        hipLaunchKernelGGL((k_step_psf<SC_KMAX, SC_BMAX>), grid, dim3(SC_BLOCK), 0, st, a);
        prof_stop(st);
    }
    HIP_TRY(hipGetLastError());
    return SCARLET_OK;
}

extern "C" int scarlet_match_psfs(const float *psf1, int n, int P1y, int P1x, const float *psf2, int n2,
                                  int P2y, int P2x, float *out, void *stream)
{
    if (!psf1 || !psf2 || !out || n <= 0 || (n2 != n && n2 != 1) || P1y <= 0 || P1x <= 0 || P2y <= 0 || P2x <= 0)
        return set_err(SCARLET_E_ARG, "bad match_psfs arguments");
    // fft._get_fft_shape(psf1, psf2, padding=3): 5-smooth length of the SUM of the sizes + 3, last axis even.
    // (A spectrum ratio is a deconvolution on the periodic domain: unlike the convolutions of the fit it
    // depends on the FFT shape, so the reference's shape is used as is.)
    const int Fy = scarlet_next_fast_len(P1y + P2y + 3);
    int Fx = scarlet_next_fast_len(P1x + P2x + 3);
    while (Fx & 1) Fx = scarlet_next_fast_len(Fx + 1);
    const int64_t plane = (int64_t)Fy * Fx, splane = (int64_t)Fy * (Fx / 2 + 1);
    hipStream_t st = (hipStream_t)stream;
    DevBuf br1, br2, bs1, bs2;
    DEV_ALLOC(br1, n * plane * sizeof(float));
    DEV_ALLOC(br2, n2 * plane * sizeof(float));
    DEV_ALLOC(bs1, n * splane * sizeof(float2));
    DEV_ALLOC(bs2, n2 * splane * sizeof(float2));
    float *r1 = br1.as<float>(), *r2 = br2.as<float>(); float2 *s1 = bs1.as<float2>(), *s2 = bs2.as<float2>();
    const int o1y = (Fy - P1y + 1) / 2 - Fy / 2, o1x = (Fx - P1x + 1) / 2 - Fx / 2;
    const int o2y = (Fy - P2y + 1) / 2 - Fy / 2, o2x = (Fx - P2x + 1) / 2 - Fx / 2;
    hipLaunchKernelGGL(k_plane_pad, dim3(grid_for(n * plane)), dim3(SC_BLOCK), 0, st, psf1, n, P1y, P1x, Fy, Fx, o1y, o1x, r1);
    hipLaunchKernelGGL(k_plane_pad, dim3(grid_for(n2 * plane)), dim3(SC_BLOCK), 0, st, psf2, n2, P2y, P2x, Fy, Fx, o2y, o2x, r2);
    FftPlans pa, pb;
    int rc;
    if ((rc = get_plans(Fy, Fx, n, &pa)) == SCARLET_OK && (rc = get_plans(Fy, Fx, n2, &pb)) == SCARLET_OK &&
        (rc = fft_r2c(pa, r1, s1, st)) == SCARLET_OK && (rc = fft_r2c(pb, r2, s2, st)) == SCARLET_OK) {
        hipLaunchKernelGGL(k_spec_div, dim3(grid_for(n * splane)), dim3(SC_BLOCK), 0, st, s1, (const float2 *)s2, n2,
                           (int)splane, n * splane, 1.0f / ((float)Fy * (float)Fx));
        if ((rc = fft_c2r(pa, s1, r1, st)) == SCARLET_OK)
            hipLaunchKernelGGL(k_plane_crop, dim3(grid_for((int64_t)n * P1y * P1x)), dim3(SC_BLOCK), 0, st,
                               (const float *)r1, n, P1y, P1x, Fy, Fx, o1y, o1x, out);
    }
    const hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(st);    // buffers outlive the kernels
    if (rc) return rc;
    HIP_TRY(e1); HIP_TRY(e2);
    return SCARLET_OK;
}

extern "C" int scarlet_convolve_same(const float *model, int n, int H, int W, const float *kernel, int nk,
                                     int Py, int Px, float *out, void *stream)
{
    if (!model || !kernel || !out || n <= 0 || H <= 0 || W <= 0 || Py <= 0 || Px <= 0 || (nk != n && nk != 1))
        return set_err(SCARLET_E_ARG, "bad convolve arguments");
    const PsfGeom g = psf_geom(H, W, Py, Px);
    hipStream_t st = (hipStream_t)stream;
    FftPlan fp;
    if (fft_make_plan(H, W, Py, Px, &fp) && !opt(OPT_PSF_HIPFFT)) {
        // LDS-resident transform (fftconv.h), one workgroup per plane
        std::vector<float2> tab;
        fft_fill_tables(fp, tab);
        DevBuf dtab, dkhat;
        DEV_ALLOC(dtab, tab.size() * sizeof(float2));
        DEV_ALLOC(dkhat, (size_t)nk * fp.Fy * (fp.M + 1) * sizeof(float2));
        HIP_TRY(hipMemcpy(dtab.p, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice));
        fp.tables = dtab.as<float2>();
        const size_t lds = fft_lds_bytes(fp.Fy, fp.M, fp.RS);
        int rc;
        if ((rc = allow_lds(k_fft_khat, lds)) || (rc = allow_lds(k_fft_convolve, lds))) return rc;
        hipLaunchKernelGGL(k_fft_khat, dim3(nk), dim3(SC_FFT_NT), lds, st, kernel, fp, dkhat.as<float2>());
        hipLaunchKernelGGL(k_fft_convolve, dim3(n), dim3(SC_FFT_NT), lds, st, model, fp, (const float2 *)dkhat.as<float2>(), nk, out);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));            // the temporaries outlive the kernels
        return SCARLET_OK;
    }
    const int64_t plane = (int64_t)g.Fy * g.Fx, splane = (int64_t)g.Fy * g.Fxh;
    DevBuf breal, bkreal, bspec, bkspec;
    DEV_ALLOC(breal, n * plane * sizeof(float));
    DEV_ALLOC(bkreal, nk * plane * sizeof(float));
    DEV_ALLOC(bspec, n * splane * sizeof(float2));
    DEV_ALLOC(bkspec, nk * splane * sizeof(float2));
    float *real = breal.as<float>(), *kreal = bkreal.as<float>(); float2 *spec = bspec.as<float2>(), *kspec = bkspec.as<float2>();
    const int oky = (g.Fry - Py + 1) / 2 - g.Fry / 2, okx = (g.Frx - Px + 1) / 2 - g.Frx / 2;
    hipLaunchKernelGGL(k_plane_pad, dim3(grid_for(n * plane)), dim3(SC_BLOCK), 0, st, model, n, H, W, g.Fy, g.Fx, g.oy, g.ox, real);
    hipLaunchKernelGGL(k_psf_pad_kernel, dim3(grid_for(nk * plane)), dim3(SC_BLOCK), 0, st, kernel, nk, Py, Px, g.Fy, g.Fx, oky, okx, kreal);
    FftPlans pm, pk;
    int rc;
    if ((rc = get_plans(g.Fy, g.Fx, n, &pm)) == SCARLET_OK && (rc = get_plans(g.Fy, g.Fx, nk, &pk)) == SCARLET_OK &&
        (rc = fft_r2c(pm, real, spec, st)) == SCARLET_OK && (rc = fft_r2c(pk, kreal, kspec, st)) == SCARLET_OK) {
        hipLaunchKernelGGL(k_spec_mul, dim3(grid_for(n * splane)), dim3(SC_BLOCK), 0, st, spec, kspec, nk, (int)splane,
                           n * splane, 0, 1.0f / ((float)g.Fy * (float)g.Fx));
        if ((rc = fft_c2r(pm, spec, real, st)) == SCARLET_OK)
            hipLaunchKernelGGL(k_plane_crop, dim3(grid_for((int64_t)n * H * W)), dim3(SC_BLOCK), 0, st, real, n, H, W,
                               g.Fy, g.Fx, g.oy, g.ox, out);
    }
    const hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(st);    // buffers outlive the kernels
    if (rc) return rc;
    HIP_TRY(e1); HIP_TRY(e2);
    return SCARLET_OK;
}

// raw_gradient = 0: buffer 1-cur receives the stepped factors; 1: the gradients themselves
static int backward_impl(scarlet_batch *b, int approximate_L, int raw_gradient, void *stream)
{
    int rc = check_batch(b);
    if (rc) return rc;
    if (b->diff_kernel) return backward_step_psf(b, approximate_L, raw_gradient, stream);
    GradArgs a = grad_args(b, approximate_L, raw_gradient);
    dim3 grid(a.T, a.S);
    hipStream_t st = (hipStream_t)stream;
    if (b->K > SC_KMAX) {
        // many components per scene: passes over chunks of eight (bigk.h)
        const int nch = (b->K + SC_CHUNK - 1) / SC_CHUNK;
        float *resid = ws_resid(b);
        SideStream *side = nullptr;
        if (!opt(OPT_NO_SIDE_STREAM) && (rc = side_stream(&side))) return rc;
        if (!approximate_L && (a.HW & 63) == 0 && !opt(OPT_NO_BIGK_FUSED)) {
            // residual + morphology step + SED-gradient sums in one pass on the matrix cores (k_bigk_fused); the Gram
            // matrix and its eigenvalue beside it on the second stream:
            //   stream : lmorph . fused ..... (join) sed (+ loss record)
            //   side   : gram . lipschitz           (exact constants do not need the loss)
            if (side) {
                HIP_TRY(hipEventRecord(side->ev[0], st));
                HIP_TRY(hipStreamWaitEvent(side->st, side->ev[0], 0));
                launch_bigk_gram(a, nch, side->st);
                hipLaunchKernelGGL(k_bigk_lipschitz, dim3(a.S), dim3(SC_BLOCK), 0, side->st, a, 2);
                HIP_TRY(hipEventRecord(side->ev[2], side->st));
            }
            prof_start(0, st);
            hipLaunchKernelGGL(k_bigk_lmorph, dim3(a.S), dim3(SC_WAVE), 0, st, a);
            prof_stop(st); prof_start(1, st);
            hipLaunchKernelGGL(k_bigk_fused, grid, dim3(SC_BLOCK), 0, st, a);
            if (side) HIP_TRY(hipStreamWaitEvent(st, side->ev[2], 0));
            else {
                launch_bigk_gram(a, nch, st);
                hipLaunchKernelGGL(k_bigk_lipschitz, dim3(a.S), dim3(SC_BLOCK), 0, st, a, 2);
            }
            hipLaunchKernelGGL(k_bigk_sed, dim3(a.S), dim3(SC_BLOCK), 0, st, a, 1);
            prof_stop(st);
            HIP_TRY(hipGetLastError());
            return SCARLET_OK;
        }
        if (side) {
            // The morphology step needs the residual planes and lambda_max(A^T A) only; the Gram matrix S S^T and
            // its largest eigenvalue (for the SED step) run beside it on a second stream:
            //   stream : resid . lmorph . step ............ (join) sed
            //   side   : gram ......... (after resid: loss) lipschitz
            HIP_TRY(hipEventRecord(side->ev[0], st));
            HIP_TRY(hipStreamWaitEvent(side->st, side->ev[0], 0));
            launch_bigk_gram(a, nch, side->st);
            prof_start(0, st);
            hipLaunchKernelGGL(k_bigk_resid, grid, dim3(SC_BLOCK), 0, st, a, resid);
            HIP_TRY(hipEventRecord(side->ev[1], st));
            HIP_TRY(hipStreamWaitEvent(side->st, side->ev[1], 0));
            hipLaunchKernelGGL(k_bigk_lipschitz, dim3(a.S), dim3(SC_BLOCK), 0, side->st, a, 1);
            HIP_TRY(hipEventRecord(side->ev[2], side->st));
            hipLaunchKernelGGL(k_bigk_lmorph, dim3(a.S), dim3(SC_WAVE), 0, st, a);
            prof_stop(st); prof_start(1, st);
            launch_bigk_step(a, nch, resid, st);
            HIP_TRY(hipStreamWaitEvent(st, side->ev[2], 0));
            hipLaunchKernelGGL(k_bigk_sed, dim3(a.S), dim3(SC_BLOCK), 0, st, a, 0);
            prof_stop(st);
            HIP_TRY(hipGetLastError());
            return SCARLET_OK;
        }
        prof_start(0, st);
        hipLaunchKernelGGL(k_bigk_resid, grid, dim3(SC_BLOCK), 0, st, a, resid);
        launch_bigk_gram(a, nch, st);
        hipLaunchKernelGGL(k_bigk_lipschitz, dim3(a.S), dim3(SC_BLOCK), 0, st, a, 0);
        prof_stop(st); prof_start(1, st);
        launch_bigk_step(a, nch, resid, st);
        hipLaunchKernelGGL(k_bigk_sed, dim3(a.S), dim3(SC_BLOCK), 0, st, a, 0);
        prof_stop(st);
        HIP_TRY(hipGetLastError());
        return SCARLET_OK;
    }
    if (b->K <= 4) {
        prof_start(0, st);
        hipLaunchKernelGGL((k_grad<4, SC_BMAX>), grid, dim3(SC_BLOCK), 0, st, a);
        prof_stop(st); prof_start(1, st);
        hipLaunchKernelGGL((k_step<4, SC_BMAX>), grid, dim3(SC_BLOCK), 0, st, a);
        prof_stop(st);
    } else {
        prof_start(0, st);
        hipLaunchKernelGGL((k_grad<SC_KMAX, SC_BMAX>), grid, dim3(SC_BLOCK), 0, st, a);
        prof_stop(st); prof_start(1, st);
        hipLaunchKernelGGL((k_step<SC_KMAX, SC_BMAX>), grid, dim3(SC_BLOCK), 0, st, a);
        prof_stop(st);
    }
    HIP_TRY(hipGetLastError());
    return SCARLET_OK;
}

extern "C" int scarlet_backward_step(scarlet_batch *b, int approximate_L, void *stream)
{
    return backward_impl(b, approximate_L, 0, stream);
}

extern "C" int scarlet_backward_gradients(scarlet_batch *b, int approximate_L, void *stream)
{
    return backward_impl(b, approximate_L, 1, stream);
}

__global__ void k_zero_int(int *p) { *p = 0; }

static int launch_update(scarlet_batch *b, int in_iteration, int force_it0, void *stream)
{
    int rc = ensure_tables();
    if (rc) return rc;
    UpdateArgs u;
    u.S = b->S; u.K = b->K; u.B = b->B; u.H = b->H; u.W = b->W;
    u.sed[0] = b->sed[0]; u.sed[1] = b->sed[1]; u.morph[0] = b->morph[0]; u.morph[1] = b->morph[1];
    u.cur = b->cur; u.in_iteration = in_iteration;
    u.centers = b->centers; u.shifts = b->shifts; u.lipschitz = b->lipschitz; u.it = b->it; u.active = b->active;
    u.status = b->status; u.symmetric = b->symmetric; u.monotonic = b->monotonic;
    u.l0_thresh = b->l0_thresh; u.l1_thresh = b->l1_thresh;
    u.centroid_psf = b->centroid_psf; u.centroid_P = b->centroid_P; u.conv = ws_conv(b); u.force_it0 = force_it0;
    u.gscratch = nullptr;
    u.only_flagged = nullptr;
    u.group = b->group;
    if (b->group) {
        // MultiComponentSource: the shared centre of every source first (one wave per scene)
        const int R = b->centroid_P / 2 + 2;
        const size_t ldsg = sizeof(float) * (size_t)(2 * R + 1) * (2 * R + 2);
        if ((rc = allow_lds(k_group_centers, ldsg))) return rc;
        hipLaunchKernelGGL(k_group_centers, dim3(b->S), dim3(SC_WAVE), ldsg, (hipStream_t)stream, u);
    }
    u.hybrid_sweep = opt(OPT_NO_HYBRID_SWEEP) ? 0 : 1;
    if ((b->H > 64 || b->W > 64) && b->H <= 256 && b->W <= 256 && b->monotonic && !opt(OPT_NO_BOX) &&
        sizeof(float) * ub_lds_floats(b->H, b->W, 63) <= LDS_LIMIT) {
        // frames beyond the wave-level tile: the pipeline on the box around each peak (boxupdate.h) -- 63 x 63 for
        // every component, 127 x 127 for those whose footprint left it; the kernels below then run only for the
        // components that left the second box too
        const size_t lds1 = sizeof(float) * ub_lds_floats(b->H, b->W, 31), lds2 = sizeof(float) * ub_lds_floats(b->H, b->W, 63);
        long long *dbg = debug_stamps((size_t)b->S * b->K * 16);
        int *fb = ws_box_fallback(b);
        int *list = ws_box_list(b), *count = list + (size_t)b->S * b->K;
        hipStream_t st = (hipStream_t)stream;
        const bool second = !opt(OPT_NO_BOX2);
        if (second) hipLaunchKernelGGL(k_zero_int, dim3(1), dim3(1), 0, st, count);   // (a 4-byte hipMemsetAsync costs 16 us)
        // the large box: workgroup i takes list[i]
        const int listed_grid = b->S * b->K;
        // instances: bands of X per frame height (8: up to 128 rows, 16: up to 256), and the two BASELINE frame shapes
        // (128 x 128, 256 x 256) as compile-time constants
        auto run = [&](auto small_k, auto listed_k) -> int {
            int r2;
            if ((r2 = allow_lds(small_k, lds1)) || (r2 = allow_lds(listed_k, lds2))) return r2;
            hipLaunchKernelGGL(small_k, dim3(b->S * b->K), dim3(SC_BLOCK), lds1, st, u, fb, second ? list : nullptr, count, dbg);
            if (second)
                hipLaunchKernelGGL(listed_k, dim3(listed_grid), dim3(SC_BLOCK), lds2, st, u, fb, (const int *)list, (const int *)count, dbg);
            return SCARLET_OK;
        };
        const bool exact = !opt(OPT_NO_EXACT);
        if (exact && b->H == 128 && b->W == 128) rc = run(k_source_update_box<8, 128>, k_source_update_box_listed<8, 128>);
        else if (exact && b->H == 256 && b->W == 256) rc = run(k_source_update_box<16, 256>, k_source_update_box_listed<16, 256>);
        else if (b->H <= 128 && b->W <= 128) rc = run(k_source_update_box<8, 0>, k_source_update_box_listed<8, 0>);
        else rc = run(k_source_update_box<16, 0>, k_source_update_box_listed<16, 0>);
        if (rc) return rc;
        u.only_flagged = fb;
    }
    if (b->H <= 64 && b->W <= 64 && !opt(OPT_FORCE_BLOCK_UPDATE)) {
        // one wave per component, four components per workgroup (wave_ops.h)
        const size_t lds = sizeof(float) * SC_NWAVES * ((size_t)b->H * tile_stride(b->W) + SC_WAVE_VEC_FLOATS);
        rc = allow_lds(k_source_update_w, lds);
        if (rc) return rc;
        const int n = b->S * b->K;
        hipLaunchKernelGGL(k_source_update_w, dim3((n + SC_NWAVES - 1) / SC_NWAVES), dim3(SC_BLOCK), lds,
                           (hipStream_t)stream, u);
    } else if (update_lds_bytes(b->H, b->W) <= LDS_LIMIT) {
        const size_t lds = update_lds_bytes(b->H, b->W);
        const size_t lds1 = sizeof(float) * ((size_t)b->H * tile_stride(b->W) + 2 * round16(b->H) + 5 * round16(b->W) +
                                             stage_floats(round16(b->H), round16(b->W)));
        if (lds > 80 * 1024 && lds1 <= 78 * 1024 && gscratch_bytes(b) > 0) {
            // scratch in HBM: two workgroups per CU instead of one
            u.gscratch = ws_gscratch(b);
            rc = allow_lds(k_source_update<1>, lds1);
            if (rc) return rc;
            hipLaunchKernelGGL(k_source_update<1>, dim3(b->S * b->K), dim3(SC_BLOCK), lds1, (hipStream_t)stream, u);
        } else {
            rc = allow_lds(k_source_update<0>, lds);
            if (rc) return rc;
            hipLaunchKernelGGL(k_source_update<0>, dim3(b->S * b->K), dim3(SC_BLOCK), lds, (hipStream_t)stream, u);
        }
    } else {
        // frames beyond the LDS tile (up to 256 x 256): operators in place on the plane in HBM / L2
        if (b->H > 256 || b->W > 256) return set_err(SCARLET_E_TOO_LARGE, "frames larger than 256 x 256 are not supported");
        u.gscratch = ws_gscratch(b);
        const size_t lds = sizeof(float) * (2 * round16(b->H) + 5 * round16(b->W) +       // av, bv, cv, zv, stage
                                            stage_floats(round16(b->H), round16(b->W)));
        hipLaunchKernelGGL(k_source_update<2>, dim3(b->S * b->K), dim3(SC_BLOCK), lds, (hipStream_t)stream, u);
    }
    HIP_TRY(hipGetLastError());
    return SCARLET_OK;
}

extern "C" int scarlet_source_update(scarlet_batch *b, int in_iteration, void *stream)
{
    int rc = check_batch(b);
    if (rc) return rc;
    return launch_update(b, in_iteration ? 1 : 0, in_iteration ? 0 : 1, stream);
}

extern "C" int scarlet_check_convergence(scarlet_batch *b, double e_rel, void *stream)
{
    int rc = check_batch(b);
    if (rc) return rc;
    hipLaunchKernelGGL(k_converge, dim3((b->S + SC_BLOCK - 1) / SC_BLOCK), dim3(SC_BLOCK), 0, (hipStream_t)stream,
                       b->S, b->K, ws_conv(b), b->flags, b->active, b->it, b->cur, e_rel * e_rel);
    HIP_TRY(hipGetLastError());
    return SCARLET_OK;
}

// workgroups of k_fit2x the current device keeps resident at once (occupancy query x compute units), per device
static int fit2x_resident_workgroups(size_t lds, int *out)
{
    static std::mutex mu;
    static int cached[64] = {0};
    static size_t cached_lds[64] = {0};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return set_err(SCARLET_E_HIP, "device index out of range");
    std::lock_guard<std::mutex> lock(mu);
    if (!cached[dev] || cached_lds[dev] != lds) {
        int per_cu = 0, cus = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fit2x, SC_FB2, lds));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (per_cu < 1) per_cu = 1;
        cached[dev] = per_cu * cus; cached_lds[dev] = lds;
    }
    *out = cached[dev];
    return SCARLET_OK;
}

// ---- fused one-kernel iteration (fused.h)
static size_t fused_lds_bytes(const scarlet_batch *b)
{
    return sizeof(float) * ((size_t)b->K * b->H * tile_stride(b->W) + SC_NWAVES * SC_WAVE_VEC_FLOATS);
}
static bool fused_ok(const scarlet_batch *b, int approximate_L)
{
    // K > 4: eight tiles leave one workgroup per CU and the general path is faster (measured at K = 6, 8:
    // 2.44 vs 2.58 ms and 3.25 vs 3.90 ms per iteration of 4000 scenes)
    if (approximate_L || b->diff_kernel || b->K > 4 || b->group || opt(OPT_NO_FUSED)) return false;
    if (b->H > 64 || b->W > 64 || (b->W & 3) || b->H < 3 || b->W < 3) return false;
    return fused_lds_bytes(b) <= LDS_LIMIT - 4096;
}
// n_iter > 1: that many iterations in ONE launch where the persistent form exists (k_fit2: the headline shape's
// exact instance); *done receives the number of iterations the launch covers
static int launch_fused(scarlet_batch *b, double e_rel, void *stream, int n_iter = 1, int *done = nullptr)
{
    if (done) *done = 1;
    int rc = ensure_tables();
    if (rc) return rc;
    FusedArgs f;
    f.S = b->S; f.K = b->K; f.B = b->B; f.H = b->H; f.W = b->W;
    f.images = b->images; f.weights = b->weights; f.weight_scalar = b->weight_scalar;
    f.sed[0] = b->sed[0]; f.sed[1] = b->sed[1]; f.morph[0] = b->morph[0]; f.morph[1] = b->morph[1];
    f.cur = b->cur; f.fix_sed = b->fix_sed; f.fix_morph = b->fix_morph;
    f.centers = b->centers; f.shifts = b->shifts; f.flags = b->flags;
    f.lipschitz = b->lipschitz; f.mse = b->mse; f.mse_capacity = b->mse_capacity;
    f.it = b->it; f.active = b->active; f.status = b->status;
    f.symmetric = b->symmetric; f.monotonic = b->monotonic; f.l0_thresh = b->l0_thresh; f.l1_thresh = b->l1_thresh;
    f.centroid_psf = b->centroid_psf; f.centroid_P = b->centroid_P; f.e_rel2 = e_rel * e_rel;
    // diagnostics: SCARLET_STAMPS=1 writes phase stamps into the (otherwise unused) partials area
    f.kscache = (b->diff_kernel || opt(OPT_NO_KSCACHE)) ? nullptr : ws_kscache(b);
    f.stamps = (opt(OPT_STAMPS) && n_partials(b->K, b->B) >= 16) ? (long long *)ws_partials(b) : nullptr;
    // experiment knob: SCARLET_PAD_LDS=<bytes> lowers the number of co-resident workgroups
    const size_t lds = fused_lds_bytes(b) + (size_t)opt(OPT_PAD_LDS);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_ITERATE(KM_, BM_)                                                                       \
    do {                                                                                               \
        rc = allow_lds(k_iterate<KM_, BM_>, lds);                                                      \
        if (rc) return rc;                                                                             \
        prof_start(4, st);                                                                             \
        hipLaunchKernelGGL((k_iterate<KM_, BM_>), dim3(b->S), dim3(SC_BLOCK), lds, st, f);             \
        prof_stop(st);                                                                                 \
    } while (0)
    // K <= 4, B <= 5: eight waves per scene, a pair of waves per component (fused2.h; its 128-VGPR
    // budget does not hold a sixth band's accumulators)
    if (b->K <= 4 && b->B <= 5 && !opt(OPT_FUSED_V1)) {
        const size_t lds2 = sizeof(float) * ((size_t)b->K * b->H * tile_stride(b->W) + (size_t)b->K * SC_PAIR_VEC_FLOATS) +
                            (size_t)opt(OPT_PAD_LDS);
#define LAUNCH_ITERATE2(BM_)                                                                           \
    do {                                                                                               \
        rc = allow_lds(k_iterate2<4, BM_>, lds2);                                                      \
        if (rc) return rc;                                                                             \
        prof_start(4, st);                                                                             \
        hipLaunchKernelGGL((k_iterate2<4, BM_>), dim3(b->S), dim3(SC_FB2), lds2, st, f);               \
        prof_stop(st);                                                                                 \
    } while (0)
        // the headline shape (BASELINE configs[1]/[3]: 4 sources, 5 bands, 64 x 64, default pipeline) has
        // an instance with every shape and switch folded at compile time
        const bool exact64 = b->K == 4 && b->B == 5 && b->H == 64 && b->W == 64 && !b->weights && b->weight_scalar == 1.0f && b->symmetric &&
                             b->monotonic && b->l0_thresh < 0.f && b->l1_thresh < 0.f && !opt(OPT_NO_EXACT);
        // (the exact-shape instances use their own tile stride, common.h SC_XS_STRIDE)
        const size_t lds2x = sizeof(float) * ((size_t)4 * 64 * SC_XS_STRIDE + (size_t)4 * SC_PAIR_VEC_FLOATS) + (size_t)opt(OPT_PAD_LDS);
        if (n_iter > 0xffffff) n_iter = 0xffffff;
        if (exact64 && (n_iter > 1 || (opt(OPT_PERSIST_DBG) & 2)) && !opt(OPT_NO_PERSIST)) {
            rc = allow_lds(k_fit2x, lds2x);
            if (rc) return rc;
            // as many workgroups as the chip keeps resident; the scenes beyond them come from the launch's queue
            // (a counter in the workspace, zeroed on the stream in front of the launch)
            int n_wg = 0;
            if ((rc = fit2x_resident_workgroups(lds2x, &n_wg))) return rc;
            if (n_wg > b->S || (opt(OPT_PERSIST_DBG) & 4)) n_wg = b->S;            // (4: diagnostic, one workgroup per scene)
            int *queue = (int *)((char *)b->workspace + base_workspace_bytes(b) - 128);
            HIP_TRY(hipMemsetAsync(queue, 0, sizeof(int), st));
            prof_start(4, st, n_iter);
            hipLaunchKernelGGL(k_fit2x, dim3(n_wg), dim3(SC_FB2), lds2x, st, f, n_iter | ((opt(OPT_PERSIST_DBG) & 1) << 30), queue, n_wg);
            prof_stop(st);
            if (done) *done = n_iter;
        } else if (exact64) {
            rc = allow_lds(k_iterate2<4, 5, 64>, lds2x);
            if (rc) return rc;
            prof_start(4, st);
            hipLaunchKernelGGL((k_iterate2<4, 5, 64>), dim3(b->S), dim3(SC_FB2), lds2x, st, f);
            prof_stop(st);
        } else
            LAUNCH_ITERATE2(5);
#undef LAUNCH_ITERATE2
        HIP_TRY(hipGetLastError());
        return SCARLET_OK;
    }
    if (b->B <= 6) LAUNCH_ITERATE(4, 6); else LAUNCH_ITERATE(4, SC_BMAX);
#undef LAUNCH_ITERATE
    HIP_TRY(hipGetLastError());
    return SCARLET_OK;
}

__global__ void k_count_active(const int *active, int S, int *out)
{
    int c = 0;
    for (int i = threadIdx.x; i < S; i += blockDim.x) c += active[i] != 0;
    c = (int)wave_sum((float)c);
    __shared__ int tot;
    if (threadIdx.x == 0) tot = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) atomicAdd(&tot, c);
    __syncthreads();
    if (threadIdx.x == 0) *out = tot;
}

extern "C" int scarlet_fit(scarlet_batch *b, int max_iter, double e_rel, int approximate_L,
                           int check_every, void *stream)
{
    int rc = check_batch(b);
    if (rc) return rc;
    if (max_iter < 0) return set_err(SCARLET_E_ARG, "max_iter < 0");
    hipStream_t st = (hipStream_t)stream;
    int launched = 0;
    int *d_count = (int *)((char *)b->workspace + base_workspace_bytes(b) - 64);
    const bool fused = fused_ok(b, approximate_L);
    if (!fused && scarlet_batch_pipelines(b) == 2) {
        // two half-batches, two streams (see split_possible)
        SideStream *side = nullptr;
        if ((rc = side_stream(&side))) return rc;
        if (side) {
            scarlet_batch v[2];
            split_views(b, v);
            hipStream_t sv[2] = {st, side->st};
            bool forked = false;
            // (an error between fork and join: the caller's stream still gets the second stream's work ordered before
            // anything it enqueues next)
            auto bail = [&](int code) {
                if (forked && hipEventRecord(side->ev[1], side->st) == hipSuccess) (void)hipStreamWaitEvent(st, side->ev[1], 0);
                return code;
            };
            for (int i = 0; i < max_iter; ++i) {
#define HIP_TRY_BAIL(expr) do { if ((expr) != hipSuccess) { set_err(SCARLET_E_HIP, "HIP call failed in the two-pipeline loop: " #expr); return bail(SCARLET_E_HIP); } } while (0)
                if (!forked) {
                    HIP_TRY(hipEventRecord(side->ev[0], st));
                    HIP_TRY_BAIL(hipStreamWaitEvent(side->st, side->ev[0], 0));
                    forked = true;
                }
                for (int h = 0; h < 2; ++h) {
                    if ((rc = scarlet_backward_step(&v[h], approximate_L, sv[h]))) return bail(rc);
                    prof_start(2, sv[h]);
                    if ((rc = launch_update(&v[h], 1, 0, sv[h]))) return bail(rc);
                    prof_stop(sv[h]); prof_start(3, sv[h]);
                    if ((rc = scarlet_check_convergence(&v[h], e_rel, sv[h]))) return bail(rc);
                    prof_stop(sv[h]);
                }
                ++launched;
                const bool check = check_every > 0 && (i + 1) % check_every == 0 && i + 1 < max_iter;
                if (check || i + 1 == max_iter) {
                    HIP_TRY_BAIL(hipEventRecord(side->ev[1], side->st));
                    HIP_TRY_BAIL(hipStreamWaitEvent(st, side->ev[1], 0));
                    forked = false;
                }
                if (check) {
                    int h_count = 0;
                    hipLaunchKernelGGL(k_count_active, dim3(1), dim3(SC_BLOCK), 0, st, b->active, b->S, d_count);
                    HIP_TRY(hipMemcpyAsync(&h_count, d_count, sizeof(int), hipMemcpyDeviceToHost, st));
                    HIP_TRY(hipStreamSynchronize(st));
                    if (h_count == 0) break;
                }
            }
#undef HIP_TRY_BAIL
            return launched;
        }
    }
    for (int i = 0; i < max_iter; ++i) {
        if (fused) {
            // up to the next host check (or the end) in one launch where the persistent kernel applies
            int want = max_iter - i, did = 1;
            if (check_every > 0) { const int to_check = check_every - i % check_every; if (to_check < want) want = to_check; }
            if ((rc = launch_fused(b, e_rel, stream, want, &did))) return rc;
            i += did - 1; launched += did - 1;
        } else {
            if ((rc = scarlet_backward_step(b, approximate_L, stream))) return rc;
            prof_start(2, st);
            if ((rc = launch_update(b, 1, 0, stream))) return rc;
            prof_stop(st); prof_start(3, st);
            if ((rc = scarlet_check_convergence(b, e_rel, stream))) return rc;
            prof_stop(st);
        }
        ++launched;
        if (check_every > 0 && (i + 1) % check_every == 0 && i + 1 < max_iter) {
            int h_count = 0;
            hipLaunchKernelGGL(k_count_active, dim3(1), dim3(SC_BLOCK), 0, st, b->active, b->S, d_count);
            HIP_TRY(hipMemcpyAsync(&h_count, d_count, sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (h_count == 0) break;
        }
    }
    return launched;
}

// ---- several observations per blend (extras.h: MultiArgs)
extern "C" int scarlet_fit_multi(scarlet_batch *state, scarlet_batch *const *obs, const int32_t *band0, int n_obs,
                                 int max_iter, double e_rel, int approximate_L, int check_every, void *stream)
{
    int rc = check_batch(state);
    if (rc) return rc;
    if (!obs || !band0 || n_obs < 1 || n_obs > SC_MULTI_MAX) return set_err(SCARLET_E_ARG, "1 to 8 observations");
    if (max_iter < 0) return set_err(SCARLET_E_ARG, "max_iter < 0");
    MultiArgs m;
    m.S = state->S; m.K = state->K; m.C = state->B; m.HW = state->H * state->W; m.n_obs = n_obs;
    m.sed[0] = state->sed[0]; m.sed[1] = state->sed[1]; m.morph[0] = state->morph[0]; m.morph[1] = state->morph[1];
    m.cur = state->cur; m.active = state->active; m.it = state->it;
    m.lipschitz = state->lipschitz; m.mse = state->mse; m.mse_capacity = state->mse_capacity;
    m.fix_sed = state->fix_sed; m.fix_morph = state->fix_morph; m.approximate_L = approximate_L;
    for (int o = 0; o < n_obs; ++o) {
        if ((rc = check_batch(obs[o]))) return rc;
        const scarlet_batch *ob = obs[o];
        if (ob->S != state->S || ob->K != state->K || ob->H != state->H || ob->W != state->W || band0[o] < 0 ||
            band0[o] + ob->B > state->B || ob->mse_capacity < 1)
            return set_err(SCARLET_E_ARG, "an observation does not fit the model frame");
        m.obs[o].sed[0] = ob->sed[0]; m.obs[o].sed[1] = ob->sed[1]; m.obs[o].morph[0] = ob->morph[0]; m.obs[o].morph[1] = ob->morph[1];
        m.obs[o].cur = ob->cur; m.obs[o].it = ob->it; m.obs[o].active = ob->active;
        m.obs[o].mse = ob->mse; m.obs[o].mse_capacity = ob->mse_capacity; m.obs[o].B = ob->B; m.obs[o].band0 = band0[o];
    }
    hipStream_t st = (hipStream_t)stream;
    // scratch for the approximate Lipschitz sums: the state's convergence-sum area is free until the update runs
    double *approx = ws_conv(state);
    int *d_count = (int *)((char *)state->workspace + base_workspace_bytes(state) - 64);
    const dim3 gridc((m.HW + 4 * SC_BLOCK - 1) / (4 * SC_BLOCK), m.S * m.K);
    int launched = 0;
    for (int i = 0; i < max_iter; ++i) {
        hipLaunchKernelGGL(k_multi_scatter, gridc, dim3(SC_BLOCK), 0, st, m);
        for (int o = 0; o < n_obs; ++o)
            if ((rc = scarlet_backward_gradients(obs[o], 0, stream))) return rc;
        // exact L of the FULL factors (blend.py:205-218): the state's own gradient pass (its loss and gradients
        // are overwritten below); approximate L: the two sums of squares
        if (approximate_L) hipLaunchKernelGGL(k_multi_approx, dim3(m.S), dim3(SC_BLOCK), 0, st, m, approx);
        else if ((rc = scarlet_backward_gradients(state, 0, stream))) return rc;
        hipLaunchKernelGGL(k_multi_loss, dim3((m.S + SC_BLOCK - 1) / SC_BLOCK), dim3(SC_BLOCK), 0, st, m, (const double *)approx);
        hipLaunchKernelGGL(k_multi_step, gridc, dim3(SC_BLOCK), 0, st, m);
        if ((rc = launch_update(state, 1, 0, stream))) return rc;
        if ((rc = scarlet_check_convergence(state, e_rel, stream))) return rc;
        ++launched;
        if (check_every > 0 && (i + 1) % check_every == 0 && i + 1 < max_iter) {
            int h_count = 0;
            hipLaunchKernelGGL(k_count_active, dim3(1), dim3(SC_BLOCK), 0, st, state->active, state->S, d_count);
            HIP_TRY(hipMemcpyAsync(&h_count, d_count, sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (h_count == 0) break;
        }
    }
    HIP_TRY(hipGetLastError());
    return launched;
}

// ------------------------------------------------------------------------------------
// ExtendedSource initialisation (source.py:139-180), float64 tile like the reference's
// float64 coadd (bg_rms is float64 there), one workgroup per component.
struct InitArgs {
    int S, K, B, H, W;
    const float *images;
    float *sed[2], *morph[2];
    const int *cur;
    const int *centers;
    int *flags;
    int *status;
    double bg_rms[SC_BMAX];
    double sed_scale[SC_BMAX];
    int has_scale;
    int do_symmetric, do_monotonic;
    double thresh;
    int no_hybrid;                    // diagnostics: SCARLET_NO_HYBRID_SWEEP
};

// GT: frames whose float64 tile does not fit LDS work on a tile in a temporary HBM buffer
template <bool GT>
__global__ __launch_bounds__(SC_BLOCK) void k_init_extended(InitArgs a, double *gtile)
{
    extern __shared__ __align__(16) double ldsd[];
    const int c = blockIdx.x, s = c / a.K, H = a.H, W = a.W, HW = H * W, B = a.B;
    TileT<double> t; t.H = H; t.W = W; t.LW = W + 1;
    t.m = GT ? gtile + (size_t)c * H * (W + 1) : ldsd;
    __shared__ double red[SC_NWAVES];
    __shared__ float sed_s[SC_BMAX];
    const int cy = a.centers[2 * c], cx = a.centers[2 * c + 1];
    if (cy < 0 || cy >= H || cx < 0 || cx >= W) {
        // a source outside the frame (IndexError in the reference, ValueError in BlendBatch): empty component
        const int wb = a.cur[s];
        float *gm0 = a.morph[wb] + (size_t)c * HW;
        for (int i = threadIdx.x; i < HW; i += SC_BLOCK) gm0[i] = 0.f;
        if (threadIdx.x < B) a.sed[wb][(size_t)c * B + threadIdx.x] = 0.f;
        if (threadIdx.x == 0) {
            a.flags[c] = SCARLET_FLAG_SED_NOT_CONVERGED | SCARLET_FLAG_MORPH_NOT_CONVERGED | SCARLET_FLAG_NO_VALID_PIXELS;
            atomicOr(&a.status[s], SCARLET_STATUS_CENTER_AT_EDGE);
        }
        return;
    }
    const float *img = a.images + (size_t)s * B * HW;
    // get_psf_sed (source.py:41-71): float32 like the reference (images.dtype)
    if (threadIdx.x < B) {
        float v = img[(size_t)threadIdx.x * HW + cy * W + cx];
        if (a.has_scale) v = v * (float)a.sed_scale[threadIdx.x];
        sed_s[threadIdx.x] = v;
    }
    __syncthreads();
    // build_detection_coadd (source.py:101-136): bands with positive SED only
    double wb[SC_BMAX], jac = 0, var = 0;
#pragma unroll
    for (int b = 0; b < SC_BMAX; ++b) {
        wb[b] = 0;
        if (b < B && sed_s[b] > 0.f) {
            const double sd = (double)sed_s[b], bg = a.bg_rms[b];
            wb[b] = sd / (bg * bg);
            jac += sd * sd / (bg * bg);
            var += wb[b] * wb[b] * bg * bg;
        }
    }
    const double cutoff = a.thresh * sqrt(var) / jac;
    for (int i = threadIdx.x; i < HW; i += SC_BLOCK) {
        double acc = 0;
#pragma unroll
        for (int b = 0; b < SC_BMAX; ++b)
            if (b < B && wb[b] != 0) acc += wb[b] * (double)img[(size_t)b * HW + i];
        t.m[(i / W) * t.LW + (i % W)] = acc / jac;
    }
    __syncthreads();
    const SymWindow sw = sym_window(H, W, cy, cx);
    if (a.do_symmetric) flip_symmetry_tile<double>(t, sw, true, 1.0);      // sdss (source.py:162)
    // thresh=.1 (source.py:165-167).  Everything <= cutoff is zeroed below (source.py:170-175), so the sweep
    // may stop once three levels hold nothing above the cutoff: the levels beyond cannot exceed it either
    __shared__ int lastpos_s;
    int lstop = 1 << 30;
    if (a.do_monotonic) {
        if (!GT && cutoff >= 0 && !a.no_hybrid) {
            // levels 1 .. 46 on one wave without barriers (wave_ops.h), the rest on the workgroup if needed
            __shared__ int hyb[2];
            if (threadIdx.x < SC_WAVE) {
                int done, quiet;
                wave_monotonic<double>(t, cy, cx, 0.1, &done, &quiet, cutoff);
                if (threadIdx.x == 0) { hyb[0] = done; hyb[1] = quiet; }
            }
            __syncthreads();
            lstop = hyb[0];
            if (lstop == (1 << 30))
                lstop = monotonic_tile<false, double>(t, cy, cx, 0.1, &lastpos_s, cutoff, SC_COMPACT_LAST + 1,
                                                      SC_COMPACT_LAST - hyb[1]);
        } else
            lstop = monotonic_tile<false, double>(t, cy, cx, 0.1, cutoff >= 0 ? &lastpos_s : nullptr, cutoff);
    }
    double cnt = 0;
    for (int i = threadIdx.x; i < HW; i += SC_BLOCK)
        if (t.m[(i / W) * t.LW + (i % W)] > cutoff && sweep_level(i / W, i % W, cy, cx) <= lstop) cnt += 1;
    cnt = block_sum(cnt, red);
    // morph[~mask] = 0 happens BEFORE the centre pixel is read (source.py:174-178)
    const double centre = t.m[cy * t.LW + cx] > cutoff ? t.m[cy * t.LW + cx] : 0.0;
    const int wbuf = a.cur[s];
    float *gm = a.morph[wbuf] + (size_t)c * HW;
    for (int i = threadIdx.x; i < HW; i += SC_BLOCK) {
        const double v = t.m[(i / W) * t.LW + (i % W)];
        gm[i] = (float)((v > cutoff && sweep_level(i / W, i % W, cy, cx) <= lstop) ? v / centre : 0.0);
    }
    if (threadIdx.x < B) a.sed[wbuf][(size_t)c * B + threadIdx.x] = sed_s[threadIdx.x];
    if (threadIdx.x == 0) {
        int f = SCARLET_FLAG_SED_NOT_CONVERGED | SCARLET_FLAG_MORPH_NOT_CONVERGED;
        if (cnt == 0) f |= SCARLET_FLAG_NO_VALID_PIXELS;          // SourceInitError in the reference
        a.flags[c] = f;
    }
}

extern "C" int scarlet_init_extended(scarlet_batch *b, const float *bg_rms_host, float thresh,
                                     const float *sed_scale_host, int init_symmetric, int init_monotonic,
                                     int run_update, void *stream)
{
    int rc = check_batch(b);
    if (rc) return rc;
    if (!bg_rms_host) return set_err(SCARLET_E_ARG, "bg_rms is required");
    InitArgs a;
    a.S = b->S; a.K = b->K; a.B = b->B; a.H = b->H; a.W = b->W;
    a.images = b->images; a.sed[0] = b->sed[0]; a.sed[1] = b->sed[1];
    a.morph[0] = b->morph[0]; a.morph[1] = b->morph[1]; a.cur = b->cur; a.centers = b->centers; a.flags = b->flags;
    a.status = b->status;
    a.has_scale = sed_scale_host != nullptr; a.thresh = thresh;
    a.do_symmetric = init_symmetric; a.do_monotonic = init_monotonic;
    a.no_hybrid = opt(OPT_NO_HYBRID_SWEEP) ? 1 : 0;
    for (int i = 0; i < SC_BMAX; ++i) {
        a.bg_rms[i] = i < b->B ? (double)bg_rms_host[i] : 1.0;
        a.sed_scale[i] = (i < b->B && sed_scale_host) ? (double)sed_scale_host[i] : 1.0;
        if (i < b->B && !(a.bg_rms[i] > 0))
            return set_err(SCARLET_E_ARG, "bg_rms must be greater than zero in all channels");
    }
    const size_t lds = sizeof(double) * (size_t)b->H * (b->W + 1);
    if (lds <= LDS_LIMIT) {
        rc = allow_lds(k_init_extended<false>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k_init_extended<false>, dim3(b->S * b->K), dim3(SC_BLOCK), lds, (hipStream_t)stream, a,
                           (double *)nullptr);
    } else {
        if (b->H > 256 || b->W > 256) return set_err(SCARLET_E_TOO_LARGE, "frames larger than 256 x 256 are not supported");
        DevBuf gtile;                                  // one-time setup: a temporary float64 tile per component
        DEV_ALLOC(gtile, lds * (size_t)b->S * b->K);
        hipLaunchKernelGGL(k_init_extended<true>, dim3(b->S * b->K), dim3(SC_BLOCK), 0, (hipStream_t)stream, a, gtile.as<double>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    }
    HIP_TRY(hipGetLastError());
    return run_update ? launch_update(b, 0, 1, stream) : SCARLET_OK;   // constructor's self.update()
}

// ---- convergence sums for the Python-override path (the built-in pipeline computes them itself)
__global__ __launch_bounds__(SC_BLOCK) void k_conv_sums(int K, int B, int HW, float *const sed0, float *const sed1,
                                                        float *const morph0, float *const morph1, const int *cur,
                                                        const int *active, double *conv)
{
    __shared__ double red[SC_NWAVES];
    const int c = blockIdx.x, s = c / K;
    if (!active[s]) return;
    const int c0 = cur[s];
    const float *mn = (c0 ? morph0 : morph1) + (size_t)c * HW, *ml = (c0 ? morph1 : morph0) + (size_t)c * HW;
    const float *sn = (c0 ? sed0 : sed1) + (size_t)c * B, *sl = (c0 ? sed1 : sed0) + (size_t)c * B;
    double d2 = 0, n2 = 0, d2s = 0, n2s = 0;
    for (int i = threadIdx.x; i < HW; i += SC_BLOCK) {
        const float v = mn[i], d = ml[i] - v;
        d2 += (double)(d * d); n2 += (double)(v * v);
    }
    for (int i = threadIdx.x; i < B; i += SC_BLOCK) {
        const float v = sn[i], d = sl[i] - v;
        d2s += (double)(d * d); n2s += (double)(v * v);
    }
    d2 = block_sum(d2, red); n2 = block_sum(n2, red); d2s = block_sum(d2s, red); n2s = block_sum(n2s, red);
    if (threadIdx.x == 0) { conv[4 * c] = d2s; conv[4 * c + 1] = n2s; conv[4 * c + 2] = d2; conv[4 * c + 3] = n2; }
}

extern "C" int scarlet_convergence_sums(scarlet_batch *b, void *stream)
{
    int rc = check_batch(b);
    if (rc) return rc;
    hipLaunchKernelGGL(k_conv_sums, dim3(b->S * b->K), dim3(SC_BLOCK), 0, (hipStream_t)stream, b->K, b->B,
                       b->H * b->W, b->sed[0], b->sed[1], b->morph[0], b->morph[1], b->cur, b->active, ws_conv(b));
    HIP_TRY(hipGetLastError());
    return SCARLET_OK;
}
