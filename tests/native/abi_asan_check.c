/* Host-side sanitizer check of the C ABI (tests/test_abi.py::test_asan_error_paths builds and runs it).
 * Links the AddressSanitizer build of the library (make -C scarlet_amd/csrc asan: host code instrumented,
 * device code untouched) and walks the argument-error and HIP-error exit paths of every entry point that
 * allocates.  On a machine without a GPU every hipMalloc fails, i.e. exactly the early-return paths that
 * used to leak run; with a GPU the calls succeed on tiny inputs.  ASan/LSan abort on a leak or overflow. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "scarlet_hip.h"

#define EXPECT(cond) do { if (!(cond)) { fprintf(stderr, "FAILED line %d: %s (last error: %s)\n", __LINE__, #cond, scarlet_last_error()); return 1; } } while (0)

int main(void)
{
    enum { H = 6, W = 5, N = H * W };
    float x32[N], w32[8 * N], res[N], vals[3] = {1.f, .5f, .25f};
    double x64[N], w64[8 * N];
    int offsets[8] = {-W - 1, -W, -W + 1, -1, 1, W - 1, W, W + 1}, dist[N - 1], ref[N];
    int ys[3] = {0, 1, 0}, ye[3] = {0, 0, 1}, xs[3] = {0, 0, 1}, xe[3] = {0, 1, 0};
    for (int i = 0; i < N; ++i) { x32[i] = (float)(i % 7); x64[i] = x32[i]; ref[i] = N / 2; }
    for (int i = 0; i < 8 * N; ++i) { w32[i] = 0.125f; w64[i] = 0.125; }
    for (int i = 0, d = 0; i < N; ++i) if (i != N / 2) dist[d++] = i;
    /* keep the sweeps inside the array: zero the weights of neighbours that fall outside */
    for (int k = 0; k < 8; ++k) for (int i = 0; i < N; ++i) if (i + offsets[k] < 0 || i + offsets[k] >= N) { w32[k * N + i] = 0; w64[k * N + i] = 0; }

    EXPECT(scarlet_version() != NULL);
    EXPECT(scarlet_next_fast_len(172) == 180);
    EXPECT(scarlet_set_option("NO_SUCH_OPTION", 1) == SCARLET_E_ARG);
    EXPECT(scarlet_set_option("NO_FUSED", 1) >= 0 && scarlet_set_option("NO_FUSED", 0) == 1);
    /* argument errors: detected before any allocation */
    EXPECT(scarlet_host_prox_weighted_monotonic_f32(NULL, N, w32, offsets, dist, N - 1, 0.f) == SCARLET_E_ARG);
    EXPECT(scarlet_host_prox_monotonic_f64(x64, N, NULL, dist, N - 1, 0.) == SCARLET_E_ARG);
    EXPECT(scarlet_host_apply_filter_f32(x32, H, W, vals, ys, ye, xs, xe, -1, res) == SCARLET_E_ARG);
    EXPECT(scarlet_match_psfs(NULL, 1, 5, 5, x32, 1, 5, 5, res, NULL) == SCARLET_E_ARG);
    EXPECT(scarlet_convolve_same(x32, 1, H, W, NULL, 1, 3, 3, res, NULL) == SCARLET_E_ARG);
    dist[0] = N + 5;                                   /* out-of-range order index */
    EXPECT(scarlet_host_prox_weighted_monotonic_f64(x64, N, w64, offsets, dist, N - 1, 0.) == SCARLET_E_ARG);
    dist[0] = 0;
    /* calls that allocate: SCARLET_OK with a GPU, SCARLET_E_HIP without; either way nothing may leak */
    int rc[6];
    rc[0] = scarlet_host_prox_weighted_monotonic_f32(x32, N, w32, offsets, dist, N - 1, 0.f);
    rc[1] = scarlet_host_prox_weighted_monotonic_f64(x64, N, w64, offsets, dist, N - 1, 0.1);
    rc[2] = scarlet_host_prox_monotonic_f64(x64, N, ref, dist, N - 1, 0.);
    rc[3] = scarlet_host_apply_filter_f32(x32, H, W, vals, ys, ye, xs, xe, 3, res);
    int gpu = rc[0] == SCARLET_OK;
    for (int i = 0; i < 4; ++i) EXPECT(rc[i] == (gpu ? SCARLET_OK : SCARLET_E_HIP));
    if (!gpu) {
        /* device-pointer entry points with their own temporaries: with no device the first allocation fails */
        rc[4] = scarlet_convolve_same(x32, 1, H, W, x32, 1, 3, 3, res, NULL);
        rc[5] = scarlet_match_psfs(x32, 1, 5, 5, x32, 1, 5, 5, res, NULL);
        EXPECT(rc[4] == SCARLET_E_HIP && rc[5] == SCARLET_E_HIP);
        EXPECT(strlen(scarlet_last_error()) > 0);
    }
    printf("abi_asan_check ok (%s)\n", gpu ? "gpu" : "no gpu: error paths");
    return 0;
}
