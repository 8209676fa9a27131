/*
 * oracle/sweeps.c -- CPU restatement of the reference's three native loops.
 *
 * TEST INFRASTRUCTURE: this file is the checker / CPU baseline, never the product
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load the library built from it (oracle/_build/liboracle_sweeps.so).
 *
 * Restates /root/reference/scarlet/operators_pybind11.cc:
 *   :11-25  prox_monotonic           (nearest-neighbour sweep, double only)
 *   :27-50  prox_weighted_monotonic  (8-neighbour weighted sweep, float and double)
 *   :53-70  apply_filter             (sum of shifted, scaled copies of an image)
 * The reference's file cannot be compiled here (needs Eigen headers, absent); the
 * loops are short enough to restate.  Parity is pinned by the golden 5x5 arrays of
 * the reference's tests (tests/test_operator.py:10-56, tests/test_update.py:120-176)
 * and by fixtures generated with the reference's Python driving transcribed loops.
 *
 * weights are row-major [8][n] (numpy C order of the reference's 8xN array).
 */
#include <stddef.h>

void oracle_prox_monotonic_f64(double *x, const int *ref_idx, const int *dist_idx,
                               int n_dist, double thresh)
{
    for (int d = 0; d < n_dist; ++d) {
        int p = dist_idx[d];
        double r = x[ref_idx[p]] * (1 - thresh);
        if (r < x[p]) x[p] = r;
    }
}

#define WEIGHTED_SWEEP(NAME, T)                                                         \
void NAME(T *x, const T *weights, int n, const int *offsets, int n_off,                 \
          const int *dist_idx, int n_dist, T thresh)                                    \
{                                                                                       \
    for (int d = 0; d < n_dist; ++d) {                                                  \
        int p = dist_idx[d];                                                            \
        T ref = 0;                                                                      \
        for (int i = 0; i < n_off; ++i) {                                               \
            T w = weights[(size_t)i * n + p];                                           \
            if (w > 0) ref += x[p + offsets[i]] * w;                                    \
        }                                                                               \
        T cap = ref * (1 - thresh);                                                     \
        if (cap < x[p]) x[p] = cap;                                                     \
    }                                                                                   \
}

WEIGHTED_SWEEP(oracle_prox_weighted_monotonic_f32, float)
WEIGHTED_SWEEP(oracle_prox_weighted_monotonic_f64, double)

#define APPLY_FILTER(NAME, T)                                                           \
void NAME(const T *image, int H, int W, const T *values, const int *y_start,            \
          const int *y_end, const int *x_start, const int *x_end, int n, T *result)     \
{                                                                                       \
    for (int i = 0; i < H * W; ++i) result[i] = 0;                                      \
    for (int k = 0; k < n; ++k) {                                                       \
        int rows = H - y_start[k] - y_end[k];                                           \
        int cols = W - x_start[k] - x_end[k];                                           \
        for (int r = 0; r < rows; ++r)                                                  \
            for (int c = 0; c < cols; ++c)                                              \
                result[(y_start[k] + r) * W + x_start[k] + c] +=                        \
                    values[k] * image[(y_end[k] + r) * W + x_end[k] + c];               \
    }                                                                                   \
}

APPLY_FILTER(oracle_apply_filter_f32, float)
APPLY_FILTER(oracle_apply_filter_f64, double)
