"""ctypes bindings of oracle/sweeps.c (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
import ctypes

import numpy as np

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_build.build())
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def prox_monotonic(x, ref_idx, dist_idx, thresh=0.0):
    """operators_pybind11.cc:11-25 -- x float64 1-D, mutated in place."""
    assert x.dtype == np.float64 and x.flags.c_contiguous
    ref_idx = np.ascontiguousarray(ref_idx, dtype=np.int32)
    dist_idx = np.ascontiguousarray(dist_idx, dtype=np.int32)
    lib().oracle_prox_monotonic_f64(_p(x, ctypes.c_double), _p(ref_idx, ctypes.c_int),
                                    _p(dist_idx, ctypes.c_int), ctypes.c_int(len(dist_idx)),
                                    ctypes.c_double(thresh))
    return x


def prox_weighted_monotonic(x, weights, offsets, dist_idx, thresh=0.0):
    """operators_pybind11.cc:27-50 -- x float32/float64 1-D, mutated in place.

    ``weights`` (8, N) is cast to x's dtype like pybind11's Eigen conversion does.
    """
    assert x.ndim == 1 and x.flags.c_contiguous
    w = np.ascontiguousarray(weights, dtype=x.dtype)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    dist_idx = np.ascontiguousarray(dist_idx, dtype=np.int32)
    if x.dtype == np.float32:
        fn, ct, th = lib().oracle_prox_weighted_monotonic_f32, ctypes.c_float, ctypes.c_float(thresh)
    elif x.dtype == np.float64:
        fn, ct, th = lib().oracle_prox_weighted_monotonic_f64, ctypes.c_double, ctypes.c_double(thresh)
    else:
        raise TypeError("prox_weighted_monotonic: float32 or float64 only")
    fn(_p(x, ct), _p(w, ct), ctypes.c_int(x.size), _p(offsets, ctypes.c_int),
       ctypes.c_int(len(offsets)), _p(dist_idx, ctypes.c_int), ctypes.c_int(len(dist_idx)), th)
    return x


def apply_filter(image, values, y_start, y_end, x_start, x_end, result):
    """operators_pybind11.cc:53-70 -- result (H, W) overwritten."""
    assert image.dtype == result.dtype and image.shape == result.shape
    image = np.ascontiguousarray(image)
    values = np.ascontiguousarray(values, dtype=image.dtype)
    idx = [np.ascontiguousarray(a, dtype=np.int32) for a in (y_start, y_end, x_start, x_end)]
    if image.dtype == np.float32:
        fn, ct = lib().oracle_apply_filter_f32, ctypes.c_float
    else:
        fn, ct = lib().oracle_apply_filter_f64, ctypes.c_double
    out = np.ascontiguousarray(result)
    fn(_p(image, ct), ctypes.c_int(image.shape[0]), ctypes.c_int(image.shape[1]),
       _p(values, ct), *[_p(a, ctypes.c_int) for a in idx], ctypes.c_int(len(values)),
       _p(out, ct))
    if out is not result:
        result[:] = out
    return result
