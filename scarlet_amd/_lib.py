"""ctypes binding of the C-ABI HIP library (include/scarlet_hip.h).

There is NO CPU fallback: importing this module without the built
``scarlet_amd/csrc/libscarlet_hip.so`` raises, and every compute entry point needs a
ROCm device.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C scarlet_amd/csrc``.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64,
                    c_uint8, c_void_p)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libscarlet_hip.so")
if os.environ.get("SCARLET_LIB_PATH"):      # development: an A/B build of the same library (tools/ab_variants.sh)
    LIB_PATH = os.environ["SCARLET_LIB_PATH"]


class HipLibraryMissing(ImportError):
    pass


if not os.path.exists(LIB_PATH):
    raise HipLibraryMissing(
        "scarlet_amd: {} not found.  The engine has no CPU fallback; build the HIP "
        "library first (make -C scarlet_amd/csrc).".format(LIB_PATH))

# PyTorch first.  The torch wheel carries its own HIP / HSA runtime libraries and libscarlet_hip.so is linked against
# the system's (/opt/rocm/lib): when the system copies are mapped into the process BEFORE torch's, every kernel launch of
# this library later fails with "no ROCm-capable device is detected" (seen with `import scarlet_amd` as a script's first
# import; tests/test_gpu_api.py::test_package_imported_before_torch).  torch owns device memory and streams for this
# package anyway, so its runtime is the one to load first.
import torch  # noqa: E402,F401  (import order matters, see above)

lib = ctypes.CDLL(LIB_PATH)

# error codes / enums (mirror include/scarlet_hip.h)
OK, E_ARG, E_TOO_LARGE, E_HIP, E_NOTIMPL = 0, -1, -2, -3, -4
FLAG_SED_NOT_CONVERGED, FLAG_MORPH_NOT_CONVERGED, FLAG_EDGE_PIXELS, FLAG_NO_VALID_PIXELS = 1, 2, 4, 8
STATUS_CENTER_AT_EDGE, STATUS_NONFINITE = 1, 2
SYM_KSPACE, SYM_SOFT, SYM_SDSS = 0, 1, 2
SYM_FULL_WINDOW = 16
NORM_SED, NORM_MORPH, NORM_MORPH_MAX = 0, 1, 2


class ScarletBatch(Structure):
    """struct scarlet_batch of include/scarlet_hip.h (field order must match)."""
    _fields_ = [
        ("S", c_int32), ("K", c_int32), ("B", c_int32), ("H", c_int32), ("W", c_int32),
        ("images", c_void_p), ("weights", c_void_p), ("weight_scalar", c_float),
        ("sed", c_void_p * 2), ("morph", c_void_p * 2), ("cur", c_void_p),
        ("centers", c_void_p), ("shifts", c_void_p), ("flags", c_void_p),
        ("fix_sed", c_void_p), ("fix_morph", c_void_p),
        ("lipschitz", c_void_p), ("mse", c_void_p), ("mse_capacity", c_int32),
        ("it", c_void_p), ("active", c_void_p), ("status", c_void_p),
        ("symmetric", c_int32), ("monotonic", c_int32),
        ("l0_thresh", c_float), ("l1_thresh", c_float),
        ("centroid_psf", c_void_p), ("centroid_P", c_int32),
        ("diff_kernel", c_void_p), ("psf_h", c_int32), ("psf_w", c_int32),
        ("diff_kernel_per_scene", c_int32),
        ("workspace", c_void_p),
        ("group", c_void_p),
    ]


_P = c_void_p
_SIGNATURES = {
    "scarlet_version": (c_char_p, []),
    "scarlet_last_error": (c_char_p, []),
    "scarlet_set_option": (c_int, [c_char_p, c_int]),
    "scarlet_debug_stamps": (c_int64, [_P, c_int64]),
    "scarlet_next_fast_len": (c_int, [c_int]),
    "scarlet_host_prox_monotonic_f64": (c_int, [_P, c_int, _P, _P, c_int, c_double]),
    "scarlet_host_prox_weighted_monotonic_f32": (c_int, [_P, c_int, _P, _P, _P, c_int, c_float]),
    "scarlet_host_prox_weighted_monotonic_f64": (c_int, [_P, c_int, _P, _P, _P, c_int, c_double]),
    "scarlet_host_apply_filter_f32": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P]),
    "scarlet_host_apply_filter_f64": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P]),
    "scarlet_prox_weighted_monotonic": (c_int, [_P, c_int, c_int, c_int, _P, c_float, _P]),
    "scarlet_prox_nearest_monotonic": (c_int, [_P, c_int, c_int, c_int, _P, c_float, _P]),
    "scarlet_prox_symmetry": (c_int, [_P, c_int, c_int, c_int, _P, _P, c_int, c_float, c_int, c_float, _P]),
    "scarlet_max_pixel": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "scarlet_psf_weighted_centroid": (c_int, [_P, c_int, c_int, c_int, _P, c_int, _P, _P, _P, _P]),
    "scarlet_match_psfs": (c_int, [_P, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P, _P]),
    "scarlet_prox_plus": (c_int, [_P, c_int64, _P]),
    "scarlet_prox_hard": (c_int, [_P, c_int64, c_float, _P]),
    "scarlet_prox_soft": (c_int, [_P, c_int64, c_float, _P]),
    "scarlet_normalize": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "scarlet_log_range": (c_int, [_P, c_int, c_int64, _P, _P]),
    "scarlet_log_hist": (c_int, [_P, c_int, c_int64, _P, _P, _P, _P]),
    "scarlet_cut_below": (c_int, [_P, c_int64, c_double, _P]),
    "scarlet_trim": (c_int, [_P, c_int, c_int, c_int, c_float, _P, _P]),
    "scarlet_resample": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, c_int, c_int, _P]),
    "scarlet_apply_filter": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P, _P]),
    "scarlet_batch_workspace_bytes": (c_int64, [POINTER(ScarletBatch)]),
    "scarlet_batch_pipelines": (c_int, [POINTER(ScarletBatch)]),
    "scarlet_fit": (c_int, [POINTER(ScarletBatch), c_int, c_double, c_int, c_int, _P]),
    "scarlet_fit_multi": (c_int, [POINTER(ScarletBatch), POINTER(POINTER(ScarletBatch)), _P, c_int, c_int, c_double, c_int, c_int, _P]),
    "scarlet_backward_step": (c_int, [POINTER(ScarletBatch), c_int, _P]),
    "scarlet_backward_gradients": (c_int, [POINTER(ScarletBatch), c_int, _P]),
    "scarlet_source_update": (c_int, [POINTER(ScarletBatch), c_int, _P]),
    "scarlet_check_convergence": (c_int, [POINTER(ScarletBatch), c_double, _P]),
    "scarlet_debug_psf_stamps_offset": (c_int64, [_P]),
    "scarlet_debug_psf_plan": (c_int, [_P, _P]),
    "scarlet_profile_begin": (c_int, [c_int]),
    "scarlet_profile_end": (c_int, [_P, _P]),
    "scarlet_profile_end_ex": (c_int, [_P, _P, _P]),
    "scarlet_init_extended": (c_int, [POINTER(ScarletBatch), _P, c_float, _P, c_int, c_int, c_int, _P]),
    "scarlet_convergence_sums": (c_int, [POINTER(ScarletBatch), _P]),
    "scarlet_batch_prepare_psf": (c_int, [POINTER(ScarletBatch), _P]),
    "scarlet_convolve_same": (c_int, [_P, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P, _P]),
}

EXPORTS = tuple(_SIGNATURES)

for _name, (_res, _args) in _SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here = header/library mismatch
    _fn.restype = _res
    _fn.argtypes = _args


def set_option(name, value):
    """Diagnostic switch of the library (scarlet_set_option); returns the previous value."""
    return check(lib.scarlet_set_option(name.encode(), int(value)))


def last_error():
    return lib.scarlet_last_error().decode("utf-8", "replace")


def check(rc):
    """Map a C status to the exception type the reference raises in the same situation."""
    if rc >= 0:
        return rc
    msg = last_error()
    if rc == E_ARG:
        raise ValueError(msg)
    if rc == E_NOTIMPL:
        raise NotImplementedError(msg)
    if rc == E_TOO_LARGE:
        raise ValueError(msg)
    raise RuntimeError("scarlet_hip: " + msg)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("scarlet_amd needs a ROCm device (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
    return torch


def stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else c_void_p(t.data_ptr())
