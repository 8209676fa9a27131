"""Component-level constraint functions ``f(component, **params) -> component`` (in place).
Mirrors the reference's ``scarlet/update.py``; each one is a call into the HIP library on
the component's device tensors."""
import ctypes

import numpy as np

from . import _lib
from . import operator
from . import measurement
from . import interpolation
from .bbox import trim


def _elementwise(fn, t, *args):
    _lib.check(fn(_lib.ptr(t), t.numel(), *args, _lib.stream_ptr()))


def positive_sed(component):
    """SED >= 0 (proxmin prox_plus; reference update.py:13-17)."""
    _elementwise(_lib.lib.scarlet_prox_plus, component.sed)
    return component


def positive_morph(component):
    """morphology >= 0 (reference update.py:20-24)."""
    _elementwise(_lib.lib.scarlet_prox_plus, component.morph)
    return component


def positive(component):
    """SED and morphology >= 0 (reference update.py:27-32)."""
    _elementwise(_lib.lib.scarlet_prox_plus, component.sed)
    _elementwise(_lib.lib.scarlet_prox_plus, component.morph)
    return component


def normalized(component, type='morph_max'):
    """Break the SED/morphology scale degeneracy: 'sed' (SED sums to one), 'morph'
    (morphology sums to one) or 'morph_max' (peak of the morphology is one)
    (reference update.py:35-68)."""
    kinds = {'sed': _lib.NORM_SED, 'morph': _lib.NORM_MORPH, 'morph_max': _lib.NORM_MORPH_MAX}
    t = type.lower()
    if t not in kinds:
        raise ValueError("Unrecognized normalization '{0}'".format(type))
    sed, morph = component.sed, component.morph
    _lib.check(_lib.lib.scarlet_normalize(_lib.ptr(sed), _lib.ptr(morph), 1, sed.numel(), morph.numel(),
                                          kinds[t], _lib.stream_ptr()))
    return component


def sparse_l0(component, thresh):
    """L0 sparsity: zero where |morph| < thresh * step_morph (reference update.py:71-75)."""
    _elementwise(_lib.lib.scarlet_prox_hard, component.morph, ctypes.c_float(thresh * component.step_morph))
    return component


def sparse_l1(component, thresh):
    """L1 sparsity: soft threshold by thresh * step_morph (reference update.py:78-82)."""
    _elementwise(_lib.lib.scarlet_prox_soft, component.morph, ctypes.c_float(thresh * component.step_morph))
    return component


def threshold(component):
    """Zero the pixels below the log-histogram noise cut and record the tight box of what is
    left in component.bboxes["thresh"] (reference update.py:85-103)."""
    thresh, _bins = measurement.threshold(component.morph)
    m = component.morph
    _lib.check(_lib.lib.scarlet_cut_below(_lib.ptr(m), m.numel(), ctypes.c_double(float(thresh)), _lib.stream_ptr()))
    bbox = trim(m)
    if not hasattr(component, "bboxes"):
        component.bboxes = {}
    component.bboxes["thresh"] = bbox
    return component


def _bbox_window(component, pixel_center, bbox):
    if bbox is None:
        return component.morph, pixel_center, None
    sl = bbox.slices
    view = component.morph[sl]
    return view, (pixel_center[0] - bbox.bottom, pixel_center[1] - bbox.left), sl


def monotonic(component, pixel_center, use_nearest=False, thresh=0, exact=False, bbox=None):
    """Radially monotonic morphology about `pixel_center`; with a `bbox` only inside it, the
    rest is zeroed (reference update.py:106-156)."""
    view, center, sl = _bbox_window(component, pixel_center, bbox)
    if sl is not None and (view.shape[0] <= 1 or view.shape[1] <= 1):
        return component
    if exact:
        # Exact monotonicity is not supported by the reference either (update.py:143)
        raise NotImplementedError("Exact monotonicity is not currently supported")
    prox = operator.prox_strict_monotonic(tuple(view.shape), use_nearest=use_nearest, thresh=thresh,
                                          center=center)
    work = view.clone().contiguous()
    prox(work, component.step_morph)
    if sl is not None:
        component.morph.zero_()
    view.copy_(work)
    return component


def translation(component, direction=1, kernel=interpolation.lanczos, padding=3):
    """Shift the morphology by direction * component.shift with a separable resampling kernel
    (reference update.py:159-167; Lanczos-3 by default)."""
    dy, dx = component.shift
    dy *= direction
    dx *= direction
    component.morph[:] = interpolation.fft_resample(component.morph, dy, dx, kernel=kernel)
    return component


def symmetric(component, pixel_center, algorithm="kspace", bbox=None, fill=None, strength=.5):
    """Point symmetry about `pixel_center` (reference update.py:170-199).

    Reference quirk kept: with a `bbox` the reference operates on a *view* of
    component.morph, zeroes component.morph (which also zeroes the view) and writes the view
    back -- the morphology ends up all zero (update.py:194-196)."""
    view, center, sl = _bbox_window(component, pixel_center, bbox)
    if sl is not None and (view.shape[0] <= 1 or view.shape[1] <= 1):
        return component
    shift = getattr(component, "shift", None)
    if sl is None:
        operator.prox_uncentered_symmetry(view, component.step_morph, center, algorithm, fill, shift, strength)
    else:
        if algorithm not in ("kspace", "soft", "sdss"):
            raise ValueError("algorithm must be one of 'soft', 'sdss', 'kspace', recieved '{0}''".format(algorithm))
        component.morph.zero_()
    return component
