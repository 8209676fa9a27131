"""Container-only harness that lets the *unmodified* reference package import and run.

TEST INFRASTRUCTURE -- not product code.  Only ``oracle/gen_golden.py`` (fixture
generation, run in the build container where ``/root/reference`` exists) uses this
module.  Nothing here travels to the GPU box in a form that is needed at run time:
the fixtures it produces are committed under ``tests/golden/``.

The reference (``/root/reference/scarlet``) depends on two third-party packages that
are not installed in this image and cannot be fetched (no network):

* ``autograd>=1.3``   (``scarlet/blend.py:1-2``, ``component.py:6``, ``fft.py:3``, ...)
* ``proxmin>=0.5.5``  (``scarlet/update.py:4``, ``operator.py:4-5``)

and on one compiled module, ``scarlet.operators_pybind11`` (needs Eigen, absent).
SURVEY.md Appendix B lists the minimal stand-ins; they are restated here.  None of it is
reference code: ``autograd.numpy`` *is* numpy on the forward pass, the four proxmin
operators are restated from the behaviour pinned by the reference's own tests
(``tests/test_update.py:23-44,74-97``), the three C++ loops are restated from
``scarlet/operators_pybind11.cc:11-70`` and the gradient is the analytic adjoint of
``Observation.get_loss`` (``scarlet/observation.py:222-239``).
"""
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


def _prox_plus(X, step):
    X[X < 0] = 0
    return X


def _prox_hard(X, step, thresh=0):
    t = thresh(step) if callable(thresh) else thresh * step
    X[np.abs(X) < t] = 0
    return X


def _prox_soft(X, step, thresh=0):
    t = thresh(step) if callable(thresh) else thresh * step
    X[:] = np.sign(X) * _prox_plus(np.abs(X) - t, step)
    return X


def _prox_unity_plus(X, step, axis=0):
    _prox_plus(X, step)
    X[:] = X / X.sum(axis=axis, keepdims=True)
    return X


class _MatrixAdapter(object):
    def __init__(self, L, axis=None):
        self.L = L
        self.axis = axis

    @property
    def spectral_norm(self):
        return None


def _pybind_prox_monotonic(X, step, ref_idx, dist_idx, thresh):
    # operators_pybind11.cc:11-25 (double only, sequential)
    for d in dist_idx:
        X[d] = min(X[d], X[ref_idx[d]] * (1 - thresh))


def _pybind_prox_weighted_monotonic(flat, step, weights, offsets, dist_idx, thresh):
    # operators_pybind11.cc:27-50; accumulate in the array's own type, i = 0..7
    T = flat.dtype.type
    w = weights.astype(flat.dtype)
    one_minus = T(1) - T(thresh)
    for d in dist_idx:
        ref = T(0)
        for i in range(len(offsets)):
            if w[i, d] > 0:
                ref = T(ref + flat[d + offsets[i]] * w[i, d])
        flat[d] = min(flat[d], T(ref * one_minus))


def _pybind_apply_filter(image, values, y_start, y_end, x_start, x_end, result):
    # operators_pybind11.cc:53-70
    result[:] = 0
    for n in range(len(values)):
        rows = image.shape[0] - y_start[n] - y_end[n]
        cols = image.shape[1] - x_start[n] - x_end[n]
        result[y_start[n]:y_start[n] + rows, x_start[n]:x_start[n] + cols] += (
            values[n] * image[y_end[n]:y_end[n] + rows, x_end[n]:x_end[n] + cols])


def _analytic_grad(fun, argnums):
    """Stand-in for ``autograd.grad(self._loss, range(2K))`` (blend.py:44-45).

    ``fun`` is the bound ``Blend._loss``; the returned callable evaluates the loss,
    appends it to ``blend.mse`` (blend.py:138 does that from inside ``_loss``) and
    returns d loss / d sed_k, d loss / d morph_k analytically.
    """
    blend = fun.__self__
    from scarlet import fft as rfft

    def adjoint_render(obs, g):
        if obs._diff_kernels is None:
            return g
        kern = obs._diff_kernels
        F = rfft._get_fft_shape(g, kern.image, 3, (1, 2))
        Khat = kern.fft(F, (1, 2))
        gp = np.fft.ifftshift(rfft._pad(g, F, (1, 2)), (1, 2))
        ghat = np.fft.rfftn(gp, axes=(1, 2)) * np.conj(Khat)
        out = np.fft.fftshift(np.fft.irfftn(ghat, F, axes=(1, 2)), (1, 2))
        return rfft._centered(out, g.shape)

    def g(*params):
        K = blend.K
        seds, morphs = params[:K], params[K:]
        model = blend.get_model(seds, morphs)
        G = np.zeros(model.shape, dtype=model.dtype)
        loss = 0
        for obs in blend.observations:
            d = obs.weights * (obs.render(model) - obs.images)
            loss = loss + 0.5 * np.sum(d ** 2)
            G[obs._band_slice] += adjoint_render(obs, obs.weights * d)
        blend.mse.append(loss)
        sed_grads = tuple((G * m[None]).sum(axis=(1, 2)) for m in morphs)
        morph_grads = tuple((G * s[:, None, None]).sum(axis=0) for s in seds)
        return sed_grads + morph_grads

    return g


_loaded = None


def load_reference():
    """Import the reference ``scarlet`` package under the stand-ins; returns the module."""
    global _loaded
    if _loaded is not None:
        return _loaded
    sys.dont_write_bytecode = True   # the reference tree is read-only

    ag = types.ModuleType("autograd")
    ag.numpy = np
    ag.grad = _analytic_grad
    boxes = types.ModuleType("autograd.numpy.numpy_boxes")
    boxes.ArrayBox = type("ArrayBox", (), {})
    sys.modules["autograd"] = ag
    sys.modules["autograd.numpy"] = np
    sys.modules["autograd.numpy.numpy_boxes"] = boxes

    px = types.ModuleType("proxmin")
    pxo = types.ModuleType("proxmin.operators")
    pxo.prox_plus, pxo.prox_hard, pxo.prox_soft = _prox_plus, _prox_hard, _prox_soft
    pxo.prox_unity_plus = _prox_unity_plus
    pxu = types.ModuleType("proxmin.utils")
    pxu.MatrixAdapter = _MatrixAdapter
    px.operators, px.utils = pxo, pxu
    sys.modules["proxmin"] = px
    sys.modules["proxmin.operators"] = pxo
    sys.modules["proxmin.utils"] = pxu

    # environment drift (not reference behaviour): numpy>=1.24 dropped np.int,
    # scipy>=1.x rejects float arguments to next_fast_len
    if not hasattr(np, "int"):
        np.int = int
    import scipy.fftpack
    import scipy.fftpack.helper as helper
    helper.next_fast_len = lambda s: scipy.fftpack.next_fast_len(int(s))

    for name in ("astropy", "astropy.wcs", "astropy.visualization"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)

    pyb = types.ModuleType("scarlet.operators_pybind11")
    pyb.prox_monotonic = _pybind_prox_monotonic
    pyb.prox_weighted_monotonic = _pybind_prox_weighted_monotonic
    pyb.apply_filter = _pybind_apply_filter
    sys.modules["scarlet.operators_pybind11"] = pyb

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import scarlet
    scarlet.operators_pybind11 = pyb
    import scarlet.blend
    scarlet.blend.grad = _analytic_grad
    _loaded = scarlet
    return scarlet
