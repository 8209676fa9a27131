"""Centre measurements (reference ``scarlet/measurement.py``): ``max_pixel`` and
``psf_weighted_centroid`` run in the HIP library; results are Python ints/floats."""
import ctypes

import numpy as np

from . import _lib
from .operator import _OnDevice, _i32


def max_pixel(morph, center=None, window=None):
    """Brightest pixel of the 5x5 window around `center` (first hit, row-major), as
    absolute (y, x).  A custom `window` (tuple of slices) is evaluated with one device
    argmax (reference measurement.py:3-29)."""
    torch = _lib.require_gpu()
    if center is None:
        center = (morph.shape[0] // 2, morph.shape[1] // 2)
    cy, cx = int(center[0]), int(center[1])
    if window is not None:
        t = morph if torch.is_tensor(morph) else torch.as_tensor(np.asarray(morph)).cuda()
        sub = t[window]
        flat = int(sub.argmax())
        y0 = window[0].start or 0
        x0 = window[1].start or 0
        return (flat // sub.shape[1] + y0, flat % sub.shape[1] + x0)
    with _OnDevice(morph) as t:
        c = _i32((cy, cx))
        st = torch.zeros(1, dtype=torch.int32, device="cuda")
        H, W = t.shape
        _lib.check(_lib.lib.scarlet_max_pixel(_lib.ptr(t), 1, H, W, _lib.ptr(c), _lib.ptr(st), _lib.stream_ptr()))
        out, status = c.cpu().numpy()[0], int(st.item())
    if status & _lib.STATUS_CENTER_AT_EDGE:
        # the reference's slice(cy-2, cy+3) goes negative here and its argmax fails
        raise ValueError("attempt to get argmax of an empty sequence")
    return (int(out[0]), int(out[1]))


def psf_weighted_centroid(morph, psf, pixel_center):
    """PSF-weighted first moments around `pixel_center`: returns the rounded centre and the
    sub-pixel shift (dy, dx) = rounded - centroid (reference measurement.py:32-94)."""
    torch = _lib.require_gpu()
    p = psf.detach().cpu().numpy() if torch.is_tensor(psf) else np.asarray(psf)
    assert p.ndim == 2 and p.shape[0] == p.shape[1] and p.shape[0] % 2 == 1, "psf must be square and odd"
    pd = torch.as_tensor(np.ascontiguousarray(p, dtype=np.float64)).cuda()
    with _OnDevice(morph) as t:
        c = _i32(pixel_center)
        sh = torch.zeros((1, 2), dtype=torch.float64, device="cuda")
        st = torch.zeros(1, dtype=torch.int32, device="cuda")
        H, W = t.shape
        _lib.check(_lib.lib.scarlet_psf_weighted_centroid(_lib.ptr(t), 1, H, W, _lib.ptr(pd), p.shape[0],
                                                          _lib.ptr(c), _lib.ptr(sh), _lib.ptr(st), _lib.stream_ptr()))
        cen, shift = c.cpu().numpy()[0], sh.cpu().numpy()[0]
    return (int(cen[0]), int(cen[1])), (float(shift[0]), float(shift[1]))


def threshold(morph):
    """Log-histogram cut of the reference (measurement.py:97-112): disabled in the reference's
    own pipeline (source.py:416-418) and outside the hot path."""
    raise NotImplementedError("measurement.threshold is outside the accelerated path")
