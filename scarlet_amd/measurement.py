"""Centre measurements (reference ``scarlet/measurement.py``): ``max_pixel`` and
``psf_weighted_centroid`` run in the HIP library; results are Python ints/floats."""
import ctypes

import numpy as np

from . import _lib
from .operator import _OnDevice, _i32, _i32_center


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
        H, W = t.shape
        c = _i32_center(pixel_center, H, W)
        sh = torch.zeros((1, 2), dtype=torch.float64, device="cuda")
        st = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.check(_lib.lib.scarlet_psf_weighted_centroid(_lib.ptr(t), 1, H, W, _lib.ptr(pd), p.shape[0],
                                                          _lib.ptr(c), _lib.ptr(sh), _lib.ptr(st), _lib.stream_ptr()))
        cen, shift = c.cpu().numpy()[0], sh.cpu().numpy()[0]
    return (int(cen[0]), int(cen[1])), (float(shift[0]), float(shift[1]))


def threshold(morph):
    """Noise cut from the histogram of log10(positive pixels) (reference measurement.py:97-112):
    50 equal bins, size/10 when fewer than 500 pixels are positive (a single bin means no cut);
    the cut is the lower edge of the last empty bin, 0 when no bin is empty.  Returns
    (thresh, bins).  Range and histogram are device kernels; the 51 edges are tabulated with
    np.linspace exactly as np.histogram does."""
    torch = _lib.require_gpu()
    with _OnDevice(morph) as t:
        rng = torch.zeros((1, 3), dtype=torch.float64, device="cuda")
        _lib.check(_lib.lib.scarlet_log_range(_lib.ptr(t), 1, t.numel(), _lib.ptr(rng), _lib.stream_ptr()))
        size, lo, hi = rng.cpu().numpy()[0]
        bins = 50
        if size < 500:
            bins = max(int(size / 10), 1)
            if bins == 1:
                return 0, bins
        if lo == hi:                      # np.histogram widens a degenerate range by +-0.5
            lo, hi = lo - 0.5, hi + 0.5
        edges = np.zeros((1, 51), dtype=np.float64)
        edges[0, :bins + 1] = np.linspace(lo, hi, bins + 1, endpoint=True, dtype=np.float64)
        e = torch.as_tensor(edges).cuda()
        nb = torch.as_tensor(np.array([bins], dtype=np.int32)).cuda()
        hist = torch.zeros((1, 50), dtype=torch.int32, device="cuda")
        _lib.check(_lib.lib.scarlet_log_hist(_lib.ptr(t), 1, t.numel(), _lib.ptr(e), _lib.ptr(nb),
                                             _lib.ptr(hist), _lib.stream_ptr()))
        h = hist.cpu().numpy()[0, :bins]
    cutoff = np.where(h == 0)[0]
    if len(cutoff) == 0:
        return 0, bins
    return 10 ** edges[0, cutoff[-1]], bins
