"""Sub-pixel resampling kernels (reference ``scarlet/interpolation.py``): the separable kernels
``bilinear`` / ``lanczos`` (a handful of taps, tabulated on the host like the reference does) and
``fft_resample``, whose convolution runs in the HIP library.  The reference multiplies FFTs of the
image and the kernel after padding both by kernel size + 3 -- more than the taps reach -- so its
result is the plain linear convolution; the device evaluates that directly in float64."""
import numpy as np

from . import _lib
from .operator import _OnDevice


def bilinear(dx):
    """Two-tap linear kernel for a shift by dx in [-1, 1] (reference interpolation.py:139-165)."""
    if np.abs(dx) > 1:
        raise ValueError("The fractional shift dx must be between -1 and 1")
    if dx >= 0:
        return np.array([1 - dx, dx]), np.arange(2)
    return np.array([-dx, 1 + dx]), np.array([-1, 0])


def lanczos(dx, a=3):
    """Lanczos kernel with 2a taps at floor(dx) + (-a+1 .. a) (reference interpolation.py:233-252)."""
    if np.abs(dx) > 1:
        raise ValueError("The fractional shift dx must be between -1 and 1")
    window = np.arange(-a + 1, a + 1) + np.floor(dx)
    y = np.sinc(dx - window) * np.sinc((dx - window) / a)
    return y, window.astype(int)


def get_separable_kernel(dy, dx, kernel=lanczos, **kwargs):
    """2-D kernel = outer(ky, kx) with its pixel windows (reference interpolation.py:273-299)."""
    kx, x_window = kernel(dx, **kwargs)
    ky, y_window = kernel(dy, **kwargs)
    return np.outer(ky, kx), y_window, x_window


def fft_resample(img, dy, dx, kernel=lanczos, **kwargs):
    """Translate `img` by the fraction of a pixel (dy, dx) (reference interpolation.py:408-448).
    Returns a new array of the same kind as `img` (device tensor in, device tensor out)."""
    torch = _lib.require_gpu()
    kx, xwin = kernel(dx, **kwargs)
    ky, ywin = kernel(dy, **kwargs)
    if len(ky) > 8 or len(kx) > 8:
        raise ValueError("resampling kernels have at most 8 taps")
    taps = np.zeros((1, 2, 8), dtype=np.float64)
    taps[0, 0, :len(ky)] = ky
    taps[0, 1, :len(kx)] = kx
    win0 = np.array([[int(ywin[0]), int(xwin[0])]], dtype=np.int32)
    is_tensor = torch.is_tensor(img)
    src = img if is_tensor else torch.as_tensor(np.ascontiguousarray(img))
    t = src.to(device="cuda", dtype=torch.float32).contiguous()
    out = torch.empty_like(t)
    tp = torch.as_tensor(taps).cuda()
    w0 = torch.as_tensor(win0).cuda()
    H, W = t.shape
    _lib.check(_lib.lib.scarlet_resample(_lib.ptr(t), _lib.ptr(out), 1, H, W, _lib.ptr(tp), _lib.ptr(w0),
                                         len(ky), len(kx), _lib.stream_ptr()))
    if is_tensor:
        return out.to(device=img.device, dtype=img.dtype)
    return out.cpu().numpy().astype(np.asarray(img).dtype if np.asarray(img).dtype.kind == "f" else np.float64)
