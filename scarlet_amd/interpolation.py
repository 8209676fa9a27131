"""Sub-pixel resampling kernels (reference ``scarlet/interpolation.py``): the separable kernels
``bilinear`` / ``lanczos`` (a handful of taps, tabulated on the host like the reference does) and
``fft_resample``, whose convolution runs in the HIP library.  The reference multiplies FFTs of the
image and the kernel after padding both by kernel size + 3 -- more than the taps reach -- so its
result is the plain linear convolution; the device evaluates that directly in float64."""
import numpy as np

from . import _lib
from .operator import _OnDevice


def get_projection_slices(image, shape, yx0=None):
    """Slices that place `image` into an array of `shape` -- its first pixel at shape // 2 + yx0, centred
    (yx0 = -(image.shape // 2)) when yx0 is None -- clipped at the borders: (slices into the
    projection, slices into the image, (bottom, top, left, right)) (reference interpolation.py:6-54).
    Integer logic on the host."""
    Ny, Nx = shape
    iNy, iNx = image.shape
    if yx0 is None:
        yx0 = (-(iNy // 2), -(iNx // 2))
    bottom = yx0[0] + (Ny >> 1)
    left = yx0[1] + (Nx >> 1)
    top, right = bottom + iNy, left + iNx
    yslice = slice(max(0, bottom), min(Ny, top))
    iyslice = slice(max(0, -bottom), max(Ny - bottom, -top))
    xslice = slice(max(0, left), min(Nx, right))
    ixslice = slice(max(0, -left), max(Nx - left, -right))
    return (yslice, xslice), (iyslice, ixslice), (bottom, top, left, right)


def project_image(image, shape, yx0=None):
    """`image` padded with zeros and / or trimmed to `shape` (reference interpolation.py:57-84)."""
    result = np.zeros(shape)
    bb, ibb, _ = get_projection_slices(image, shape, yx0)
    result[bb] = image[ibb]
    return result


def common_projections(img1, img2):
    """Both images projected onto the smallest shape that holds them (reference interpolation.py:87-111)."""
    shape = (max(img1.shape[0], img2.shape[0]), max(img1.shape[1], img2.shape[1]))
    return project_image(img1, shape), project_image(img2, shape)


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


def cubic_spline(dx, a=1, b=0):
    """Four-tap cubic (Mitchell-Netravali family, sharpness a, shape b) at floor(dx) + (-1 .. 2)
    (reference interpolation.py:168-213)."""
    if np.abs(dx) > 1:
        raise ValueError("The fractional shift dx must be between -1 and 1")
    window = np.arange(-1, 3) + np.floor(dx)
    x = np.abs(dx - window)
    near = ((12 - 6 * a - 9 * b) * x ** 3 + (6 * a + 12 * b - 18) * x ** 2 + (6 - 2 * b)) / 6        # |x| <= 1
    far = ((-6 * a - b) * x ** 3 + (30 * a + 6 * b) * x ** 2 - (48 * a + 12 * b) * x + (24 * a + 8 * b)) / 6   # 1 < |x| < 2
    y = np.where(x <= 1, near, np.where(x < 2, far, 0.0))
    return y, window.astype(int)


def catmull_rom(dx):
    """cubic_spline with a = 1/2, b = 0 (reference interpolation.py:216-221)."""
    return cubic_spline(dx, a=.5, b=0)


def mitchel_netravali(dx):
    """cubic_spline with a = b = 1/3 (reference interpolation.py:224-230)."""
    return cubic_spline(dx, a=1 / 3, b=1 / 3)


def quintic_spline(dx, dtype=np.float64):
    """Seven-tap quintic spline on the fixed window -3 .. 3 (reference interpolation.py:255-270)."""
    window = np.arange(-3, 4)
    x = np.abs(dx - window).astype(dtype)
    inner = 1 + x ** 3 / 12 * (-95 + 138 * x - 55 * x ** 2)
    middle = (x - 1) * (x - 2) / 24 * (-138 + 348 * x - 249 * x ** 2 + 55 * x ** 3)
    outer = (x - 2) * (x - 3) ** 2 / 24 * (-54 + 50 * x - 11 * x ** 2)
    y = np.where(x <= 1, inner, np.where(x <= 2, middle, np.where(x <= 3, outer, 0.0)))
    return y, window


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
    if len(ky) > 12 or len(kx) > 12:
        raise ValueError("resampling kernels have at most 12 taps")
    taps = np.zeros((1, 2, 12), dtype=np.float64)
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
