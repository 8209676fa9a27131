"""PSF profile helpers (host side, numpy): setup-time only, not on the per-iteration path.

Mirrors the reference's ``scarlet/psf.py`` API (``moffat``, ``gaussian``, ``double_gaussian``,
``generate_psf_image``); SURVEY.md section 2 row 13 marks it out of scope for kernels.
"""
from functools import partial

import numpy as np


def moffat(y, x, y0, x0, amplitude, alpha, beta=1.5):
    """Symmetric 2-D Moffat profile sampled on the grid y (rows) x x (columns)
    (reference psf.py:8-21)."""
    xx, yy = np.meshgrid(x, y)
    r2 = (xx - x0) ** 2 + (yy - y0) ** 2
    return amplitude * (1 + r2 / alpha ** 2) ** -beta


def gaussian(y, x, y0=0, x0=0, amplitude=None, sigma=1):
    """Circular Gaussian; the default amplitude is the reference's 1/(pi^2 sigma^2)
    (reference psf.py:24-47)."""
    if amplitude is None:
        amplitude = 1 / (np.pi ** 2 * sigma ** 2)
    xx, yy = np.meshgrid(x, y)
    return amplitude * np.exp(-((xx - x0) ** 2 + (yy - y0) ** 2) / (2 * sigma ** 2))


def double_gaussian(y, x, y0=0, x0=0, A1=None, sigma1=1, A2=None, sigma2=1):
    """Sum of two circular Gaussians (reference psf.py:50-53)."""
    return gaussian(y, x, y0, x0, A1, sigma1) + gaussian(y, x, y0, x0, A2, sigma2)


def integrate_pixels(y, x, func, subsamples):
    """Sub-sampled pixel integration, the reference's 2-D "trapezoid" rule including its
    0.4 corner factor (reference interpolation.py:506-552)."""
    n = int(subsamples)
    assert n % 2 == 0, "subsamples must be even, received {0}".format(n)
    dy, dx = y[1] - y[0], x[1] - x[0]
    fine_y = np.linspace(y[0] - dy / 2, y[-1] + dy / 2, len(y) * n + 1)
    fine_x = np.linspace(x[0] - dx / 2, x[-1] + dx / 2, len(x) * n + 1)
    z = func(fine_y, fine_x)
    corners = z[:-1, :-1] + z[1:, :-1] + z[:-1, 1:] + z[1:, 1:]
    cells = dy * dx * (0.4 * corners) / n / n
    return cells.reshape(len(y), n, len(x), n).sum(axis=(1, 3))


def generate_psf_image(func, shape, subsamples=10, normalize=True, **kwargs):
    """Pixel-integrated PSF image of `func` on a grid of `shape`, returned as a
    `scarlet_amd.fft.Fourier` (reference psf.py:55-89)."""
    from .fft import Fourier
    ry, rx = np.array(shape) // 2
    y = np.linspace(-ry, ry, shape[0])
    x = np.linspace(-rx, rx, shape[1])
    img = integrate_pixels(y, x, partial(func, **kwargs), subsamples)
    if normalize:
        img /= img.sum()
    return Fourier(img)
