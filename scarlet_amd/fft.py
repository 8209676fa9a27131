"""Image + cached spectra container and the pad/centre conventions of the FFT path.

Mirrors the reference's ``scarlet/fft.py`` names (``Fourier``, ``_pad``, ``_centered``,
``_get_fft_shape``).  The integer layout logic lives here (host side); the transforms of
the per-iteration path run on the device.
"""
import numpy as np

from . import _lib


def next_fast_len(n):
    """5-smooth FFT length >= n, from the C-ABI library (reference fft.py:99)."""
    return int(_lib.lib.scarlet_next_fast_len(int(n)))


def _centered(arr, newshape):
    """Central `newshape` crop of `arr`; start index (cur-new+1)//2 so that an odd array
    cropped to an even shape keeps its centre on the centre-right pixel
    (reference fft.py:7-35)."""
    new = np.asarray(newshape)
    cur = np.array(arr.shape)
    if not np.all(new <= cur):
        msg = "arr must be larger than newshape in both dimensions, received {0}, and {1}"
        raise ValueError(msg.format(arr.shape, newshape))
    lo = (cur - new + 1) // 2
    return arr[tuple(slice(a, a + n) for a, n in zip(lo, new))]


def _pad(arr, newshape, axes=None):
    """Zero-pad `arr` to `newshape` (only along `axes` if given); leading pad (dS+1)//2
    (reference fft.py:38-65)."""
    if axes is None:
        axes = range(arr.ndim)
    else:
        try:
            len(axes)
        except TypeError:
            axes = [axes]
    widths = [(0, 0)] * arr.ndim
    for n, ax in enumerate(axes):
        extra = newshape[n] - arr.shape[ax]
        lead = (extra + 1) // 2
        widths[ax] = (lead, extra - lead)
    return np.pad(arr, widths, mode="constant")


def _get_fft_shape(img1, img2, padding=3, axes=None, max=False):
    """Fast FFT shape for combining img1 and img2 along `axes`; the last axis is forced
    even (reference fft.py:68-106)."""
    s1, s2 = np.asarray(img1.shape), np.asarray(img2.shape)
    if len(s1) != len(s2):
        msg = "img1 and img2 must have the same number of dimensions, but got {0} and {1}"
        raise ValueError(msg.format(len(s1), len(s2)))
    if axes is None:
        axes = range(len(s1))
    else:
        try:
            len(axes)
        except TypeError:
            axes = [axes]
    combine = (lambda a, b: builtins_max(a, b)) if max else (lambda a, b: a + b)
    shape = [next_fast_len(combine(int(s1[a]), int(s2[a])) + padding) for a in axes]
    while shape[-1] % 2 != 0:
        shape[-1] = next_fast_len(shape[-1] + 1)
    return shape


def builtins_max(a, b):
    return a if a > b else b


class Fourier(object):
    """A real-space image plus a dictionary of its spectra keyed by (fft_shape, axes)
    (reference fft.py:109-261).  Host-side holder for PSFs."""

    def __init__(self, image, image_fft=None):
        self._image = image
        self._fft = {} if image_fft is None else image_fft

    @property
    def image(self):
        return self._image

    @property
    def shape(self):
        return self._image.shape

    def __len__(self):
        return len(self._image)

    def sum(self, axis=None):
        return self._image.sum(axis)

    def max(self, axis=None):
        return self._image.max(axis=axis)

    def normalize(self, axes=None):
        """Scale the image to unit sum (over all axes when `axes` is None, as the
        reference does: fft.py:216-229)."""
        total = self._image.sum(axis=axes)
        if axes is None:
            self._image = self._image * (1 / total)
        else:
            idx = [slice(None)] * self._image.ndim
            for a in axes:
                idx[a] = None
            self._image = self._image * (1 / total)[tuple(idx)]
        self._fft = {}

    def update_dtype(self, dtype):
        if self._image.dtype != dtype:
            self._image = self._image.astype(dtype)
            self._fft = {}

    def __getitem__(self, index):
        return Fourier(self._image[index])


# ---------------------------------------------------------------------------------------
# Setup-time k-space helpers (host, numpy).  They run once per scene in Observation.match,
# not in the iteration (SURVEY.md 8f rank 2 moves them to the device later); the
# per-iteration convolution of the model (row a3b) is done by the HIP engine.
def _kspace(image, fshape, axes):
    """pad -> ifftshift -> rfftn (reference Fourier.fft, fft.py:193-211)."""
    return np.fft.rfftn(np.fft.ifftshift(_pad(image, fshape, axes), axes), axes=axes)


def _from_kspace(spec, fshape, image_shape, axes):
    """irfftn -> fftshift -> central crop (reference Fourier.from_fft, fft.py:138-181)."""
    return _centered(np.fft.fftshift(np.fft.irfftn(spec, fshape, axes=axes), axes=axes), image_shape)


def match_psfs(psf1, psf2, padding=3, axes=(-2, -1)):
    """Difference kernel that turns `psf2` into `psf1`: ratio of the spectra, cropped to the
    larger image shape (reference fft.match_psfs, fft.py:282-301).  Fourier in, Fourier out."""
    shape = psf2.shape if psf1.shape[0] < psf2.shape[0] else psf1.shape
    F = _get_fft_shape(psf1.image, psf2.image, padding, axes)
    spec = _kspace(psf1.image, F, axes) / _kspace(psf2.image, F, axes)
    return Fourier(_from_kspace(spec, F, shape, axes))


def match_psfs_device(psf1, psf2):
    """`match_psfs` on the device for a batch of PSFs: `psf1` (n, P1y, P1x)
    or (S, B, P1y, P1x), `psf2` (1 | n, P2y, P2x); numpy arrays or device tensors in, device tensor of
    psf1's shape out.  Same FFT shape and centring as the host function above."""
    import ctypes
    from . import _lib
    torch = _lib.require_gpu()
    as_t = lambda a: (a if torch.is_tensor(a) else torch.as_tensor(np.ascontiguousarray(a))).to(
        device="cuda", dtype=torch.float32).contiguous()
    t1, t2 = as_t(psf1), as_t(psf2)
    shape1 = tuple(t1.shape)
    t1 = t1.reshape(-1, shape1[-2], shape1[-1])
    t2 = t2.reshape(-1, t2.shape[-2], t2.shape[-1])
    if t2.shape[0] not in (1, t1.shape[0]):
        raise ValueError("psf2 must hold one PSF or one per PSF of psf1")
    out = torch.empty_like(t1)
    _lib.check(_lib.lib.scarlet_match_psfs(_lib.ptr(t1), t1.shape[0], t1.shape[1], t1.shape[2],
                                           _lib.ptr(t2), t2.shape[0], t2.shape[1], t2.shape[2],
                                           _lib.ptr(out), _lib.stream_ptr()))
    return out.reshape(shape1)


def convolve(image1, image2, padding=3, axes=(-2, -1)):
    """Linear convolution of two `Fourier` images cropped to image1's shape
    (reference fft.convolve, fft.py:304-317).  Host helper for setup and tests."""
    F = _get_fft_shape(image1.image, image2.image, padding, axes)
    spec = _kspace(image1.image, F, axes) * _kspace(image2.image, F, axes)
    return Fourier(_from_kspace(spec, F, image1.shape, axes))
