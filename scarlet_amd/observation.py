"""Frame and Observation (reference ``scarlet/observation.py:13-239`` API).

`Frame` describes the model (shape, PSF, channels, dtype); `Observation` holds the data
(images, weights, PSFs) and, after ``match(model_frame)``, the band slice and the PSF
difference kernel that map the model into the observed frame.  Data live on the device;
the engine computes in float32: a float64 DATA frame is cast to the model frame's dtype by ``match``
(as in the reference), a float64 MODEL frame is accepted with one warning -- the factors are stored in float32
(component._require_float32_frame; a strict opt-in refuses it).
"""
import logging

import numpy as np

from . import _lib
from . import fft

logger = logging.getLogger("scarlet_amd.observation")


class Frame(object):
    """Spatial and spectral characteristics of a model or of data
    (reference observation.py:13-99)."""

    def __init__(self, shape, wcs=None, psfs=None, channels=None, dtype=np.float32):
        assert len(shape) == 3
        self._shape = tuple(shape)
        self.wcs = wcs
        if psfs is None:
            logger.warning('No PSFs specified. Possible, but dangerous!')
        else:
            msg = 'PSFs need to have shape (1,Ny,Nx) for Blend and (B,Ny,Nx) for Observation'
            assert len(psfs) == 1 or len(psfs) == shape[0], msg
            if not isinstance(psfs, fft.Fourier):
                psfs = fft.Fourier(np.array(psfs))
            if not np.allclose(psfs.sum(axis=(1, 2)), 1):
                logger.warning('PSFs not normalized. Normalizing now..')
                psfs.normalize()
            if dtype != psfs.image.dtype:
                logger.warning("Dtypes of PSFs and Frame different. Casting PSFs to {}".format(dtype))
                psfs.update_dtype(dtype)
        self._psfs = psfs
        assert channels is None or len(channels) == shape[0]
        self.channels = channels
        self.dtype = dtype

    @property
    def C(self):
        return self._shape[0]

    @property
    def Ny(self):
        return self._shape[1]

    @property
    def Nx(self):
        return self._shape[2]

    @property
    def shape(self):
        return self._shape

    @property
    def psfs(self):
        return self._psfs

    def get_pixel(self, sky_coord):
        """Pixel (y, x) of a sky coordinate: integer truncation without a WCS
        (reference observation.py:84-99)."""
        if self.wcs is not None:
            if self.wcs.naxis == 3:
                coord = self.wcs.wcs_world2pix(sky_coord[0], sky_coord[1], 0, 0)
            elif self.wcs.naxis == 2:
                coord = self.wcs.wcs_world2pix(sky_coord[0], sky_coord[1], 0)
            else:
                raise ValueError("Invalid number of wcs dimensions: {0}".format(self.wcs.naxis))
            return (int(coord[0].item()), int(coord[1].item()))
        return tuple(int(coord) for coord in sky_coord)


class Observation(object):
    """Images, weights and PSFs of one data set (reference observation.py:102-239)."""

    def __init__(self, images, psfs=None, weights=None, wcs=None, channels=None, padding=10):
        images = np.asarray(images) if not hasattr(images, "detach") else images.detach().cpu().numpy()
        self.frame = Frame(images.shape, wcs=wcs, psfs=psfs, channels=channels, dtype=images.dtype)
        self.images = np.array(images)
        self.weights = np.array(weights) if weights is not None else 1
        self._padding = padding
        self._band_slice = slice(None)
        self._diff_kernels = None
        self._device = {}

    def match(self, model_frame):
        """Set up the mapping from the model frame to this observation: dtype, band slice and
        PSF difference kernel (reference observation.py:155-196)."""
        if self.frame.dtype != model_frame.dtype:
            msg = "Dtypes of model and observation different. Casting observation to {}"
            logger.warning(msg.format(model_frame.dtype))
            self.frame.dtype = model_frame.dtype
            self.images = self.images.astype(model_frame.dtype)
            if type(self.weights) is np.ndarray:
                self.weights = self.weights.astype(model_frame.dtype)
            if self.frame._psfs is not None:
                self.frame.psfs.update_dtype(model_frame.dtype)
        self._band_slice = slice(None)
        if self.frame.channels is not model_frame.channels:
            assert self.frame.channels is not None and model_frame.channels is not None
            bmin = list(model_frame.channels).index(self.frame.channels[0])
            bmax = list(model_frame.channels).index(self.frame.channels[-1])
            self._band_slice = slice(bmin, bmax + 1)
        self._diff_kernels = None
        if self.frame.psfs is not model_frame.psfs:
            assert self.frame.psfs is not None and model_frame.psfs is not None
            self._diff_kernels = fft.match_psfs(self.frame.psfs, model_frame.psfs)
        self._device = {}
        return self

    # ---- device copies used by the engine
    def _images_device(self):
        torch = _lib.require_gpu()
        if "images" not in self._device:
            self._device["images"] = torch.as_tensor(np.ascontiguousarray(self.images, dtype=np.float32)).cuda()
        return self._device["images"]

    def _weights_device(self):
        torch = _lib.require_gpu()
        if type(self.weights) is not np.ndarray:
            return None
        if "weights" not in self._device:
            w = np.broadcast_to(self.weights, self.images.shape)
            self._device["weights"] = torch.as_tensor(np.ascontiguousarray(w, dtype=np.float32)).cuda()
        return self._device["weights"]

    def render(self, model):
        """Map a model (bands, height, width) into the observed frame: band slice, then
        convolution with the difference kernel if there is one (reference observation.py:203-220)."""
        model_ = model[self._band_slice, :, :]
        if self._diff_kernels is not None:
            from .blend import render_with_kernel
            model_ = render_with_kernel(model_, self._diff_kernels.image)
        return model_

    def get_loss(self, model):
        """0.5 * sum (weights * (render(model) - images))^2 (reference observation.py:222-239)."""
        torch = _lib.require_gpu()
        m = self.render(model)
        m = m if torch.is_tensor(m) else torch.as_tensor(np.asarray(m)).cuda()
        w = self._weights_device()
        d = m.to(torch.float32) - self._images_device()
        if w is not None:
            d = w * d
        elif self.weights != 1:
            d = float(self.weights) * d
        return 0.5 * (d.double() ** 2).sum()
