"""Sources: components with an initialisation recipe and a constraint pipeline
(reference ``scarlet/source.py`` API: ``PointSource``, ``ExtendedSource``, the ``get_*_sed``
/ ``build_detection_coadd`` / ``init_extended_source`` helpers, ``SourceInitError``).

Initialisation of an `ExtendedSource` runs in the HIP library (`scarlet_init_extended`); the
small public helpers that return host arrays (pixel SEDs, detection coadd) are setup-time
numpy, outside the per-iteration path.
"""
import logging

import numpy as np

from . import _lib
from . import measurement
from . import update
from .component import Component, ComponentTree
from .psf import generate_psf_image, gaussian

logger = logging.getLogger("scarlet_amd.source")


class SourceInitError(Exception):
    """Error during source initialization"""
    pass


def _host(a):
    return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)


def get_pixel_sed(sky_coord, observation):
    """SED at `sky_coord` read off the observed images (reference source.py:21-38)."""
    pixel = observation.frame.get_pixel(sky_coord)
    return _host(observation.images)[:, pixel[0], pixel[1]].copy()


def _psf_sed_scale(observation, frame):
    """Per-band factor of get_psf_sed: 1/max(obs psf_b) * max(model psf) (source.py:62-69)."""
    B = observation.frame.C
    scale = np.ones(B, dtype=observation.images.dtype)
    if observation.frame.psfs is not None:
        scale = scale / observation.frame.psfs.max(axis=(1, 2)).astype(scale.dtype)
    if frame.psfs is not None:
        scale = scale * frame.psfs[0].max()
    return scale


def get_psf_sed(sky_coord, observation, frame):
    """Pixel SED corrected for the PSF peak heights (reference source.py:41-71)."""
    sed = get_pixel_sed(sky_coord, observation)
    if observation.frame.psfs is not None:
        sed /= observation.frame.psfs.max(axis=(1, 2))
    if frame.psfs is not None:
        sed = sed * frame.psfs[0].max()
    return sed


def get_best_fit_seds(morphs, frame, observation):
    """Least-squares SEDs for fixed morphologies (reference source.py:74-98)."""
    morphs = _host(morphs)
    K = len(morphs)
    S = morphs.reshape(K, -1)
    data = _host(observation.images).reshape(observation.frame.C, -1)
    return np.dot(np.linalg.inv(np.dot(S, S.T)), np.dot(S, data.T))


def build_detection_coadd(sed, bg_rms, observation, thresh=1):
    """SED-weighted coadd of the bands with positive SED and its noise cutoff
    (reference source.py:101-136)."""
    sed = _host(sed)
    bg_rms = np.asarray(bg_rms)
    if np.any(bg_rms <= 0):
        raise ValueError("bg_rms must be greater than zero in all channels")
    images = _host(observation.images)
    pos = [c for c in range(len(sed)) if sed[c] > 0]
    w = np.array([sed[c] / bg_rms[c] ** 2 for c in pos])
    jacobian = np.array([sed[c] ** 2 / bg_rms[c] ** 2 for c in pos]).sum()
    detect = np.einsum('i,i...', w, [images[c] for c in pos]) / jacobian
    bg_cutoff = thresh * np.sqrt((w ** 2 * np.array([bg_rms[c] for c in pos]) ** 2).sum()) / jacobian
    return detect, bg_cutoff


def _device_init(sky_coord, frame, observation, bg_rms, thresh, symmetric, monotonic):
    """Run scarlet_init_extended for one source; returns (batch, pixel)."""
    from .batch import BlendBatch
    pixel = frame.get_pixel(sky_coord)
    b = BlendBatch(observation._images_device()[None], np.array([[pixel]], dtype=np.int32),
                   symmetric=False, monotonic=bool(monotonic))
    scale = None
    if observation.frame.psfs is not None or frame.psfs is not None:
        scale = _psf_sed_scale(observation, frame)
    b.init_extended(np.asarray(bg_rms, dtype=np.float32), thresh=thresh, sed_scale=scale,
                    init_symmetric=symmetric, init_monotonic=monotonic, run_update=False)
    if int(b.flags[0, 0].item()) & _lib.FLAG_NO_VALID_PIXELS:
        _, cutoff = build_detection_coadd(b.sed[0][0, 0].cpu().numpy(), bg_rms, observation, thresh)
        msg = "No flux above threshold={2} for source at y={0} x={1}"
        raise SourceInitError(msg.format(sky_coord[0], sky_coord[1], cutoff))
    return b, pixel


def init_extended_source(sky_coord, frame, observation, bg_rms, thresh=1., symmetric=True, monotonic=True):
    """(sed, morph) of a source that is symmetric and monotonic around `sky_coord`
    (reference source.py:139-180); device tensors."""
    b, _ = _device_init(sky_coord, frame, observation, bg_rms, thresh, symmetric, monotonic)
    sed = b.sed[0][0, 0].clone()
    if bool((sed <= 0).any()):
        msg = "Zero or negative SED {} at y={}, x={}".format(sed.cpu().numpy(), *sky_coord)
        (logger.warning if bool((sed <= 0).all()) else logger.info)(msg)
    return sed, b.morph[0][0, 0].clone()


def _default_centroid_weight(frame):
    """Centroid weight (reference source.py:483-490): the model PSF, or a 41x41 sigma=.9
    Gaussian scaled to peak 1 when the frame has none."""
    if frame.psfs is None:
        psf = generate_psf_image(gaussian, (41, 41), amplitude=1, sigma=.9, normalize=False).image
        return psf / psf.max()
    return np.asarray(frame.psfs[0].image)


class PointSource(Component):
    """Source initialised as a single pixel (or as the model PSF centred on it); default
    constraints: symmetry and monotonicity (reference source.py:340-440)."""

    def __init__(self, frame, sky_coord, observation, symmetric=True, monotonic=True,
                 center_step=5, delay_thresh=10, **component_kwargs):
        C, Ny, Nx = frame.shape
        images = _host(observation.images)
        morph = np.zeros((Ny, Nx), images.dtype)
        pixel = frame.get_pixel(sky_coord)
        if frame.psfs is None:
            morph[pixel] = 1
        else:
            # paste the model PSF so that its centre lands on `pixel` (source.py:380-387)
            psf = np.asarray(frame.psfs[0].image)
            assert psf.ndim == 2
            py, px = pixel
            sy, sx = (np.array(psf.shape) - 1) // 2
            y0, x0 = py - sy, px - sx
            ys = slice(max(0, y0), min(Ny, y0 + psf.shape[0]))
            xs = slice(max(0, x0), min(Nx, x0 + psf.shape[1]))
            morph[ys, xs] = psf[ys.start - y0:ys.stop - y0, xs.start - x0:xs.stop - x0]
        self.pixel_center = pixel
        opix = observation.frame.get_pixel(sky_coord)
        sed = images[:, opix[0], opix[1]].copy()
        if observation.frame.psfs is not None:
            sed /= observation.frame.psfs.max(axis=(1, 2))
        super().__init__(frame, sed, morph, **component_kwargs)
        self.symmetric = symmetric
        self.monotonic = monotonic
        self.center_step = center_step
        self.delay_thresh = delay_thresh
        if self.symmetric:
            self._centroid_weight = _default_centroid_weight(self.frame)
        self.update()

    def update(self):
        """The reference's default constraint pipeline (source.py:402-440): recentre ->
        (every 5th iteration) centroid -> k-space symmetry -> weighted monotonicity ->
        positivity -> peak normalisation.  Inside `Blend.fit` this exact pipeline runs fused
        on the device unless a subclass overrides this method."""
        it = 0 if self._parent is None else self._parent.it
        self.pixel_center = measurement.max_pixel(self.morph, self.pixel_center)
        bbox = self.bboxes["thresh"] if hasattr(self, "bboxes") and "thresh" in self.bboxes else None
        if self.symmetric:
            if it % 5 == 0:
                self.pixel_center, self.shift = measurement.psf_weighted_centroid(
                    self.morph, self._centroid_weight, self.pixel_center)
            update.symmetric(self, self.pixel_center, algorithm="kspace", bbox=bbox)
        if self.monotonic:
            update.monotonic(self, self.pixel_center, bbox=bbox)
        update.positive(self)
        update.normalized(self)
        return self


class ExtendedSource(PointSource):
    """Extended source initialised from the detection coadd around `sky_coord`
    (reference source.py:443-492)."""

    def __init__(self, frame, sky_coord, observation, bg_rms, thresh=1, symmetric=True, monotonic=True,
                 center_step=5, delay_thresh=10, **component_kwargs):
        self.symmetric = symmetric
        self.monotonic = monotonic
        self.coords = sky_coord
        self.pixel_center = frame.get_pixel(sky_coord)
        self.center_step = center_step
        self.delay_thresh = delay_thresh
        sed, morph = init_extended_source(sky_coord, frame, observation, bg_rms, thresh, True, monotonic)
        Component.__init__(self, frame, sed, morph, **component_kwargs)
        if self.symmetric:
            self._centroid_weight = _default_centroid_weight(self.frame)
        self.update()


def init_combined_extended_source(sky_coord, frame, observations, bg_rms, obs_idx=0, thresh=1.,
                                  symmetric=True, monotonic=True):
    """(sed, morph) for a source seen by several observations (reference source.py:183-240): the SED is
    the concatenation of the per-observation PSF-scaled SEDs, the morphology comes from the detection
    coadd of observation `obs_idx` (device initialisation, as for ExtendedSource)."""
    import torch
    try:
        iter(observations)
    except TypeError:
        observations = [observations]
    seds = [torch.as_tensor(_host(get_psf_sed(sky_coord, obs, frame))).reshape(-1) for obs in observations]
    sed = torch.cat(seds).to(device="cuda", dtype=torch.float32)
    if bool((sed <= 0).any()):
        msg = "Zero or negative SED {} at y={}, x={}".format(sed.cpu().numpy(), *sky_coord)
        (logger.warning if bool((sed <= 0).all()) else logger.info)(msg)
    b, _ = _device_init(sky_coord, frame, observations[obs_idx], bg_rms[obs_idx], thresh, symmetric, monotonic)
    return sed, b.morph[0][0, 0].clone()


class CombinedExtendedSource(PointSource):
    """Extended source initialised to match a set of observations (reference source.py:495-536).  As in the
    reference the constructor does not run update(), `symmetric` defaults to False and the initial
    morphology is always made symmetric."""

    def __init__(self, frame, sky_coord, observations, bg_rms, obs_idx=0, thresh=1, symmetric=False,
                 monotonic=True, center_step=5, delay_thresh=0, **component_kwargs):
        self.symmetric = symmetric
        self.monotonic = monotonic
        self.coords = sky_coord
        self.pixel_center = frame.get_pixel(sky_coord)
        self.center_step = center_step
        self.delay_thresh = delay_thresh
        sed, morph = init_combined_extended_source(sky_coord, frame, observations, bg_rms, obs_idx, thresh,
                                                   True, monotonic)
        Component.__init__(self, frame, sed, morph, **component_kwargs)


def init_multicomponent_source(sky_coord, frame, observation, bg_rms, flux_percentiles=None,
                               thresh=1., symmetric=True, monotonic=True):
    """(seds, morphs) of a source split into layered components at the given flux percentiles
    (reference source.py:242-295).  One-time setup: the base morphology comes from the device
    initialisation, the layering and the least-squares SEDs are host numpy."""
    if flux_percentiles is None:
        flux_percentiles = [25]
    sed, morph = init_extended_source(sky_coord, frame, observation, bg_rms, thresh, symmetric, monotonic)
    morph = _host(morph)
    K = len(flux_percentiles) + 1
    Ny, Nx = morph.shape
    morphs = np.zeros((K, Ny, Nx), dtype=morph.dtype)
    morphs[0, :, :] = morph[:, :]
    max_flux = morph.max()
    percentiles_ = np.sort(flux_percentiles)
    last_thresh = 0
    for k in range(1, K):
        perc = percentiles_[k - 1]
        flux_thresh = perc * max_flux / 100
        mask_ = morph > flux_thresh
        morphs[k - 1][mask_] = flux_thresh - last_thresh
        morphs[k][mask_] = morph[mask_] - flux_thresh
        last_thresh = flux_thresh
    for k in range(K):
        if np.all(morphs[k] <= 0):
            logger.warning("Zero or negative morphology for component {} at y={}, x={}".format(k, *sky_coord))
        morphs[k] /= morphs[k].max()
    seds = get_best_fit_seds(morphs, frame, observation)
    for k in range(K):
        if np.any(seds[k] <= 0):
            msg = "Zero or negative SED {} for component {} at y={}, x={}".format(seds[k], k, *sky_coord)
            (logger.warning if bool(np.all(_host(sed) <= 0)) else logger.info)(msg)
    return seds, morphs


class RandomSource(Component):
    """Uniform random morphology, SED random or fitted to an observation (reference source.py:298-327)."""

    def __init__(self, frame, observation=None, **component_kwargs):
        C, Ny, Nx = frame.shape
        morph = np.random.rand(Ny, Nx)
        if observation is None:
            sed = np.random.rand(C)
        else:
            sed = get_best_fit_seds(morph[None], frame, observation)[0]
        super().__init__(frame, sed, morph, **component_kwargs)


class MultiComponentSource(ComponentTree):
    """Extended source made of layered components that share one centre (reference
    source.py:495-641).  Its update() measures the centre on the flux-weighted sum of the
    components and constrains every component around it; inside `Blend.fit` it runs through the
    Python pipeline (device gradient step, these device operators, device convergence check)."""

    def __init__(self, frame, sky_coord, observation, bg_rms, flux_percentiles=None, thresh=1.,
                 symmetric=True, monotonic=True, center_step=5, delay_thresh=0, **component_kwargs):
        self.symmetric = symmetric
        self.monotonic = monotonic
        self.pixel_center = frame.get_pixel(sky_coord)
        self.center_step = center_step
        self.delay_thresh = delay_thresh
        seds, morphs = init_multicomponent_source(sky_coord, frame, observation, bg_rms, flux_percentiles,
                                                  thresh, symmetric, monotonic)
        components = [Component(frame, seds[k], morphs[k], **component_kwargs) for k in range(len(seds))]
        super().__init__(components)
        if self.symmetric:
            self._centroid_weight = _default_centroid_weight(self.frame)
        self.update()

    def update(self):
        it = 0 if self._parent is None else self._parent.it
        # centre from the flux-weighted mean of all components (source.py:613-617)
        _morph = sum(c.morph * c.sed.sum() for c in self.components)
        self.pixel_center = measurement.max_pixel(_morph, self.pixel_center)
        bbox = self.bboxes["thresh"] if hasattr(self, "bboxes") and "thresh" in self.bboxes else None
        if self.symmetric and it % 5 == 0:
            self.pixel_center, self.shift = measurement.psf_weighted_centroid(
                _morph, self._centroid_weight, self.pixel_center)
        for c in self.components:
            if self.symmetric:
                update.symmetric(c, self.pixel_center, algorithm="kspace", bbox=bbox)
            if self.monotonic:
                update.monotonic(c, self.pixel_center, bbox=bbox)
            update.positive(c)
            update.normalized(c, type='morph_max')
        return self
