"""The blended scene and its proximal-gradient fit (reference ``scarlet/blend.py`` API).

``Blend(sources, observations).fit(max_iter, e_rel, approximate_L)`` runs the reference's
iteration (blend.py:65-102) on the GPU.  A `Blend` is a batch of ONE scene on top of
`scarlet_amd.batch.BlendBatch`; many scenes at once go through `BlendBatch` directly.

Two execution modes, chosen automatically:

* built-in pipeline: every source is a `PointSource`/`ExtendedSource` whose ``update`` is not
  overridden -> the whole fit stays on the device (fused kernels, no per-iteration sync);
* Python pipeline: some source overrides ``update()`` (the reference's extension point,
  component.py:189-196) or is a plain `Component` -> per iteration the gradient step runs on
  the device, then each source's Python ``update()`` (whose `update.*`/`operator.*` calls are
  HIP kernels on the same device memory), then the device convergence check.
"""
import ctypes
import logging

import numpy as np

from . import _lib
from .component import ComponentTree, BlendFlag

logger = logging.getLogger("scarlet_amd.blend")


def render_with_kernel(model, kernel_image):
    """Observation.render's convolution (reference observation.py:198-201) on the device."""
    from .psfconv import convolve_same
    return convolve_same(model, kernel_image)


class Blend(ComponentTree):
    """The blended scene (reference blend.py:11-63).

    Attributes: ``mse`` (loss before each step), ``it``, ``converged``, ``observations``.
    """

    def __init__(self, sources, observations):
        ComponentTree.__init__(self, sources)
        try:
            iter(observations)
        except TypeError:
            observations = (observations,)
        self.observations = tuple(observations)
        self._batch = None
        self._view_buf = 0            # which ping-pong buffer the component views show
        self.L_sed = 1
        self.L_morph = 1

    # ------------------------------------------------------------------ device state
    def _source_attr(self, name, default):
        vals = [getattr(s, name, default) for s in self.sources]
        return vals

    def _builtin_pipeline(self):
        """True when every source uses the reference's stock update() (source.py:402-440)."""
        from .source import PointSource, MultiComponentSource
        if getattr(self, "python_pipeline", False):        # tests: force the per-source Python update() path
            return False
        for s in self.sources:
            if isinstance(s, MultiComponentSource):
                # layered sources: shared centre on the device (k_group_centers), stock update only
                if type(s).update is not MultiComponentSource.update or hasattr(s, "bboxes"):
                    return False
                if any(c.prior is not None for c in s.components):
                    return False
                continue
            if not isinstance(s, PointSource) or type(s).update is not PointSource.update:
                return False
            if getattr(s, "prior", None) is not None or hasattr(s, "bboxes"):
                return False
        sym = {bool(s.symmetric) for s in self.sources}
        mono = {bool(s.monotonic) for s in self.sources}
        return len(sym) == 1 and len(mono) == 1

    def _ensure_batch(self):
        if self._batch is not None:
            return self._batch
        torch = _lib.require_gpu()
        from .batch import BlendBatch
        multi = len(self.observations) != 1 or self.observations[0]._band_slice != slice(None)
        obs = self.observations[0]
        comps = self.components
        from .source import MultiComponentSource
        centers, group, owner = [], [], []
        for si, src in enumerate(self.sources):
            layered = isinstance(src, MultiComponentSource)
            for c in (src.components if hasattr(src, "components") else [src]):
                pc = getattr(src if layered else c, "pixel_center", None)
                if pc is None:
                    pc = (self.frame.Ny // 2, self.frame.Nx // 2)
                centers.append((int(pc[0]), int(pc[1])))
                group.append(si if layered else -1)
                owner.append(src if layered else None)
        assert len(centers) == len(comps)
        self._group_owner = owner
        cw = None
        for s in self.sources:
            if getattr(s, "_centroid_weight", None) is not None:
                cw = np.asarray(s._centroid_weight, dtype=np.float64)
                break
        builtin = self._builtin_pipeline()

        def obs_batch(o, images=None):
            ob = BlendBatch(o._images_device()[None] if images is None else images,
                            np.array(centers, dtype=np.int32)[None],
                            weights=None if (images is not None or o._weights_device() is None) else o._weights_device()[None],
                            symmetric=bool(self.sources[0].symmetric) if builtin else False,
                            monotonic=bool(self.sources[0].monotonic) if builtin else False,
                            centroid_weight=cw,
                            group=np.array(group, dtype=np.int32)[None] if (builtin and any(g >= 0 for g in group)) else None)
            if images is None:
                if type(o.weights) is not np.ndarray and o.weights != 1:
                    ob.weight_scalar = float(o.weights)
                    ob._fill_struct()
                if o._diff_kernels is not None:
                    ob.set_diff_kernel(np.asarray(o._diff_kernels.image, dtype=np.float32))
            return ob

        self._obs_batches = None
        if multi:
            # several observations and / or band slices (reference blend.py:120-139, 219-220): the state
            # (factors, centres, flags, convergence) lives in a batch over the model frame's channels,
            # every observation has a gradient-only batch over its own channels
            C, Ny, Nx = self.frame.shape
            b = obs_batch(obs, images=torch.zeros((1, C, Ny, Nx), dtype=torch.float32, device="cuda"))
            self._obs_batches = [(obs_batch(o), o._band_slice) for o in self.observations]
        else:
            b = obs_batch(obs)
        sed = torch.stack([c._own_sed for c in comps])[None]
        morph = torch.stack([c._own_morph for c in comps])[None]
        shifts = np.full((1, len(comps), 2), np.nan)
        for k, c in enumerate(comps):
            sh = getattr(c, "shift", None)
            if sh is not None:
                shifts[0, k] = (float(sh[0]), float(sh[1]))
        b.set_state(sed, morph, shifts=shifts)
        if any(c.fix_sed for c in comps) or any(c.fix_morph for c in comps):
            b.fix_sed = torch.tensor([[int(bool(c.fix_sed)) for c in comps]], dtype=torch.uint8, device="cuda")
            b.fix_morph = torch.tensor([[int(bool(c.fix_morph)) for c in comps]], dtype=torch.uint8, device="cuda")
            b._fill_struct()
        b.flags[0] = torch.tensor([c._flags.value for c in comps], dtype=torch.int32, device="cuda")
        self._batch = b
        self._view_buf = 0
        for k, c in enumerate(comps):
            c._binding = (self, k)
        return b

    def _factor_view(self, which, k):
        pair = self._batch.sed if which == "sed" else self._batch.morph
        return pair[self._view_buf][0, k]

    def _sync_sources(self):
        """Copy per-component scalars the device updated back to the Python objects."""
        b = self._batch
        self._view_buf = int(b.cur[0].item())
        cen = b.centers[0].cpu().numpy()
        sh = b.shifts[0].cpu().numpy()
        L = b.lipschitz[0].cpu().numpy()
        self.L_sed, self.L_morph = float(L[0]), float(L[1])
        for k, c in enumerate(self.components):
            c.L_sed, c.L_morph = self.L_sed, self.L_morph
            tgt = c if hasattr(c, "pixel_center") else getattr(self, "_group_owner", [None] * (k + 1))[k]
            if tgt is not None and self._builtin_pipeline():
                tgt.pixel_center = (int(cen[k, 0]), int(cen[k, 1]))
                if not np.isnan(sh[k, 0]):
                    tgt.shift = (float(sh[k, 0]), float(sh[k, 1]))
            elif hasattr(c, "pixel_center"):
                c.pixel_center = (int(cen[k, 0]), int(cen[k, 1]))
                if not np.isnan(sh[k, 0]):
                    c.shift = (float(sh[k, 0]), float(sh[k, 1]))

    # ------------------------------------------------------------------ reference API
    @property
    def mse(self):
        """Loss before each iteration's step (reference blend.py:138)."""
        return [] if self._batch is None else self._batch.mse(0)

    @property
    def it(self):
        """Number of iterations run so far (= len(mse); inside a source's update() it already
        counts the iteration in progress, as in the reference where _backward appended first)."""
        if hasattr(self, "_it_in_progress"):
            return self._it_in_progress
        return 0 if self._batch is None else int(self._batch.it[0].item())

    @property
    def converged(self):
        for c in self.components:
            if (c.flags & (BlendFlag.SED_NOT_CONVERGED | BlendFlag.MORPH_NOT_CONVERGED)).value > 0:
                return False
        return True

    def fit(self, max_iter=200, e_rel=1e-2, approximate_L=False):
        """Fit the model of every source to the data (reference blend.py:65-102)."""
        b = self._ensure_batch()
        if self._builtin_pipeline():
            if self._obs_batches is None:
                b.fit(max_iter, e_rel=e_rel, approximate_L=approximate_L, check_every=4)
            else:
                # several observations / band slices (reference blend.py:120-139, 219-220): gradients of every
                # observation, their sum, L * n_obs, step, constraints and convergence in one device loop
                b._ensure_mse_capacity(max_iter)
                b.active.fill_(1)
                n = len(self._obs_batches)
                ptrs = (ctypes.POINTER(_lib.ScarletBatch) * n)(*[ctypes.pointer(ob._c) for ob, _ in self._obs_batches])
                band0 = np.array([(sl.start or 0) for _, sl in self._obs_batches], dtype=np.int32)
                _lib.check(_lib.lib.scarlet_fit_multi(ctypes.byref(b._c), ptrs, band0.ctypes.data_as(ctypes.c_void_p), n,
                                                      int(max_iter), float(e_rel), int(bool(approximate_L)), 4,
                                                      _lib.stream_ptr()))
            b.raise_on_status()
            self._sync_sources()
            return self
        # Python pipeline: device gradient step, Python update() per source, device check
        s = _lib.stream_ptr
        b._ensure_mse_capacity(max_iter)
        b.active.fill_(1)
        multi = self._obs_batches is not None
        for _ in range(max_iter):
            cur = int(b.cur[0].item())
            # Prior hooks (reference blend.py:86-90, component.py:177-187) run on the factors
            # BEFORE the step: evaluate them now, add them to the likelihood gradients below
            priors = {}
            for k, c in enumerate(self.components):
                if c.prior is not None:
                    self._view_buf = cur
                    c.prior.compute_grad(c)
                    priors[k] = c
            if not priors and not multi:
                # plain case: the device takes the step itself
                _lib.check(_lib.lib.scarlet_backward_step(ctypes.byref(b._c), int(bool(approximate_L)), s()))
                self._view_buf = 1 - cur                 # sources see the stepped factors
                L = b.lipschitz[0].cpu().numpy()
                self.L_sed, self.L_morph = float(L[0]), float(L[1])
                for c in self.components:
                    c.L_sed, c.L_morph = self.L_sed, self.L_morph
            else:
                # gradients from the device (scarlet_backward_gradients), combination on the host side
                # of the ABI: sum over observations, L * len(observations) (blend.py:219-220), priors
                # (component.py:177-187), fix_sed / fix_morph (blend.py:91-96)
                import torch
                x_sed, x_morph = b.sed[cur][0], b.morph[cur][0]
                it_idx = int(b.it[0].item())
                _lib.check(_lib.lib.scarlet_backward_gradients(ctypes.byref(b._c), int(bool(approximate_L)), s()))
                L = b.lipschitz[0].cpu().numpy()
                if multi:
                    g_sed = torch.zeros_like(x_sed); g_morph = torch.zeros_like(x_morph)
                    loss = 0.0
                    for ob, sl in self._obs_batches:
                        ob.sed[0][0].copy_(x_sed[:, sl]); ob.morph[0][0].copy_(x_morph)
                        ob.cur.zero_(); ob.it.zero_(); ob.active.fill_(1)
                        _lib.check(_lib.lib.scarlet_backward_gradients(ctypes.byref(ob._c), 0, s()))
                        g_sed[:, sl] += ob.sed[1][0]; g_morph += ob.morph[1][0]
                        loss += float(ob.mse_buf[0, 0].item())
                    b.mse_buf[0, it_idx] = loss
                    if approximate_L:
                        # blend.py:189-201 on the summed loss: crude bound, doubled when the loss rose
                        L = np.array([float((x_morph.double() ** 2).sum().item()),
                                      float((x_sed.double() ** 2).sum().item())])
                        if it_idx >= 1 and loss > float(b.mse_buf[0, it_idx - 1].item()):
                            L *= 2
                else:
                    g_sed, g_morph = b.sed[1 - cur][0].clone(), b.morph[1 - cur][0].clone()
                n_obs = len(self.observations)
                self.L_sed, self.L_morph = float(L[0]) * n_obs, float(L[1]) * n_obs
                dev = x_sed.device
                as_t = lambda v: v.to(dev) if hasattr(v, "to") else torch.as_tensor(np.asarray(v, dtype=np.float32), device=dev)
                new_sed, new_morph = b.sed[1 - cur][0], b.morph[1 - cur][0]
                for k, c in enumerate(self.components):
                    c.L_sed, c.L_morph = self.L_sed, self.L_morph
                    gs, gm = g_sed[k], g_morph[k]
                    if k in priors:
                        if not c.fix_morph:
                            gm = gm + as_t(c.prior.morph_grad); c.L_morph = self.L_morph + float(c.prior.L_morph)
                        if not c.fix_sed:
                            gs = gs + as_t(c.prior.sed_grad); c.L_sed = self.L_sed + float(c.prior.L_sed)
                    new_sed[k].copy_(x_sed[k] if c.fix_sed else x_sed[k] - (1 / c.L_sed) * gs)
                    new_morph[k].copy_(x_morph[k] if c.fix_morph else x_morph[k] - (1 / c.L_morph) * gm)
                self._view_buf = 1 - cur                 # sources see the stepped factors
            self._it_in_progress = int(b.it[0].item()) + 1
            self.update()
            del self._it_in_progress
            _lib.check(_lib.lib.scarlet_convergence_sums(ctypes.byref(b._c), s()))
            _lib.check(_lib.lib.scarlet_check_convergence(ctypes.byref(b._c), float(e_rel), s()))
            self._view_buf = int(b.cur[0].item())
            if int(b.active[0].item()) == 0:
                break
        return self
