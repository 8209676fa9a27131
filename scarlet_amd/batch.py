"""Batched Blend.fit() on one GPU: S independent scenes of identical shape.

This is the new, batched entry point the reference lacks (it fits one scene at a time in
Python).  `scarlet_amd.Blend` (single scene, reference API) is a thin view on a batch
of size 1.  State lives in PyTorch-ROCm tensors; all arithmetic is in the C-ABI HIP
library (`scarlet_amd/_lib.py`).  No CPU fallback.
"""
import ctypes

import numpy as np

from . import _lib


def default_centroid_weight():
    """Centroid weight used when the model frame has no PSF (reference source.py:483-490):
    41x41 integrated Gaussian, sigma=0.9, scaled to peak 1, float64."""
    from .psf import generate_psf_image, gaussian
    psf = generate_psf_image(gaussian, (41, 41), amplitude=1, sigma=.9, normalize=False).image
    return psf / psf.max()


class BlendBatch(object):
    """S scenes x K components x B bands x H x W pixels, all float32 on one device.

    Parameters
    ----------
    images : (S, B, H, W) array or tensor
    centers : (S, K, 2) integer pixel centres (y, x) of the sources
    weights : None (scalar 1, reference observation.py:148-151), a Python scalar, or (S, B, H, W)
    symmetric, monotonic : constraint switches of PointSource/ExtendedSource.update
    l0_thresh, l1_thresh : None or sparsity thresholds (update.sparse_l0 / sparse_l1)
    centroid_weight : (P, P) float64 centroid PSF; default = reference default
    group : None, or (S, K) integers: -1 = the component is a source of its own, g >= 0 = it is a layer of
        multi-component source g of its scene (reference MultiComponentSource, source.py:538-641): the layers
        of a source (adjacent components) share one centre measured on their flux-weighted sum
    """

    def __init__(self, images, centers, weights=None, symmetric=True, monotonic=True,
                 l0_thresh=None, l1_thresh=None, centroid_weight=None, mse_capacity=256,
                 device=None, group=None):
        torch = _lib.require_gpu()
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        f32 = dict(dtype=torch.float32, device=self.device)
        i32 = dict(dtype=torch.int32, device=self.device)
        f64 = dict(dtype=torch.float64, device=self.device)
        self.images = torch.as_tensor(images).to(**f32).contiguous()
        assert self.images.ndim == 4, "images must be (S, B, H, W)"
        S, B, H, W = self.images.shape
        self.centers = torch.as_tensor(np.asarray(centers) if not torch.is_tensor(centers) else centers).to(**i32).contiguous()
        assert self.centers.ndim == 3 and self.centers.shape[0] == S and self.centers.shape[2] == 2
        K = self.centers.shape[1]
        self.S, self.K, self.B, self.H, self.W = S, K, B, H, W
        self._check_centers(self.centers)
        self.weight_scalar = 1.0
        if weights is not None and np.ndim(weights) == 0 and not torch.is_tensor(weights):
            self.weight_scalar, weights = float(weights), None
        self.weights = None if weights is None else torch.as_tensor(weights).to(**f32).contiguous()
        self.sed = [torch.zeros((S, K, B), **f32) for _ in range(2)]
        self.morph = [torch.zeros((S, K, H, W), **f32) for _ in range(2)]
        self.cur = torch.zeros((S,), **i32)
        self.shifts = torch.full((S, K, 2), float("nan"), **f64)      # NaN == reference's "no shift yet"
        self.flags = torch.full((S, K), _lib.FLAG_SED_NOT_CONVERGED | _lib.FLAG_MORPH_NOT_CONVERGED, **i32)
        self.lipschitz = torch.ones((S, 2), **f64)
        self.mse_capacity = int(mse_capacity)
        self.mse_buf = torch.zeros((S, self.mse_capacity), **f64)
        self.it = torch.zeros((S,), **i32)
        self.active = torch.ones((S,), **i32)
        self.status = torch.zeros((S,), **i32)
        self.fix_sed = None
        self.fix_morph = None
        self.group = None
        if group is not None:
            g = np.asarray(group, dtype=np.int32).reshape(S, K)
            for row in g:                    # the layers of a source are adjacent components
                seen = set()
                for k, v in enumerate(row):
                    if v >= 0 and v in seen and row[k - 1] != v:
                        raise ValueError("the components of a multi-component source must be adjacent")
                    seen.add(int(v))
            self.group = torch.as_tensor(g).to(**i32).contiguous()
        self.symmetric, self.monotonic = bool(symmetric), bool(monotonic)
        self.l0_thresh, self.l1_thresh = l0_thresh, l1_thresh
        cw = default_centroid_weight() if centroid_weight is None else np.asarray(centroid_weight, dtype=np.float64)
        assert cw.ndim == 2 and cw.shape[0] == cw.shape[1] and cw.shape[0] % 2 == 1
        self.centroid_weight = torch.as_tensor(cw).to(**f64).contiguous()
        self._c = _lib.ScarletBatch()
        self._fill_struct()
        nbytes = _lib.lib.scarlet_batch_workspace_bytes(ctypes.byref(self._c))
        self.workspace = torch.zeros((int(nbytes),), dtype=torch.uint8, device=self.device)
        self._c.workspace = self.workspace.data_ptr()

    # ------------------------------------------------------------------ plumbing
    def _check_centers(self, centers):
        """Source centres index the frame directly (init, max_pixel window, sweeps): the reference raises
        IndexError for a source outside the image; here it is a ValueError before anything is launched."""
        c = centers
        bad = (c[..., 0] < 0) | (c[..., 0] >= self.H) | (c[..., 1] < 0) | (c[..., 1] >= self.W)
        if bool(bad.any().item()):
            s, k = [int(v[0]) for v in self.torch.nonzero(bad, as_tuple=True)]
            raise ValueError("centre %s of scene %d, source %d lies outside the %d x %d frame"
                             % (tuple(int(v) for v in c[s, k].tolist()), s, k, self.H, self.W))

    def _fill_struct(self):
        c, p = self._c, (lambda t: None if t is None else t.data_ptr())
        c.S, c.K, c.B, c.H, c.W = self.S, self.K, self.B, self.H, self.W
        c.images, c.weights, c.weight_scalar = p(self.images), p(self.weights), float(self.weight_scalar)
        c.sed[0], c.sed[1] = p(self.sed[0]), p(self.sed[1])
        c.morph[0], c.morph[1] = p(self.morph[0]), p(self.morph[1])
        c.cur = p(self.cur)
        c.centers, c.shifts, c.flags = p(self.centers), p(self.shifts), p(self.flags)
        c.fix_sed, c.fix_morph = p(self.fix_sed), p(self.fix_morph)
        c.lipschitz, c.mse, c.mse_capacity = p(self.lipschitz), p(self.mse_buf), self.mse_capacity
        c.it, c.active, c.status = p(self.it), p(self.active), p(self.status)
        c.symmetric, c.monotonic = int(self.symmetric), int(self.monotonic)
        c.l0_thresh = -1.0 if self.l0_thresh is None else float(self.l0_thresh)
        c.l1_thresh = -1.0 if self.l1_thresh is None else float(self.l1_thresh)
        c.centroid_psf, c.centroid_P = p(self.centroid_weight), int(self.centroid_weight.shape[0])
        c.group = p(self.group)
        if getattr(self, "workspace", None) is not None:
            c.workspace = self.workspace.data_ptr()

    def _ensure_mse_capacity(self, extra):
        need = int(self.it.max().item()) + int(extra)
        if need > self.mse_capacity:
            new_cap = max(need, 2 * self.mse_capacity)
            buf = self.torch.zeros((self.S, new_cap), dtype=self.torch.float64, device=self.device)
            buf[:, :self.mse_capacity] = self.mse_buf
            self.mse_buf, self.mse_capacity = buf, new_cap
            self._fill_struct()

    def _pick(self, pair):
        """Current-buffer view per scene (scenes flip their buffer index independently)."""
        cur = self.cur.to(self.torch.bool)
        shape = (-1,) + (1,) * (pair[0].ndim - 1)
        return self.torch.where(cur.view(shape), pair[1], pair[0])

    # ------------------------------------------------------------------ state access
    def set_state(self, sed, morph, centers=None, shifts=None):
        """Load factors into the current buffers (e.g. a state produced elsewhere)."""
        t = self.torch
        as_t = lambda a: a if t.is_tensor(a) else t.as_tensor(np.asarray(a))
        sed = as_t(sed).to(self.sed[0])
        morph = as_t(morph).to(self.morph[0])
        for b in range(2):
            self.sed[b].copy_(sed)
            self.morph[b].copy_(morph)
        if centers is not None:
            new = t.as_tensor(np.asarray(centers)).to(self.centers)
            self._check_centers(new)
            self.centers.copy_(new)
        if shifts is not None:
            self.shifts.copy_(t.as_tensor(np.asarray(shifts, dtype=np.float64)).to(self.shifts))

    def set_diff_kernel(self, kernel):
        """PSF difference kernel (what Observation.match computes, reference observation.py:191-194):
        (B, Py, Px) shared by all scenes, or (S, B, Py, Px) when every scene was observed with its own
        PSFs (`fft.match_psfs_device` makes them in one call).  Enables the FFT-convolution render
        (row a3b): grows the workspace by the FFT buffers and transforms the kernels once."""
        t = self.torch
        k = (kernel if t.is_tensor(kernel) else t.as_tensor(np.ascontiguousarray(kernel, dtype=np.float32)))
        self.diff_kernel = k.to(device=self.device, dtype=t.float32).contiguous()
        per_scene = self.diff_kernel.ndim == 4
        assert (self.diff_kernel.ndim == 3 and self.diff_kernel.shape[0] == self.B) or \
               (per_scene and tuple(self.diff_kernel.shape[:2]) == (self.S, self.B))
        self._c.diff_kernel = self.diff_kernel.data_ptr()
        self._c.diff_kernel_per_scene = int(per_scene)
        self._c.psf_h, self._c.psf_w = int(self.diff_kernel.shape[-2]), int(self.diff_kernel.shape[-1])
        nbytes = _lib.lib.scarlet_batch_workspace_bytes(ctypes.byref(self._c))
        self.workspace = t.zeros((int(nbytes),), dtype=t.uint8, device=self.device)
        self._c.workspace = self.workspace.data_ptr()
        _lib.check(_lib.lib.scarlet_batch_prepare_psf(ctypes.byref(self._c), _lib.stream_ptr()))
        return self

    @property
    def sed_current(self):
        return self._pick(self.sed)

    @property
    def morph_current(self):
        return self._pick(self.morph)

    def mse(self, s=0):
        """List of losses of scene `s`, one per iteration (reference Blend.mse)."""
        n = int(self.it[s].item())
        return self.mse_buf[s, :n].cpu().numpy().tolist()

    def raise_on_status(self):
        st = self.status.cpu().numpy()
        if (st & _lib.STATUS_CENTER_AT_EDGE).any():
            bad = np.nonzero(st & _lib.STATUS_CENTER_AT_EDGE)[0][:5]
            raise ValueError("max_pixel window left the image in scenes {} (the reference fails "
                             "there too: measurement.py:24-29)".format(bad.tolist()))

    # ------------------------------------------------------------------ operations
    def init_extended(self, bg_rms, thresh=1.0, sed_scale=None, init_symmetric=True, init_monotonic=None,
                      run_update=True):
        """ExtendedSource initialisation for every component (reference source.py:139-180,
        444-492) followed by the constructor's update() call."""
        bg = np.ascontiguousarray(bg_rms, dtype=np.float32)
        assert bg.shape == (self.B,)
        sc = None if sed_scale is None else np.ascontiguousarray(sed_scale, dtype=np.float32)
        rc = _lib.lib.scarlet_init_extended(
            ctypes.byref(self._c), bg.ctypes.data_as(ctypes.c_void_p), float(thresh),
            None if sc is None else sc.ctypes.data_as(ctypes.c_void_p), int(bool(init_symmetric)),
            int(self.monotonic if init_monotonic is None else bool(init_monotonic)), int(bool(run_update)),
            _lib.stream_ptr())
        _lib.check(rc)
        return self

    def update_sources(self):
        """Run the constraint pipeline once with it=0 (what the source constructors do)."""
        _lib.check(_lib.lib.scarlet_source_update(ctypes.byref(self._c), 0, _lib.stream_ptr()))
        return self

    def fit(self, max_iter=200, e_rel=1e-2, approximate_L=False, check_every=10):
        """Blend.fit for every scene (reference blend.py:65-102).  Scenes that reach e_rel
        stop iterating individually.  Returns the number of iterations launched."""
        self._ensure_mse_capacity(max_iter)
        self.active.fill_(1)          # a new fit() call iterates again, like the reference
        rc = _lib.lib.scarlet_fit(ctypes.byref(self._c), int(max_iter), float(e_rel),
                                  int(bool(approximate_L)), int(check_every), _lib.stream_ptr())
        return _lib.check(rc)

    def step(self, e_rel=1e-2, approximate_L=False):
        """One iteration in three separately callable phases (used by tests and by the
        Python-level update() override path)."""
        self._ensure_mse_capacity(1)
        s = _lib.stream_ptr()
        _lib.check(_lib.lib.scarlet_backward_step(ctypes.byref(self._c), int(bool(approximate_L)), s))
        _lib.check(_lib.lib.scarlet_source_update(ctypes.byref(self._c), 1, s))
        _lib.check(_lib.lib.scarlet_check_convergence(ctypes.byref(self._c), float(e_rel), s))
