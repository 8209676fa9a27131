"""Components and component trees (reference ``scarlet/component.py`` API).

A `Component` is one (SED, morphology) factor pair.  Its arrays are float32 PyTorch-ROCm
tensors; once the component belongs to a `Blend` they are views into the blend's batched
device state, so the HIP engine and user code see the same memory.
"""
import logging
from enum import Flag, auto

import numpy as np

from . import _lib

logger = logging.getLogger("scarlet_amd.component")


class BlendFlag(Flag):
    """Status bits of a component (reference component.py:13-36); the numeric values are the
    SCARLET_FLAG_* constants of include/scarlet_hip.h."""
    NONE = 0
    SED_NOT_CONVERGED = auto()
    MORPH_NOT_CONVERGED = auto()
    EDGE_PIXELS = auto()
    NO_VALID_PIXELS = auto()


class Prior(object):
    """Differentiable prior: ``grad_func(sed, morph) -> (sed_grad, morph_grad)`` and
    ``L_func(sed, morph) -> (L_sed, L_morph)`` (reference component.py:39-67).  Priors run
    as Python callbacks between the device gradient and the device step."""

    def __init__(self, grad_func, L_func):
        self._grad_func = grad_func
        self._L_func = L_func
        self.sed_grad = 0
        self.morph_grad = 0

    def compute_grad(self, component):
        self.sed_grad, self.morph_grad = self._grad_func(component._sed, component._morph)
        self.L_sed, self.L_morph = self._L_func(component._sed, component._morph)


def _to_device(a, shape=None):
    torch = _lib.require_gpu()
    if torch.is_tensor(a):
        t = a.detach().to(device="cuda", dtype=torch.float32)
    else:
        t = torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).cuda()
    return t.contiguous().clone()


STRICT_FLOAT32_FRAME = False     # opt-in: refuse a float64 model Frame instead of accepting it with a warning
_warned_float64_frame = False


def _require_float32_frame(frame):
    """The HIP engine stores and steps the factors in float32 (reductions -- loss, Gram matrices, moments,
    convergence sums -- accumulate in float64).  The reference casts the factors to the model frame's dtype
    (observation.py:29-54, component.py:94-95) and its own tests build ``Frame(..., dtype=np.float64)``
    (tests/test_blend.py:63, 83, 104), so a float64 MODEL frame is accepted as the reference accepts it: ONE
    warning says that the factors are stored and stepped in float32, and the components report float32.
    ``scarlet_amd.component.STRICT_FLOAT32_FRAME = True`` turns the warning into a TypeError for callers that
    must not lose precision silently.  float64 DATA is cast by Observation.match as in the reference
    (observation.py:172-181)."""
    global _warned_float64_frame
    if np.dtype(getattr(frame, "dtype", np.float32)) == np.dtype(np.float32):
        return
    if STRICT_FLOAT32_FRAME:
        raise TypeError("scarlet_amd computes in float32: a model Frame with dtype %s is not supported "
                        "(build the Frame with dtype=numpy.float32; float64 images are cast by Observation.match)"
                        % np.dtype(frame.dtype))
    if not _warned_float64_frame:
        _warned_float64_frame = True
        logger.warning("model Frame has dtype %s: scarlet_amd stores and steps the factors in float32 "
                       "(reductions accumulate in float64); results agree with a float64 run to ~1e-6",
                       np.dtype(frame.dtype))


class Component(object):
    """A single component of a blend: ``model = sed[:, None, None] * morph[None]``.

    Parameters follow the reference (component.py:70-112): `frame`, `sed` (bands,),
    `morph` (height, width), optional `prior`, `fix_sed`, `fix_morph`.
    """

    def __init__(self, frame, sed, morph, prior=None, fix_sed=False, fix_morph=False):
        self._frame = frame
        _require_float32_frame(frame)
        self._own_sed = _to_device(sed)
        self._own_morph = _to_device(morph)
        self._binding = None            # (blend, k) once adopted by a Blend
        self.sed_grad = 0
        self.morph_grad = 0
        self.prior = prior
        self.L_sed = 1
        self.L_morph = 1
        self._flags = BlendFlag.SED_NOT_CONVERGED | BlendFlag.MORPH_NOT_CONVERGED
        self._index = None
        self._parent = None
        self.fix_sed = fix_sed
        self.fix_morph = fix_morph

    # ---- storage: own tensors until a Blend adopts the component
    @property
    def _sed(self):
        if self._binding is not None:
            blend, k = self._binding
            return blend._factor_view("sed", k)
        return self._own_sed

    @_sed.setter
    def _sed(self, value):
        self._sed.copy_(_to_device(value))

    @property
    def _morph(self):
        if self._binding is not None:
            blend, k = self._binding
            return blend._factor_view("morph", k)
        return self._own_morph

    @_morph.setter
    def _morph(self, value):
        self._morph.copy_(_to_device(value))

    @property
    def flags(self):
        if self._binding is not None:
            blend, k = self._binding
            return BlendFlag(int(blend._batch.flags[0, k].item()))
        return self._flags

    @flags.setter
    def flags(self, value):
        if self._binding is not None:
            blend, k = self._binding
            blend._batch.flags[0, k] = int(value.value)
        self._flags = value

    @property
    def shape(self):
        """(channels, height, width) of the model frame."""
        return self._frame.shape

    @property
    def coord(self):
        """Coordinate of this node in its `ComponentTree`."""
        if self._index is not None:
            if self._parent._index is not None:
                return tuple(self._parent.coord) + (self._index,)
            return (self._index,)

    @property
    def frame(self):
        return self._frame

    @property
    def sed(self):
        """Device view of the SED."""
        return self._sed

    @property
    def morph(self):
        """Device view of the morphology."""
        return self._morph

    def get_model(self, sed=None, morph=None):
        """(bands, height, width) model of this component (reference component.py:148-170)."""
        if sed is not None and morph is not None:
            return sed[:, None, None] * morph[None, :, :]
        if sed is None and morph is None:
            return self._sed[:, None, None] * self._morph[None, :, :]
        raise ValueError("You need to supply `sed` and `morph` or neither")

    def get_flux(self):
        """Total flux in every band."""
        return self.morph.sum() * self.sed

    def backward_prior(self):
        """Add the prior's gradients and Lipschitz constants (reference component.py:177-187)."""
        if self.prior is not None:
            self.prior.compute_grad(self)
            if not self.fix_morph:
                self.morph_grad += self.prior.morph_grad
                self.L_morph += self.prior.L_morph
            if not self.fix_sed:
                self.sed_grad += self.prior.sed_grad
                self.L_sed += self.prior.L_sed

    def update(self):
        """Constraint hook run once per iteration after the gradient step; the base class
        applies no constraint (reference component.py:189-196)."""
        return self

    @property
    def step_morph(self):
        try:
            return 1 / self.L_morph
        except AttributeError:
            return None

    @property
    def step_sed(self):
        try:
            return 1 / self.L_sed
        except AttributeError:
            return None


class ComponentTree(object):
    """Hierarchy of components / sub-trees (reference component.py:213-407)."""

    def __init__(self, components):
        if not hasattr(components, "__iter__"):
            components = (components,)
        self._tree = tuple(components)
        self._index = None
        self._parent = None
        for i, c in enumerate(self._tree):
            if not isinstance(c, ComponentTree) and not isinstance(c, Component):
                raise NotImplementedError("argument needs to be list of Components or ComponentTrees")
            assert c.frame is self.frame, "All components need to share the same Frame"
            c._index = i
            c._parent = self
        self._components = None

    @property
    def components(self):
        """Flattened tuple of the leaf components, each one once (first occurrence wins)."""
        if self._components is None:
            found = []
            for node in self._tree:
                leaves = node.components if isinstance(node, ComponentTree) else [node]
                for leaf in leaves:
                    if not any(leaf is f for f in found):
                        found.append(leaf)
            self._components = tuple(found)
        return self._components

    @property
    def n_components(self):
        return len(self.components)

    @property
    def K(self):
        return self.n_components

    @property
    def frame(self):
        return self._tree[0].frame

    @property
    def sources(self):
        """The nodes the tree was built from (a source may hold several components)."""
        return self._tree

    @property
    def n_sources(self):
        return len(self._tree)

    @property
    def coord(self):
        if self._index is not None:
            if self._parent._index is not None:
                return tuple(self._parent.coord) + (self._index,)
            return (self._index,)

    def get_model(self, seds=None, morphs=None):
        """Sum of the component models (reference component.py:321-347)."""
        torch = _lib.require_gpu()
        model = torch.zeros(tuple(self.frame.shape), dtype=torch.float32, device="cuda")
        for k, c in enumerate(self.components):
            if seds is not None and morphs is not None:
                model = model + c.get_model(seds[k], morphs[k])
            else:
                model = model + c.get_model()
        return model

    def get_flux(self):
        total = None
        for c in self.components:
            total = c.get_flux() if total is None else total + c.get_flux()
        return total

    def update(self):
        """Run every top-level node's update() (reference component.py:359-367)."""
        for node in self._tree:
            node.update()

    def __iadd__(self, c):
        c_index = self.n_sources
        if isinstance(c, ComponentTree):
            self._tree = self._tree + c._tree
        elif isinstance(c, Component):
            self._tree = self._tree + (c,)
        else:
            raise NotImplementedError("argument needs to be Component or ComponentTree")
        c._index = c_index
        c._parent = self
        self._components = None
        return self

    def __getitem__(self, coord):
        if isinstance(coord, (tuple, list)):
            if len(coord) > 1:
                return self._tree[coord[0]].__getitem__(coord[1:])
            return self._tree[coord[0]]
        if isinstance(coord, int):
            return self._tree[coord]
        raise NotImplementedError("coord needs to be index or list of indices")
