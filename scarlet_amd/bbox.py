"""Integer bounding boxes (host side).  Mirrors the reference's ``scarlet/bbox.py`` API:
``Box``, ``trim``, ``flux_at_edge``.  Pure integer logic; arrays may be numpy arrays or
torch tensors (device tensors are reduced on the device, only the 4 bounds come back)."""
import numpy as np


class Box(object):
    """Axis-aligned box: lower-left corner ``yx0``, ``height``, ``width``; a box with a
    non-positive extent is empty (reference bbox.py:4-29)."""

    def __init__(self, yx0, height, width):
        if width <= 0 or height <= 0:
            self.yx0, self.height, self.width = None, 0, 0
        else:
            self.yx0, self.height, self.width = yx0, height, width

    @staticmethod
    def from_bounds(bottom, top, left, right):
        """Box spanning rows bottom..top and columns left..right inclusive."""
        return Box((bottom, left), top + 1 - bottom, right + 1 - left)

    @property
    def is_empty(self):
        return self.width == 0 or self.height == 0

    @property
    def slices(self):
        if self.is_empty:
            return slice(0, 0), slice(0, 0)
        b, l = self.yx0
        return slice(b, b + self.height), slice(l, l + self.width)

    @property
    def bottom(self):
        return None if self.is_empty else self.yx0[0]

    @property
    def left(self):
        return None if self.is_empty else self.yx0[1]

    @property
    def top(self):
        return None if self.is_empty else self.yx0[0] + self.height - 1

    @property
    def right(self):
        return None if self.is_empty else self.yx0[1] + self.width - 1

    @property
    def shape(self):
        return (self.height, self.width)

    def __or__(self, other):
        """Smallest box containing both."""
        return Box.from_bounds(min(self.bottom, other.bottom), max(self.top, other.top),
                               min(self.left, other.left), max(self.right, other.right))

    def __and__(self, other):
        """Overlap of the two boxes (empty box if none)."""
        b, t = max(self.bottom, other.bottom), min(self.top, other.top)
        l, r = max(self.left, other.left), min(self.right, other.right)
        if t < b or r < l:
            return Box((0, 0), width=0, height=0)
        return Box.from_bounds(b, t, l, r)

    def __str__(self):
        return "(({0}, {1}), ({2}, {3}))".format(self.bottom, self.top, self.left, self.right)

    def __repr__(self):
        return "<Box yx0={0}, height={1}, width={2}>".format(self.yx0, self.height, self.width)

    def copy(self):
        return Box((self.yx0[0], self.yx0[1]), self.height, self.width)

    def __eq__(self, other):
        return (self.left == other.left and self.right == other.right and
                self.top == other.top and self.bottom == other.bottom)


def _as_numpy_mask(mask):
    return mask.detach().cpu().numpy() if hasattr(mask, "detach") else np.asarray(mask)


def trim(X, min_value=0):
    """Tight box around the pixels of the 2-D array X that exceed `min_value`
    (reference bbox.py:174-193).  Device tensors are reduced by the HIP library."""
    if hasattr(X, "is_cuda") and X.is_cuda:
        import torch
        from . import _lib
        t = X if (X.dtype == torch.float32 and X.is_contiguous()) else X.to(torch.float32).contiguous()
        box = torch.zeros(4, dtype=torch.int32, device=t.device)
        _lib.check(_lib.lib.scarlet_trim(_lib.ptr(t), 1, t.shape[0], t.shape[1], float(min_value), _lib.ptr(box),
                                         _lib.stream_ptr()))
        b = box.cpu().numpy()
        if b[1] < 0:
            # the reference takes min() of an empty index list here
            raise ValueError("zero-size array to reduction operation minimum which has no identity")
        return Box.from_bounds(int(b[0]), int(b[1]), int(b[2]), int(b[3]))
    ys, xs = np.nonzero(_as_numpy_mask(X > min_value))
    return Box.from_bounds(int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max()))


def flux_at_edge(X, min_value=0):
    """True if any pixel on the border of X exceeds `min_value` (reference bbox.py:196-210)."""
    edge = max(X[:, 0].max(), X[:, -1].max(), X[0].max(), X[-1].max())
    return bool(edge > min_value)
