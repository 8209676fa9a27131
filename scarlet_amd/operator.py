"""Array-level proximal operators, proxmin style: ``prox(X, step, **params) -> X`` mutating X
in place.  Mirrors the reference's ``scarlet/operator.py`` API; X is a 2-D float32
PyTorch-ROCm tensor (numpy input is accepted, moved to the device, and the result copied
back into the caller's array).  All arithmetic runs in the C-ABI HIP library; there is no
CPU fallback.
"""
import ctypes
from functools import partial

import numpy as np

from . import _lib

_ALG = {"kspace": _lib.SYM_KSPACE, "soft": _lib.SYM_SOFT, "sdss": _lib.SYM_SDSS}


# ------------------------------------------------------------------ tensor plumbing
class _OnDevice(object):
    """Context: gives a contiguous float32 CUDA tensor for X and writes the result back into X
    (numpy array, CPU tensor, non-contiguous or non-float32 tensor) on exit."""

    def __init__(self, X):
        self.X = X

    def __enter__(self):
        torch = _lib.require_gpu()
        X = self.X
        if torch.is_tensor(X) and X.is_cuda and X.dtype == torch.float32 and X.is_contiguous():
            self.t, self.copy_back = X, False
        else:
            src = X if torch.is_tensor(X) else torch.as_tensor(np.ascontiguousarray(X))
            self.t = src.to(device="cuda", dtype=torch.float32).contiguous()
            self.copy_back = True
        return self.t

    def __exit__(self, *exc):
        if exc[0] is None and self.copy_back:
            torch = _lib.require_gpu()
            if torch.is_tensor(self.X):
                self.X.copy_(self.t.to(device=self.X.device, dtype=self.X.dtype))
            else:
                self.X[...] = self.t.cpu().numpy().astype(self.X.dtype, copy=False)
        return False


def _i32(pairs):
    torch = _lib.require_gpu()
    return torch.as_tensor(np.asarray(pairs, dtype=np.int32).reshape(-1, 2)).cuda().contiguous()


def _i32_center(pairs, H, W):
    """Centres that index an H x W array (the reference raises IndexError outside it)."""
    a = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
    if ((a[:, 0] < 0) | (a[:, 0] >= H) | (a[:, 1] < 0) | (a[:, 1] >= W)).any():
        raise IndexError("center %s is outside the %d x %d array" % (a.tolist(), H, W))
    return _i32(a)


def _f64(pairs):
    torch = _lib.require_gpu()
    return torch.as_tensor(np.asarray(pairs, dtype=np.float64).reshape(-1, 2)).cuda().contiguous()


# ------------------------------------------------------------------ simple operators
def prox_max_unity(X, step):
    """Scale X so that its maximum is one (reference operator.py:17-21)."""
    with _OnDevice(X) as t:
        t.div_(t.max())
    return X


def prox_center_on(X, step, tiny=1e-10):
    """Keep the centre pixel (shape//2) at least `tiny` (reference operator.py:149-158)."""
    cy, cx = X.shape[0] // 2, X.shape[1] // 2
    X[cy, cx] = max(float(X[cy, cx]), tiny)
    return X


def prox_sed_on(X, step, tiny=1e-10):
    """If no element of X is positive set all of them to `tiny` (reference operator.py:161-172)."""
    if bool((X <= 0).all()):
        X[:] = tiny
    return X


# ------------------------------------------------------------------ monotonicity
def _prox_weighted_monotonic(X, step, center, thresh=0):
    """operators_pybind11.prox_weighted_monotonic with the weights and the radial order
    generated on the device from `center` (reference operator.py:32-37, 540-621)."""
    with _OnDevice(X) as t:
        H, W = t.shape[-2:]
        c = _i32_center(center, H, W)
        _lib.check(_lib.lib.scarlet_prox_weighted_monotonic(_lib.ptr(t), 1, H, W, _lib.ptr(c),
                                                            ctypes.c_float(thresh), _lib.stream_ptr()))
    return X


def _prox_strict_monotonic(X, step, center, thresh=0):
    """operators_pybind11.prox_monotonic: nearest-neighbour reference pixel
    (reference operator.py:24-29)."""
    with _OnDevice(X) as t:
        H, W = t.shape[-2:]
        c = _i32_center(center, H, W)
        _lib.check(_lib.lib.scarlet_prox_nearest_monotonic(_lib.ptr(t), 1, H, W, _lib.ptr(c),
                                                           ctypes.c_float(thresh), _lib.stream_ptr()))
    return X


def prox_strict_monotonic(shape, use_nearest=False, thresh=0, center=None):
    """Build the monotonicity operator for images of `shape` with the peak at `center`
    (default: (shape-1)//2).  Returns ``prox(X, step)`` (reference operator.py:81-122).

    Unlike the reference no N-sized weight table or argsort is built: the callable only
    captures (center, thresh)."""
    if center is None:
        center = ((shape[0] - 1) >> 1, (shape[1] - 1) >> 1)
    center = (int(center[0]), int(center[1]))
    if use_nearest:
        if thresh != 0:
            # thresh and nearest neighbours are not compatible (reference operator.py:107-110)
            raise ValueError("Thresholding does not work with nearest neighbor monotonicity")
        return partial(_prox_strict_monotonic, center=center, thresh=thresh)
    return partial(_prox_weighted_monotonic, center=center, thresh=thresh)


# ------------------------------------------------------------------ symmetry
def _symmetry(X, center, algorithm, strength, fill, shift, full=False):
    with _OnDevice(X) as t:
        H, W = t.shape
        c = _i32_center((0, 0) if center is None else center, H, W)
        sh = None if shift is None else _f64(shift)
        alg = _ALG[algorithm] | (_lib.SYM_FULL_WINDOW if full else 0)
        _lib.check(_lib.lib.scarlet_prox_symmetry(
            _lib.ptr(t), 1, H, W, _lib.ptr(c), _lib.ptr(sh), alg, ctypes.c_float(strength),
            int(fill is not None), ctypes.c_float(0.0 if fill is None else fill), _lib.stream_ptr()))
    return X


def prox_sdss_symmetry(X, step):
    """min(X, X rotated by 180 degrees), in place (reference operator.py:231-239)."""
    return _symmetry(X, None, "sdss", 1.0, None, None, full=True)


def prox_soft_symmetry(X, step, strength=1):
    """X <- strength/2 (X + X rotated) + (1 - strength) X, in place (reference operator.py:242-251)."""
    return _symmetry(X, None, "soft", strength, None, None, full=True)


def prox_kspace_symmetry(X, step, shift=None, padding=10):
    """Symmetrise X about the sub-pixel position (shape//2 - shift): the reference does it by
    dropping the imaginary part in Fourier space; evaluated here as the equivalent dense
    real-space operator on the MFMA units.  Returns a NEW array like the
    reference (reference operator.py:253-288)."""
    if padding != 10:
        raise NotImplementedError("only the reference's default padding=10 is supported")
    torch = _lib.require_gpu()
    out = X.clone() if torch.is_tensor(X) else np.array(X, copy=True)
    _symmetry(out, None, "kspace", 1.0, None, shift, full=True)
    return out


def uncentered_operator(X, func, center=None, fill=None, **kwargs):
    """Apply `func` to the largest window of X that is point-symmetric about `center`
    (reference operator.py:175-228), including its quirk that for a centred peak the
    return value of `func` is returned without being written into X."""
    if center is None:
        flat = int(X.argmax())
        py, px = flat // X.shape[1], flat % X.shape[1]
    else:
        py, px = center
    cy, cx = X.shape[0] // 2, X.shape[1] // 2
    if py == cy and px == cx:
        return func(X, **kwargs)
    dy = int(2 * (py - cy)) + (0 if X.shape[0] % 2 else 1)
    dx = int(2 * (px - cx)) + (0 if X.shape[1] % 2 else 1)
    ys = slice(None, dy) if dy < 0 else slice(dy, None)
    xs = slice(None, dx) if dx < 0 else slice(dx, None)
    torch = _lib.require_gpu()
    window = X[ys, xs]
    work = window.clone() if torch.is_tensor(X) else window.copy()
    result = func(work, **kwargs)
    if fill is not None:
        X[:] = fill
    X[ys, xs] = result
    return X


def prox_uncentered_symmetry(X, step, center=None, algorithm="kspace", fill=None, shift=None, strength=.5):
    """Symmetry about an off-centre peak (reference operator.py:291-350).  The rule
    "kspace falls back to soft with strength 1 when `shift` is None or all zero" is evaluated
    exactly like the reference does -- ``np.all(shift == 0)`` on the object passed in."""
    if algorithm == "kspace" and (shift is None or np.all(shift == 0)):
        algorithm, strength = "soft", 1
    if algorithm not in _ALG:
        msg = "algorithm must be one of 'soft', 'sdss', 'kspace', recieved '{0}''"
        raise ValueError(msg.format(algorithm))
    if center is None:
        flat = int(X.argmax())
        center = (flat // X.shape[1], flat % X.shape[1])
    return _symmetry(X, center, algorithm, strength, fill, shift if algorithm == "kspace" else None)


# ------------------------------------------------------------------ operator geometry (host)
def getOffsets(width, coords=None):
    """Flat offsets of the 8 neighbours and the slices that align a shifted copy
    (reference operator.py:462-477)."""
    if coords is None:
        coords = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]
    offsets = [width * y + x for y, x in coords]
    slices = [slice(None, s) if s < 0 else slice(s, None) for s in offsets]
    slicesInv = [slice(-s, None) if s < 0 else slice(None, -s) for s in offsets]
    return offsets, slices, slicesInv


def sort_by_radius(shape, center=None):
    """Flat pixel indices ordered by distance from `center` (reference operator.py:40-78).
    Only needed by callers of the host-pointer drop-ins; the device sweep does not use it."""
    if center is None:
        cy, cx = (shape[0] - 1) >> 1, (shape[1] - 1) >> 1
    else:
        cy, cx = int(center[0]), int(center[1])
    yy, xx = np.mgrid[:shape[0], :shape[1]]
    return np.argsort(np.sqrt((xx - cx) ** 2 + (yy - cy) ** 2).flatten())


def diagonalizeArray(arr, shape=None, dtype=np.float64):
    """The 8 x N table of each pixel's eight neighbour values (row i = neighbour i in getOffsets'
    order) and the mask of entries that do not exist, i.e. neighbours beyond the array
    (reference operator.py:481-522).  Host-side set-up helper: the device kernels never need it.
    The mask is the geometric one (a neighbour outside the array), which is what the reference's
    slice and index arithmetic marks."""
    if shape is None:
        height, width = arr.shape
        data = np.asarray(arr).reshape(-1)
    elif np.ndim(arr) == 1:
        height, width = shape
        data = np.asarray(arr)
    else:
        raise ValueError("Expected either a 2D array or a 1D array and a shape")
    plane = data.reshape(height, width)
    yy, xx = np.mgrid[:height, :width]
    diagonals = np.zeros((8, height * width), dtype=dtype)
    mask = np.ones((8, height * width), dtype=bool)
    for n, (oy, ox) in enumerate([(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]):
        ok = (yy + oy >= 0) & (yy + oy < height) & (xx + ox >= 0) & (xx + ox < width)
        src = plane[np.clip(yy + oy, 0, height - 1), np.clip(xx + ox, 0, width - 1)]
        diagonals[n] = np.where(ok, src, 0).reshape(-1)
        mask[n] = ~ok.reshape(-1)
    return diagonals, mask


def getRadialMonotonicWeights(shape, useNearest=True, minGradient=1, center=None):
    """8 x N float64 table of the radial monotonicity operator (reference operator.py:540-621): for
    every pixel the weights of its eight neighbours -- zero unless the neighbour exists and is
    strictly closer to the peak; `useNearest`: `minGradient` on the single neighbour best aligned
    with the direction to the peak; else cos(angle to that direction), normalised to sum one.
    Memoised per (shape, centre, useNearest, minGradient) like the reference.  The HIP sweeps
    generate these weights on the fly; this table exists for callers of the
    reference API and for the host drop-ins of operators_pybind11."""
    from .cache import Cache
    if center is None:
        center = ((shape[0] - 1) // 2, (shape[1] - 1) // 2)
    name = "RadialMonotonicWeights"
    key = tuple(shape) + tuple(center) + (useNearest, minGradient)
    try:
        return Cache.check(name, key)
    except KeyError:
        pass
    H, W = int(shape[0]), int(shape[1])
    py, px = int(center[0]), int(center[1])
    yy, xx = np.mgrid[:H, :W]
    Y = (yy - py).astype(np.float64)
    X = (xx - px).astype(np.float64)
    r2 = X * X + Y * Y
    cosw = np.zeros((8, H * W), dtype=np.float64)
    for n, (oy, ox) in enumerate([(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]):
        exists = (yy + oy >= 0) & (yy + oy < H) & (xx + ox >= 0) & (xx + ox < W)
        closer = (X + ox) ** 2 + (Y + oy) ** 2 < r2
        with np.errstate(invalid="ignore", divide="ignore"):
            c = (-X * ox - Y * oy) / (np.sqrt(r2) * np.sqrt(float(ox * ox + oy * oy)))
        cosw[n] = np.where(exists & closer, c, 0.0).reshape(-1)
    if useNearest:
        out = np.zeros_like(cosw)
        out[np.argmax(cosw, axis=0), np.arange(H * W)] = minGradient
        out[:, px + py * W] = 0
    else:
        norm = cosw.sum(axis=0)
        norm[norm == 0] = 1
        out = cosw / norm[None, :]
    Cache.set(name, key, out)
    return out
