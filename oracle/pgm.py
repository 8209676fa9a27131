"""CPU restatement (numpy + oracle/sweeps.c) of the reference's Blend.fit() hot path.

TEST INFRASTRUCTURE -- the *checker* and the timed CPU baseline, never the product.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product (``scarlet_amd``) never does, and fails loudly when
its HIP library is missing.

Every function cites the reference lines it follows (paths relative to
``/root/reference``).  The code is a restatement, written from the behaviour of those
lines, not a copy; it is pinned against

* the golden vectors the reference's own tests hold (``tests/test_*.py``), and
* fixtures produced by running the reference itself under ``oracle/refshim.py``
  (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``),

in ``tests/test_oracle_golden.py``.

Third-party arithmetic absent from ``/root/reference``:
``proxmin>=0.5.5`` (``prox_plus``/``prox_hard``/``prox_soft``, setup.py:135) is restated
from the behaviour pinned by ``tests/test_update.py:23-44,74-97``; ``autograd>=1.3``'s
reverse pass is replaced by the analytic adjoint of ``Observation.get_loss`` (checked
against finite differences in ``tests/test_oracle_golden.py``).
"""
import numpy as np

from . import native

# BlendFlag bit values (scarlet/component.py:13-36; enum.auto() -> 1, 2, 4, 8)
SED_NOT_CONVERGED = 1
MORPH_NOT_CONVERGED = 2
EDGE_PIXELS = 4
NO_VALID_PIXELS = 8

NEIGHBOURS = ((-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1))


# --------------------------------------------------------------------------- fft.py
def next_fast_len(n):
    """Smallest 5-smooth integer >= n (scipy.fftpack.next_fast_len, used at fft.py:99)."""
    n = int(n)
    if n <= 6:
        return max(n, 1) if n > 0 else 1
    best = None
    p5 = 1
    while p5 < 2 * n:
        p35 = p5
        while p35 < 2 * n:
            v = p35
            while v < n:
                v *= 2
            if best is None or v < best:
                best = v
            p35 *= 3
        p5 *= 5
    return best


def fft_shape(shape1, shape2, padding=3, axes=None):
    """fft.py:68-106 -- fast shape per transformed axis, last one forced even."""
    if axes is None:
        axes = range(len(shape1))
    out = [next_fast_len(shape1[a] + shape2[a] + padding) for a in axes]
    while out[-1] % 2:
        out[-1] = next_fast_len(out[-1] + 1)
    return out


def pad_to(arr, newshape, axes=None):
    """fft.py:38-65 -- zero-pad; the leading pad is (dS+1)//2."""
    if axes is None:
        axes = range(arr.ndim)
    width = [(0, 0)] * arr.ndim
    for n, ax in enumerate(axes):
        d = newshape[n] - arr.shape[ax]
        lead = (d + 1) // 2
        width[ax] = (lead, d - lead)
    return np.pad(arr, width, mode="constant")


def centered(arr, newshape):
    """fft.py:7-35 -- central crop starting at (cur-new+1)//2; ValueError if too small."""
    cur = np.array(arr.shape)
    new = np.asarray(newshape)
    if not np.all(new <= cur):
        raise ValueError("arr must be larger than newshape in both dimensions, received "
                         "{0}, and {1}".format(arr.shape, tuple(newshape)))
    start = (cur - new + 1) // 2
    return arr[tuple(slice(s, s + n) for s, n in zip(start, new))]


def to_kspace(image, fshape, axes):
    """Fourier.fft (fft.py:193-211): pad -> ifftshift -> rfftn over `axes`."""
    return np.fft.rfftn(np.fft.ifftshift(pad_to(image, fshape, axes), axes), axes=axes)


def from_kspace(spec, fshape, image_shape, axes):
    """Fourier.from_fft (fft.py:138-181): irfftn -> fftshift -> central crop."""
    img = np.fft.fftshift(np.fft.irfftn(spec, fshape, axes=axes), axes=axes)
    return centered(img, image_shape)


def match_psfs(psf1, psf2, padding=3):
    """fft.match_psfs (fft.py:282-301): difference kernel psf1 / psf2 in k-space, cropped
    to the larger of the two image shapes (compared on axis 0, as the reference does)."""
    axes = (-2, -1)
    shape = psf2.shape if psf1.shape[0] < psf2.shape[0] else psf1.shape
    F = fft_shape(psf1.shape, psf2.shape, padding, axes)
    spec = to_kspace(psf1, F, axes) / to_kspace(psf2, F, axes)
    return from_kspace(spec, F, shape, axes)


def convolve(image, kernel, padding=3, axes=(-2, -1)):
    """fft.convolve (fft.py:304-317): linear convolution, cropped to image.shape."""
    F = fft_shape(image.shape, kernel.shape, padding, axes)
    spec = to_kspace(image, F, axes) * to_kspace(kernel, F, axes)
    return from_kspace(spec, F, image.shape, axes)


# ----------------------------------------------------------- psf.py / interpolation.py
def gaussian(y, x, y0=0, x0=0, amplitude=None, sigma=1):
    """psf.gaussian (psf.py:24-47)."""
    if amplitude is None:
        amplitude = 1 / (np.pi ** 2 * sigma ** 2)
    X, Y = np.meshgrid(x, y)
    return amplitude * np.exp(-((X - x0) ** 2 + (Y - y0) ** 2) / (2 * sigma ** 2))


def moffat(y, x, y0, x0, amplitude, alpha, beta=1.5):
    """psf.moffat (psf.py:8-21)."""
    X, Y = np.meshgrid(x, y)
    return amplitude * (1 + ((X - x0) ** 2 + (Y - y0) ** 2) / alpha ** 2) ** -beta


def generate_psf_image(func, shape, subsamples=10, normalize=True, **kwargs):
    """psf.generate_psf_image (psf.py:55-89) + interpolation.apply_2D_trapezoid_rule
    (interpolation.py:506-552).  Note the reference's 0.4 (not 0.25) corner factor."""
    ry, rx = np.array(shape) // 2
    y = np.linspace(-ry, ry, shape[0])
    x = np.linspace(-rx, rx, shape[1])
    dy, dx = y[1] - y[0], x[1] - x[0]
    n = subsamples
    ys = np.linspace(y[0] - dy / 2, y[-1] + dy / 2, len(y) * n + 1)
    xs = np.linspace(x[0] - dx / 2, x[-1] + dx / 2, len(x) * n + 1)
    z = func(ys, xs, **kwargs)
    cell = 0.4 * (z[:-1, :-1] + z[1:, :-1] + z[:-1, 1:] + z[1:, 1:])
    vol = dy * dx * cell / n / n
    img = vol.reshape(len(y), n, len(x), n).sum(axis=(1, 3))
    if normalize:
        img /= img.sum()
    return img


def default_centroid_weight():
    """source.py:483-490: 41x41 sigma=0.9 Gaussian scaled to peak 1 (float64)."""
    psf = generate_psf_image(gaussian, (41, 41), amplitude=1, sigma=.9, normalize=False)
    return psf / psf.max()


# ------------------------------------------------------------------- component / loss
def component_model(sed, morph):
    """Component.get_model (component.py:148-170)."""
    return sed[:, None, None] * morph[None, :, :]


def scene_model(seds, morphs, shape, dtype):
    """ComponentTree.get_model (component.py:321-347): running sum over components."""
    model = np.zeros(shape, dtype=dtype)
    for s, m in zip(seds, morphs):
        model = model + component_model(s, m)
    return model


def render(model, diff_kernel):
    """Observation.render (observation.py:203-220), single observation, all bands."""
    if diff_kernel is None:
        return model
    return convolve(model, diff_kernel, axes=(1, 2))


def render_adjoint(g, diff_kernel):
    """Adjoint of `render`: same pad/shift/crop chain with conj(K-hat)."""
    if diff_kernel is None:
        return g
    axes = (1, 2)
    F = fft_shape(g.shape, diff_kernel.shape, 3, axes)
    spec = to_kspace(g, F, axes) * np.conj(to_kspace(diff_kernel, F, axes))
    return from_kspace(spec, F, g.shape, axes)


def loss_and_gradients(seds, morphs, images, weights, diff_kernel):
    """Blend._loss + autograd reverse pass (blend.py:105-139, observation.py:222-239).

    Returns (loss, [d/d sed_k], [d/d morph_k]) with
      d = w (render(M) - I), loss = 0.5 sum d^2, G = render^T(w d),
      d/d sed_k[b] = sum_yx G[b] morph_k,  d/d morph_k = sum_b sed_k[b] G[b].
    """
    model = scene_model(seds, morphs, images.shape, images.dtype)
    d = weights * (render(model, diff_kernel) - images)
    loss = 0.5 * np.sum(d ** 2)
    G = render_adjoint(weights * d, diff_kernel)
    sed_grads = [(G * m[None]).sum(axis=(1, 2)) for m in morphs]
    morph_grads = [(G * s[:, None, None]).sum(axis=0) for s in seds]
    return loss, sed_grads, morph_grads


def lipschitz(seds, morphs, n_obs=1, approximate=False, mse=None):
    """Blend._set_lipschitz (blend.py:186-223).  Returns (L_sed, L_morph); as in the
    reference L_sed comes from the morphology Gram and L_morph from the SED Gram."""
    if approximate:
        LS = sum((s ** 2).sum().item() for s in seds)
        LA = sum((m ** 2).sum().item() for m in morphs)
        if mse is not None and len(mse) > 1 and mse[-1] > mse[-2]:
            LS *= 2
            LA *= 2
    else:
        A = np.array(seds)
        S = np.array(morphs).reshape(len(morphs), -1)
        LA = np.real(np.linalg.eigvals(S.dot(S.T)).max())
        LS = np.real(np.linalg.eigvals(A.T.dot(A)).max())
    return LA * n_obs, LS * n_obs


# -------------------------------------------------------------------- measurement.py
def max_pixel(morph, center):
    """measurement.max_pixel (measurement.py:3-29): first maximum (row-major) of the
    5x5 window [cy-2, cy+3) x [cx-2, cx+3); numpy clips the high edge."""
    cy, cx = int(center[0]), int(center[1])
    win = morph[cy - 2:cy + 3, cx - 2:cx + 3]
    iy, ix = np.unravel_index(np.argmax(win), win.shape)
    return (int(iy + cy - 2), int(ix + cx - 2))


def psf_weighted_centroid(morph, psf, center):
    """measurement.psf_weighted_centroid (measurement.py:32-94).

    Window = +-min(distance to the first/last pixel, psf radius) about `center`;
    first moments of morph*psf in window index units; returns the rounded (half-even)
    centre in morph coordinates and shift = rounded - moment."""
    cy, cx = center
    rad = psf.shape[1] // 2
    ry = min(abs(0 - cy), abs(morph.shape[0] - 1 - cy), rad)
    rx = min(abs(0 - cx), abs(morph.shape[1] - 1 - cx), rad)
    mview = morph[cy - ry:cy + ry + 1, cx - rx:cx + rx + 1]
    pview = psf[rad - ry:rad + ry + 1, rad - rx:rad + rx + 1]
    w = mview * pview
    total = np.sum(w)
    iy, ix = np.indices(w.shape)
    my = np.sum(iy * w) / total
    mx = np.sum(ix * w) / total
    whole = np.round((my, mx))
    dy, dx = whole - (my, mx)
    new_center = tuple((whole + (cy - ry, cx - rx)).astype(int))
    return new_center, (dy, dx)


# ----------------------------------------------------------------------- operator.py
def radius_order(shape, center):
    """operator.sort_by_radius (operator.py:40-78): flat indices by distance."""
    cy, cx = int(center[0]), int(center[1])
    Y, X = np.mgrid[:shape[0], :shape[1]]
    dist = np.sqrt((X - cx) ** 2 + (Y - cy) ** 2)
    return np.argsort(dist.flatten())


def _neighbour_geometry(shape, center):
    """Per (neighbour i, pixel p): in-bounds & strictly closer mask, and cos of the angle
    between (pixel -> peak) and (pixel -> neighbour)."""
    H, W = shape
    cy, cx = int(center[0]), int(center[1])
    Y, X = np.mgrid[:H, :W]
    Y = (Y - cy).astype(np.float64)
    X = (X - cx).astype(np.float64)
    r2 = X ** 2 + Y ** 2
    valid = np.zeros((8, H, W), dtype=bool)
    cosw = np.zeros((8, H, W), dtype=np.float64)
    yy, xx = np.mgrid[:H, :W]
    for i, (oy, ox) in enumerate(NEIGHBOURS):
        inb = (yy + oy >= 0) & (yy + oy < H) & (xx + ox >= 0) & (xx + ox < W)
        closer = (X + ox) ** 2 + (Y + oy) ** 2 < r2
        valid[i] = inb & closer
        with np.errstate(invalid="ignore", divide="ignore"):
            c = (-X * ox - Y * oy) / (np.sqrt(r2) * np.sqrt(ox * ox + oy * oy))
        cosw[i] = np.where(valid[i], c, 0.0)
    return valid.reshape(8, -1), cosw.reshape(8, -1)


def radial_weights(shape, center, use_nearest=False):
    """operator.getRadialMonotonicWeights (operator.py:540-621): (8, N) float64.

    weighted: cos-weights over strictly closer in-bounds neighbours, normalised to sum 1;
    nearest : 1 on the first neighbour with the largest cos-weight, peak column zeroed."""
    H, W = shape
    valid, cosw = _neighbour_geometry(shape, center)
    if use_nearest:
        out = np.zeros_like(cosw)
        best = np.argmax(cosw, axis=0)
        out[best, np.arange(H * W)] = 1
        out[:, int(center[1]) + int(center[0]) * W] = 0
        return out
    norm = cosw.sum(axis=0)
    norm[norm == 0] = 1
    return cosw / norm[None, :]


def neighbour_offsets(width):
    """operator.py:117-118: flat offsets of the 8 neighbours."""
    return np.array([width * y + x for y, x in NEIGHBOURS])


_SWEEP_CACHE = {}


def _sweep_operator(shape, center, dtype):
    """Memo of (order, weights, offsets) per (shape, centre, dtype), the analogue of the
    reference's Cache (update.py:131-147, cache.py): building them costs an argsort."""
    key = (tuple(shape), (int(center[0]), int(center[1])), np.dtype(dtype).str)
    hit = _SWEEP_CACHE.get(key)
    if hit is None:
        if len(_SWEEP_CACHE) > 4096:
            _SWEEP_CACHE.clear()
        order = radius_order(shape, center)[1:].astype(np.int32)
        w = np.ascontiguousarray(radial_weights(shape, center), dtype=dtype)
        hit = (order, w, neighbour_offsets(shape[1]).astype(np.int32))
        _SWEEP_CACHE[key] = hit
    return hit


def prox_weighted_monotonic(X, center, thresh=0.0):
    """update.monotonic default path -> operator.prox_strict_monotonic(use_nearest=False)
    -> operators_pybind11.prox_weighted_monotonic (update.py:106-156, operator.py:81-122,
    32-37).  X (H, W) float32/float64, mutated in place (must be C-contiguous)."""
    order, w, offs = _sweep_operator(X.shape, center, X.dtype)
    native.prox_weighted_monotonic(X.reshape(-1), w, offs, order, thresh)
    return X


def nearest_reference(shape, center):
    """ref_idx of the nearest-neighbour operator (operator.py:104-113): per pixel the
    flat index of its reference pixel; the peak references itself."""
    H, W = shape
    w = radial_weights(shape, center, use_nearest=True)
    offs = neighbour_offsets(W)
    ref = np.arange(H * W)
    rows, cols = np.nonzero(w)
    ref[cols] = cols + offs[rows]
    return ref


def prox_nearest_monotonic(X, center, thresh=0.0):
    """operator.prox_strict_monotonic(use_nearest=True) -> operators_pybind11.prox_monotonic
    (operator.py:24-30,104-113).  float64 only, like the reference."""
    if thresh != 0:
        raise ValueError("Thresholding does not work with nearest neighbor monotonicity")
    shape = X.shape
    order = radius_order(shape, center)
    native.prox_monotonic(X.reshape(-1), nearest_reference(shape, center), order, thresh)
    return X


def symmetric_window(shape, center):
    """Integer window selection of operator.uncentered_operator (operator.py:197-219).

    Returns None when `center` is the array middle (shape//2) -- the reference then
    applies the operator to the whole array -- else (yslice, xslice)."""
    py, px = int(center[0]), int(center[1])
    cy, cx = shape[0] // 2, shape[1] // 2
    if py == cy and px == cx:
        return None
    dy = 2 * (py - cy) + (0 if shape[0] % 2 else 1)
    dx = 2 * (px - cx) + (0 if shape[1] % 2 else 1)
    ys = slice(None, dy) if dy < 0 else slice(dy, None)
    xs = slice(None, dx) if dx < 0 else slice(dx, None)
    return ys, xs


def uncentered(X, func, center, fill=None):
    """operator.uncentered_operator (operator.py:175-228).  When `center` is the array
    middle the reference returns func(X) WITHOUT writing into X (operator.py:203-204):
    in-place operators still act, value-returning ones (k-space) are a no-op."""
    win = symmetric_window(X.shape, center)
    if win is None:
        return func(X)
    if fill is not None:
        full = np.ones(X.shape, X.dtype) * fill
        full[win] = func(X[win])
        X[:] = full
    else:
        X[win] = func(X[win])
    return X


def soft_symmetry(X, strength=1):
    """operator.prox_soft_symmetry (operator.py:242-251), in place."""
    Xs = X[::-1, ::-1]
    X[:] = 0.5 * strength * (X + Xs) + (1 - strength) * X
    return X


def sdss_symmetry(X):
    """operator.prox_sdss_symmetry (operator.py:231-239), in place."""
    X[:] = np.minimum(X, X[::-1, ::-1].copy())
    return X


def kspace_symmetry(X, shift, padding=10):
    """operator.prox_kspace_symmetry (operator.py:253-288) + interpolation.mk_shifter
    (interpolation.py:302-340).  Returns a NEW array."""
    F = fft_shape(X.shape, X.shape, padding)
    dy, dx = shift
    spec = to_kspace(X, F, (0, 1))
    zero = X <= 0
    sy = np.exp(-1j * 2 * np.pi * np.fft.fftfreq(F[0]))
    sx = np.exp(-1j * 2 * np.pi * np.fft.rfftfreq(F[1]))
    r = spec * sy[:, None] ** (-dy)
    r *= sx[None, :] ** (-dx)
    r = r.real
    r = r * sy[:, None] ** dy
    r = r * sx[None, :] ** dx
    out = from_kspace(r, F, X.shape, [0, 1]).copy()
    out[zero] = 0
    return np.real(out)


def prox_symmetry(X, center, algorithm="kspace", fill=None, shift=None, strength=.5):
    """operator.prox_uncentered_symmetry (operator.py:291-350), X mutated in place."""
    # NB the reference evaluates `np.all(shift == 0)` on whatever object it is handed: the
    # pipeline passes component.shift, a *tuple*, for which `shift == 0` is plain False, so
    # inside Blend.fit the soft fallback only triggers while shift is None.  Keep that.
    if algorithm == "kspace" and (shift is None or np.all(shift == 0)):
        algorithm, strength = "soft", 1
    if algorithm == "kspace":
        return uncentered(X, lambda w: kspace_symmetry(w, shift), center, fill)
    if algorithm == "sdss":
        return uncentered(X, sdss_symmetry, center, fill)
    if algorithm == "soft":
        return uncentered(X, lambda w: soft_symmetry(w, strength), center, fill)
    raise ValueError("algorithm must be one of 'soft', 'sdss', 'kspace', recieved '{0}''"
                     .format(algorithm))


# ------------------------------------------------------------------------- update.py
def prox_plus(X):
    """proxmin.operators.prox_plus as pinned by tests/test_update.py:23-44."""
    X[X < 0] = 0
    return X


def prox_hard(X, step, thresh):
    """proxmin prox_hard (tests/test_update.py:80-87): zero where |x| < thresh*step."""
    X[np.abs(X) < thresh * step] = 0
    return X


def prox_soft(X, step, thresh):
    """proxmin prox_soft (tests/test_update.py:89-97), in place."""
    X[:] = np.sign(X) * prox_plus(np.abs(X) - thresh * step)
    return X


def normalize(sed, morph, kind="morph_max"):
    """update.normalized (update.py:35-68), in place."""
    t = kind.lower()
    if t == "sed":
        n = sed.sum()
        sed[:] = sed / n
        morph[:] = morph * n
    elif t == "morph":
        n = morph.sum()
        sed[:] = sed * n
        morph[:] = morph / n
    elif t == "morph_max":
        n = morph.max()
        sed[:] = sed * n
        morph[:] = morph / n
    else:
        raise ValueError("Unrecognized normalization '{0}'".format(kind))


def _bbox_view(morph, center, bbox):
    """Common bbox preamble of update.monotonic / update.symmetric (update.py:118-127,
    176-185).  bbox = (bottom, top, left, right) inclusive or None."""
    if bbox is None:
        return morph, center, None
    b, t, l, r = bbox
    sl = (slice(b, t + 1), slice(l, r + 1))
    return morph[sl], (center[0] - b, center[1] - l), sl


def update_monotonic(morph, center, use_nearest=False, thresh=0, bbox=None):
    """update.monotonic (update.py:106-156): works on a copy, writes back; with a bbox
    everything outside is zeroed."""
    view, c, sl = _bbox_view(morph, center, bbox)
    if sl is not None and (view.shape[0] <= 1 or view.shape[1] <= 1):
        return morph
    work = np.ascontiguousarray(view.copy())
    if use_nearest:
        prox_nearest_monotonic(work, c, thresh)
    else:
        prox_weighted_monotonic(work, c, thresh)
    if sl is not None:
        morph[:] = 0
        morph[sl] = work
    else:
        morph[:] = work
    return morph


def update_symmetric(morph, center, shift=None, algorithm="kspace", bbox=None, fill=None,
                     strength=.5):
    """update.symmetric (update.py:170-199)."""
    view, c, sl = _bbox_view(morph, center, bbox)
    if sl is not None and (view.shape[0] <= 1 or view.shape[1] <= 1):
        return morph
    prox_symmetry(view, c, algorithm, fill, shift, strength)
    if sl is not None:
        # Reference quirk (update.py:194-196): `morph` there is a *view* of
        # component.morph, so zeroing component.morph also zeroes the view before it is
        # written back -- with a bbox the whole morphology ends up zero.  Reproduced.
        morph[:] = 0
        morph[sl] = view
    return morph


# ------------------------------------------------------------------------- source.py
class Source(object):
    """State of one component + the options of PointSource/ExtendedSource.update."""

    def __init__(self, sed, morph, center, dtype, symmetric=True, monotonic=True,
                 centroid_weight=None, l0_thresh=None, l1_thresh=None, fix_sed=False,
                 fix_morph=False):
        self.sed = np.array(sed, dtype=dtype)
        self.morph = np.array(morph, dtype=dtype)
        self.center = (int(center[0]), int(center[1]))
        self.shift = None
        self.symmetric = symmetric
        self.monotonic = monotonic
        self.l0_thresh = l0_thresh
        self.l1_thresh = l1_thresh
        self.fix_sed = fix_sed
        self.fix_morph = fix_morph
        self.centroid_weight = centroid_weight
        self.flags = SED_NOT_CONVERGED | MORPH_NOT_CONVERGED
        self.last_sed = np.zeros_like(self.sed)
        self.last_morph = np.zeros_like(self.morph)
        self.L_sed = 1
        self.L_morph = 1


def source_update(src, it):
    """PointSource.update (source.py:402-440): centre -> (every 5th it) centroid ->
    symmetric(kspace) -> monotonic(weighted) -> [sparse] -> positive -> morph_max."""
    if getattr(src, "trace", None) is not None:
        # test hook (tests/test_gpu_parity_long.py): the stepped morphology, i.e. what the k-space symmetry's
        # `X <= 0` mask (operator.py:285-287) is about to see
        src.trace["step"].append(src.morph.copy())
    src.center = max_pixel(src.morph, src.center)
    if src.symmetric:
        if it % 5 == 0:
            src.center, src.shift = psf_weighted_centroid(src.morph, src.centroid_weight,
                                                          src.center)
        update_symmetric(src.morph, src.center, src.shift, algorithm="kspace")
    if src.monotonic:
        update_monotonic(src.morph, src.center)
    if src.l0_thresh is not None:
        prox_hard(src.morph, 1 / src.L_morph, src.l0_thresh)
    if src.l1_thresh is not None:
        prox_soft(src.morph, 1 / src.L_morph, src.l1_thresh)
    if getattr(src, "trace", None) is not None:
        # ... and the morphology as prox_plus is about to see it
        src.trace["pre_plus"].append(src.morph.copy())
    prox_plus(src.sed)
    prox_plus(src.morph)
    normalize(src.sed, src.morph, "morph_max")


def pixel_sed(images, pixel):
    """source.get_pixel_sed (source.py:21-38)."""
    return images[:, pixel[0], pixel[1]].copy()


def psf_sed(images, pixel, obs_psfs=None, frame_psf=None):
    """source.get_psf_sed (source.py:41-71)."""
    sed = pixel_sed(images, pixel)
    if obs_psfs is not None:
        sed /= obs_psfs.max(axis=(1, 2))
    if frame_psf is not None:
        sed = sed * frame_psf.max()
    return sed


def detection_coadd(sed, bg_rms, images, thresh=1):
    """source.build_detection_coadd (source.py:101-136)."""
    bg_rms = np.asarray(bg_rms)
    if np.any(bg_rms <= 0):
        raise ValueError("bg_rms must be greater than zero in all channels")
    pos = [c for c in range(len(sed)) if sed[c] > 0]
    w = np.array([sed[c] / bg_rms[c] ** 2 for c in pos])
    jac = np.array([sed[c] ** 2 / bg_rms[c] ** 2 for c in pos]).sum()
    detect = np.einsum('i,i...', w, [images[c] for c in pos]) / jac
    cutoff = thresh * np.sqrt((w ** 2 * np.array([bg_rms[c] for c in pos]) ** 2).sum()) / jac
    return detect, cutoff


class SourceInitError(Exception):
    pass


def init_extended_source(pixel, images, bg_rms, obs_psfs=None, frame_psf=None, thresh=1.,
                         symmetric=True, monotonic=True):
    """source.init_extended_source (source.py:139-180)."""
    sed = psf_sed(images, pixel, obs_psfs, frame_psf)
    morph, cutoff = detection_coadd(sed, bg_rms, images, thresh)
    morph = np.ascontiguousarray(morph)
    if symmetric:
        prox_symmetry(morph, pixel, algorithm="sdss")
    if monotonic:
        prox_weighted_monotonic(morph, pixel, thresh=.1)
    mask = morph > cutoff
    if mask.sum() == 0:
        raise SourceInitError("No flux above threshold={2} for source at y={0} x={1}"
                              .format(pixel[0], pixel[1], cutoff))
    morph[~mask] = 0
    morph /= morph[int(pixel[0]), int(pixel[1])]
    return sed, morph


def init_combined_extended_source(pixel, images_list, bg_rms_list, obs_idx=0, thresh=1., symmetric=True,
                                  monotonic=True):
    """source.init_combined_extended_source (source.py:183-240), observations without PSFs: the SED runs over
    the channels of all observations, the morphology comes from observation `obs_idx` alone."""
    sed = np.concatenate([psf_sed(im, pixel) for im in images_list])
    _, morph = init_extended_source(pixel, images_list[obs_idx], bg_rms_list[obs_idx], None, None, thresh,
                                    symmetric, monotonic)
    return sed, morph


# -------------------------------------------------------------------------- blend.py
class Scene(object):
    """One blend: data + sources.  Mirrors Blend + one matched Observation."""

    def __init__(self, images, sources, weights=1, diff_kernel=None):
        self.images = images
        self.weights = weights
        self.diff_kernel = diff_kernel
        self.sources = sources
        self.mse = []

    @property
    def it(self):
        return len(self.mse)


def make_extended_scene(images, pixels, bg_rms, obs_psfs=None, frame_psf=None, weights=1,
                        l0_thresh=None):
    """docs/quickstart.ipynb cells 7-11 for one scene: Frame/Observation.match ->
    ExtendedSource per catalogue pixel (source.py:444-492; its constructor runs
    update() once with it=0)."""
    dtype = images.dtype
    diff = None
    cw = default_centroid_weight()
    if obs_psfs is not None:
        fp = np.asarray(frame_psf, dtype=dtype)
        op = np.asarray(obs_psfs, dtype=dtype)
        diff = match_psfs(op, fp[None] if fp.ndim == 2 else fp)
        fp2 = fp if fp.ndim == 2 else fp[0]
        cw = fp2
    else:
        fp2 = None
    sources = []
    for px in pixels:
        px = (int(px[0]), int(px[1]))
        sed, morph = init_extended_source(px, images, bg_rms, obs_psfs if obs_psfs is None else op,
                                          fp2)
        s = Source(sed, morph, px, dtype, centroid_weight=cw, l0_thresh=l0_thresh)
        source_update(s, 0)
        sources.append(s)
    return Scene(images, sources, weights, diff)


def scene_from_state(images, seds, morphs, centers, shifts, weights=1, diff_kernel=None,
                     centroid_weight=None, l0_thresh=None):
    """Scene whose sources start from a given (sed, morph, centre, shift) state, e.g. the
    state right after the ExtendedSource constructors ran.  `shifts` may be None."""
    dtype = images.dtype
    cw = default_centroid_weight() if centroid_weight is None else centroid_weight
    sources = []
    for k in range(len(seds)):
        s = Source(seds[k], morphs[k], centers[k], dtype, centroid_weight=cw, l0_thresh=l0_thresh)
        if shifts is not None:
            s.shift = (float(shifts[k][0]), float(shifts[k][1]))
        sources.append(s)
    return Scene(images, sources, weights, diff_kernel)


def check_convergence(scene, e_rel):
    """Blend._check_convergence (blend.py:141-184)."""
    e2 = e_rel ** 2
    if scene.it > 1:
        done = True
        for c in scene.sources:
            if ((c.last_sed - c.sed) ** 2).sum() <= e2 * (c.sed ** 2).sum():
                c.flags &= ~SED_NOT_CONVERGED
            else:
                c.flags |= SED_NOT_CONVERGED
                done = False
            if ((c.last_morph - c.morph) ** 2).sum() <= e2 * (c.morph ** 2).sum():
                c.flags &= ~MORPH_NOT_CONVERGED
            else:
                c.flags |= MORPH_NOT_CONVERGED
                done = False
    else:
        done = False
    for c in scene.sources:
        c.last_sed = c.sed.copy()
        c.last_morph = c.morph.copy()
    return done


def fit(scene, max_iter=200, e_rel=1e-2, approximate_L=False, callback=None):
    """Blend.fit (blend.py:65-102), one scene, single observation.  `callback(scene)` (test hook) runs
    after every iteration's update, before the convergence check."""
    for _ in range(max_iter):
        seds = [c.sed for c in scene.sources]
        morphs = [c.morph for c in scene.sources]
        extra = getattr(scene, "observations", None)
        if extra is None:
            loss, gs, gm = loss_and_gradients(seds, morphs, scene.images, scene.weights,
                                              scene.diff_kernel)
            n_obs = 1
        else:
            # several observations (blend.py:120-139): the losses add up; an observation sees the
            # channels of its band slice (observation.py:184-189, 222-224)
            loss = 0
            gs = [np.zeros_like(sd) for sd in seds]
            gm = [np.zeros_like(m) for m in morphs]
            for ob in extra:
                sl = ob.get("band_slice", slice(None))
                l_, gs_, gm_ = loss_and_gradients([sd[sl] for sd in seds], morphs, ob["images"],
                                                  ob.get("weights", 1), ob.get("diff_kernel"))
                loss = loss + l_
                for k in range(len(seds)):
                    gs[k][sl] += gs_[k]
                    gm[k] = gm[k] + gm_[k]
            n_obs = len(extra)
        scene.mse.append(loss)
        L_sed, L_morph = lipschitz(seds, morphs, n_obs, approximate_L, scene.mse)
        for c, g_s, g_m in zip(scene.sources, gs, gm):
            c.L_sed, c.L_morph = L_sed, L_morph
            prior = getattr(c, "prior", None)
            if prior is not None:
                # Component.backward_prior (component.py:177-187): prior = (grad_func, L_func),
                # both called on the factors before the step (Prior.compute_grad, component.py:58-67)
                p_gs, p_gm = prior[0](c.sed, c.morph)
                p_Ls, p_Lm = prior[1](c.sed, c.morph)
                if not c.fix_morph:
                    g_m = g_m + p_gm
                    c.L_morph = c.L_morph + p_Lm
                if not c.fix_sed:
                    g_s = g_s + p_gs
                    c.L_sed = c.L_sed + p_Ls
            if not c.fix_sed:
                c.sed = c.sed - (1 / c.L_sed) * g_s
            if not c.fix_morph:
                c.morph = c.morph - (1 / c.L_morph) * g_m
        it = scene.it
        trees = getattr(scene, "trees", None)
        if trees is not None:              # sources made of several components (MultiSource)
            for t in trees:
                multi_source_update(t, it) if isinstance(t, MultiSource) else source_update(t, it)
        else:
            for c in scene.sources:
                source_update(c, it)
        if callback is not None:
            callback(scene)
        if check_convergence(scene, e_rel):
            break
    return scene



# --------------------------------------------------------------------------- MultiComponentSource
def best_fit_seds(morphs, images):
    """source.get_best_fit_seds (source.py:74-98)."""
    K = len(morphs)
    S = morphs.reshape(K, -1)
    data = images.reshape(images.shape[0], -1)
    return np.dot(np.linalg.inv(np.dot(S, S.T)), np.dot(S, data.T))


def init_multicomponent_source(pixel, images, bg_rms, flux_percentiles=None, obs_psfs=None,
                               frame_psf=None, thresh=1., symmetric=True, monotonic=True):
    """source.init_multicomponent_source (source.py:242-295)."""
    if flux_percentiles is None:
        flux_percentiles = [25]
    sed, morph = init_extended_source(pixel, images, bg_rms, obs_psfs, frame_psf, thresh, symmetric, monotonic)
    K = len(flux_percentiles) + 1
    morphs = np.zeros((K,) + morph.shape, dtype=morph.dtype)
    morphs[0] = morph
    max_flux = morph.max()
    last_thresh = 0
    for k, perc in enumerate(np.sort(flux_percentiles), start=1):
        flux_thresh = perc * max_flux / 100
        mask_ = morph > flux_thresh
        morphs[k - 1][mask_] = flux_thresh - last_thresh
        morphs[k][mask_] = morph[mask_] - flux_thresh
        last_thresh = flux_thresh
    for k in range(K):
        morphs[k] /= morphs[k].max()
    return best_fit_seds(morphs, images), morphs


class MultiSource(object):
    """MultiComponentSource (source.py:495-641): components share centre and shift."""

    def __init__(self, components, center, symmetric=True, monotonic=True, centroid_weight=None):
        self.components = components
        self.center = (int(center[0]), int(center[1]))
        self.shift = None
        self.symmetric = symmetric
        self.monotonic = monotonic
        self.centroid_weight = default_centroid_weight() if centroid_weight is None else centroid_weight


def multi_source_update(ms, it):
    """MultiComponentSource.update (source.py:605-641).  The components themselves never get a
    `shift` attribute, so update.symmetric passes shift=None (update.py:187-190)."""
    _morph = np.sum([c.morph * c.sed.sum() for c in ms.components], axis=0)
    ms.center = max_pixel(_morph, ms.center)
    if ms.symmetric and it % 5 == 0:
        ms.center, ms.shift = psf_weighted_centroid(_morph, ms.centroid_weight, ms.center)
    for c in ms.components:
        if ms.symmetric:
            update_symmetric(c.morph, ms.center, None, algorithm="kspace")
        if ms.monotonic:
            update_monotonic(c.morph, ms.center)
        prox_plus(c.sed)
        prox_plus(c.morph)
        normalize(c.sed, c.morph, "morph_max")
        c.center = ms.center

# --------------------------------------------------------------------------- bbox.py
def trim_bounds(X, min_value=0):
    """bbox.trim (bbox.py:174-193): (bottom, top, left, right) of X > min_value."""
    ys, xs = np.where(X > min_value)
    return int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max())


def flux_at_edge(X, min_value=0):
    """bbox.flux_at_edge (bbox.py:196-210)."""
    return bool(max(X[:, 0].max(), X[:, -1].max(), X[0].max(), X[-1].max()) > min_value)


# --------------------------------------------------------------------------- measurement.threshold / update.threshold
def threshold(morph):
    """measurement.threshold (measurement.py:97-112): noise cut from the histogram of
    log10(positive pixels).  50 bins, or size/10 when fewer than 500 pixels are positive
    (one bin -> no cut); the cut is the lower edge of the LAST empty bin, 0 if none is empty.
    Returns (thresh, bins)."""
    pos = morph[morph > 0]
    bins = 50
    if pos.size < 500:
        bins = max(int(pos.size / 10), 1)
        if bins == 1:
            return 0, bins
    hist, edges = np.histogram(np.log10(pos).reshape(-1), bins)
    empty = np.where(hist == 0)[0]
    if len(empty) == 0:
        return 0, bins
    return 10 ** edges[empty[-1]], bins


def update_threshold(morph):
    """update.threshold (update.py:85-103): zero below the cut (in place), return the cut and
    the tight box (bottom, top, left, right) of what is left (bbox.trim)."""
    thresh, _ = threshold(morph)
    morph[morph < thresh] = 0
    return thresh, trim_bounds(morph)


# --------------------------------------------------------------------------- interpolation.py / update.translation
def lanczos(dx, a=3):
    """interpolation.lanczos (interpolation.py:233-252): taps at floor(dx) + (-a+1 .. a)."""
    if np.abs(dx) > 1:
        raise ValueError("The fractional shift dx must be between -1 and 1")
    window = np.arange(-a + 1, a + 1) + np.floor(dx)
    return np.sinc(dx - window) * np.sinc((dx - window) / a), window.astype(int)


def bilinear(dx):
    """interpolation.bilinear (interpolation.py:139-165)."""
    if np.abs(dx) > 1:
        raise ValueError("The fractional shift dx must be between -1 and 1")
    if dx >= 0:
        return np.array([1 - dx, dx]), np.arange(2)
    return np.array([-dx, 1 + dx]), np.array([-1, 0])


def cubic_spline(dx, a=1, b=0):
    """interpolation.cubic_spline (interpolation.py:168-213): piecewise cubic on |x| <= 1 and 1 < |x| < 2."""
    if np.abs(dx) > 1:
        raise ValueError("The fractional shift dx must be between -1 and 1")
    window = np.arange(-1, 3) + np.floor(dx)
    x = np.abs(dx - window)
    out = np.zeros(4)
    for i, v in enumerate(x):
        if v <= 1:
            out[i] = ((-6 * a - 9 * b + 12) * v ** 3 + (6 * a + 12 * b - 18) * v ** 2 + (-2 * b + 6)) / 6
        elif v < 2:
            out[i] = ((-6 * a - b) * v ** 3 + (30 * a + 6 * b) * v ** 2 + (-48 * a - 12 * b) * v + (24 * a + 8 * b)) / 6
    return out, window.astype(int)


def quintic_spline(dx):
    """interpolation.quintic_spline (interpolation.py:255-270), window -3 .. 3."""
    window = np.arange(-3, 4)
    out = np.zeros(7)
    for i, v in enumerate(np.abs(dx - window)):
        if v <= 1:
            out[i] = 1 + v ** 3 / 12 * (-95 + 138 * v - 55 * v ** 2)
        elif v <= 2:
            out[i] = (v - 1) * (v - 2) / 24 * (-138 + 348 * v - 249 * v ** 2 + 55 * v ** 3)
        elif v <= 3:
            out[i] = (v - 2) * (v - 3) ** 2 / 24 * (-54 + 50 * v - 11 * v ** 2)
    return out, window


def _project(image, shape, yx0=None):
    """interpolation.project_image (interpolation.py:6-84): place `image` in zeros(shape) with its
    first pixel at shape//2 + yx0 (yx0 = -(image.shape//2) when None), clipped at the borders."""
    out = np.zeros(shape)
    if yx0 is None:
        yx0 = (-(image.shape[0] // 2), -(image.shape[1] // 2))
    sl_out, sl_in = [], []
    for n, i, o in zip(shape, image.shape, yx0):
        lo = o + (n >> 1)
        hi = lo + i
        sl_out.append(slice(max(0, lo), min(n, hi)))
        sl_in.append(slice(max(0, -lo), max(n - lo, -hi)))
    out[tuple(sl_out)] = image[tuple(sl_in)]
    return out


def fft_resample(img, dy, dx, kernel=lanczos, **kwargs):
    """interpolation.fft_resample (interpolation.py:408-448) with fft_convolve (:114-136):
    circular convolution of the zero-padded image with the separable kernel, both ifftshift-ed,
    the real part fftshift-ed back and cropped to the image."""
    ky, ywin = kernel(dy, **kwargs)
    kx, xwin = kernel(dx, **kwargs)
    k2 = np.outer(ky, kx)
    shape = (img.shape[0] + k2.shape[0] + 3, img.shape[1] + k2.shape[1] + 3)
    K = _project(k2, shape, (ywin[0], xwin[0]))
    I = _project(img, shape)
    prod = np.fft.fft2(np.fft.ifftshift(I)) * np.fft.fft2(np.fft.ifftshift(K))
    res = np.fft.fftshift(np.real(np.fft.ifft2(prod)))
    return _project(res, img.shape)


def update_translation(morph, shift, direction=1):
    """update.translation (update.py:159-167): morph[:] = fft_resample(morph, dy, dx), Lanczos-3."""
    morph[:] = fft_resample(morph, shift[0] * direction, shift[1] * direction)
    return morph
