"""Synthetic blend scenes -- plain numpy, host side.

Shared by ``bench.py``, the tests, the fixture generator and the CPU-baseline leg so
that every path sees bit-identical inputs.  This is data generation, not the hot path.

Per scene ``s`` the generator is ``numpy.random.Generator(PCG64(20260000 + s))``:
K integer centres in [8, H-9] x [8, W-9] rejection-sampled to >= `min_sep` px Chebyshev
separation; morphology = elliptical Gaussian (sigma_major ~ U(1.5, 3.5), axis ratio
~ U(0.5, 1), angle ~ U(0, pi)) sampled at pixel centres, peak 1; SED = a * c with
a ~ logU(5, 50), c ~ U(0.2, 1)^B; images = sum_k sed_k (x) morph_k (convolved per band
with a PSF when `psfs` is given) + N(0, noise^2).
"""
import numpy as np

SEED0 = 20260000


def gaussian_psf(shape, sigma):
    """Pixel-integrated circular Gaussian (erf form), normalised to sum 1, float64."""
    from math import erf, sqrt
    ry, rx = shape[0] // 2, shape[1] // 2
    def prof(n, r):
        e = np.array([erf((i - r + 0.5) / (sqrt(2) * sigma)) - erf((i - r - 0.5) / (sqrt(2) * sigma))
                      for i in range(n)])
        return 0.5 * e
    img = np.outer(prof(shape[0], ry), prof(shape[1], rx))
    return img / img.sum()


def make_scene(index, B=5, H=64, W=64, K=4, noise=0.1, min_sep=4, psfs=None,
               dtype=np.float32):
    """Returns dict(images (B,H,W), centers (K,2) int, true_seds (K,B), true_morphs (K,H,W))."""
    rng = np.random.Generator(np.random.PCG64(SEED0 + int(index)))
    centers = []
    while len(centers) < K:
        cy = int(rng.integers(8, H - 8))
        cx = int(rng.integers(8, W - 8))
        if all(max(abs(cy - y), abs(cx - x)) >= min_sep for y, x in centers):
            centers.append((cy, cx))
    yy, xx = np.mgrid[:H, :W].astype(np.float64)
    morphs = np.zeros((K, H, W))
    seds = np.zeros((K, B))
    for k, (cy, cx) in enumerate(centers):
        smaj = rng.uniform(1.5, 3.5)
        q = rng.uniform(0.5, 1.0)
        th = rng.uniform(0, np.pi)
        dy, dx = yy - cy, xx - cx
        u = np.cos(th) * dx + np.sin(th) * dy
        v = -np.sin(th) * dx + np.cos(th) * dy
        morphs[k] = np.exp(-0.5 * ((u / smaj) ** 2 + (v / (smaj * q)) ** 2))
        amp = np.exp(rng.uniform(np.log(5.0), np.log(50.0)))
        seds[k] = amp * rng.uniform(0.2, 1.0, size=B)
    model = np.einsum('kb,kyx->byx', seds, morphs)
    if psfs is not None:
        from scipy.signal import fftconvolve
        model = np.array([fftconvolve(model[b], psfs[b], mode="same") for b in range(B)])
    images = model + rng.normal(0.0, noise, size=model.shape)
    return dict(images=images.astype(dtype), centers=np.array(centers, dtype=np.int32),
                true_seds=seds, true_morphs=morphs)


def make_batch(start, count, **kw):
    """Stack `count` scenes: images (S,B,H,W), centers (S,K,2)."""
    scenes = [make_scene(start + i, **kw) for i in range(count)]
    return dict(images=np.stack([s["images"] for s in scenes]),
                centers=np.stack([s["centers"] for s in scenes]))
