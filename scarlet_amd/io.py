"""Scene files: the ``.npz`` layout of the reference's data files (reference data/*.npz, used by
docs/quickstart.ipynb cell 3): ``images`` (bands, H, W), optional ``psfs`` (bands, P, P),
``variance`` and ``mask`` (bands, H, W), ``filters`` (bands,) and a structured ``catalog`` with
``x`` / ``y`` columns.  SURVEY.md 8f rank 4: the callers' format on the input side of the path.

Files are read with ``numpy.load(allow_pickle=False)`` only.
"""
import numpy as np


def load_scene(path):
    """Read one scene file.  Returns a dict with
    images (B,H,W) float32, psfs (B,P,P) float32 or None, weights (B,H,W) float32 or None
    (1 / sqrt(variance), zero where masked or variance <= 0: `weights` MULTIPLIES the residual in
    Observation.get_loss -- 0.5 sum (w (model - image))^2, reference observation.py:239 -- so this
    is the inverse-variance chi-square), channels (list of str) or None,
    centers (K,2) int32 -- catalog positions rounded to pixels, (y, x) order."""
    with np.load(path, allow_pickle=False) as d:
        files = set(d.files)
        images = np.ascontiguousarray(d["images"], dtype=np.float32)
        psfs = np.ascontiguousarray(d["psfs"], dtype=np.float32) if "psfs" in files else None
        weights = None
        if "variance" in files:
            var = np.asarray(d["variance"], dtype=np.float32)
            weights = np.where(var > 0, 1.0 / np.sqrt(np.where(var > 0, var, 1)), 0).astype(np.float32)
        if "mask" in files:
            good = np.asarray(d["mask"]) == 0
            weights = good.astype(np.float32) if weights is None else weights * good
        channels = [str(f) for f in d["filters"]] if "filters" in files else None
        if "catalog" in files:
            cat = d["catalog"]
            yx = np.stack([cat["y"], cat["x"]], axis=1)
        elif "catalog_yx" in files:
            yx = np.asarray(d["catalog_yx"])
        else:
            yx = np.zeros((0, 2))
    centers = np.rint(yx).astype(np.int32)
    H, W = images.shape[-2:]
    if len(centers) and ((centers[:, 0] < 0) | (centers[:, 0] >= H) | (centers[:, 1] < 0) | (centers[:, 1] >= W)).any():
        raise ValueError("catalog position outside the %d x %d frame in %s (the reference raises IndexError "
                         "when a source is initialised there)" % (H, W, path))
    return dict(images=images, psfs=psfs, weights=weights, channels=channels, centers=centers)


def group_by_shape(scenes):
    """{(B, H, W, K): [scene indices]} -- one BlendBatch per group (a batch is rectangular)."""
    groups = {}
    for i, s in enumerate(scenes):
        key = tuple(s["images"].shape) + (len(s["centers"]),)
        groups.setdefault(key, []).append(i)
    return groups


def stack_scenes(scenes):
    """Stack scenes of one shape group into the arrays `BlendBatch` takes:
    images (S,B,H,W), centers (S,K,2) and weights (S,B,H,W) or None."""
    shapes = {tuple(s["images"].shape) + (len(s["centers"]),) for s in scenes}
    if len(shapes) != 1:
        raise ValueError("scenes of different shapes cannot share a batch: %s (see group_by_shape)" % sorted(shapes))
    images = np.stack([s["images"] for s in scenes])
    centers = np.stack([s["centers"] for s in scenes]).astype(np.int32)
    has_w = [s.get("weights") is not None for s in scenes]
    weights = None
    if any(has_w):
        weights = np.stack([s["weights"] if w else np.ones_like(s["images"]) for s, w in zip(scenes, has_w)])
    return images, centers, weights


def build_blend(scene, bg_rms, model_psf=None, source="extended"):
    """The quickstart recipe (reference docs/quickstart.ipynb cells 3-9) on a loaded scene:
    Frame + Observation.match + one source per catalog entry + Blend."""
    from . import Frame, Observation, ExtendedSource, PointSource, Blend
    images = scene["images"]
    frame = Frame(images.shape, psfs=model_psf, channels=scene["channels"])
    obs = Observation(images, psfs=scene["psfs"], weights=scene["weights"], channels=scene["channels"]).match(frame)
    if source == "extended":
        sources = [ExtendedSource(frame, tuple(int(v) for v in p), obs, bg_rms) for p in scene["centers"]]
    else:
        sources = [PointSource(frame, tuple(int(v) for v in p), obs) for p in scene["centers"]]
    return Blend(sources, obs)
