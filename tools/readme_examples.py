"""The two usage examples of README.md, run as written (first import: the package), on synthetic inputs.
    python tools/readme_examples.py        (needs an MI355X)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scarlet_amd as scarlet
from scarlet_amd import synth

B, H, W, K = 5, 64, 64, 3
psfs = np.array([synth.gaussian_psf((15, 15), 1.2 + 0.15 * b) for b in range(B)])
model_psf = synth.gaussian_psf((15, 15), 0.9)
sc = synth.make_scene(7, B=B, H=H, W=W, K=K, psfs=psfs)
images, catalog = sc["images"], [tuple(int(v) for v in c) for c in sc["centers"]]
filters = list("grizy")
weights = np.ones_like(images)
bg_rms = np.ones(B) * 0.1

# the reference's API, one scene (docs/quickstart.ipynb of the reference)
frame = scarlet.Frame(images.shape, psfs=model_psf[None], channels=filters)
obs = scarlet.Observation(images, psfs=psfs, weights=weights, channels=filters).match(frame)
sources = [scarlet.ExtendedSource(frame, (y, x), obs, bg_rms) for (y, x) in catalog]
blend = scarlet.Blend(sources, obs).fit(30, e_rel=1e-3)
model = blend.get_model()
print("Blend.fit: iterations", blend.it, "model", tuple(model.shape), "loss", float(blend.mse[-1]))   # (a device tensor)
assert bool(model.isfinite().all()) and blend.mse[-1] < blend.mse[0]

# the batched entry point: S scenes of one shape at once
S = 6
scs = [synth.make_scene(100 + i, B=B, H=H, W=W, K=K, psfs=psfs) for i in range(S)]
images_SBHW = np.stack([s["images"] for s in scs]); centers_SK2 = np.stack([s["centers"] for s in scs])
psfs_SBPP = np.tile(psfs[None], (S, 1, 1, 1))
batch = scarlet.BlendBatch(images_SBHW, centers_SK2)
batch.set_diff_kernel(scarlet.fft.match_psfs_device(psfs_SBPP, model_psf[None]))
batch.init_extended(bg_rms)
batch.fit(30, e_rel=1e-3)
sed, morph = batch.sed_current, batch.morph_current
print("BlendBatch.fit: iterations per scene", batch.it.tolist(), "sed", tuple(sed.shape), "morph", tuple(morph.shape))
assert bool(sed.isfinite().all()) and bool(morph.isfinite().all()) and int(batch.status.abs().sum().item()) == 0
print("readme examples ok")
