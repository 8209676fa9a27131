"""Driver for rocprofv3 passes on BASELINE config 3's shape: S scenes 5 x 128 x 128, K = 8, PSF 41 x 41; 3 iterations."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from scarlet_amd import synth, fft as fftmod
from scarlet_amd.batch import BlendBatch
S = int(os.environ.get("PMC_SCENES", "4096")); B, H, W, K = 5, 128, 128, 8
obs = np.array([synth.gaussian_psf((41, 41), 1.2 + 0.15 * b) for b in range(B)]); model = synth.gaussian_psf((41, 41), 0.9)
diff = fftmod.match_psfs(fftmod.Fourier(obs.astype(np.float32)), fftmod.Fourier(model[None].astype(np.float32))).image
scenes = [synth.make_scene(300 + i, B=B, H=H, W=W, K=K, psfs=obs) for i in range(32)]
reps = (S + 31) // 32
images = np.tile(np.stack([s["images"] for s in scenes]), (reps, 1, 1, 1))[:S]
centers = np.tile(np.stack([s["centers"] for s in scenes]), (reps, 1, 1))[:S]
b = BlendBatch(images, centers, centroid_weight=model.astype(np.float32))
b.set_diff_kernel(np.asarray(diff, dtype=np.float32))
b.init_extended(np.ones(B) * 0.1, sed_scale=(model.max() / obs.max(axis=(1, 2))).astype(np.float32))
b.fit(int(os.environ.get("PMC_ITERS", "3")), e_rel=0, check_every=0)
torch.cuda.synchronize()
