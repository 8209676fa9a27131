"""BASELINE config 3 timing (not the round's bench line): S scenes of 5 x 128 x 128, K = 8, per-band PSF 41 x 41
(FFT shape 180 x 180), general path = hipFFT convolution + gradient / step / constraint / convergence kernels.
    python tools/bench_c3.py [--scenes 4096] [--steps 10]"""
import argparse, ctypes, json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from scarlet_amd import synth, _lib, psf as psfmod, fft as fftmod
from scarlet_amd.batch import BlendBatch

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=4096)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--unique", type=int, default=64, help="distinct synthetic scenes (tiled to --scenes)")
a = ap.parse_args()
B, H, W, K, S = 5, 128, 128, 8, a.scenes
obs_psfs = np.array([synth.gaussian_psf((41, 41), 1.2 + 0.15 * b) for b in range(B)])
model_psf = synth.gaussian_psf((41, 41), 0.9)
diff = fftmod.match_psfs(fftmod.Fourier(obs_psfs.astype(np.float32)), fftmod.Fourier(model_psf[None].astype(np.float32))).image
t0 = time.perf_counter()
scenes = [synth.make_scene(300 + i, B=B, H=H, W=W, K=K, psfs=obs_psfs) for i in range(a.unique)]
reps = (S + a.unique - 1) // a.unique
images = np.tile(np.stack([s["images"] for s in scenes]), (reps, 1, 1, 1))[:S]
centers = np.tile(np.stack([s["centers"] for s in scenes]), (reps, 1, 1))[:S]
print("host scene generation %.1f s" % (time.perf_counter() - t0), file=sys.stderr)
b = BlendBatch(images, centers, centroid_weight=model_psf.astype(np.float32), mse_capacity=a.steps + a.warmup + 1)
b.set_diff_kernel(np.asarray(diff, dtype=np.float32))
scale = (model_psf.max() / obs_psfs.max(axis=(1, 2))).astype(np.float32)
b.init_extended(np.ones(B) * 0.1, sed_scale=scale)
b.fit(a.warmup, e_rel=0, check_every=0)
torch.cuda.synchronize()
_lib.check(_lib.lib.scarlet_profile_begin(a.steps))
t0 = time.perf_counter()
b.fit(a.steps, e_rel=0, check_every=0)
torch.cuda.synchronize()
el = time.perf_counter() - t0
ms = (ctypes.c_double * 8)(); cnt = (ctypes.c_int64 * 8)()
_lib.check(_lib.lib.scarlet_profile_end(ms, cnt))
names = ["grad", "step", "source_update", "converge", "iterate(fused)", "psf_chain", "6", "7"]
alg = 4 * H * W * (B + 2 * K) + 8 * K * B
print(json.dumps({"config": "c3: %d scenes 5x128x128 K=8 PSF 41x41" % S, "ms_per_iteration": 1e3 * el / a.steps,
                  "scene_iterations_per_s": S * a.steps / el,
                  "algorithmic_GBps": alg * S * a.steps / el / 1e9,
                  "per_class_ms_per_iteration": {names[i]: ms[i] / a.steps for i in range(8) if cnt[i]},
                  "status_nonzero": int((b.status != 0).sum().item())}))
