"""BASELINE config 5 timing (not the round's bench line): S scenes of 6 x 256 x 256, K = 30 overlapping sources,
symmetry + monotonicity + L0; gradient passes of bigk.h, constraints in place in HBM (k_source_update<2>).
    python tools/bench_c5.py [--scenes 64] [--steps 5]      (512 scenes over 8 GPUs = 64 per GPU)"""
import argparse, ctypes, json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from scarlet_amd import synth, _lib
from scarlet_amd.batch import BlendBatch

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=64)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--unique", type=int, default=8)
a = ap.parse_args()
B, H, W, K, S = 6, 256, 256, 30, a.scenes
t0 = time.perf_counter()
scenes = [synth.make_scene(5000 + i, B=B, H=H, W=W, K=K, min_sep=3) for i in range(a.unique)]
reps = (S + a.unique - 1) // a.unique
images = np.tile(np.stack([s["images"] for s in scenes]), (reps, 1, 1, 1))[:S]
centers = np.tile(np.stack([s["centers"] for s in scenes]), (reps, 1, 1))[:S]
print("host scene generation %.1f s" % (time.perf_counter() - t0), file=sys.stderr)
b = BlendBatch(images, centers, l0_thresh=0.05, mse_capacity=a.steps + a.warmup + 1)
t0 = time.perf_counter()
b.init_extended(np.ones(B) * 0.1)
torch.cuda.synchronize()
t_init = time.perf_counter() - t0
b.fit(a.warmup, e_rel=0, check_every=0)
torch.cuda.synchronize()
_lib.check(_lib.lib.scarlet_profile_begin(a.steps))
t0 = time.perf_counter()
b.fit(a.steps, e_rel=0, check_every=0)
torch.cuda.synchronize()
el = time.perf_counter() - t0
ms = (ctypes.c_double * 8)(); cnt = (ctypes.c_int64 * 8)()
_lib.check(_lib.lib.scarlet_profile_end(ms, cnt))
names = ["grad(resid+gram+lipschitz)", "step(step+sed)", "source_update", "converge", "iterate(fused)", "psf_chain", "6", "7"]
alg = 4 * H * W * (B + 2 * K) + 8 * K * B
print(json.dumps({"config": "c5: %d scenes 6x256x256 K=30 L0" % S, "init_s": t_init, "ms_per_iteration": 1e3 * el / a.steps,
                  "scene_iterations_per_s": S * a.steps / el, "algorithmic_GBps": alg * S * a.steps / el / 1e9,
                  "per_class_ms_per_iteration": {names[i]: ms[i] / a.steps for i in range(8) if cnt[i]},
                  "status_nonzero": int((b.status != 0).sum().item())}))
