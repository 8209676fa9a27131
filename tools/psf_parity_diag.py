"""Diagnostic (TEST INFRASTRUCTURE, imports oracle/): actual max-norm errors of the PSF path on BASELINE
config 1 against the f64 reference fixture, per iteration count, next to the f32 CPU oracle's own error."""
import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, rel_err
import torch
import scarlet_amd
from oracle import pgm

g = load_golden("fit_hsc"); d = load_golden("hsc_inputs")
tag = "f64"
def gpu_run(its):
    b = scarlet_amd.BlendBatch(d["images"][None], g["init_center_" + tag][None].astype(np.int32), centroid_weight=g["model_psf"][0])
    b.set_diff_kernel(g["diff_kernel"].astype(np.float32))
    b.set_state(g["init_sed_" + tag][None], g["init_morph_" + tag][None], shifts=g["init_shift_" + tag][None])
    b.fit(its, e_rel=0); torch.cuda.synchronize()
    return b.sed_current[0].cpu().numpy(), b.morph_current[0].cpu().numpy(), np.array(b.mse(0))
def cpu_run(its, dt):
    sc = pgm.scene_from_state(d["images"].astype(dt), g["init_sed_" + tag].astype(dt), g["init_morph_" + tag].astype(dt),
                              g["init_center_" + tag], g["init_shift_" + tag], diff_kernel=g["diff_kernel"].astype(dt),
                              centroid_weight=g["model_psf"][0])
    pgm.fit(sc, its, e_rel=0)
    return np.array([s.sed for s in sc.sources]), np.array([s.morph for s in sc.sources]), np.array(sc.mse)
for its in (1, 2, 5, 10, 20, 50):
    gs, gm, gmse = gpu_run(its)
    r64 = cpu_run(its, np.float64); r32 = cpu_run(its, np.float32)
    print("its %2d  GPU vs f64: sed %.2e morph %.2e mse %.2e | CPU f32 vs f64: sed %.2e morph %.2e mse %.2e" % (
        its, rel_err(gs, r64[0]), rel_err(gm, r64[1]), rel_err(gmse, r64[2]),
        rel_err(r32[0], r64[0]), rel_err(r32[1], r64[1]), rel_err(r32[2], r64[2])), flush=True)
    if its == 50:
        e = np.abs(gm - r64[1]) / np.abs(r64[1]).max()
        idx = np.argsort(e.ravel())[::-1][:8]
        for i in idx:
            k, y, x = np.unravel_index(i, e.shape)
            print("   worst px comp %d (%d,%d): gpu %.6e ref %.6e err %.2e  (peak at %s)" % (k, y, x, gm[k, y, x], r64[1][k, y, x], e[k, y, x], g["center_" + tag][k]))
        print("   per-iteration mse rel err:", np.array2string(np.abs(gmse - r64[2]) / np.abs(r64[2]), precision=1))
# single convolution precision
from scarlet_amd.psfconv import convolve_same
rng = np.random.RandomState(1)
img = rng.rand(5, 58, 48); ker = g["diff_kernel"]
ref = pgm.convolve(img, ker, axes=(1, 2))
print("single conv (58x48 * 43x43) rel err: gpu %.2e  cpu-f32 %.2e" % (
    rel_err(convolve_same(img, ker).cpu().numpy(), ref),
    rel_err(pgm.convolve(img.astype(np.float32), ker.astype(np.float32), axes=(1, 2)), ref)))
