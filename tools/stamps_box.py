"""Phase stamps of k_source_update_box (SCARLET_STAMPS=1) on config 3's or config 5's shape: cycles per phase."""
import sys, os, ctypes
os.environ["SCARLET_STAMPS"] = "1"
os.environ["SCARLET_NO_PIPELINE"] = "1"      # (the stamp buffer is indexed by the batch's own scene numbers: one pipeline)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from scarlet_amd import synth, _lib
from scarlet_amd.batch import BlendBatch
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
if cfg in ("c3", "c3psf"):
    S, B, H, W, K, kw, first = 2048, 5, 128, 128, 8, {}, 300
else:
    S, B, H, W, K, kw, first = 64, 6, 256, 256, 30, dict(l0_thresh=0.05), 5000
if cfg == "c3psf":
    # the bench's config-3 workload (PSF-convolved scenes, difference kernels), 1024 distinct scenes
    from scarlet_amd import fft as fftmod
    S, B, H, W, K, kw = 1024, 5, 128, 128, 8, {}
    obs = np.array([synth.gaussian_psf((41, 41), 1.2 + 0.15 * b) for b in range(B)])
    model = synth.gaussian_psf((41, 41), 0.9)
    d = synth.make_batch(0, S, B=B, H=H, W=W, K=K, psfs=obs)
    b = BlendBatch(d["images"], d["centers"], centroid_weight=model.astype(np.float32))
    diff = fftmod.match_psfs(fftmod.Fourier(obs.astype(np.float32)), fftmod.Fourier(model[None].astype(np.float32))).image
    b.set_diff_kernel(np.asarray(diff, dtype=np.float32))
    b.init_extended(np.ones(B) * 0.1, sed_scale=(model.max() / obs.max(axis=(1, 2))).astype(np.float32))
else:
    scenes = [synth.make_scene(first + i, B=B, H=H, W=W, K=K, min_sep=3 if cfg == "c5" else 4) for i in range(16)]
    reps = (S + 15) // 16
    images = np.tile(np.stack([s["images"] for s in scenes]), (reps, 1, 1, 1))[:S]
    centers = np.tile(np.stack([s["centers"] for s in scenes]), (reps, 1, 1))[:S]
    b = BlendBatch(images, centers, **kw)
    b.init_extended(np.ones(B) * 0.1)
for its in (3, 8):
    b.fit(its, e_rel=0, check_every=0)
    torch.cuda.synchronize()
    n = S * K * 16
    buf = (ctypes.c_int64 * n)()
    got = _lib.lib.scarlet_debug_stamps(buf, n)
    st = np.frombuffer(buf, dtype=np.int64).reshape(-1, 16)
    ok = st[:, 7] > 0
    fb = (st[:, 6] > 0) & (st[:, 7] == 0)
    print("after %d more iterations: %d components, completed in the box %d, left to the full path %d" % (its, len(st), ok.sum(), fb.sum()))
    ls = st[ok, 8]
    print("   stop level percentiles 10/50/90/99/max:", np.percentile(ls, [10, 50, 90, 99, 100]))
    big = ok & (st[:, 9] == 63)
    print("   completed in the 127 x 127 box: %d (%.1f %%), their stop levels 10/50/90/max: %s, cycles per component %.0f"
          % (big.sum(), 100.0 * big.sum() / len(st), np.percentile(st[big, 8], [10, 50, 90, 100]) if big.any() else "-",
             (st[big, 7] - st[big, 0]).mean() if big.any() else 0))
    if big.any():
        dur = st[big, 7] - st[big, 0]
        # (the cycle counters of the eight XCDs are not synchronised: only durations inside a workgroup mean anything)
        print("      127-box per-component cycles 50/90/99/max: %s" % np.percentile(dur, [50, 90, 99, 100]).astype(int))
        for i, nm in enumerate(["max_pixel(+centroid)", "box load + vectors", "GEMM1 (X through LDS)", "rank-1 z", "GEMM2 + epilogue", "sweep", "final pass"]):
            print("      127-box %-24s %8.0f" % (nm, (st[big, i + 1] - st[big, i]).mean()))
    ok = ok & (st[:, 9] != 63)
    ls = st[ok, 8]
    names = ["max_pixel(+centroid)", "box load + vectors", "GEMM1 (X through LDS)", "rank-1 z", "GEMM2 + epilogue", "sweep", "final pass"]
    tot = (st[ok, 7] - st[ok, 0]).mean()
    print("   total cycles per component %.0f" % tot)
    for i, nm in enumerate(names):
        dd = st[ok, i + 1] - st[ok, i]
        dd = dd[(dd >= 0) & (dd < 10 ** 7)]         # (a component that skips a phase keeps an older launch's stamp there)
        d = dd.mean() if len(dd) else 0.0
        print("   %-24s %8.0f %5.1f%%" % (nm, d, 100 * d / tot))
