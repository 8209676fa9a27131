"""Phase stamps of k_source_update_box (SCARLET_STAMPS=1) on config 3's or config 5's shape: cycles per phase."""
import sys, os, ctypes
os.environ["SCARLET_STAMPS"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from scarlet_amd import synth, _lib
from scarlet_amd.batch import BlendBatch
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
if cfg == "c3":
    S, B, H, W, K, kw, first = 2048, 5, 128, 128, 8, {}, 300
else:
    S, B, H, W, K, kw, first = 64, 6, 256, 256, 30, dict(l0_thresh=0.05), 5000
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
    names = ["max_pixel(+centroid)", "box load + vectors", "GEMM1 (X through LDS)", "rank-1 z", "GEMM2 + epilogue", "sweep", "final pass"]
    tot = (st[ok, 7] - st[ok, 0]).mean()
    print("   total cycles per component %.0f" % tot)
    for i, nm in enumerate(names):
        d = (st[ok, i + 1] - st[ok, i]).mean()
        print("   %-24s %8.0f %5.1f%%" % (nm, d, 100 * d / tot))
