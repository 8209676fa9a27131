"""A/B for shapes with 5 < B <= 8 bands, K <= 4 (k_iterate v1, which spills) against the general path."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from scarlet_amd import synth, _lib
from scarlet_amd.batch import BlendBatch
S = 4000
for B in (6, 8):
    d = synth.make_batch(0, 256, B=B)
    reps = (S + 255) // 256
    imgs = np.tile(d["images"], (reps, 1, 1, 1))[:S]; cen = np.tile(d["centers"], (reps, 1, 1))[:S]
    for nofused in (0, 1):
        _lib.set_option("NO_FUSED", nofused)
        b = BlendBatch(imgs, cen); b.init_extended(np.ones(B) * .1)
        b.fit(5, e_rel=0, check_every=0); torch.cuda.synchronize()
        t0 = time.perf_counter(); b.fit(20, e_rel=0, check_every=0); torch.cuda.synchronize()
        print("B=%d NO_FUSED=%d: %.3f ms per iteration of %d scenes" % (B, nofused, 1e3 * (time.perf_counter() - t0) / 20, S))
_lib.set_option("NO_FUSED", 0)
