"""diagnostic: where does a 10 000-scene multi-iteration launch differ from per-iteration launches?"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from scarlet_amd import synth, _lib
from scarlet_amd.batch import BlendBatch
U, S = 256, int(os.environ.get("DIAG_S", "10000"))
d = synth.make_batch(4000, U)
reps = (S + U - 1) // U
images = np.tile(d["images"], (reps, 1, 1, 1))[:S]; centers = np.tile(d["centers"], (reps, 1, 1))[:S]
def run(img, cen, iters, per_iteration, chunk=0):
    _lib.set_option("NO_PERSIST", 1 if per_iteration else 0)
    b = BlendBatch(img, cen)
    b.init_extended(np.ones(5) * 0.1)
    b.fit(iters, e_rel=0, check_every=chunk)
    torch.cuda.synchronize()
    return dict(morph=b.morph_current.clone(), sed=b.sed_current.clone(), mse=b.mse_buf[:, :iters].clone(), cen=b.centers.clone(),
                sh=b.shifts.clone(), flags=b.flags.clone(), lip=b.lipschitz.clone())
for iters in (1, 2, 3, 4, 5, 6):
    ref = run(images, centers, iters, True)
    for label, kw in (("persist", dict()), ("persist chunk2", dict(chunk=2))):
        got = run(images, centers, iters, False, **kw)
        line = "iters %d %-15s" % (iters, label)
        for k in ref:
            a, c = ref[k].reshape(S, -1), got[k].reshape(S, -1)
            bad = ((a != c) & ~(torch.isnan(a) & torch.isnan(c))).any(dim=1)
            n = int(bad.sum())
            line += "  %s:%d" % (k, n)
            if n and k in ("morph", "mse"):
                idx = torch.nonzero(bad)[:8, 0].tolist()
                line += str(idx)
        print(line, flush=True)
