"""diagnostic 2: nature of the differences of a 2-iteration k_fit2x launch against two k_iterate2 launches"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from scarlet_amd import synth, _lib
from scarlet_amd.batch import BlendBatch
U = 256
d = synth.make_batch(4000, U)
def run(S, iters, per_iteration):
    reps = (S + U - 1) // U
    img = np.tile(d["images"], (reps, 1, 1, 1))[:S]; cen = np.tile(d["centers"], (reps, 1, 1))[:S]
    _lib.set_option("NO_PERSIST", 1 if per_iteration else 0)
    b = BlendBatch(img, cen)
    b.init_extended(np.ones(5) * 0.1)
    b.fit(iters, e_rel=0, check_every=0)
    torch.cuda.synchronize()
    return b.morph_current.cpu().numpy(), b.sed_current.cpu().numpy(), b.mse_buf[:, :iters].cpu().numpy(), b.centers.cpu().numpy()
for S in [int(x) for x in os.environ.get("DIAG_SIZES", "256,512,700,10000").split(",")]:
    ref = run(S, 2, True)
    got = run(S, 2, False)
    bad = np.nonzero((ref[0] != got[0]).reshape(S, -1).any(axis=1))[0]
    print("S=%d: %d scenes differ in morph; mse differs in %d" % (S, len(bad), (ref[2] != got[2]).any(axis=1).sum()), flush=True)
    for s in bad[:6]:
        for k in range(4):
            df = ref[0][s, k] != got[0][s, k]
            if df.any():
                ys, xs = np.nonzero(df)
                cy, cx = got[3][s, k]
                print("  scene %d comp %d centre (%d,%d): %d px differ, max abs %.3e (max|ref| %.3e); rows %d..%d cols %d..%d; first (%d,%d): ref %.6e got %.6e; got==0: %d ref==0: %d"
                      % (s, k, cy, cx, df.sum(), np.abs(ref[0][s, k] - got[0][s, k]).max(), np.abs(ref[0][s, k]).max(), ys.min(), ys.max(), xs.min(), xs.max(),
                         ys[0], xs[0], ref[0][s, k][ys[0], xs[0]], got[0][s, k][ys[0], xs[0]], (got[0][s, k][df] == 0).sum(), (ref[0][s, k][df] == 0).sum()))
        print("  scene %d: mse ref %s got %s; sed differs: %s" % (s, ref[2][s], got[2][s], (ref[1][s] != got[1][s]).any()))
