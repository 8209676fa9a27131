"""diagnostic 3: which iteration of a k_fit2x launch goes wrong first (both factor buffers are compared)"""
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
    cur = b.cur.cpu().numpy()
    m = [b.morph[0].cpu().numpy(), b.morph[1].cpu().numpy()]
    sd = [b.sed[0].cpu().numpy(), b.sed[1].cpu().numpy()]
    idx = np.arange(S)
    return dict(morph_cur=np.where(cur[:, None, None, None] == 0, m[0], m[1]), morph_last=np.where(cur[:, None, None, None] == 0, m[1], m[0]),
                sed_cur=np.where(cur[:, None, None] == 0, sd[0], sd[1]), sed_last=np.where(cur[:, None, None] == 0, sd[1], sd[0]),
                mse=b.mse_buf[:, :iters].cpu().numpy(), cen=b.centers.cpu().numpy(), lip=b.lipschitz.cpu().numpy())
S = int(os.environ.get("DIAG_S", "3000"))
for iters in [int(x) for x in os.environ.get("DIAG_ITERS", "1,2,3").split(",")]:
    ref, got = run(S, iters, True), run(S, iters, False)
    line = "S=%d iters=%d:" % (S, iters)
    for k in ref:
        a, c = ref[k].reshape(S, -1), got[k].reshape(S, -1)
        bad = ((a != c) & ~(np.isnan(a) & np.isnan(c))).any(axis=1)
        line += "  %s:%d" % (k, bad.sum())
        if k == "mse" and bad.any():
            first = [(int(s), int(np.nonzero(ref[k][s] != got[k][s])[0][0]) + 1) for s in np.nonzero(bad)[0][:10]]
            line += " first differing iteration per scene %s" % first
    print(line, flush=True)
