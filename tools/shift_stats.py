import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from scarlet_amd import synth
from scarlet_amd.batch import BlendBatch
d = synth.make_batch(0, 512)
b = BlendBatch(d["images"], d["centers"])
b.init_extended(np.ones(5) * .1)
for it in (1, 4, 5, 10, 20, 30):
    b.fit(it - int(b.it[0].item()), e_rel=0, check_every=0)
    torch.cuda.synchronize()
    sh = b.shifts.cpu().numpy()
    a = np.abs(sh).max(axis=2)
    print("it", int(b.it[0].item()), "frac |shift|<1e-9: %.4f  <1e-6: %.4f  nan: %.4f  median %.2e  p99 %.2e  max %.3f" % (
        (a < 1e-9).mean(), (a < 1e-6).mean(), np.isnan(a).mean(), np.nanmedian(a), np.nanpercentile(a, 99), np.nanmax(a)))
