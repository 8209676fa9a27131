"""Small driver for rocprofv3 --pmc passes: 3 warm-up + 2 measured iterations on 10k scenes."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from scarlet_amd import synth
from scarlet_amd.batch import BlendBatch
S = int(os.environ.get("PMC_SCENES", "10000"))
d = synth.make_batch(0, 512)
reps = (S + 511) // 512
imgs = np.tile(d["images"], (reps, 1, 1, 1))[:S]; cen = np.tile(d["centers"], (reps, 1, 1))[:S]
kw = {}
if os.environ.get('PMC_NOSYM'): kw['symmetric'] = False
if os.environ.get('PMC_NOMONO'): kw['monotonic'] = False
b = BlendBatch(imgs, cen, **kw)
b.init_extended(np.ones(5) * .1)
# PMC_WARM iterations as single-iteration launches (k_iterate2: another kernel name), then ONE launch of PMC_ITERS
# iterations (k_fit2x) in the steady state -- the first four iterations of a fit run the flip symmetry, not the GEMMs
from scarlet_amd import _lib
warm, iters = int(os.environ.get("PMC_WARM", "10")), int(os.environ.get("PMC_ITERS", "10"))
if warm:
    _lib.set_option("NO_PERSIST", 1)
    b.fit(warm, e_rel=0, check_every=0)
    _lib.set_option("NO_PERSIST", int(os.environ.get("SCARLET_NO_PERSIST", "0") or 0))
b.fit(iters, e_rel=0, check_every=0)
torch.cuda.synchronize()
