"""Phase stamps of k_psf_conv (SCARLET_STAMPS=1): shader-clock cycles per pass, mean over planes."""
import sys, os
os.environ["SCARLET_STAMPS"] = "1"
os.environ["SCARLET_NO_PIPELINE"] = "1"      # (the stamp buffer is indexed by the batch's own scene numbers: one pipeline)
os.environ.setdefault("PMC_SCENES", "4096"); os.environ["PMC_ITERS"] = "3"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import runpy, numpy as np, torch
g = runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools", "pmc_run_c3.py"))
b = g["b"]
n = b.S * b.B * 32
ws = b.workspace
import ctypes
from scarlet_amd import _lib
off = int(_lib.lib.scarlet_debug_psf_stamps_offset(ctypes.byref(b._c)))
assert off >= 0, "no stamps region (SCARLET_STAMPS=1 must be set before the library is loaded)"
st = ws[off:off + n * 8].view(torch.int64).view(-1, 32).cpu().numpy()
ok = st[:, 15] > 0
st = st[ok]
names = ["start", "tables (+ model plane)", "rowsA (+ model load)", "rowsB", "colsA+unt", "colsB*K*Binv", "colsAinv+tangle", "rowsBinv",
         "rowsAinv+residual+rowsA", "rowsB", "colsA+unt", "colsB*K*Binv", "colsAinv+tangle", "rowsBinv", "rowsAinv+store G"]
seq = [30, 31, 0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13, 15]
prev = st[:, 30]
tot = (st[:, 15] - st[:, 30]).mean()
print("planes stamped:", len(st), " total cycles per plane: %.0f" % tot)
for nm, i in zip(names[1:], seq[1:]):
    d = (st[:, i] - prev).mean(); prev = st[:, i]
    print("%-24s %8.0f  %5.1f%%" % (nm, d, 100 * d / tot))
