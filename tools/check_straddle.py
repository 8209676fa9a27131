"""Is a scene that misses the 1e-5 tolerance in tools/parity_report.py a threshold straddle in the sense of
tests/parity_common.py (GPU and float32 oracle agree until a pixel whose float64 value sits on a <= 0 test within tolerance)?
    python tools/check_straddle.py <synthetic scene index> [iterations]
TEST INFRASTRUCTURE (imports oracle/ through tests/parity_common.py)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.zeros(1, device="cuda")            # (the device context exists before the library's first call)
import scarlet_amd as scarlet
import parity_common as pc
idx = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
wl = pc.Workload()
images, centers = wl.scenes(idx, 1)
ok, msg = pc.straddles_threshold(scarlet, wl, images[0], centers[0], iters)
print("scene %d: %s -- %s" % (idx, "threshold straddle" if ok else "NOT explained", msg))
