"""Driver for rocprofv3 passes on BASELINE config 5's shape: 64 scenes 6 x 256 x 256, K = 30, L0; 4 iterations."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from scarlet_amd import synth
from scarlet_amd.batch import BlendBatch
S, B, H, W, K = 64, 6, 256, 256, 30
scenes = [synth.make_scene(5000 + i, B=B, H=H, W=W, K=K, min_sep=3) for i in range(16)]
images = np.tile(np.stack([s["images"] for s in scenes]), (4, 1, 1, 1))[:S]
centers = np.tile(np.stack([s["centers"] for s in scenes]), (4, 1, 1))[:S]
b = BlendBatch(images, centers, l0_thresh=0.05)
b.init_extended(np.ones(B) * 0.1)
b.fit(int(os.environ.get("PMC_ITERS", "4")), e_rel=0, check_every=0)
torch.cuda.synchronize()
