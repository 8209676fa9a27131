import os, sys, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, warnings
warnings.filterwarnings("ignore")
from scarlet_amd import synth, fft as fftmod, _lib
from scarlet_amd.batch import BlendBatch
B, H, W, K = 5, 128, 128, 8
S = 64
obs_psfs = np.array([synth.gaussian_psf((41, 41), 1.2 + 0.15 * b) for b in range(B)])
model_psf = synth.gaussian_psf((41, 41), 0.9)
diff = np.asarray(fftmod.match_psfs(fftmod.Fourier(obs_psfs.astype(np.float32)), fftmod.Fourier(model_psf[None].astype(np.float32))).image, dtype=np.float32)
scale = (model_psf.max() / obs_psfs.max(axis=(1, 2))).astype(np.float32)
scenes = [synth.make_scene(300 + i, B=B, H=H, W=W, K=K, psfs=obs_psfs) for i in range(S)]
images = np.stack([s["images"] for s in scenes]); centers = np.stack([s["centers"] for s in scenes])
b = BlendBatch(images, centers, centroid_weight=model_psf.astype(np.float32), mse_capacity=20)
b.set_diff_kernel(diff)
b.init_extended(np.ones(B) * 0.1, sed_scale=scale)
st = _lib.stream_ptr()
_lib.check(_lib.lib.scarlet_backward_step(ctypes.byref(b._c), 0, st)); torch.cuda.synchronize()
cur = b.cur.cpu().numpy()
print("cur", np.unique(cur))
other = b.morph[1] if cur[0] == 0 else b.morph[0]
osed = b.sed[1] if cur[0] == 0 else b.sed[0]
stepped = other.clone(); ssed = osed.clone()
cen0 = b.centers.clone(); sh0 = b.shifts.clone()
print("stepped morph nan:", int(torch.isnan(stepped).sum()), "sed nan", int(torch.isnan(ssed).sum()))
ref = None
for trial in range(6):
    other.copy_(stepped); osed.copy_(ssed); b.centers.copy_(cen0); b.shifts.copy_(sh0); b.status.zero_()
    torch.cuda.synchronize()
    _lib.check(_lib.lib.scarlet_source_update(ctypes.byref(b._c), 1, st)); torch.cuda.synchronize()
    out = other.cpu().numpy()
    nanc = [(i, k) for i in range(S) for k in range(K) if np.isnan(out[i, k]).any()]
    if ref is None and not nanc: ref = out.copy()
    diffc = [] if ref is None else [(i, k) for i in range(S) for k in range(K) if not np.array_equal(out[i, k], ref[i, k], equal_nan=True)]
    print("trial", trial, "nan comps", nanc, "differs from clean run", diffc[:8], "status", np.nonzero(b.status.cpu().numpy())[0].tolist())
    for (i, k) in nanc[:2]:
        print("    comp", i, k, "nan count", int(np.isnan(out[i, k]).sum()), "center", b.centers[i, k].cpu().numpy().tolist(), "shift", b.shifts[i, k].cpu().numpy().tolist(),
              "stepped max", float(stepped[i, k].max()), "at center", float(stepped[i, k, cen0[i, k, 0], cen0[i, k, 1]]))
