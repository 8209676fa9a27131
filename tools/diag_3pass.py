"""Which outputs of the three-pass PSF iteration differ from the four-pass form, and after how many iterations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import scarlet_amd as scarlet
from scarlet_amd import synth, fft as fftmod, _lib
B, H, W, S, K = 5, 128, 128, 3, 8
obs_psfs = np.array([synth.gaussian_psf((41, 41), 1.2 + 0.15 * b) for b in range(B)])
model_psf = synth.gaussian_psf((41, 41), 0.9)
diff = np.asarray(fftmod.match_psfs(fftmod.Fourier(obs_psfs.astype(np.float32)),
                                    fftmod.Fourier(model_psf[None].astype(np.float32))).image, dtype=np.float32)
scenes = [synth.make_scene(900 + i, B=B, H=H, W=W, K=K, psfs=obs_psfs) for i in range(S)]
for iters in (1, 2, 3):
    out = []
    for four in (1, 0):
        _lib.set_option("NO_PSF3PASS", four)
        b = scarlet.BlendBatch(np.stack([s["images"] for s in scenes]), np.stack([s["centers"] for s in scenes]),
                               centroid_weight=model_psf.astype(np.float32))
        b.set_diff_kernel(diff)
        b.init_extended(np.ones(B) * 0.1)
        b.fit(iters, e_rel=0)
        torch.cuda.synchronize()
        out.append(dict(morph=b.morph_current.cpu().numpy().copy(), sed=b.sed_current.cpu().numpy().copy(),
                        mse=np.array([b.mse(i) for i in range(S)]), L=b.lipschitz.cpu().numpy().copy()))
    for k in out[0]:
        x, y = out[0][k], out[1][k]
        print(iters, k, "differing:", int((x != y).sum()), "max abs", float(np.abs(x.astype(np.float64) - y).max()))
    print(out[0]["L"][0], out[1]["L"][0])
