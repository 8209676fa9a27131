"""Row a5 pinned independently of the hand-written adjoint: the reference obtains d loss / d sed, d loss / d morph from
autograd over its forward model (blend.py:105-139, observation.py:198-239, fft.py:138-211, 304-317).  The oracle (and
the reference-run fixtures, whose autograd stand-in is the same formula: oracle/refshim.py) use the analytic adjoint
G = render^T(w^2 (render(M) - I)).  Here the FORWARD chain alone is restated in torch (float64: zero-pad with the
leading pad (dS + 1) // 2, ifftshift, rfftn, x K-hat, irfftn, fftshift, central crop from (cur - new + 1) // 2) and
torch.autograd differentiates it: the oracle's gradients must equal autograd's to float64 round-off -- with and without
a PSF, even- and odd-sized kernels, non-square frames, per-pixel weights with masked pixels."""
import numpy as np
import pytest
import torch

from oracle import pgm


def _pad_to(x, newshape):
    pads = []
    for ax in (2, 1):                                   # F.pad takes the last axis first
        d = newshape[ax - 1] - x.shape[ax]
        lead = (d + 1) // 2
        pads += [lead, d - lead]
    return torch.nn.functional.pad(x, pads)


def _centered(x, newshape):
    sl = [slice(None)]
    for ax in (1, 2):
        start = (x.shape[ax] - newshape[ax - 1] + 1) // 2
        sl.append(slice(start, start + newshape[ax - 1]))
    return x[tuple(sl)]


def _render_torch(model, diff_kernel):
    if diff_kernel is None:
        return model
    F = pgm.fft_shape(model.shape, diff_kernel.shape, 3, (1, 2))          # integer shape arithmetic only
    spec = torch.fft.rfftn(torch.fft.ifftshift(_pad_to(model, F), dim=(1, 2)), dim=(1, 2)) * \
        torch.fft.rfftn(torch.fft.ifftshift(_pad_to(diff_kernel, F), dim=(1, 2)), dim=(1, 2))
    img = torch.fft.fftshift(torch.fft.irfftn(spec, s=tuple(F), dim=(1, 2)), dim=(1, 2))
    return _centered(img, model.shape[1:])


@pytest.mark.parametrize("B,K,H,W,P", [(5, 4, 24, 24, None), (3, 3, 21, 30, (9, 9)), (4, 2, 32, 20, (8, 11)), (2, 5, 17, 17, (6, 6))])
@pytest.mark.parametrize("weighted", [False, True])
def test_oracle_gradients_equal_autograd_of_the_forward_model(B, K, H, W, P, weighted):
    rng = np.random.default_rng(B * 1000 + K * 100 + H)
    seds = rng.uniform(0.1, 2.0, size=(K, B))
    morphs = rng.uniform(0.0, 1.0, size=(K, H, W))
    images = rng.normal(size=(B, H, W))
    weights = rng.uniform(0.2, 2.0, size=(B, H, W)) if weighted else np.ones((B, H, W))
    if weighted:
        weights[rng.uniform(size=weights.shape) < 0.1] = 0.0               # masked pixels
    diff = None
    if P is not None:
        diff = rng.normal(size=(B,) + P) * 0.1
        diff[:, P[0] // 2, P[1] // 2] += 1.0
    loss, g_sed, g_morph = pgm.loss_and_gradients(list(seds), list(morphs), images, weights, diff)

    ts = torch.tensor(seds, dtype=torch.float64, requires_grad=True)
    tm = torch.tensor(morphs, dtype=torch.float64, requires_grad=True)
    model = torch.einsum("kb,kyx->byx", ts, tm)
    rendered = _render_torch(model, None if diff is None else torch.tensor(diff, dtype=torch.float64))
    d = torch.tensor(weights) * (rendered - torch.tensor(images))
    tl = 0.5 * (d ** 2).sum()
    ag_sed, ag_morph = torch.autograd.grad(tl, (ts, tm))

    def rel(a, b):
        return float(np.abs(a - b).max() / np.abs(b).max())
    assert abs(loss - float(tl)) <= 1e-12 * abs(float(tl))
    assert rel(np.array(g_sed), ag_sed.numpy()) < 1e-11
    assert rel(np.array(g_morph), ag_morph.numpy()) < 1e-11
