"""-m gpu: the PSF (FFT-convolution) path, row a3b: render, its adjoint inside the fit, and
BASELINE config 1 (hsc_cosmos_35, 5x58x48, PSF 43x43, K=2, 50 iterations)."""
import numpy as np
import pytest
from numpy.testing import assert_almost_equal, assert_array_equal

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def scarlet():
    import scarlet_amd
    scarlet_amd._lib.require_gpu()
    return scarlet_amd


def npy(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


def test_convolve_same_matches_reference(scarlet):
    from scarlet_amd.psfconv import convolve_same
    g = load_golden("fft")
    out = npy(convolve_same(g["img"], g["ker"]))
    assert rel_err(out, g["conv"]) < 1e-5
    # narrow -> wide PSF round trip (reference tests/test_fft.py:73-90)
    out = npy(convolve_same(g["psf1"][None], g["k12"][None]))[0]
    assert_almost_equal(out, g["psf2"], decimal=6)
    # odd/even shape mix against the oracle
    from oracle import pgm
    rng = np.random.RandomState(0)
    for shape, ks in (((2, 31, 55), (2, 41, 41)), ((3, 58, 48), (1, 43, 43)), ((1, 64, 64), (1, 8, 3)), ((2, 33, 20), (2, 6, 9))):
        img = rng.rand(*shape); ker = rng.rand(*ks)
        ref = pgm.convolve(img, np.broadcast_to(ker, (shape[0],) + ks[1:]), axes=(1, 2))
        assert rel_err(npy(convolve_same(img, ker)), ref) < 1e-5, (shape, ks)


def test_gradient_step_with_psf_matches_oracle(scarlet):
    """one backward+step with a difference kernel and per-pixel weights vs the CPU oracle"""
    from oracle import pgm
    g = load_golden("grad")
    images = g["images"].astype(np.float32); weights = g["weights"].astype(np.float32)
    B, H, W = images.shape
    K = 2
    diff = g["diff_psf"].astype(np.float32)
    b = scarlet.BlendBatch(images[None], np.array([[[10, 12], [5, 6]]], dtype=np.int32), weights=weights[None],
                           symmetric=False, monotonic=False)
    b.set_diff_kernel(diff)
    b.set_state(g["seds"][None], g["morphs"][None])
    import ctypes
    from scarlet_amd import _lib
    _lib.check(_lib.lib.scarlet_backward_step(ctypes.byref(b._c), 0, _lib.stream_ptr()))
    torch.cuda.synchronize()
    seds = [s.astype(np.float32) for s in g["seds"]]; morphs = [m.astype(np.float32) for m in g["morphs"]]
    loss, gs, gm = pgm.loss_and_gradients(seds, morphs, images, weights, diff)
    L_sed, L_morph = pgm.lipschitz(seds, morphs)
    assert rel_err(b.lipschitz[0].cpu().numpy(), [L_sed, L_morph]) < 1e-5
    assert abs(float(b.mse_buf[0, 0].item()) - loss) < 1e-5 * abs(loss)
    want_sed = np.array(seds) - np.array(gs) / L_sed
    want_morph = np.array(morphs) - np.array(gm) / L_morph
    assert rel_err(b.sed[1][0].cpu().numpy(), want_sed) < 1e-5
    assert rel_err(b.morph[1][0].cpu().numpy(), want_morph) < 1e-5


@pytest.mark.parametrize("tag,tol", [("f32", 1e-5), ("f64", 1e-5)])
def test_config1_hsc_fifty_iterations(scarlet, tag, tol):
    """BASELINE config 1 through the batched engine, from the reference's initial state."""
    g = load_golden("fit_hsc")
    d = load_golden("hsc_inputs")
    b = scarlet.BlendBatch(d["images"][None], g["init_center_" + tag][None].astype(np.int32),
                           centroid_weight=g["model_psf"][0])
    b.set_diff_kernel(g["diff_kernel"].astype(np.float32))
    b.set_state(g["init_sed_" + tag][None], g["init_morph_" + tag][None], shifts=g["init_shift_" + tag][None])
    b.fit(50, e_rel=0)
    torch.cuda.synchronize()
    assert_array_equal(b.centers[0].cpu().numpy(), g["center_" + tag])
    assert rel_err(b.mse(0), g["mse_" + tag]) < tol
    assert rel_err(b.sed_current[0].cpu().numpy(), g["sed_" + tag]) < tol
    assert rel_err(b.morph_current[0].cpu().numpy(), g["morph_" + tag]) < tol
    assert int(b.status.abs().sum().item()) == 0


def test_config1_through_the_scarlet_api(scarlet):
    """docs/quickstart.ipynb cells 3-14 with our imports: Frame / Observation.match /
    ExtendedSource / Blend.fit on the hsc_cosmos_35 data."""
    g = load_golden("fit_hsc")
    d = load_golden("hsc_inputs")
    images = d["images"]
    psfs = g["obs_psfs"].astype(np.float32)
    frame = scarlet.Frame(images.shape, psfs=g["model_psf"].astype(np.float32))
    obs = scarlet.Observation(images, psfs=psfs).match(frame)
    assert rel_err(obs._diff_kernels.image, g["diff_kernel"]) < 1e-5
    bg = np.ones(5) * 0.1
    srcs = [scarlet.ExtendedSource(frame, tuple(int(v) for v in p), obs, bg) for p in g["pixels"]]
    assert rel_err(np.array([npy(s.morph) for s in srcs]), g["init_morph_f32"]) < 1e-5
    assert rel_err(np.array([npy(s.sed) for s in srcs]), g["init_sed_f32"]) < 1e-5
    blend = scarlet.Blend(srcs, obs).fit(50, e_rel=0)
    assert blend.it == 50
    assert rel_err(blend.mse, g["mse_f32"]) < 1e-5
    assert rel_err(np.array([npy(c.morph) for c in blend.components]), g["morph_f32"]) < 1e-5
    assert_array_equal(np.array([c.pixel_center for c in blend.components]), g["center_f32"])
    model = obs.render(blend.get_model())
    assert model.shape == images.shape


def init_data(scarlet, shape, coords, amplitudes, dtype=np.float32):
    """reference tests/test_blend.py:7-53 (data only)"""
    import scipy.signal
    B, Ny, Nx = shape
    K = len(coords)
    _seds = [np.arange(B, dtype=dtype), np.arange(B, dtype=dtype)[::-1], np.ones((B,), dtype=dtype)]
    seds = np.array([_seds[n % 3] * amplitudes[n] for n in range(K)])
    morphs = np.zeros((K, Ny, Nx))
    for k, coord in enumerate(coords):
        morphs[k, coord[0], coord[1]] = 1
    images = seds.T.dot(morphs.reshape(K, -1)).reshape(shape)
    psf_shape = (41, 41)
    target_psf = scarlet.psf.generate_psf_image(scarlet.psf.gaussian, psf_shape, sigma=.9).image
    target_psf /= target_psf.sum()
    psfs = np.array([scarlet.psf.generate_psf_image(scarlet.psf.gaussian, psf_shape, sigma=1 + .2 * b).image
                     for b in range(B)], dtype=dtype)
    psfs /= psfs.max(axis=(1, 2))[:, None, None]
    images = np.array([scipy.signal.convolve(img, psf, method="direct", mode="same")
                       for img, psf in zip(images, psfs)], dtype=dtype)
    psfs /= psfs.sum(axis=(1, 2))[:, None, None]
    return target_psf, psfs, images, seds


def test_reference_blend_tests(scarlet):
    """reference tests/test_blend.py:56-122 (float32 engine: tolerances adapted)"""
    shape = (6, 31, 55)
    coords = [(20, 10), (10, 30), (17, 42)]
    target_psf, psfs, images, seds = init_data(scarlet, shape, coords, [3, 2, 1])
    frame = scarlet.Frame(images.shape, psfs=target_psf[None].astype(np.float32))
    obs = scarlet.Observation(images, psfs=psfs).match(frame)
    sources = [scarlet.PointSource(frame, coord, obs) for coord in coords]
    blend = scarlet.Blend(sources, obs)
    model = npy(obs.render(blend.get_model()))
    assert_almost_equal(images, model, decimal=4)
    blend.fit(10)
    assert blend.it == 2
    assert max(blend.mse) < 1e-8
    # ExtendedSource (tests/test_blend.py:96-122)
    bg_rms = np.ones((6,))
    sources = [scarlet.ExtendedSource(frame, coord, obs, bg_rms) for coord in coords]
    blend = scarlet.Blend(sources, obs)
    psf_scale = obs.frame.psfs.max(axis=(1, 2)) / frame.psfs[0].max()
    scaled_seds = np.array([npy(c.sed) * psf_scale for c in blend.components])
    assert_almost_equal(scaled_seds, seds, decimal=4)
    blend.fit(100)
    assert blend.it < 20
    mse = np.array(blend.mse)
    assert np.all(mse[:-1] - mse[1:] >= -1e-9)


def test_reference_blend_tests_with_their_float64_frames(scarlet, caplog):
    """reference tests/test_blend.py:63, 83, 104 build ``Frame(..., dtype=np.float64)`` and float64 data: ported
    user code must construct and run (in float32, after ONE warning) instead of stopping at the first Component."""
    import logging
    from scarlet_amd import component
    shape = (6, 31, 55)
    coords = [(20, 10), (10, 30), (17, 42)]
    target_psf, psfs, images, seds = init_data(scarlet, shape, coords, [3, 2, 1], dtype=np.float64)
    component._warned_float64_frame = False
    with caplog.at_level(logging.WARNING, logger="scarlet_amd.component"):
        frame = scarlet.Frame(images.shape, psfs=target_psf[None], dtype=np.float64)
        obs = scarlet.Observation(images, psfs=psfs).match(frame)
        sources = [scarlet.PointSource(frame, coord, obs) for coord in coords]
        blend = scarlet.Blend(sources, obs)
    assert sum("float32" in r.getMessage() for r in caplog.records if r.name == "scarlet_amd.component") == 1
    assert_almost_equal(images, npy(obs.render(blend.get_model())), decimal=4)
    blend.fit(10)
    assert blend.it == 2
    assert max(blend.mse) < 1e-8
    sources = [scarlet.ExtendedSource(frame, coord, obs, np.ones((6,))) for coord in coords]
    blend = scarlet.Blend(sources, obs)
    psf_scale = obs.frame.psfs.max(axis=(1, 2)) / frame.psfs[0].max()
    assert_almost_equal(np.array([npy(c.sed) * psf_scale for c in blend.components]), seds, decimal=4)
    blend.fit(100)
    assert blend.it < 20
    mse = np.array(blend.mse)
    assert np.all(mse[:-1] - mse[1:] >= -1e-9)
    # the opt-in strict mode refuses the frame
    component.STRICT_FLOAT32_FRAME = True
    try:
        with pytest.raises(TypeError, match="float32"):
            scarlet.PointSource(frame, coords[0], obs)
    finally:
        component.STRICT_FLOAT32_FRAME = False


def test_config3_shape_128_psf_k8_vs_oracle(scarlet):
    """BASELINE config 3 shape (5x128x128, per-band PSF 41x41, 8 sources) at a 2-scene batch:
    general path = hipFFT convolution + 8-component gradient kernels + workgroup-level
    constraint kernel (H > 64).  6 iterations against the CPU oracle."""
    from oracle import pgm
    from scarlet_amd import synth
    B, H, W, K = 5, 128, 128, 8
    obs_psfs = np.array([synth.gaussian_psf((41, 41), 1.2 + 0.15 * b) for b in range(B)])
    model_psf = synth.gaussian_psf((41, 41), 0.9)
    diff = pgm.match_psfs(obs_psfs.astype(np.float32), model_psf[None].astype(np.float32))
    scenes = [synth.make_scene(300 + i, B=B, H=H, W=W, K=K, psfs=obs_psfs) for i in range(2)]
    b = scarlet.BlendBatch(np.stack([s["images"] for s in scenes]), np.stack([s["centers"] for s in scenes]),
                           centroid_weight=model_psf.astype(np.float32))
    b.set_diff_kernel(diff)
    scale = (model_psf.max() / obs_psfs.max(axis=(1, 2))).astype(np.float32)
    b.init_extended(np.ones(B) * 0.1, sed_scale=scale)
    sed0 = b.sed_current.cpu().numpy(); morph0 = b.morph_current.cpu().numpy()
    cen0 = b.centers.cpu().numpy(); sh0 = b.shifts.cpu().numpy()
    iters = 6
    b.fit(iters, e_rel=0)
    torch.cuda.synchronize()
    assert int(b.status.abs().sum().item()) == 0
    worst = 0
    for i in range(2):
        sc = pgm.scene_from_state(scenes[i]["images"], sed0[i], morph0[i], cen0[i], sh0[i],
                                  diff_kernel=diff, centroid_weight=model_psf.astype(np.float32))
        pgm.fit(sc, iters, e_rel=0)
        assert_array_equal(b.centers[i].cpu().numpy(), np.array([s.center for s in sc.sources]))
        worst = max(worst, rel_err(b.morph_current[i].cpu().numpy(), np.array([s.morph for s in sc.sources])),
                    rel_err(b.sed_current[i].cpu().numpy(), np.array([s.sed for s in sc.sources])),
                    rel_err(b.mse(i), sc.mse))
    assert worst < 1e-5, worst


def test_config3_batch_equals_single_scene_runs(scarlet):
    """Scenes are independent: a 12-scene batch of the config-3 shape must reproduce, bit for bit, the
    twelve one-scene runs (regression: an uninitialised LDS entry read by the PSF step kernel turned
    random components into NaN in large batches only)."""
    from scarlet_amd import synth, fft as fftmod
    B, H, W, K, S = 5, 128, 128, 8, 12
    obs_psfs = np.array([synth.gaussian_psf((41, 41), 1.2 + 0.15 * b) for b in range(B)])
    model_psf = synth.gaussian_psf((41, 41), 0.9)
    diff = np.asarray(fftmod.match_psfs(fftmod.Fourier(obs_psfs.astype(np.float32)),
                                        fftmod.Fourier(model_psf[None].astype(np.float32))).image, dtype=np.float32)
    scale = (model_psf.max() / obs_psfs.max(axis=(1, 2))).astype(np.float32)
    scenes = [synth.make_scene(306 + i, B=B, H=H, W=W, K=K, psfs=obs_psfs) for i in range(S)]
    images = np.stack([s["images"] for s in scenes]); centers = np.stack([s["centers"] for s in scenes])

    def run(img, cen):
        b = scarlet.BlendBatch(img, cen, centroid_weight=model_psf.astype(np.float32))
        b.set_diff_kernel(diff)
        b.init_extended(np.ones(B) * 0.1, sed_scale=scale)
        b.fit(3, e_rel=0)
        torch.cuda.synchronize()
        return npy(b.morph_current), npy(b.sed_current), npy(b.status), npy(b.centers)

    m, s, st, c = run(images, centers)
    assert not st.any() and np.isfinite(m).all() and np.isfinite(s).all()
    for i in range(S):
        mi, si, sti, ci = run(images[i:i + 1], centers[i:i + 1])
        assert_array_equal(m[i], mi[0]); assert_array_equal(s[i], si[0]); assert_array_equal(c[i], ci[0])


def test_large_frame_with_psf_vs_oracle(scarlet):
    """PSF path on a frame beyond the LDS tile (3 x 160 x 144, kernel 21 x 21): FFT convolution at
    the 7-smooth length, gradient kernels, constraints in place in HBM; 3 iterations vs the oracle."""
    from oracle import pgm
    from scarlet_amd import synth, fft as fftmod
    B, H, W, K = 3, 160, 144, 3
    obs_psfs = np.array([synth.gaussian_psf((21, 21), 1.3 + 0.2 * b) for b in range(B)])
    model_psf = synth.gaussian_psf((21, 21), 0.9)
    diff = np.asarray(fftmod.match_psfs(fftmod.Fourier(obs_psfs.astype(np.float32)),
                                        fftmod.Fourier(model_psf[None].astype(np.float32))).image, dtype=np.float32)
    scn = synth.make_scene(2100, B=B, H=H, W=W, K=K, psfs=obs_psfs)
    init = pgm.make_extended_scene(scn["images"], scn["centers"], np.ones(B) * 0.1,
                                   obs_psfs=obs_psfs.astype(np.float32), frame_psf=model_psf[None].astype(np.float32))
    b = scarlet.BlendBatch(scn["images"][None], scn["centers"][None], centroid_weight=model_psf.astype(np.float32))
    b.set_diff_kernel(diff)
    b.set_state(np.array([[c.sed for c in init.sources]]), np.array([[c.morph for c in init.sources]]),
                centers=np.array([[c.center for c in init.sources]]), shifts=np.array([[c.shift for c in init.sources]]))
    b.fit(3, e_rel=0)
    torch.cuda.synchronize()
    assert int(b.status.abs().sum().item()) == 0
    pgm.fit(init, 3, e_rel=0)
    assert_array_equal(npy(b.centers[0]), np.array([s.center for s in init.sources]))
    assert rel_err(npy(b.morph_current[0]), np.array([s.morph for s in init.sources])) < 1e-5
    assert rel_err(npy(b.sed_current[0]), np.array([s.sed for s in init.sources])) < 1e-5
    assert rel_err(b.mse(0), init.mse) < 1e-5


def test_many_components_with_psf_vs_oracle(scarlet):
    """K > 8 with a PSF: the adjoint render G is cropped out of the FFT buffers once and the chunked
    gradient passes of bigk.h take over.  3 bands, 10 sources, 64 x 64, kernel 11 x 11, 4 iterations."""
    from oracle import pgm
    from scarlet_amd import synth, fft as fftmod
    B, H, W, K = 3, 64, 64, 10
    obs_psfs = np.array([synth.gaussian_psf((11, 11), 1.2 + 0.2 * b) for b in range(B)])
    model_psf = synth.gaussian_psf((11, 11), 0.9)
    diff = np.asarray(fftmod.match_psfs(fftmod.Fourier(obs_psfs.astype(np.float32)),
                                        fftmod.Fourier(model_psf[None].astype(np.float32))).image, dtype=np.float32)
    scn = synth.make_scene(2200, B=B, H=H, W=W, K=K, psfs=obs_psfs)
    b = scarlet.BlendBatch(scn["images"][None], scn["centers"][None], centroid_weight=model_psf.astype(np.float32))
    b.set_diff_kernel(diff)
    scale = (model_psf.max() / obs_psfs.max(axis=(1, 2))).astype(np.float32)
    b.init_extended(np.ones(B) * 0.1, sed_scale=scale)
    sed0 = npy(b.sed_current)[0]; morph0 = npy(b.morph_current)[0]; cen0 = npy(b.centers)[0]; sh0 = npy(b.shifts)[0]
    b.fit(4, e_rel=0)
    torch.cuda.synchronize()
    assert int(b.status.abs().sum().item()) == 0
    sc = pgm.scene_from_state(scn["images"], sed0, morph0, cen0, sh0, diff_kernel=diff,
                              centroid_weight=model_psf.astype(np.float32))
    pgm.fit(sc, 4, e_rel=0)
    assert_array_equal(npy(b.centers[0]), np.array([s.center for s in sc.sources]))
    assert rel_err(npy(b.morph_current[0]), np.array([s.morph for s in sc.sources])) < 1e-5
    assert rel_err(npy(b.sed_current[0]), np.array([s.sed for s in sc.sources])) < 1e-5
    assert rel_err(b.mse(0), sc.mse) < 1e-5


def test_match_psfs_on_device_and_per_scene_kernels(scarlet):
    """fft.match_psfs batched on the device (against the host function, which is
    pinned on the reference's fixture), and a batch whose scenes each have their own PSFs: every scene
    must equal its own single-scene run with that scene's kernels."""
    from scarlet_amd import synth, fft as fftmod
    B, H, W, K, S = 3, 48, 56, 3, 3
    model_psf = synth.gaussian_psf((15, 15), 0.9).astype(np.float32)
    psfs = np.array([[synth.gaussian_psf((15, 15), 1.2 + 0.15 * b + 0.1 * s) for b in range(B)] for s in range(S)],
                    dtype=np.float32)
    dev = scarlet.fft.match_psfs_device(psfs, model_psf[None])
    assert tuple(dev.shape) == (S, B, 15, 15)
    for s in range(S):
        host = fftmod.match_psfs(fftmod.Fourier(psfs[s]), fftmod.Fourier(model_psf[None])).image
        assert rel_err(npy(dev[s]), host) < 1e-5
    # a different kernel size for psf2, one psf2 per psf1
    p2 = np.array([synth.gaussian_psf((11, 13), 0.8 + 0.05 * b) for b in range(B)], dtype=np.float32)
    host = fftmod.match_psfs(fftmod.Fourier(psfs[0]), fftmod.Fourier(p2)).image
    assert rel_err(npy(scarlet.fft.match_psfs_device(psfs[0], p2)), host) < 1e-5

    scenes = [synth.make_scene(2300 + s, B=B, H=H, W=W, K=K, psfs=psfs[s].astype(np.float64)) for s in range(S)]
    images = np.stack([sc["images"] for sc in scenes]); centers = np.stack([sc["centers"] for sc in scenes])

    def run(img, cen, kern):
        b = scarlet.BlendBatch(img, cen, centroid_weight=model_psf)
        b.set_diff_kernel(kern)
        b.init_extended(np.ones(B) * 0.1)
        b.fit(4, e_rel=0)
        torch.cuda.synchronize()
        assert int(b.status.abs().sum().item()) == 0
        return npy(b.morph_current), npy(b.sed_current)

    m, sd = run(images, centers, dev)
    for s in range(S):
        ms, ss = run(images[s:s + 1], centers[s:s + 1], dev[s])
        assert_array_equal(m[s], ms[0]); assert_array_equal(sd[s], ss[0])
    # and the kernels matter: scene 1 with scene 0's kernels gives a different answer
    m_wrong, _ = run(images[1:2], centers[1:2], dev[0])
    assert rel_err(m_wrong[0], m[1]) > 1e-4


def test_config3_full_size_properties(scarlet):
    """BASELINE config 3 at its full batch (4096 scenes of 5 x 128 x 128, 8 sources, PSF 41 x 41):
    32 distinct scenes tiled to 4096 -- every copy bit-identical wherever it sits, identical to the
    32-scene run, no status bits, loss decreasing."""
    from scarlet_amd import synth, fft as fftmod
    B, H, W, K, U, S, iters = 5, 128, 128, 8, 32, 4096, 3
    obs_psfs = np.array([synth.gaussian_psf((41, 41), 1.2 + 0.15 * b) for b in range(B)])
    model_psf = synth.gaussian_psf((41, 41), 0.9)
    diff = np.asarray(fftmod.match_psfs(fftmod.Fourier(obs_psfs.astype(np.float32)),
                                        fftmod.Fourier(model_psf[None].astype(np.float32))).image, dtype=np.float32)
    scale = (model_psf.max() / obs_psfs.max(axis=(1, 2))).astype(np.float32)
    scenes = [synth.make_scene(300 + i, B=B, H=H, W=W, K=K, psfs=obs_psfs) for i in range(U)]
    ui = np.stack([s["images"] for s in scenes]); uc = np.stack([s["centers"] for s in scenes])

    def run(img, cen):
        b = scarlet.BlendBatch(img, cen, centroid_weight=model_psf.astype(np.float32))
        b.set_diff_kernel(diff)
        b.init_extended(np.ones(B) * 0.1, sed_scale=scale)
        b.fit(iters, e_rel=0)
        torch.cuda.synchronize()
        assert int(b.status.abs().sum().item()) == 0
        return b.morph_current, b.sed_current, b.mse_buf[:, :iters].clone()

    reps = S // U
    m, s, mse = run(np.tile(ui, (reps, 1, 1, 1)), np.tile(uc, (reps, 1, 1)))
    ms, ss, mses = run(ui, uc)
    idx = torch.arange(S, device=m.device) % U
    assert torch.equal(m, ms[idx]) and torch.equal(s, ss[idx]) and torch.equal(mse, mses[idx])
    assert bool((mse[:, -1] < mse[:, 0]).all())


@pytest.mark.parametrize("K,approx", [(8, False), (3, False), (8, True)])
def test_three_pass_iteration_equals_four_pass(scarlet, K, approx):
    """The PSF iteration without the separate reduction pass (model + Gram, convolution, step + SED-gradient
    sums, scalar head: psf_path.h) keeps every sum's order; what differs from the four-pass form is the
    compiler's choice of fused / unfused multiply-adds in the two sets of kernels (measured: one SED entry off
    by one ulp after the first iteration): same iteration counts, values equal to 2e-6 of the array's maximum."""
    from scarlet_amd import synth, fft as fftmod, _lib
    B, H, W, S = 5, 128, 128, 3
    obs_psfs = np.array([synth.gaussian_psf((41, 41), 1.2 + 0.15 * b) for b in range(B)])
    model_psf = synth.gaussian_psf((41, 41), 0.9)
    diff = np.asarray(fftmod.match_psfs(fftmod.Fourier(obs_psfs.astype(np.float32)),
                                        fftmod.Fourier(model_psf[None].astype(np.float32))).image, dtype=np.float32)
    scenes = [synth.make_scene(900 + i, B=B, H=H, W=W, K=K, psfs=obs_psfs) for i in range(S)]
    out = []
    for four in (1, 0):
        _lib.set_option("NO_PSF3PASS", four)
        try:
            b = scarlet.BlendBatch(np.stack([s["images"] for s in scenes]), np.stack([s["centers"] for s in scenes]),
                                   centroid_weight=model_psf.astype(np.float32))
            b.set_diff_kernel(diff)
            b.init_extended(np.ones(B) * 0.1)
            b.fit(7, e_rel=1e-3, approximate_L=approx)
            torch.cuda.synchronize()
            assert int(b.status.abs().sum().item()) == 0
            out.append((b.morph_current.cpu().numpy().copy(), b.sed_current.cpu().numpy().copy(),
                        np.array([b.mse(i) for i in range(S)]), b.lipschitz.cpu().numpy().copy(),
                        b.it.cpu().numpy().copy()))
        finally:
            _lib.set_option("NO_PSF3PASS", 0)
    assert_array_equal(out[0][4], out[1][4])
    for x, y in zip(out[0][:4], out[1][:4]):
        assert rel_err(x, y) < 2e-6


def test_two_stream_pipeline_equals_single_stream(scarlet):
    """scarlet_fit on a large PSF batch runs its two halves as two pipelines on two streams (views of the batch with
    their own workspace regions and K-hat copies): scenes are independent, so every result must be bit-identical to
    the single-stream run -- including ragged convergence (e_rel = 1e-2 with a host check every 3 iterations)."""
    from scarlet_amd import synth, fft as fftmod, _lib
    B, H, W, K, S = 3, 32, 32, 2, 1032
    obs_psfs = np.array([synth.gaussian_psf((9, 9), 1.2 + 0.15 * b) for b in range(B)])
    model_psf = synth.gaussian_psf((9, 9), 0.9)
    diff = np.asarray(fftmod.match_psfs(fftmod.Fourier(obs_psfs.astype(np.float32)),
                                        fftmod.Fourier(model_psf[None].astype(np.float32))).image, dtype=np.float32)
    d = synth.make_batch(40, S, B=B, H=H, W=W, K=K, psfs=obs_psfs)
    out = []
    for off in (1, 0):
        _lib.set_option("NO_PIPELINE", off)
        try:
            b = scarlet.BlendBatch(d["images"], d["centers"], centroid_weight=model_psf.astype(np.float32))
            b.set_diff_kernel(diff)
            b.init_extended(np.ones(B) * 0.1)
            n = b.fit(12, e_rel=1e-2, check_every=3)
            torch.cuda.synchronize()
            assert int(b.status.abs().sum().item()) == 0
            out.append((npy(b.morph_current), npy(b.sed_current), npy(b.it), npy(b.active), npy(b.lipschitz),
                        npy(b.mse_buf), npy(b.centers), n))
        finally:
            _lib.set_option("NO_PIPELINE", 0)
    assert out[0][-1] == out[1][-1]
    for x, y in zip(out[0][:-1], out[1][:-1]):
        assert_array_equal(x, y)
    assert len(np.unique(out[0][2])) > 1          # the run was ragged


def test_config3_shape_twenty_five_iterations_vs_oracle(scarlet):
    """BASELINE config 3's shape beyond the first few iterations: 2 scenes, 25 iterations through the three-pass PSF
    iteration and the box kernels (exact-shape instances), against the CPU oracle from the same initial state."""
    from oracle import pgm
    from scarlet_amd import synth
    B, H, W, K = 5, 128, 128, 8
    obs_psfs = np.array([synth.gaussian_psf((41, 41), 1.2 + 0.15 * b) for b in range(B)])
    model_psf = synth.gaussian_psf((41, 41), 0.9)
    diff = pgm.match_psfs(obs_psfs.astype(np.float32), model_psf[None].astype(np.float32))
    scenes = [synth.make_scene(330 + i, B=B, H=H, W=W, K=K, psfs=obs_psfs) for i in range(2)]
    b = scarlet.BlendBatch(np.stack([s["images"] for s in scenes]), np.stack([s["centers"] for s in scenes]),
                           centroid_weight=model_psf.astype(np.float32))
    b.set_diff_kernel(diff)
    scale = (model_psf.max() / obs_psfs.max(axis=(1, 2))).astype(np.float32)
    b.init_extended(np.ones(B) * 0.1, sed_scale=scale)
    sed0 = b.sed_current.cpu().numpy(); morph0 = b.morph_current.cpu().numpy()
    cen0 = b.centers.cpu().numpy(); sh0 = b.shifts.cpu().numpy()
    iters = 25
    b.fit(iters, e_rel=0)
    torch.cuda.synchronize()
    assert int(b.status.abs().sum().item()) == 0
    worst = 0
    for i in range(2):
        sc = pgm.scene_from_state(scenes[i]["images"], sed0[i], morph0[i], cen0[i], sh0[i],
                                  diff_kernel=diff, centroid_weight=model_psf.astype(np.float32))
        pgm.fit(sc, iters, e_rel=0)
        assert_array_equal(b.centers[i].cpu().numpy(), np.array([s.center for s in sc.sources]))
        worst = max(worst, rel_err(b.morph_current[i].cpu().numpy(), np.array([s.morph for s in sc.sources])),
                    rel_err(b.sed_current[i].cpu().numpy(), np.array([s.sed for s in sc.sources])),
                    rel_err(b.mse(i), sc.mse))
    assert worst < 1e-5, worst


def test_exact_shape_convolution_instance_equals_the_generic_kernel(scarlet):
    """BASELINE config 3's plan (128 x 128 frame, 41 x 41 kernel: F = 150 = 10 x 15, M = 75 = 5 x 15) has an exact-shape
    instance of the convolution kernel (k_psf_conv_x128: 1024 threads, shapes and radices as compile-time constants, the
    model-plane load fused into the first row pass).  (a) The host-side plan report says the instance applies to this
    shape -- an instance written for a plan the shape does not have is never launched, which is how a first version went
    unnoticed.  (b) Same codelets, same operands: everything but the loss (a sum whose order follows the thread
    count) is bit-identical to the generic kernel (NO_EXACT also switches the exact-shape box instances, themselves
    bit-identical to the generic ones: test_gpu_engine)."""
    import ctypes
    from scarlet_amd import synth, fft as fftmod, _lib
    B, H, W, K, S = 5, 128, 128, 8, 9
    obs_psfs = np.array([synth.gaussian_psf((41, 41), 1.2 + 0.15 * b) for b in range(B)])
    model_psf = synth.gaussian_psf((41, 41), 0.9)
    diff = np.asarray(fftmod.match_psfs(fftmod.Fourier(obs_psfs.astype(np.float32)),
                                        fftmod.Fourier(model_psf[None].astype(np.float32))).image, dtype=np.float32)
    scenes = [synth.make_scene(700 + i, B=B, H=H, W=W, K=K, psfs=obs_psfs) for i in range(S)]
    out = []
    for generic in (1, 0):
        _lib.set_option("NO_EXACT", generic)
        try:
            b = scarlet.BlendBatch(np.stack([s["images"] for s in scenes]), np.stack([s["centers"] for s in scenes]),
                                   centroid_weight=model_psf.astype(np.float32))
            b.set_diff_kernel(diff)
            plan = (ctypes.c_int32 * 16)()
            assert _lib.lib.scarlet_debug_psf_plan(ctypes.byref(b._c), plan) == 0
            assert list(plan[:10]) == [128, 128, 150, 150, 75, 76, 10, 15, 5, 15]
            assert plan[13] == (0 if generic else 1)
            b.init_extended(np.ones(B) * 0.1)
            b.fit(9, e_rel=1e-3)
            torch.cuda.synchronize()
            assert int(b.status.abs().sum().item()) == 0
            out.append((npy(b.morph_current), npy(b.sed_current), npy(b.it), npy(b.centers), npy(b.lipschitz),
                        np.array([b.mse(i) for i in range(S)])))
        finally:
            _lib.set_option("NO_EXACT", 0)
    for x, y in zip(out[0][:5], out[1][:5]):
        assert_array_equal(x, y)
    assert rel_err(out[0][5], out[1][5]) < 1e-6
