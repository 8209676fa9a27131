"""-m gpu: the scarlet-style Python API (Frame / Observation / Component / Blend / update.* /
operator.*) on top of the HIP library.  These tests read like the reference's own tests
(tests/test_component.py, test_update.py, test_operator.py, test_source.py, test_blend.py);
the golden arrays are the reference's."""
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


# ------------------------------------------------------------------ tests/test_component.py
def test_component_methods(scarlet):
    sed = np.arange(5); morph = np.arange(20).reshape(4, 5)
    frame = scarlet.Frame((5, 4, 5))
    c = scarlet.Component(frame, sed, morph)
    truth = sed[:, None, None] * morph[None]
    assert_array_equal(npy(c.get_model()), truth)
    other = c.get_model(torch.ones(5, device="cuda"), c.morph + 10)
    assert_array_equal(npy(other), np.ones(5)[:, None, None] * (morph + 10)[None])
    assert_array_equal(npy(c.get_flux()), truth.sum(axis=(1, 2)))
    assert c.update() is c and c.coord is None and c.step_sed == 1 and c.step_morph == 1
    with pytest.raises(ValueError):
        c.get_model(sed=c.sed)


def test_component_tree(scarlet):
    class UpdateComponent(scarlet.Component):
        def __init__(self, norm="sed", *args, **kwargs):
            self.norm = norm
            super().__init__(*args, **kwargs)

        def update(self):
            scarlet.update.normalized(self, self.norm)

    frame = scarlet.Frame((3, 5, 5))
    sed = np.arange(3, dtype=np.float32); morph = np.arange(25, dtype=np.float32).reshape(5, 5)
    c1, c2, c3, c4 = [UpdateComponent(frame=frame, sed=sed, morph=morph) for _ in range(4)]
    c5 = UpdateComponent("morph", frame, sed, morph)
    tree1 = scarlet.ComponentTree([c1, c2]); tree2 = scarlet.ComponentTree([c3, c4])
    assert c1.coord == (0,) and c2._parent is tree1 and tree1.K == 2
    tree1 += tree2
    assert tree1.components == (c1, c2, c3, c4) and tree1.n_sources == 4
    tree1 += c5
    assert tree1.components == (c1, c2, c3, c4, c5) and tree2.components == (c3, c4)
    tree2.update()
    assert_array_equal(npy(c1.sed), sed); assert_array_equal(npy(c3.sed), sed / sed.sum())
    assert_array_equal(npy(c3.morph), morph * sed.sum())
    tree1.update()
    assert_array_equal(npy(c1.sed), sed / sed.sum()); assert_array_equal(npy(c5.sed), sed * morph.sum())
    assert_array_equal(npy(c5.morph), morph / morph.sum())
    t = scarlet.ComponentTree([scarlet.ComponentTree([c1]), c2])
    assert_array_equal(npy(t.get_model()), npy(c1.get_model() + c2.get_model()))
    with pytest.raises(NotImplementedError):
        scarlet.ComponentTree([1, 2])


# ------------------------------------------------------------------ tests/test_update.py
def test_update_functions(scarlet):
    update = scarlet.update
    frame = scarlet.Frame((6, 3, 3))
    sed = np.array([-.1, .1, 4, -.2, .2, 0], dtype=np.float32)
    morph = np.array([[-1, -.5, -1], [.1, 2, .3], [-.5, .3, 0]], dtype=np.float32)
    src = scarlet.Component(frame, sed.copy(), morph.copy()); update.positive_sed(src)
    assert_array_equal(npy(src.sed), np.array([0, .1, 4, 0, .2, 0], np.float32)); assert_array_equal(npy(src.morph), morph)
    src = scarlet.Component(frame, sed.copy(), morph.copy()); update.positive(src)
    assert_array_equal(npy(src.morph), np.array([[0, 0, 0], [.1, 2, .3], [0, .3, 0]], np.float32))
    frame = scarlet.Frame((6, 5, 5))
    sed = np.arange(6, dtype=np.float32); morph = np.arange(25, dtype=np.float32).reshape(5, 5)
    src = scarlet.Component(frame, sed.copy(), morph.copy()); update.normalized(src, type='sed')
    assert_array_equal(npy(src.sed), sed / 15); assert_array_equal(npy(src.morph), morph * 15)
    src = scarlet.Component(frame, sed.copy(), morph.copy()); update.normalized(src)
    assert_array_equal(npy(src.sed), sed * 24); assert_array_equal(npy(src.morph), morph / 24)
    with pytest.raises(ValueError):
        update.normalized(src, type='fubar')
    src = scarlet.Component(frame, sed.copy(), morph.copy()); src.L_morph = 1
    update.sparse_l0(src, thresh=4)
    t = morph.copy(); t[0, :-1] = 0
    assert_array_equal(npy(src.morph), t)
    src = scarlet.Component(frame, sed.copy(), morph.copy()); src.L_morph = 0.5
    update.sparse_l1(src, thresh=2)
    t = np.zeros(25, np.float32); t[5:] = np.arange(20) + 1
    assert_array_equal(npy(src.morph), t.reshape(5, 5))


def test_update_monotonic_symmetric(scarlet):
    update = scarlet.update
    frame = scarlet.Frame((6, 5, 5))
    sed = np.arange(6); morph = np.arange(25, dtype=float).reshape(5, 5)
    src = scarlet.Component(frame, sed.copy(), morph.copy()); src.L_morph = 1
    update.monotonic(src, (2, 2), use_nearest=True, exact=False, thresh=0)
    assert_array_equal(npy(src.morph), [[0, 1, 2, 3, 4], [5, 6, 7, 8, 9], [10, 11, 12, 12, 12],
                                        [11, 12, 12, 12, 12], [12, 12, 12, 12, 12]])
    src = scarlet.Component(frame, sed.copy(), morph.copy()); src.L_morph = 1
    update.monotonic(src, (2, 2))
    new_X = [[0.000000000, 1.000000000, 2.000000000, 3.000000000, 4.000000000],
             [5.000000000, 6.000000000, 7.000000000, 8.000000000, 9.000000000],
             [9.742640687, 11.000000000, 12.000000000, 12.000000000, 10.828427125],
             [11.030627697, 11.707106781, 12.000000000, 12.000000000, 11.771236166],
             [11.556349186, 11.868867239, 11.914213562, 11.983249156, 11.928090416]]
    assert_almost_equal(npy(src.morph), new_X, decimal=5)
    with pytest.raises(ValueError):
        update.monotonic(src, (2, 2), use_nearest=True, thresh=.25)
    with pytest.raises(NotImplementedError):
        update.monotonic(src, (2, 2), exact=True)
    src = scarlet.Component(frame, sed.copy(), morph.copy()); src.L_morph = 1
    update.monotonic(src, (2, 2), thresh=.25)
    assert_almost_equal(npy(src.morph)[1, 3:], [7.242640687, 5.806841831], decimal=5)
    # symmetry (reference tests/test_update.py:178-211)
    src = scarlet.Component(frame, sed.copy(), morph.copy()); src.L_morph = 1; src.pixel_center = (2, 2)
    update.symmetric(src, src.pixel_center)
    assert_array_equal(npy(src.morph), np.ones_like(morph) * 12)
    src = scarlet.Component(frame, sed.copy(), morph.copy()); src.L_morph = 1
    update.symmetric(src, (2, 2), strength=.5, algorithm="soft")
    assert_array_equal(npy(src.morph), np.arange(6, 18.5, .5).reshape(5, 5))
    src = scarlet.Component(frame, sed.copy(), morph.copy()); src.L_morph = 1
    update.symmetric(src, (1, 1))
    t = morph.copy(); t[:3, :3] = 6
    assert_array_equal(npy(src.morph), t)
    # bbox arguments: fixtures from the reference (incl. its all-zero symmetric quirk)
    g = load_golden("update")
    frame = scarlet.Frame((5, 24, 28))
    bbox = scarlet.Box.from_bounds(*[int(v) for v in g["bbox"]])
    c = scarlet.Component(frame, g["sed"], g["morph"]); c.L_morph = 2.
    update.monotonic(c, (11, 14), bbox=bbox)
    assert rel_err(npy(c.morph), g["mono_bbox"]) < 1e-5
    c = scarlet.Component(frame, g["sed"], g["morph"]); c.L_morph = 2.; c.shift = np.array((0.2, -0.1))
    update.symmetric(c, (11, 14), bbox=bbox)
    assert_array_equal(npy(c.morph), g["sym_bbox"])


# ------------------------------------------------------------------ tests/test_operator.py
def test_operator_functions(scarlet):
    op = scarlet.operator
    X = np.arange(25).reshape(5, 5).astype(np.float64)
    prox = op.prox_strict_monotonic(X.shape, use_nearest=True, thresh=0)
    _X = X.copy(); prox(_X, 0.0)
    assert_array_equal(_X[2:], [[10, 11, 12, 12, 12], [11, 12, 12, 12, 12], [12, 12, 12, 12, 12]])
    with pytest.raises(ValueError):
        op.prox_strict_monotonic(X.shape, use_nearest=True, thresh=.25)
    prox = op.prox_strict_monotonic(X.shape, use_nearest=False, thresh=0)
    _X = X.copy(); prox(_X, 0.0)
    assert_almost_equal(_X[2, 0], 9.74264069, decimal=5); assert_almost_equal(_X[4, 4], 11.92809042, decimal=5)
    Z = np.zeros((5, 5)); op.prox_center_on(Z, 0); assert Z[2, 2] == 1e-10
    Z = np.zeros((5, 5)); op.prox_sed_on(Z, 0, .1); assert_array_equal(Z, np.ones((5, 5)) * .1)
    M = np.arange(11, dtype=np.float32).reshape(1, 11); op.prox_max_unity(M, 0)
    assert_array_equal(M, (np.arange(11, dtype=np.float32) / 10).reshape(1, 11))
    _X = X.copy(); op.prox_soft_symmetry(_X, 0); assert_array_equal(_X, np.ones_like(X) * 12)
    _X = X.copy(); op.prox_soft_symmetry(_X, 0, 0); assert_array_equal(_X, X)
    _X = X.copy(); op.prox_soft_symmetry(_X, 0, .5); assert_array_equal(_X, np.arange(6, 18.5, .5).reshape(5, 5))
    x = np.zeros((21, 21))
    x[8:13, 8:13] = [[1, 2, 3, 2, 1], [2, 3, 4, 3, 1], [3, 4, 5, 1, 1], [2, 3, 1, 1, 1], [1, 1, 1, 1, 1]]
    assert_almost_equal(op.prox_kspace_symmetry(x, None, (0, 0)), (x[::-1, ::-1] + x) / 2, decimal=5)
    g = load_golden("symmetry")
    assert rel_err(op.prox_kspace_symmetry(g["kx"], 0, shift=(0.3, -0.45)), g["ky"]) < 1e-5
    assert rel_err(op.prox_kspace_symmetry(g["kx2"], 0, shift=(-0.2, 0.15)), g["ky2"]) < 1e-5
    with pytest.raises(ValueError):
        op.prox_uncentered_symmetry(X.copy(), 0, (2, 2), algorithm="fubar")


def test_uncentered_operator(scarlet):
    op = scarlet.operator

    def prox_plus(X, step):
        X[X < 0] = 0
        return X
    for flip in (False, True):
        x = np.arange(35).reshape(5, 7)
        x = (x[::-1] if flip else x) - 5
        x = x.astype(np.float32)
        shape = x.shape
        for c, region in [((2, 3), (slice(None), slice(None))), ((1, 2), (slice(0, 3), slice(0, 5))),
                          ((1, shape[1] - 3), (slice(0, 3), slice(-5, shape[1]))),
                          ((shape[0] - 2, 2), (slice(-3, shape[0]), slice(0, 5))),
                          ((shape[0] - 2, shape[1] - 3), (slice(-3, shape[0]), slice(-5, shape[1])))]:
            truth = x.copy(); truth[region][x[region] < 0] = 0
            _x = x.copy(); op.uncentered_operator(_x, prox_plus, c, step=1)
            assert_array_equal(_x, truth)
            truthf = np.zeros_like(x); truthf[region][x[region] > 0] = x[region][x[region] > 0]
            _x = x.copy(); op.uncentered_operator(_x, prox_plus, c, step=1, fill=0)
            assert_array_equal(_x, truthf)


def test_measurement(scarlet):
    morph = np.zeros((15, 15)); morph[4, 7] = 1; morph[11, 9] = 2
    assert scarlet.measurement.max_pixel(morph, (5, 5)) == (4, 7)
    assert scarlet.measurement.max_pixel(morph, (5, 5), window=(slice(0, 15),) * 2) == (11, 9)
    g = load_golden("measure")
    nc, sh = scarlet.measurement.psf_weighted_centroid(g["m0"].astype(np.float32), g["psf"], tuple(g["maxpix0"]))
    assert nc == tuple(g["cen32_0"]); assert_almost_equal(sh, g["shift32_0"], decimal=6)


# ------------------------------------------------------------------ tests/test_source.py
def test_sources(scarlet):
    shape = (5, 11, 15)
    x, y = np.meshgrid(np.linspace(-2, 2, 5), np.linspace(-2, 2, 5))
    r = np.sqrt(x ** 2 + y ** 2)
    true_sed = np.arange(5); true_morph = np.zeros(shape[1:])
    skycoord = (np.array(true_morph.shape) - 1) // 2
    cy, cx = skycoord
    true_morph[cy - 2:cy + 3, cx - 2:cx + 3] = 3 - r
    morph = true_morph.copy(); morph[5, 3] = 10
    images = (true_sed[:, None, None] * morph[None]).astype(np.float32)
    frame = scarlet.Frame(shape)
    obs = scarlet.Observation(images).match(frame)
    bg_rms = np.ones(5) * 1e-3
    sed, m = scarlet.init_extended_source(skycoord, frame, obs, bg_rms)
    assert_array_equal(npy(sed) / 3, true_sed); assert_almost_equal(npy(m) * 3, true_morph, decimal=5)
    src = scarlet.ExtendedSource(frame, skycoord, obs, bg_rms)
    assert_array_equal(src.pixel_center, skycoord)
    assert src.symmetric is True and src.monotonic is True and src.center_step == 5 and src.delay_thresh == 10
    assert_almost_equal(npy(src.morph) * 3, true_morph, decimal=5)
    morph = true_morph.copy(); morph[5, 5] = 2
    obs = scarlet.Observation((true_sed[:, None, None] * morph[None]).astype(np.float32)).match(frame)
    sed, m = scarlet.init_extended_source(skycoord, frame, obs, bg_rms, symmetric=False)
    t = true_morph.copy(); t[5, 5] = 1.5816233815926433
    assert_almost_equal(npy(m) * 3, t, decimal=5)
    sed, m = scarlet.init_extended_source(skycoord, frame, obs, bg_rms, monotonic=False)
    assert_almost_equal(npy(m) * 3, true_morph, decimal=5)
    with pytest.raises(scarlet.SourceInitError):
        scarlet.init_extended_source(skycoord, frame, obs, np.ones(5) * 1e3)
    # PointSource without PSF (reference tests/test_source.py:47-66)
    ps = scarlet.PointSource(frame, (4, 8), obs)
    t = np.zeros(shape[1:]); t[4, 8] = 1
    assert_array_equal(npy(ps.morph), t); assert ps.pixel_center == (4, 8)
    assert_array_equal(scarlet.get_pixel_sed((3, 2), obs), obs.images[:, 3, 2])


# ------------------------------------------------------------------ tests/test_blend.py (no-PSF part)
def test_blend_fit_matches_reference_and_both_pipelines_agree(scarlet):
    from scarlet_amd import synth
    g = load_golden("fit_synth")
    scn = synth.make_scene(0)
    frame = scarlet.Frame(scn["images"].shape)
    obs = scarlet.Observation(scn["images"]).match(frame)
    bg = np.ones(5) * 0.1
    srcs = [scarlet.ExtendedSource(frame, tuple(int(v) for v in p), obs, bg) for p in scn["centers"]]
    assert rel_err(np.array([npy(s.morph) for s in srcs]), g["s0_f32_init_morph"]) < 1e-5
    assert_array_equal(np.array([s.pixel_center for s in srcs]), g["s0_f32_init_center"])
    blend = scarlet.Blend(srcs, obs)
    blend.fit(30, e_rel=0)
    assert blend.it == 30 and len(blend.mse) == 30
    assert rel_err(blend.mse, g["s0_f32_mse"]) < 1e-5
    assert rel_err(np.array([npy(c.morph) for c in blend.components]), g["s0_f32_morph"]) < 1e-5
    assert rel_err(np.array([npy(c.sed) for c in blend.components]), g["s0_f32_sed"]) < 1e-5
    assert_array_equal(np.array([c.pixel_center for c in blend.components]), g["s0_f32_center"])
    mse = np.array(blend.mse)
    # re-entrant fit (reference docs/user_docs cell 46): 10 more iterations continue the run
    blend.fit(10, e_rel=0)
    assert blend.it == 40

    # the same fit through the Python pipeline: a subclass that overrides update()
    class MySource(scarlet.ExtendedSource):
        def update(self):
            return super().update()
    srcs2 = [MySource(frame, tuple(int(v) for v in p), obs, bg) for p in scn["centers"]]
    blend2 = scarlet.Blend(srcs2, obs)
    assert not blend2._builtin_pipeline()
    blend2.fit(30, e_rel=0)
    assert blend2.it == 30
    assert rel_err(blend2.mse, mse) < 1e-5
    assert rel_err(np.array([npy(c.morph) for c in blend2.components]), g["s0_f32_morph"]) < 1e-5
    assert_array_equal(np.array([c.pixel_center for c in blend2.components]), g["s0_f32_center"])
    # ragged stop: e_rel=1e-2 converges at the reference's iteration count
    srcs3 = [scarlet.ExtendedSource(frame, tuple(int(v) for v in p), obs, bg) for p in scn["centers"]]
    blend3 = scarlet.Blend(srcs3, obs).fit(200, e_rel=1e-2)
    assert blend3.it == int(g["s0_f32_erel_it"]) and blend3.converged


def test_prior_hooks_match_the_oracle(scarlet):
    """Component.backward_prior (reference component.py:177-187, blend.py:86-98): a prior adds its
    gradient and its Lipschitz constant to ONE component's step.  Quadratic prior on source 1 of a
    4-source scene, 8 iterations, against the CPU oracle with the same hook."""
    from oracle import pgm
    from scarlet_amd import synth
    scn = synth.make_scene(3)
    frame = scarlet.Frame(scn["images"].shape)
    obs = scarlet.Observation(scn["images"]).match(frame)
    bg = np.ones(5) * 0.1
    grad = lambda sed, morph: (0.3 * sed, 2.0 * morph)
    lip = lambda sed, morph: (0.3, 2.0)
    srcs = [scarlet.ExtendedSource(frame, tuple(int(v) for v in p), obs, bg,
                                   **({"prior": scarlet.Prior(grad, lip)} if k == 1 else {}))
            for k, p in enumerate(scn["centers"])]
    blend = scarlet.Blend(srcs, obs)
    assert not blend._builtin_pipeline()
    sed0 = np.array([npy(c.sed) for c in blend.components]); morph0 = np.array([npy(c.morph) for c in blend.components])
    cen0 = np.array([c.pixel_center for c in blend.components])
    sh0 = np.array([[float(v) for v in c.shift] for c in blend.components])   # set by the constructors' update()
    sc = pgm.scene_from_state(scn["images"], sed0, morph0, cen0, sh0)
    sc.sources[1].prior = (grad, lip)
    blend.fit(8, e_rel=0)
    pgm.fit(sc, 8, e_rel=0)
    assert rel_err(blend.mse, sc.mse) < 1e-5
    assert rel_err(np.array([npy(c.morph) for c in blend.components]), np.array([s.morph for s in sc.sources])) < 1e-5
    assert rel_err(np.array([npy(c.sed) for c in blend.components]), np.array([s.sed for s in sc.sources])) < 1e-5
    assert_array_equal(np.array([c.pixel_center for c in blend.components]), np.array([s.center for s in sc.sources]))
    # and against the fixture produced by the reference itself (oracle/gen_golden.py gen_fit_extras)
    g = load_golden("fit_extras")
    assert rel_err(blend.mse, g["prior_mse"]) < 1e-5
    assert rel_err(np.array([npy(c.morph) for c in blend.components]), g["prior_morph"]) < 1e-5
    assert rel_err(np.array([npy(c.sed) for c in blend.components]), g["prior_sed"]) < 1e-5
    assert_array_equal(np.array([c.pixel_center for c in blend.components]), g["prior_center"])
    # the prior changed the fit of that component
    plain = pgm.scene_from_state(scn["images"], sed0, morph0, cen0, sh0)
    pgm.fit(plain, 8, e_rel=0)
    assert rel_err(plain.sources[1].sed, sc.sources[1].sed) > 1e-3


def test_multicomponent_source_matches_the_oracle(scarlet):
    """MultiComponentSource (reference source.py:242-295, 495-641): layered initialisation, shared
    centre measured on the flux-weighted sum, per-component constraints.  One two-component source
    + two extended sources, 8 iterations -- through the DEVICE pipeline (scarlet_fit with the `group` field:
    k_group_centers + the grouped mode of the update kernels, no host synchronisation per iteration) and
    through the Python pipeline (per-source update()), both against the CPU oracle and the reference's fixture."""
    from oracle import pgm
    from scarlet_amd import synth
    scn = synth.make_scene(5)
    images = scn["images"]
    frame = scarlet.Frame(images.shape)
    obs = scarlet.Observation(images).match(frame)
    bg = np.ones(5) * 0.1
    cen = [tuple(int(v) for v in p) for p in scn["centers"]]
    multi = scarlet.MultiComponentSource(frame, cen[0], obs, bg, flux_percentiles=[30])
    assert multi.n_components == 2
    # initialisation against the oracle's restatement
    oseds, omorphs = pgm.init_multicomponent_source(cen[0], images, bg, [30])
    oms = pgm.MultiSource([pgm.Source(oseds[k], omorphs[k], cen[0], images.dtype) for k in range(2)], cen[0])
    pgm.multi_source_update(oms, 0)
    for k in range(2):
        assert rel_err(npy(multi.components[k].morph), oms.components[k].morph) < 1e-5
        assert rel_err(npy(multi.components[k].sed), oms.components[k].sed) < 1e-5
    assert tuple(multi.pixel_center) == oms.center
    others = [scarlet.ExtendedSource(frame, p, obs, bg) for p in cen[1:3]]
    blend = scarlet.Blend([multi] + others, obs)
    comps = blend.components
    assert len(comps) == 4 and blend._builtin_pipeline()
    # the same blend through the Python pipeline (fresh sources: a Blend adopts its components)
    multi_p = scarlet.MultiComponentSource(frame, cen[0], obs, bg, flux_percentiles=[30])
    blend_p = scarlet.Blend([multi_p] + [scarlet.ExtendedSource(frame, p, obs, bg) for p in cen[1:3]], obs)
    blend_p.python_pipeline = True
    assert not blend_p._builtin_pipeline()
    osrc = []
    for s in others:
        o = pgm.Source(npy(s.sed), npy(s.morph), s.pixel_center, images.dtype,
                       centroid_weight=pgm.default_centroid_weight())
        o.shift = tuple(float(v) for v in s.shift)
        osrc.append(o)
    sc = pgm.Scene(images, oms.components + osrc)
    sc.trees = [oms] + osrc
    blend.fit(8, e_rel=0)
    blend_p.fit(8, e_rel=0)
    pgm.fit(sc, 8, e_rel=0)
    g = load_golden("fit_extras")                        # produced by the reference itself
    for bl, mu in ((blend, multi), (blend_p, multi_p)):
        cs = bl.components
        assert rel_err(bl.mse, sc.mse) < 1e-5
        assert rel_err(np.array([npy(c.morph) for c in cs]), np.array([s.morph for s in sc.sources])) < 1e-5
        assert rel_err(np.array([npy(c.sed) for c in cs]), np.array([s.sed for s in sc.sources])) < 1e-5
        assert tuple(mu.pixel_center) == oms.center
        assert rel_err(bl.mse, g["multi_mse"]) < 1e-5
        assert rel_err(np.array([npy(c.morph) for c in cs]), g["multi_morph"]) < 1e-5
        assert rel_err(np.array([npy(c.sed) for c in cs]), g["multi_sed"]) < 1e-5
        assert_array_equal(np.array(mu.pixel_center), g["multi_center"])
    # a longer run across two centroid iterations (it % 5 == 0), the pipelines against each other
    blend.fit(7, e_rel=0); blend_p.fit(7, e_rel=0)
    assert rel_err(np.array([npy(c.morph) for c in blend.components]), np.array([npy(c.morph) for c in blend_p.components])) < 1e-5
    assert tuple(multi.pixel_center) == tuple(multi_p.pixel_center)
    assert np.allclose(multi.shift, multi_p.shift, atol=1e-6)


def test_several_observations_match_the_reference(scarlet):
    """Blend(sources, [obs1, obs2]) (reference blend.py:24-43, 120-139, 219-220): band-sliced
    observations of one cube and the same scene observed twice, against fixtures generated by the
    reference (oracle/gen_golden.py gen_fit_extras)."""
    from scarlet_amd import synth
    g = load_golden("fit_extras")
    scn = synth.make_scene(7)
    images = scn["images"]
    ch = list("grizy")
    bg = np.ones(5) * 0.1
    frame = scarlet.Frame(images.shape, channels=ch)
    full = scarlet.Observation(images, channels=ch).match(frame)
    cen = [tuple(int(v) for v in p) for p in scn["centers"]]
    for tag in ("sliced", "twice"):
        srcs = [scarlet.ExtendedSource(frame, p, full, bg) for p in cen]
        if tag == "sliced":
            obs = [scarlet.Observation(images[:3], channels=ch[:3]).match(frame),
                   scarlet.Observation(images[3:], channels=ch[3:]).match(frame)]
        else:
            obs = [scarlet.Observation(images, channels=ch).match(frame),
                   scarlet.Observation(g["twice_images2"], channels=ch).match(frame)]
        for python_pipeline in (False, True):          # device loop (scarlet_fit_multi) and host-side combination
            if python_pipeline:
                srcs = [scarlet.ExtendedSource(frame, p, full, bg) for p in cen]
            blend = scarlet.Blend(srcs, obs)
            blend.python_pipeline = python_pipeline
            assert blend._builtin_pipeline() == (not python_pipeline)
            blend.fit(8, e_rel=0)
            assert blend.it == 8
            assert rel_err(blend.mse, g[tag + "_mse"]) < 1e-5
            assert rel_err(np.array([npy(c.morph) for c in blend.components]), g[tag + "_morph"]) < 1e-5
            assert rel_err(np.array([npy(c.sed) for c in blend.components]), g[tag + "_sed"]) < 1e-5
            assert_array_equal(np.array([c.pixel_center for c in blend.components]), g[tag + "_center"])


def test_approximate_L_with_two_observations(scarlet):
    """fit(approximate_L=True) on a blend with two observations (reference blend.py:189-201,
    219-220), against a fixture generated by the reference (gen_fit_extras2)."""
    from scarlet_amd import synth
    g = load_golden("fit_extras2")
    scn = synth.make_scene(7)
    images = scn["images"]
    ch = list("grizy")
    frame = scarlet.Frame(images.shape, channels=ch)
    full = scarlet.Observation(images, channels=ch).match(frame)
    srcs = [scarlet.ExtendedSource(frame, tuple(int(v) for v in p), full, np.ones(5) * 0.1)
            for p in scn["centers"]]
    obs = [scarlet.Observation(images, channels=ch).match(frame),
           scarlet.Observation(g["images2"], channels=ch).match(frame)]
    blend = scarlet.Blend(srcs, obs)
    blend.fit(12, e_rel=0, approximate_L=True)
    assert blend.it == 12
    assert rel_err(blend.mse, g["mse"]) < 1e-5
    assert rel_err(np.array([npy(c.morph) for c in blend.components]), g["morph"]) < 1e-5
    assert rel_err(np.array([npy(c.sed) for c in blend.components]), g["sed"]) < 1e-5
    assert_array_equal(np.array([c.pixel_center for c in blend.components]), g["center"])


def test_threshold_matches_the_reference(scarlet):
    """measurement.threshold / update.threshold (reference measurement.py:97-112, update.py:85-103;
    the reference's tests/test_update.py:98-117 is case 0) on reference-generated fixtures: cut value
    to 1e-6 relative (float64 log10 on the device vs numpy's in the array's dtype), number of bins,
    surviving pixels and the trimmed box exactly."""
    g = load_golden("thresh_translate")
    for i in range(int(g["thr_n"])):
        for tag in ("f64", "f32"):
            m = g["thr_in%d_%s" % (i, tag)]
            want_t, want_b = g["thr_value%d_%s" % (i, tag)]
            frame = scarlet.Frame((3,) + m.shape)
            c = scarlet.Component(frame, np.arange(3.), m.copy())
            t, b = scarlet.measurement.threshold(c.morph)
            assert b == int(want_b)
            assert abs(float(t) - want_t) <= 1e-6 * abs(want_t)
            scarlet.update.threshold(c)
            assert_array_equal(npy(c.sed), np.arange(3.))
            assert_array_equal(npy(c.morph), g["thr_out%d_%s" % (i, tag)].astype(np.float32))
            bb = c.bboxes["thresh"]
            assert (bb.bottom, bb.top, bb.left, bb.right) == tuple(g["thr_box%d_%s" % (i, tag)])
    # the reference's assertion itself
    m = g["thr_in0_f64"]
    frame = scarlet.Frame((5,) + m.shape)
    src = scarlet.Component(frame, np.arange(5.), m.copy())
    scarlet.update.threshold(src)
    truth = np.zeros(m.shape)
    truth[7:14, 7:14] = m[7:14, 7:14]
    assert_almost_equal(npy(src.morph), truth, decimal=6)
    assert src.bboxes["thresh"] == scarlet.bbox.Box((7, 7), 7, 7)
    # batched device entry points: every plane of a stack gets its own range / histogram / box
    import ctypes
    L = scarlet._lib
    stack = torch.as_tensor(np.stack([g["thr_in1_f32"], g["thr_in2_f32"]]).astype(np.float32)).cuda()
    rng = torch.zeros((2, 3), dtype=torch.float64, device="cuda")
    L.check(L.lib.scarlet_log_range(L.ptr(stack), 2, 64 * 64, L.ptr(rng), L.stream_ptr()))
    box = torch.zeros((2, 4), dtype=torch.int32, device="cuda")
    L.check(L.lib.scarlet_trim(L.ptr(stack), 2, 64, 64, ctypes.c_float(0.), L.ptr(box), L.stream_ptr()))
    for k in range(2):
        pos = stack[k][stack[k] > 0].cpu().numpy().astype(np.float64)
        assert rng[k, 0].item() == pos.size
        assert_almost_equal(rng[k, 1:].cpu().numpy(), [np.log10(pos).min(), np.log10(pos).max()], decimal=12)
        ys, xs = np.nonzero(stack[k].cpu().numpy() > 0)
        assert_array_equal(box[k].cpu().numpy(), [ys.min(), ys.max(), xs.min(), xs.max()])
    with pytest.raises(ValueError):
        scarlet.bbox.trim(torch.zeros((8, 8), device="cuda"))


def test_translation_matches_the_reference(scarlet):
    """update.translation / interpolation.fft_resample (reference update.py:159-167,
    interpolation.py:408-448): reference-generated Lanczos shifts (float64 fixtures, float32 device
    planes: 1e-5 max-norm relative, the north_star tolerance) and the bilinear known answer of the
    reference's tests/test_interpolation.py:365-393."""
    g = load_golden("thresh_translate")
    img = g["tr_in"]
    for k, sh in enumerate(g["tr_shifts"]):
        for d in (1, -1):
            frame = scarlet.Frame((3,) + img.shape)
            c = scarlet.Component(frame, np.arange(3.), img.copy())
            c.shift = (float(sh[0]), float(sh[1]))
            scarlet.update.translation(c, direction=d)
            assert rel_err(npy(c.morph), g["tr_out%d_%d" % (k, d)]) < 1e-5
    im = g["bil_in"]
    for dy, dx, key in ((.217, -.026, "bil_out0"), (-.691, .321, "bil_out1")):
        res = scarlet.interpolation.fft_resample(im, dy, dx, kernel=scarlet.interpolation.bilinear)
        assert res.shape == im.shape
        assert rel_err(res, g[key]) < 1e-6
    with pytest.raises(ValueError):
        scarlet.interpolation.lanczos(1.5)


def test_resample_with_every_kernel_vs_oracle(scarlet):
    """interpolation.fft_resample with each separable kernel of the reference (bilinear, cubic spline family,
    Lanczos 3 and 5, quintic spline: 2 to 10 taps) against the oracle's FFT form (oracle/pgm.py
    fft_resample, itself pinned on reference-generated fixtures); float32 planes: 1e-6 max-norm relative."""
    from oracle import pgm
    I = scarlet.interpolation
    rng = np.random.RandomState(5)
    img = rng.rand(29, 34) - 0.1
    cases = [(I.bilinear, pgm.bilinear, {}), (I.cubic_spline, pgm.cubic_spline, {}),
             (I.catmull_rom, pgm.cubic_spline, dict(a=.5, b=0)),
             (I.mitchel_netravali, pgm.cubic_spline, dict(a=1 / 3, b=1 / 3)),
             (I.lanczos, pgm.lanczos, {}), (I.lanczos, pgm.lanczos, dict(a=5)),
             (I.quintic_spline, pgm.quintic_spline, {})]
    for kern, okern, okw in cases:
        kw = okw if kern in (I.lanczos, I.cubic_spline) else {}
        for dy, dx in ((.31, -.47), (-.9, .05)):
            got = I.fft_resample(img, dy, dx, kernel=kern, **kw)
            want = pgm.fft_resample(img, dy, dx, kernel=okern, **okw)
            assert got.shape == img.shape
            assert rel_err(got, want) < 1e-6, (kern.__name__, okw, dy, dx)


def test_combined_extended_source_matches_the_reference(scarlet):
    """CombinedExtendedSource / init_combined_extended_source (reference source.py:183-240, 495-536) on two
    band-sliced observations: initial factors for obs_idx = 0 and 1 and a 6-iteration fit, against a fixture
    generated by the reference (gen_fit_extras3).  The sources have symmetric=False and their constructor
    runs no update(), as in the reference."""
    from scarlet_amd import synth
    g = load_golden("fit_extras3")
    scn = synth.make_scene(7)
    images = scn["images"]
    ch = list("grizy")
    frame = scarlet.Frame(images.shape, channels=ch)
    obs = [scarlet.Observation(images[:3], channels=ch[:3]).match(frame),
           scarlet.Observation(images[3:], channels=ch[3:]).match(frame)]
    cen = [tuple(int(v) for v in p) for p in scn["centers"]]
    bg = [np.ones(3) * 0.1, np.ones(2) * 0.1]
    for idx in (0, 1):
        srcs = [scarlet.CombinedExtendedSource(frame, p, obs, bg, obs_idx=idx) for p in cen]
        assert srcs[0].symmetric is False
        assert rel_err(np.array([npy(c.sed) for c in srcs]), g["init%d_sed" % idx]) < 1e-6
        assert rel_err(np.array([npy(c.morph) for c in srcs]), g["init%d_morph" % idx]) < 1e-5
    srcs = [scarlet.CombinedExtendedSource(frame, p, obs, bg, obs_idx=0) for p in cen]
    blend = scarlet.Blend(srcs, obs)
    blend.fit(6, e_rel=0)
    assert blend.it == 6
    assert rel_err(blend.mse, g["mse"]) < 1e-5
    assert rel_err(np.array([npy(c.morph) for c in blend.components]), g["morph"]) < 1e-5
    assert rel_err(np.array([npy(c.sed) for c in blend.components]), g["sed"]) < 1e-5
    assert_array_equal(np.array([c.pixel_center for c in blend.components]), g["center"])


# ---------------------------------------------------------------------------------------------
# robustness of the boundary: centres outside the frame, scalar weights, threads
def test_centre_outside_the_frame_is_rejected(scarlet):
    """the reference raises IndexError for a source outside the image; BlendBatch raises ValueError before
    any launch, io.load_scene refuses the catalogue, and the C ABI itself (a caller that skipped the host
    check) flags the scene instead of reading outside the frame"""
    from scarlet_amd import synth, _lib
    import ctypes
    d = synth.make_batch(77, 3)
    bad = d["centers"].copy()
    bad[1, 2] = (64, 10)                                   # y == H: what np.rint of a catalogue can produce
    with pytest.raises(ValueError):
        scarlet.BlendBatch(d["images"], bad)
    b = scarlet.BlendBatch(d["images"], d["centers"])
    with pytest.raises(ValueError):
        b.set_state(np.zeros((3, 4, 5)), np.zeros((3, 4, 64, 64)), centers=bad)
    with pytest.raises(IndexError):
        scarlet.operator.prox_strict_monotonic((9, 9), center=(9, 4))(np.ones((9, 9), np.float32), 0)
    # straight through the C ABI: overwrite the device centres behind the host check
    b.centers.copy_(torch.as_tensor(bad).to(b.centers))
    b.init_extended(np.ones(5) * 0.1)
    b.fit(3, e_rel=0)
    torch.cuda.synchronize()
    st = b.status.cpu().numpy()
    assert st[1] & _lib.STATUS_CENTER_AT_EDGE and st[0] == 0 and st[2] == 0
    assert int(b.flags[1, 2].item()) & _lib.FLAG_NO_VALID_PIXELS
    ok = scarlet.BlendBatch(d["images"], d["centers"]).init_extended(np.ones(5) * 0.1)
    ok.fit(3, e_rel=0)
    for s in (0, 2):                                       # the healthy scenes are untouched by their neighbour
        np.testing.assert_array_equal(b.morph_current[s].cpu().numpy(), ok.morph_current[s].cpu().numpy())


def test_scalar_weight_survives_struct_refills(scarlet):
    """a Python-scalar weight != 1 (observation.py:148-151) is a batch attribute: fixed factors and a growing
    loss history both rebuild the C struct and must keep it"""
    from scarlet_amd import synth
    from oracle import pgm
    scn = synth.make_scene(4242)
    w = 0.5
    b = scarlet.BlendBatch(scn["images"][None], scn["centers"][None], weights=w, mse_capacity=4)
    assert b.weights is None and b.weight_scalar == w
    b.init_extended(np.ones(5) * 0.1)
    st = [t.cpu().numpy()[0] for t in (b.sed_current, b.morph_current, b.centers, b.shifts)]
    b.fix_sed = torch.zeros((1, 4), dtype=torch.uint8, device="cuda")
    b.fix_sed[0, 1] = 1
    b._fill_struct()
    b.fit(3, e_rel=0)
    b.fit(6, e_rel=0)                                      # grows the loss history: the struct is refilled again
    torch.cuda.synchronize()
    sc = pgm.scene_from_state(scn["images"], st[0], st[1], st[2], st[3], weights=w)
    sc.sources[1].fix_sed = True
    pgm.fit(sc, 9, e_rel=0)
    assert rel_err(b.mse(0), sc.mse) < 1e-5
    assert rel_err(b.morph_current[0].cpu().numpy(), np.array([s.morph for s in sc.sources])) < 1e-5
    assert rel_err(b.sed_current[0].cpu().numpy(), np.array([s.sed for s in sc.sources])) < 1e-5


def test_two_host_threads_on_two_streams(scarlet):
    """include/scarlet_hip.h: entry points may be called from several host threads at once on different batches
    and streams.  Two threads fit different batches concurrently (fused path and PSF path, which shares the
    plan/option/profile globals); results equal the single-threaded runs bit for bit."""
    import threading
    from scarlet_amd import synth
    d1, d2 = synth.make_batch(900, 64), synth.make_batch(1900, 8, H=48, W=40, K=3)
    ker = np.zeros((5, 7, 7), np.float32); ker[:, 3, 3] = 0.6; ker[:, 3, 2] = ker[:, 2, 3] = ker[:, 3, 4] = ker[:, 4, 3] = 0.1
    def run1(out, key):
        with torch.cuda.stream(torch.cuda.Stream()):
            b = scarlet.BlendBatch(d1["images"], d1["centers"]).init_extended(np.ones(5) * 0.1)
            for _ in range(4):
                b.fit(5, e_rel=0, check_every=0)
            torch.cuda.current_stream().synchronize()
            out[key] = b.morph_current.cpu().numpy()
    def run2(out, key):
        with torch.cuda.stream(torch.cuda.Stream()):
            b = scarlet.BlendBatch(d2["images"], d2["centers"])
            b.set_diff_kernel(ker)
            b.init_extended(np.ones(5) * 0.1)
            for _ in range(4):
                b.fit(5, e_rel=0, check_every=0)
            torch.cuda.current_stream().synchronize()
            out[key] = b.morph_current.cpu().numpy()
    ref = {}
    run1(ref, "a"); run2(ref, "b")
    got, errs = {}, []
    def guarded(fn, key):
        try:
            fn(got, key)
        except Exception as e:                              # surfaced below: a thread's exception is otherwise lost
            errs.append(e)
    th = [threading.Thread(target=guarded, args=(run1, "a")), threading.Thread(target=guarded, args=(run2, "b"))]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    np.testing.assert_array_equal(got["a"], ref["a"])
    np.testing.assert_array_equal(got["b"], ref["b"])


def test_multicomponent_device_pipeline_on_a_large_frame(scarlet):
    """the grouped mode of the box kernel (frames > 64: k_source_update_box) against the Python pipeline and the oracle"""
    from oracle import pgm
    from scarlet_amd import synth
    scn = synth.make_scene(77, H=96, W=80, K=3)
    images = scn["images"]
    frame = scarlet.Frame(images.shape)
    obs = scarlet.Observation(images).match(frame)
    bg = np.ones(5) * 0.1
    cen = [tuple(int(v) for v in p) for p in scn["centers"]]
    def make(py):
        m = scarlet.MultiComponentSource(frame, cen[0], obs, bg, flux_percentiles=[30])
        bl = scarlet.Blend([scarlet.ExtendedSource(frame, cen[1], obs, bg), m], obs)
        bl.python_pipeline = py
        return bl, m
    (bd, md), (bp, mp) = make(False), make(True)
    assert bd._builtin_pipeline() and not bp._builtin_pipeline()
    bd.fit(11, e_rel=0); bp.fit(11, e_rel=0)
    assert rel_err(bd.mse, bp.mse) < 1e-5
    assert rel_err(np.array([npy(c.morph) for c in bd.components]), np.array([npy(c.morph) for c in bp.components])) < 1e-5
    assert rel_err(np.array([npy(c.sed) for c in bd.components]), np.array([npy(c.sed) for c in bp.components])) < 1e-5
    assert tuple(md.pixel_center) == tuple(mp.pixel_center)


@pytest.mark.parametrize("case", ["many components (second stream beside the gradient pass)",
                                  "PSF batch of 1032 scenes (two half-batch pipelines)"])
def test_iteration_with_second_stream_survives_graph_capture(scarlet, case):
    """The library forks work onto a per-thread second stream and joins it back by events (include/scarlet_hip.h):
    a caller that captures its stream into a graph must get both branches.  One scarlet_fit iteration captured,
    replayed four times after one eager warm-up iteration, against five eager iterations: bit-identical."""
    import ctypes
    from scarlet_amd import synth, _lib, fft as fftmod
    if case.startswith("many"):
        B, K, H, W, S = 6, 12, 64, 64, 4
        scenes = [synth.make_scene(700 + i, B=B, H=H, W=W, K=K, min_sep=3) for i in range(S)]
        images, centers = np.stack([s["images"] for s in scenes]), np.stack([s["centers"] for s in scenes])
        kw, diff = {}, None
    else:
        B, K, H, W, S = 3, 2, 32, 32, 1032
        obs_psfs = np.array([synth.gaussian_psf((9, 9), 1.2 + 0.15 * b) for b in range(B)])
        model_psf = synth.gaussian_psf((9, 9), 0.9)
        diff = np.asarray(fftmod.match_psfs(fftmod.Fourier(obs_psfs.astype(np.float32)),
                                            fftmod.Fourier(model_psf[None].astype(np.float32))).image, dtype=np.float32)
        d = synth.make_batch(40, S, B=B, H=H, W=W, K=K, psfs=obs_psfs)
        images, centers, kw = d["images"], d["centers"], dict(centroid_weight=model_psf.astype(np.float32))

    def make():
        b = scarlet.BlendBatch(images, centers, **kw)
        if diff is not None:
            b.set_diff_kernel(diff)
        b.init_extended(np.ones(B) * 0.1)
        return b

    ref = make()
    ref.fit(5, e_rel=0, check_every=0)
    torch.cuda.synchronize()
    b = make()
    b._ensure_mse_capacity(16)
    if diff is not None:
        assert _lib.lib.scarlet_batch_pipelines(ctypes.byref(b._c)) == 2
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        sp = ctypes.c_void_p(st.cuda_stream)

        def one_iteration():
            assert _lib.lib.scarlet_fit(ctypes.byref(b._c), 1, 0.0, 0, 0, sp) == 1
        one_iteration()                      # eager: creates the second stream, sets the kernel attributes
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            one_iteration()
        torch.cuda.synchronize()
        for _ in range(4):
            g.replay()
        torch.cuda.synchronize()
    np.testing.assert_array_equal(npy(b.it), npy(ref.it))
    np.testing.assert_array_equal(npy(b.morph_current), npy(ref.morph_current))
    np.testing.assert_array_equal(npy(b.sed_current), npy(ref.sed_current))
    np.testing.assert_array_equal(npy(b.mse_buf)[:, :5], npy(ref.mse_buf)[:, :5])


def test_multi_iteration_launch_survives_graph_capture(scarlet):
    """scarlet_fit on the headline shape is ONE k_fit2x launch behind a memset of its scene-queue counter: both must be
    capturable.  A 3-iteration fit captured once and replayed three times after an eager 3-iteration warm-up, against 12
    eager iterations: bit-identical (700 scenes: more than the 512 workgroups of the launch, so the queue is used)."""
    import ctypes
    from scarlet_amd import synth, _lib
    U, S = 70, 700
    d = synth.make_batch(2600, U)
    images, centers = np.tile(d["images"], (S // U, 1, 1, 1)), np.tile(d["centers"], (S // U, 1, 1))

    def make():
        b = scarlet.BlendBatch(images, centers)
        b.init_extended(np.ones(5) * 0.1)
        return b

    ref = make()
    ref.fit(12, e_rel=0, check_every=0)
    torch.cuda.synchronize()
    b = make()
    b._ensure_mse_capacity(16)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        sp = ctypes.c_void_p(st.cuda_stream)

        def three_iterations():
            assert _lib.lib.scarlet_fit(ctypes.byref(b._c), 3, 0.0, 0, 0, sp) == 3
        three_iterations()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            three_iterations()
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
    np.testing.assert_array_equal(npy(b.it), npy(ref.it))
    np.testing.assert_array_equal(npy(b.morph_current), npy(ref.morph_current))
    np.testing.assert_array_equal(npy(b.sed_current), npy(ref.sed_current))
    np.testing.assert_array_equal(npy(b.mse_buf)[:, :12], npy(ref.mse_buf)[:, :12])
    np.testing.assert_array_equal(npy(b.flags), npy(ref.flags))


@pytest.mark.gpu
def test_package_imported_before_torch():
    """`import scarlet_amd` as the FIRST import of a process (the README's example) must work: the package loads PyTorch --
    and with it torch's own HIP runtime -- before its native library; in the other order every launch of the library failed
    with "no ROCm-capable device is detected".  Runs in a child process (import order is a property of a fresh interpreter)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "import scarlet_amd as scarlet\n"
        "from scarlet_amd import synth\n"
        "sc = synth.make_scene(11, B=5, H=64, W=64, K=4)\n"
        "b = scarlet.BlendBatch(sc['images'][None], sc['centers'][None])\n"
        "b.init_extended(np.ones(5) * 0.1)\n"
        "b.fit(3, e_rel=0)\n"
        "assert int(b.status.abs().sum().item()) == 0 and int(b.it[0].item()) == 3\n"
        "print('ok')\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
