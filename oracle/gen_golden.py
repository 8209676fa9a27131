"""Generate golden fixtures by RUNNING THE REFERENCE (container only).

TEST INFRASTRUCTURE.  Usage (from the repo root, in the build container where
/root/reference exists):   python -m oracle.gen_golden

Imports the unmodified reference package through ``oracle/refshim.py`` and stores
inputs + outputs for every row of SURVEY.md section 8a as small ``.npz`` files under
``tests/golden/``.  Only data is written: no reference source text leaves the
container.  ``tests/test_oracle_golden.py`` pins ``oracle/pgm.py`` against these
files; the ``-m gpu`` tests pin the HIP path against them too.

Frames are float64 unless the name says f32 (numpy 2.x runs float32 FFTs in single
precision while the numpy the reference was written for up-cast).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle.refshim import load_reference  # noqa: E402

# the synthetic-scene generator is plain numpy; it is loaded by file so that the fixture generator runs on a
# clean checkout (importing the `scarlet_amd` package needs the built HIP library)
import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location("_scarlet_amd_synth", os.path.join(ROOT, "scarlet_amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path), "bytes")


def gen_fft(sc):
    fft = sc.fft
    g = lambda s: sc.psf.generate_psf_image(sc.psf.gaussian, (41, 41), sigma=s)
    p1, p2 = g(1.0), g(2.0)
    k12 = fft.match_psfs(p2, p1)
    k21 = fft.match_psfs(p1, p2)
    rng = np.random.RandomState(3)
    img = rng.rand(3, 23, 30)
    ker = rng.rand(3, 7, 9)
    conv = fft.convolve(fft.Fourier(img), fft.Fourier(ker), axes=(1, 2)).image
    shapes_in = np.array([[58, 43], [48, 43], [64, 41], [128, 41], [61, 61], [33, 33], [67, 8], [72, 3]])
    shapes_out = np.array([fft._get_fft_shape(np.zeros((a, a)), np.zeros((b, b)), 3)
                           for a, b in shapes_in])
    save("fft", psf1=p1.image, psf2=p2.image, k12=k12.image, k21=k21.image,
         img=img, ker=ker, conv=conv, shapes_in=shapes_in, shapes_out=shapes_out)


def gen_monotonic(sc):
    op = sc.operator
    out = {}
    cases = [((5, 5), (2, 2)), ((9, 11), (2, 7)), ((33, 29), (10, 20)), ((16, 16), (8, 8))]
    rng = np.random.RandomState(11)
    for n, (shape, c) in enumerate(cases):
        sc.cache.Cache._cache = {}
        w = op.getRadialMonotonicWeights(shape, useNearest=False, center=c)
        out["w%d" % n] = w
        out["shape%d" % n] = np.array(shape)
        out["center%d" % n] = np.array(c)
        for dt, tag in ((np.float64, "f64"), (np.float32, "f32")):
            for th in (0.0, 0.1):
                X = (rng.rand(*shape) + np.exp(-((np.arange(shape[0])[:, None] - c[0]) ** 2 +
                                                 (np.arange(shape[1])[None, :] - c[1]) ** 2) / 20.)).astype(dt)
                out["x%d_%s_%g" % (n, tag, th)] = X.copy()
                prox = op.prox_strict_monotonic(shape, use_nearest=False, thresh=th, center=c)
                Y = X.copy()
                prox(Y, 0)
                out["y%d_%s_%g" % (n, tag, th)] = Y
    # nearest-neighbour operator (centred shapes only: the reference ignores `center`
    # when building it, operator.py:110)
    for n, shape in enumerate([(5, 5), (9, 9), (11, 15)]):
        sc.cache.Cache._cache = {}
        c = ((shape[0] - 1) // 2, (shape[1] - 1) // 2)
        prox = op.prox_strict_monotonic(shape, use_nearest=True, thresh=0, center=c)
        X = rng.rand(*shape)
        Y = X.copy()
        prox(Y, 0)
        out["nshape%d" % n] = np.array(shape)
        out["nref%d" % n] = np.array(prox.keywords["ref_idx"])
        out["nx%d" % n] = X
        out["ny%d" % n] = Y
    save("monotonic", **out)


def gen_measure(sc):
    ms = sc.measurement
    rng = np.random.RandomState(5)
    psf = sc.psf.generate_psf_image(sc.psf.gaussian, (41, 41), amplitude=1, sigma=.9, normalize=False).image
    psf /= psf.max()
    out = {"psf": psf}
    H, W = 40, 36
    yy, xx = np.mgrid[:H, :W]
    cents = [(20, 18), (5, 30), (37, 3), (12, 12), (21, 33), (2, 2), (38, 34)]
    for n, (cy, cx) in enumerate(cents):
        m = np.exp(-((yy - cy - 0.3) ** 2 + (xx - cx + 0.2) ** 2) / 8.) + 0.05 * rng.rand(H, W)
        out["m%d" % n] = m
        out["c%d" % n] = np.array((cy, cx))
        mp = ms.max_pixel(m, (cy, cx))
        out["maxpix%d" % n] = np.array(mp)
        nc, sh = ms.psf_weighted_centroid(m, psf, mp)
        out["cen%d" % n] = np.array(nc)
        out["shift%d" % n] = np.array(sh)
        m32 = m.astype(np.float32)
        nc, sh = ms.psf_weighted_centroid(m32, psf, ms.max_pixel(m32, (cy, cx)))
        out["cen32_%d" % n] = np.array(nc)
        out["shift32_%d" % n] = np.array(sh)
    out["n"] = np.array(len(cents))
    save("measure", **out)


def gen_symmetry(sc):
    op = sc.operator
    rng = np.random.RandomState(7)
    out = {}
    cases = [((21, 21), (8, 13), (0.21, -0.37)), ((32, 30), (16, 15), (0.4, 0.1)),
             ((32, 30), (10, 22), (-0.49, 0.05)), ((64, 64), (30, 40), (0.013, -0.3)),
             ((17, 24), (3, 20), (0.3, 0.3)), ((64, 64), (33, 31), (0.25, 0.5)),
             ((31, 33), (15, 16), (0.3, 0.2))]
    for n, (shape, c, sh) in enumerate(cases):
        yy, xx = np.mgrid[:shape[0], :shape[1]]
        X = np.exp(-((yy - c[0] + sh[0]) ** 2 + (xx - c[1] + sh[1]) ** 2) / 10.) + .1 * rng.rand(*shape) - .03
        out["x%d" % n] = X
        out["shape%d" % n] = np.array(shape)
        out["center%d" % n] = np.array(c)
        out["shift%d" % n] = np.array(sh)
        for alg in ("kspace", "soft", "sdss"):
            for fill in (None, 0.0):
                Y = X.copy()
                op.prox_uncentered_symmetry(Y, 0, center=c, algorithm=alg, fill=fill,
                                            shift=np.array(sh), strength=.5)
                out["y%d_%s_%s" % (n, alg, "fill" if fill is not None else "nofill")] = Y
        # zero shift falls back to soft symmetry at strength 1 (operator.py:337-339)
        Y = X.copy()
        op.prox_uncentered_symmetry(Y, 0, center=c, algorithm="kspace", shift=np.array((0., 0.)))
        out["y%d_zero" % n] = Y
        Y = X.copy()
        op.prox_uncentered_symmetry(Y, 0, center=c, algorithm="kspace", shift=None)
        out["y%d_none" % n] = Y
        # the bare k-space operator on a centred odd window
    W = rng.rand(21, 27)
    out["kx"] = W
    out["ky"] = op.prox_kspace_symmetry(W, 0, shift=(0.3, -0.45))
    W2 = rng.rand(20, 26)
    out["kx2"] = W2
    out["ky2"] = op.prox_kspace_symmetry(W2, 0, shift=(-0.2, 0.15))
    out["n"] = np.array(len(cases))
    save("symmetry", **out)


def _blend_state(blend):
    comps = blend.components
    return dict(sed=np.array([c.sed for c in comps]), morph=np.array([c.morph for c in comps]),
                center=np.array([c.pixel_center for c in comps]).astype(np.int64),
                shift=np.array([c.shift for c in comps]).astype(np.float64),
                flags=np.array([c.flags.value for c in comps]),
                mse=np.array(blend.mse, dtype=np.float64))


def gen_grad(sc):
    """loss / gradient / Lipschitz / convergence on a small scene, with and without PSF."""
    rng = np.random.RandomState(13)
    B, H, W, K = 3, 21, 25, 2
    images = rng.rand(B, H, W)
    seds = rng.rand(K, B) + .5
    morphs = rng.rand(K, H, W)
    weights = rng.rand(B, H, W) + .5
    out = dict(images=images, seds=seds, morphs=morphs, weights=weights)
    g = lambda s, n: sc.psf.generate_psf_image(sc.psf.gaussian, (n, n), sigma=s).image
    tpsf = g(.8, 11)[None]
    opsf = np.array([g(1.1 + .2 * b, 11) for b in range(B)])
    out["tpsf"], out["opsf"] = tpsf, opsf
    for tag, use_psf, w in (("nopsf", False, 1), ("nopsf_w", False, weights),
                            ("psf", True, 1), ("psf_w", True, weights)):
        if use_psf:
            frame = sc.Frame(images.shape, psfs=tpsf.copy(), dtype=np.float64)
            obs = sc.Observation(images, psfs=opsf.copy(), weights=None if w is 1 else w).match(frame)
        else:
            frame = sc.Frame(images.shape, dtype=np.float64)
            obs = sc.Observation(images, weights=None if w is 1 else w).match(frame)
        comps = [sc.Component(frame, seds[k].copy(), morphs[k].copy()) for k in range(K)]
        blend = sc.Blend(comps, obs)
        blend._backward()
        out["loss_" + tag] = np.array(blend.mse[-1])
        out["gsed_" + tag] = np.array([c.sed_grad for c in comps])
        out["gmorph_" + tag] = np.array([c.morph_grad for c in comps])
        out["render_" + tag] = obs.render(blend.get_model())
        if use_psf:
            out["diff_" + tag] = obs._diff_kernels.image
        blend._set_lipschitz(False)
        out["L_exact_" + tag] = np.array([blend.L_sed, blend.L_morph])
        blend._set_lipschitz(True)
        out["L_approx_" + tag] = np.array([blend.L_sed, blend.L_morph])
    save("grad", **out)


def gen_fit_hsc(sc):
    """BASELINE config 1: data/hsc_cosmos_35.npz, rows 0-1, 50 iterations, e_rel=0."""
    d = np.load("/root/reference/data/hsc_cosmos_35.npz")
    cat = d["catalog"]
    pix = np.array([(int(cat["y"][i]), int(cat["x"][i])) for i in range(2)])
    out = {"pixels": pix}
    # the inputs themselves (data, 5x58x48 float32 + 5x43x43 PSFs) so that the tests and the
    # GPU box can rebuild config 1 without /root/reference
    save("hsc_inputs", images=d["images"], psfs=d["psfs"].astype(np.float32),
         catalog_yx=np.array([(cat["y"][i], cat["x"][i]) for i in range(len(cat))]))
    for dt, tag in ((np.float64, "f64"), (np.float32, "f32")):
        images = d["images"].astype(dt)
        psfs = d["psfs"] / d["psfs"].sum(axis=(1, 2))[:, None, None]
        psfs = psfs.astype(dt)
        mpsf = sc.psf.generate_psf_image(sc.psf.gaussian, (43, 43), sigma=.9).image[None].astype(dt)
        frame = sc.Frame(images.shape, psfs=mpsf.copy(), dtype=dt)
        obs = sc.Observation(images, psfs=psfs.copy()).match(frame)
        bg = np.ones(5) * 0.1
        srcs = [sc.ExtendedSource(frame, tuple(p), obs, bg) for p in pix]
        out["init_sed_" + tag] = np.array([s.sed for s in srcs])
        out["init_morph_" + tag] = np.array([s.morph for s in srcs])
        out["init_center_" + tag] = np.array([s.pixel_center for s in srcs]).astype(np.int64)
        out["init_shift_" + tag] = np.array([s.shift for s in srcs])
        blend = sc.Blend(srcs, obs)
        blend.fit(50, e_rel=0)
        for k, v in _blend_state(blend).items():
            out[k + "_" + tag] = v
        if tag == "f64":
            out["model_psf"] = mpsf
            out["obs_psfs"] = psfs
            out["diff_kernel"] = obs._diff_kernels.image
    save("fit_hsc", **out)


def gen_fit_synth(sc):
    """BASELINE config 2 shaped scenes (5x64x64, K=4, no PSF): init + 30 iterations.
    Also a run with e_rel=1e-3 (ragged stop), one with approximate_L, one with L0."""
    out = {}
    for idx in (0, 1, 2):
        scn = synth.make_scene(idx)
        for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
            images = scn["images"].astype(dt)
            frame = sc.Frame(images.shape, dtype=dt)
            obs = sc.Observation(images).match(frame)
            bg = np.ones(5) * 0.1
            srcs = [sc.ExtendedSource(frame, tuple(int(v) for v in p), obs, bg) for p in scn["centers"]]
            pre = "s%d_%s_" % (idx, tag)
            out[pre + "init_sed"] = np.array([s.sed for s in srcs])
            out[pre + "init_morph"] = np.array([s.morph for s in srcs])
            out[pre + "init_center"] = np.array([s.pixel_center for s in srcs]).astype(np.int64)
            out[pre + "init_shift"] = np.array([s.shift for s in srcs])
            blend = sc.Blend(srcs, obs)
            blend.fit(30, e_rel=0)
            for k, v in _blend_state(blend).items():
                out[pre + k] = v
            # snapshots after 1 iteration for step-level parity
            srcs1 = [sc.ExtendedSource(frame, tuple(int(v) for v in p), obs, bg) for p in scn["centers"]]
            b1 = sc.Blend(srcs1, obs)
            b1.fit(1, e_rel=0)
            for k, v in _blend_state(b1).items():
                out[pre + "it1_" + k] = v
    # ragged stop + approximate_L on scene 0, f32
    scn = synth.make_scene(0)
    images = scn["images"]
    frame = sc.Frame(images.shape, dtype=np.float32)
    obs = sc.Observation(images).match(frame)
    bg = np.ones(5) * 0.1
    for tag, kw in (("erel", dict(max_iter=200, e_rel=1e-2)),
                    ("approx", dict(max_iter=30, e_rel=0, approximate_L=True))):
        srcs = [sc.ExtendedSource(frame, tuple(int(v) for v in p), obs, bg) for p in scn["centers"]]
        blend = sc.Blend(srcs, obs)
        blend.fit(**kw)
        for k, v in _blend_state(blend).items():
            out["s0_f32_%s_%s" % (tag, k)] = v
        out["s0_f32_%s_it" % tag] = np.array(blend.it)
    save("fit_synth", **out)


def gen_update(sc):
    """update.* with bbox arguments and sparsity (rows a10/a14/a15 untested upstream)."""
    upd = sc.update
    rng = np.random.RandomState(17)
    shape = (5, 24, 28)
    frame = sc.Frame(shape, dtype=np.float64)
    morph = rng.rand(24, 28) - .1
    sed = rng.rand(5)
    bbox = sc.bbox.Box.from_bounds(4, 19, 6, 25)
    out = dict(morph=morph, sed=sed, bbox=np.array([4, 19, 6, 25]), center=np.array([11, 14]))
    c = sc.Component(frame, sed.copy(), morph.copy())
    c.L_morph = 2.
    upd.monotonic(c, (11, 14), bbox=bbox)
    out["mono_bbox"] = c.morph.copy()
    c = sc.Component(frame, sed.copy(), morph.copy())
    c.L_morph = 2.
    c.shift = np.array((0.2, -0.1))
    upd.symmetric(c, (11, 14), bbox=bbox)
    out["sym_bbox"] = c.morph.copy()
    c = sc.Component(frame, sed.copy(), morph.copy())
    c.L_morph = 2.
    upd.sparse_l0(c, thresh=.5)
    out["l0"] = c.morph.copy()
    c = sc.Component(frame, sed.copy(), morph.copy())
    c.L_morph = 2.
    upd.sparse_l1(c, thresh=.5)
    out["l1"] = c.morph.copy()
    save("update", **out)


def gen_fit_extras(sc):
    """MultiComponentSource (source.py:242-295, 495-641) and Prior hooks
    (component.py:39-67, 177-187), which the reference's own tests do not exercise.  float32 frames."""
    out = {}
    bg = np.ones(5) * 0.1
    # ---- one two-component source + two extended sources, 8 iterations
    scn = synth.make_scene(5)
    images = scn["images"]
    frame = sc.Frame(images.shape, dtype=np.float32)
    obs = sc.Observation(images).match(frame)
    cen = [tuple(int(v) for v in p) for p in scn["centers"]]
    multi = sc.MultiComponentSource(frame, cen[0], obs, bg, flux_percentiles=[30])
    out["multi_init_sed"] = np.array([c.sed for c in multi.components])
    out["multi_init_morph"] = np.array([c.morph for c in multi.components])
    out["multi_init_center"] = np.array(multi.pixel_center).astype(np.int64)
    others = [sc.ExtendedSource(frame, p, obs, bg) for p in cen[1:3]]
    blend = sc.Blend([multi] + others, obs)
    blend.fit(8, e_rel=0)
    out["multi_sed"] = np.array([c.sed for c in blend.components])
    out["multi_morph"] = np.array([c.morph for c in blend.components])
    out["multi_mse"] = np.array(blend.mse)
    out["multi_center"] = np.array(multi.pixel_center).astype(np.int64)
    # ---- quadratic prior on source 1 of a 4-source scene, 8 iterations
    scn = synth.make_scene(3)
    images = scn["images"]
    frame = sc.Frame(images.shape, dtype=np.float32)
    obs = sc.Observation(images).match(frame)
    prior = sc.Prior(lambda sed, morph: (0.3 * sed, 2.0 * morph), lambda sed, morph: (0.3, 2.0))
    srcs = [sc.ExtendedSource(frame, tuple(int(v) for v in p), obs, bg, **({"prior": prior} if k == 1 else {}))
            for k, p in enumerate(scn["centers"])]
    blend = sc.Blend(srcs, obs)
    blend.fit(8, e_rel=0)
    out["prior_sed"] = np.array([c.sed for c in blend.components])
    out["prior_morph"] = np.array([c.morph for c in blend.components])
    out["prior_mse"] = np.array(blend.mse)
    out["prior_center"] = np.array([c.pixel_center for c in blend.components]).astype(np.int64)
    # ---- two observations of one 5-band scene (blend.py:120-139, 219-220): bands 0-2 and bands 3-4,
    # and the same scene observed twice with independent noise; sources initialised from the full cube
    scn = synth.make_scene(7)
    images = scn["images"]
    ch = list("grizy")
    frame = sc.Frame(images.shape, dtype=np.float32, channels=ch)
    full = sc.Observation(images, channels=ch).match(frame)
    cen = [tuple(int(v) for v in p) for p in scn["centers"]]
    for tag in ("sliced", "twice"):
        srcs = [sc.ExtendedSource(frame, p, full, bg) for p in cen]
        if tag == "sliced":
            obs = [sc.Observation(images[:3], channels=ch[:3]).match(frame),
                   sc.Observation(images[3:], channels=ch[3:]).match(frame)]
        else:
            noise = np.random.RandomState(5).normal(0, 0.1, images.shape).astype(np.float32)
            out["twice_images2"] = images + noise
            obs = [sc.Observation(images, channels=ch).match(frame),
                   sc.Observation(images + noise, channels=ch).match(frame)]
        blend = sc.Blend(srcs, obs)
        blend.fit(8, e_rel=0)
        out[tag + "_sed"] = np.array([c.sed for c in blend.components])
        out[tag + "_morph"] = np.array([c.morph for c in blend.components])
        out[tag + "_mse"] = np.array(blend.mse)
        out[tag + "_center"] = np.array([c.pixel_center for c in blend.components]).astype(np.int64)
    save("fit_extras", **out)


def gen_fit_extras2(sc):
    """approximate_L with two observations (blend.py:189-201, 219-220): the crude Lipschitz bound,
    doubled when the summed loss rose, times the number of observations."""
    out = {}
    bg = np.ones(5) * 0.1
    scn = synth.make_scene(7)
    images = scn["images"]
    ch = list("grizy")
    frame = sc.Frame(images.shape, dtype=np.float32, channels=ch)
    full = sc.Observation(images, channels=ch).match(frame)
    cen = [tuple(int(v) for v in p) for p in scn["centers"]]
    srcs = [sc.ExtendedSource(frame, p, full, bg) for p in cen]
    noise = np.random.RandomState(5).normal(0, 0.1, images.shape).astype(np.float32)
    obs = [sc.Observation(images, channels=ch).match(frame),
           sc.Observation(images + noise, channels=ch).match(frame)]
    blend = sc.Blend(srcs, obs)
    blend.fit(12, e_rel=0, approximate_L=True)
    out["images2"] = images + noise
    out["sed"] = np.array([c.sed for c in blend.components])
    out["morph"] = np.array([c.morph for c in blend.components])
    out["mse"] = np.array(blend.mse)
    out["center"] = np.array([c.pixel_center for c in blend.components]).astype(np.int64)
    save("fit_extras2", **out)


def gen_fit_extras3(sc):
    """CombinedExtendedSource / init_combined_extended_source (source.py:183-240, 495-536): SED over the channels
    of all observations, morphology from the detection coadd of observation `obs_idx`; no update() in the
    constructor, symmetric=False by default.  Two band-sliced observations, 6 iterations."""
    out = {}
    scn = synth.make_scene(7)
    images = scn["images"]
    ch = list("grizy")
    frame = sc.Frame(images.shape, dtype=np.float32, channels=ch)
    obs = [sc.Observation(images[:3], channels=ch[:3]).match(frame),
           sc.Observation(images[3:], channels=ch[3:]).match(frame)]
    cen = [tuple(int(v) for v in p) for p in scn["centers"]]
    bg = [np.ones(3) * 0.1, np.ones(2) * 0.1]
    for idx in (0, 1):
        srcs = [sc.CombinedExtendedSource(frame, p, obs, bg, obs_idx=idx) for p in cen]
        out["init%d_sed" % idx] = np.array([c.sed for c in srcs])
        out["init%d_morph" % idx] = np.array([c.morph for c in srcs])
        if idx == 0:
            blend = sc.Blend(srcs, obs)
            blend.fit(6, e_rel=0)
            out["sed"] = np.array([c.sed for c in blend.components])
            out["morph"] = np.array([c.morph for c in blend.components])
            out["mse"] = np.array(blend.mse)
            out["center"] = np.array([c.pixel_center for c in blend.components]).astype(np.int64)
    save("fit_extras3", **out)


def gen_geometry(sc):
    """Host-side geometry helpers of operator.py that the reference's tests do not pin: diagonalizeArray
    (481-522) and getRadialMonotonicWeights in nearest mode (540-621)."""
    import scarlet.operator as rop
    out = {}
    arr = np.arange(20.).reshape(4, 5) + 1
    d, m = rop.diagonalizeArray(arr)
    out["diag_in"] = arr; out["diag"] = d; out["diag_mask"] = m
    d, m = rop.diagonalizeArray(arr.reshape(-1), shape=(4, 5))
    out["diag_flat"] = d
    cases = [((5, 5), (2, 2), 1), ((7, 6), (1, 4), 1), ((9, 9), None, 0.5)]
    for n, (shape, c, mg) in enumerate(cases):
        sc.cache.Cache._cache = {}
        out["near%d" % n] = rop.getRadialMonotonicWeights(shape, useNearest=True, minGradient=mg, center=c)
        out["near%d_shape" % n] = np.array(shape)
        out["near%d_center" % n] = np.array(c if c is not None else (-1, -1))
        out["near%d_mg" % n] = np.array(mg)
    save("geometry", **out)


def gen_thresh_translate(sc):
    """update.threshold / measurement.threshold (log-histogram noise cut) and update.translation
    (Lanczos resampling by component.shift): SURVEY.md 8f rank 3, off the default pipeline."""
    upd, meas = sc.update, sc.measurement
    out = {}
    # the reference's own test case (tests/test_update.py:98-117) plus three synthetic morphologies
    np.random.seed(0)
    noise = np.random.rand(21, 21) * 2
    signal = np.zeros(noise.shape)
    signal[7:14, 7:14] = sc.psf.generate_psf_image(sc.psf.gaussian, (21, 21), normalize=False,
                                                   amplitude=10, sigma=3)[7:14, 7:14].image
    morphs = [signal + noise]
    rng = np.random.RandomState(23)
    y, x = np.mgrid[:64, :64]
    g = np.exp(-((y - 30.) ** 2 + (x - 35.) ** 2) / (2 * 3. ** 2))
    morphs.append(g + 1e-4 * rng.rand(64, 64))              # > 500 positive pixels: 50 bins
    m = g.copy(); m[g < 1e-3] = 0
    morphs.append(m)                                        # few positive pixels: size/10 bins
    m = np.zeros((16, 16)); m[5:8, 6:9] = rng.rand(3, 3)
    morphs.append(m)                                        # < 20 positive pixels: one bin, no cut
    morphs.append(rng.rand(40, 32) - 0.3)                   # negatives present
    for n, morph in enumerate(morphs):
        for dt, tag in ((np.float64, "f64"), (np.float32, "f32")):
            mm = morph.astype(dt)
            frame = sc.Frame((3,) + mm.shape, dtype=dt)
            c = sc.Component(frame, np.arange(3).astype(dt), mm.copy())
            thresh, bins = meas.threshold(c.morph)
            upd.threshold(c)
            b = c.bboxes["thresh"]
            out["thr_in%d_%s" % (n, tag)] = mm
            out["thr_value%d_%s" % (n, tag)] = np.array([float(thresh), float(bins)])
            out["thr_out%d_%s" % (n, tag)] = c.morph.copy()
            out["thr_box%d_%s" % (n, tag)] = np.array([b.bottom, b.top, b.left, b.right])
    out["thr_n"] = np.array(len(morphs))
    # translation
    shifts = [(0.217, -0.026), (-0.691, 0.321), (0.5, 0.5), (-0.3, 0.0), (0.0, 0.0)]
    img = rng.rand(31, 27) - 0.2
    for n, sh in enumerate(shifts):
        for direction in (1, -1):
            frame = sc.Frame((3,) + img.shape, dtype=np.float64)
            c = sc.Component(frame, np.arange(3.), img.copy())
            c.shift = np.array(sh)
            upd.translation(c, direction=direction)
            out["tr_out%d_%d" % (n, direction)] = c.morph.copy()
    out["tr_in"] = img
    out["tr_shifts"] = np.array(shifts)
    # interpolation.fft_resample with the bilinear kernel (the case tests/test_interpolation.py:365-393 pins)
    _img = np.arange(36).reshape(6, 6)
    im = np.zeros((11, 11)); im[2:8, 2:8] = _img
    out["bil_in"] = im
    out["bil_out0"] = sc.interpolation.fft_resample(im, .217, -.026, kernel=sc.interpolation.bilinear)
    out["bil_out1"] = sc.interpolation.fft_resample(im, -.691, .321, kernel=sc.interpolation.bilinear)
    save("thresh_translate", **out)


def main():
    sc = load_reference()
    import scarlet.cache
    sc.cache = scarlet.cache
    if len(sys.argv) > 1 and sys.argv[1] == "thresh":
        import scarlet.interpolation
        sc.interpolation = scarlet.interpolation
        gen_thresh_translate(sc)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "extras2":
        gen_fit_extras2(sc)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "extras3":
        gen_fit_extras3(sc)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "geometry":
        gen_geometry(sc)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "extras":
        gen_fit_extras(sc)
        return
    gen_fft(sc)
    gen_monotonic(sc)
    gen_measure(sc)
    gen_symmetry(sc)
    gen_grad(sc)
    gen_update(sc)
    gen_fit_hsc(sc)
    gen_fit_synth(sc)
    gen_fit_extras(sc)
    import scarlet.interpolation
    sc.interpolation = scarlet.interpolation
    gen_thresh_translate(sc)
    gen_fit_extras2(sc)
    gen_fit_extras3(sc)
    gen_geometry(sc)


if __name__ == "__main__":
    main()
