"""-m gpu: the batched Blend.fit() engine (HIP, through the C ABI) vs the CPU oracle and
the fixtures generated from the reference.  Config-2 shaped scenes (5x64x64, K=4)."""
import numpy as np
import pytest

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

TOL = 1e-5          # BASELINE.json north_star: float SED/morph arrays within 1e-5 relative


@pytest.fixture(scope="module")
def BB():
    from scarlet_amd import _lib
    _lib.require_gpu()
    from scarlet_amd.batch import BlendBatch
    return BlendBatch


def golden_batch(BB, g, idxs, tag="f32", **kw):
    from scarlet_amd import synth
    scenes = [synth.make_scene(i) for i in idxs]
    b = BB(np.stack([s["images"] for s in scenes]), np.stack([s["centers"] for s in scenes]), **kw)
    pre = ["s%d_%s_" % (i, tag) for i in idxs]
    b.set_state(np.stack([g[p + "init_sed"] for p in pre]), np.stack([g[p + "init_morph"] for p in pre]),
                np.stack([g[p + "init_center"] for p in pre]), np.stack([g[p + "init_shift"] for p in pre]))
    return b, pre


def test_init_extended_matches_reference(BB):
    g = load_golden("fit_synth")
    from scarlet_amd import synth
    idxs = [0, 1, 2]
    scenes = [synth.make_scene(i) for i in idxs]
    b = BB(np.stack([s["images"] for s in scenes]), np.stack([s["centers"] for s in scenes]))
    b.init_extended(np.ones(5) * 0.1)
    torch.cuda.synchronize()
    for n, i in enumerate(idxs):
        pre = "s%d_f32_" % i
        assert rel_err(b.morph_current[n].cpu().numpy(), g[pre + "init_morph"]) < TOL
        assert rel_err(b.sed_current[n].cpu().numpy(), g[pre + "init_sed"]) < TOL
        np.testing.assert_array_equal(b.centers[n].cpu().numpy(), g[pre + "init_center"])
        np.testing.assert_allclose(b.shifts[n].cpu().numpy(), g[pre + "init_shift"], rtol=0, atol=1e-6)
    assert int(b.status.abs().sum().item()) == 0


def test_one_iteration_matches_reference(BB):
    g = load_golden("fit_synth")
    b, pre = golden_batch(BB, g, [0, 1, 2])
    assert b.fit(1, e_rel=0) == 1
    torch.cuda.synchronize()
    for n, p in enumerate(pre):
        assert rel_err(b.morph_current[n].cpu().numpy(), g[p + "it1_morph"]) < TOL
        assert rel_err(b.sed_current[n].cpu().numpy(), g[p + "it1_sed"]) < TOL
        np.testing.assert_array_equal(b.centers[n].cpu().numpy(), g[p + "it1_center"])
        assert rel_err(b.mse(n), g[p + "it1_mse"]) < TOL


def test_thirty_iterations_match_reference(BB):
    g = load_golden("fit_synth")
    b, pre = golden_batch(BB, g, [0, 1, 2])
    b.fit(30, e_rel=0)
    torch.cuda.synchronize()
    for n, p in enumerate(pre):
        np.testing.assert_array_equal(b.centers[n].cpu().numpy(), g[p + "center"])
        assert rel_err(b.mse(n), g[p + "mse"]) < TOL
        assert rel_err(b.sed_current[n].cpu().numpy(), g[p + "sed"]) < TOL
        assert rel_err(b.morph_current[n].cpu().numpy(), g[p + "morph"]) < TOL
        np.testing.assert_allclose(b.shifts[n].cpu().numpy(), g[p + "shift"], rtol=0, atol=1e-5)
        # flags are NOT compared here: with e_rel=0 a flag clears only if a factor is bit-wise
        # unchanged between two iterations, which no independent float32 evaluation can
        # reproduce; they are compared at e_rel=1e-2 in test_ragged_convergence_...
        assert int(b.it[n].item()) == 30
    assert int(b.status.abs().sum().item()) == 0


def test_ragged_convergence_and_approximate_L(BB):
    g = load_golden("fit_synth")
    b, pre = golden_batch(BB, g, [0])
    b.fit(200, e_rel=1e-2)
    assert int(b.it[0].item()) == int(g["s0_f32_erel_it"])
    assert rel_err(b.morph_current[0].cpu().numpy(), g["s0_f32_erel_morph"]) < TOL
    np.testing.assert_array_equal(b.flags[0].cpu().numpy(), g["s0_f32_erel_flags"])
    b, pre = golden_batch(BB, g, [0])
    b.fit(30, e_rel=0, approximate_L=True)
    assert rel_err(b.mse(0), g["s0_f32_approx_mse"]) < TOL
    assert rel_err(b.morph_current[0].cpu().numpy(), g["s0_f32_approx_morph"]) < TOL


def test_batch_vs_oracle_scene_by_scene(BB):
    """24 seeded scenes, device init + 12 iterations, each compared with the CPU oracle
    run one scene at a time from the same initial state."""
    from oracle import pgm
    from scarlet_amd import synth
    S, iters = 24, 12
    scenes = [synth.make_scene(100 + i) for i in range(S)]
    b = BB(np.stack([s["images"] for s in scenes]), np.stack([s["centers"] for s in scenes]))
    b.init_extended(np.ones(5) * 0.1)
    sed0 = b.sed_current.cpu().numpy(); morph0 = b.morph_current.cpu().numpy()
    cen0 = b.centers.cpu().numpy(); sh0 = b.shifts.cpu().numpy()
    b.fit(iters, e_rel=0)
    torch.cuda.synchronize()
    sed1 = b.sed_current.cpu().numpy(); morph1 = b.morph_current.cpu().numpy()
    worst = 0
    for i in range(S):
        sc = pgm.scene_from_state(scenes[i]["images"], sed0[i], morph0[i], cen0[i], sh0[i])
        pgm.fit(sc, iters, e_rel=0)
        np.testing.assert_array_equal(b.centers[i].cpu().numpy(), np.array([s.center for s in sc.sources]))
        worst = max(worst, rel_err(morph1[i], np.array([s.morph for s in sc.sources])),
                    rel_err(sed1[i], np.array([s.sed for s in sc.sources])),
                    rel_err(b.mse(i), sc.mse))
    assert worst < TOL, worst


def test_batch_independence_and_determinism(BB):
    """Size-independent properties: a scene's result does not depend on which batch it is
    in, nor on its position; two runs are bit-identical."""
    from scarlet_amd import synth
    S = 64
    data = synth.make_batch(500, S)
    def run(sel):
        b = BB(data["images"][sel], data["centers"][sel])
        b.init_extended(np.ones(5) * 0.1)
        b.fit(7, e_rel=0)
        torch.cuda.synchronize()
        return b.morph_current.cpu().numpy(), b.sed_current.cpu().numpy(), b.mse_buf[:, :7].cpu().numpy()
    full = run(np.arange(S))
    again = run(np.arange(S))
    for a, c in zip(full, again):
        np.testing.assert_array_equal(a, c)
    perm = np.random.RandomState(0).permutation(S)[:16]
    sub = run(perm)
    for a, c in zip(full, sub):
        np.testing.assert_array_equal(a[perm], c)


def test_exact_shape_instance_is_bit_identical_to_the_generic_kernel(BB, monkeypatch):
    """k_iterate2<4,5,64> (K, B, H, W and the default pipeline folded at compile time; chosen by
    launch_fused for BASELINE's headline shape) performs the same arithmetic as k_iterate2<4,5,0>:
    factors, losses, centres, shifts and flags must agree bit for bit, also across an iteration
    with the centroid update (it % 5 == 0) and with fixed factors."""
    from scarlet_amd import synth
    S = 48
    data = synth.make_batch(1200, S)
    fix = np.zeros((S, 4), dtype=np.uint8)
    fix[::3, 1] = 1
    def run(generic, **kw):
        from scarlet_amd import _lib
        _lib.set_option("NO_EXACT", 1 if generic else 0)
        b = BB(data["images"], data["centers"])
        for name, arr in kw.items():
            setattr(b, name, torch.as_tensor(arr).cuda())
        b._fill_struct()
        b.init_extended(np.ones(5) * 0.1)
        b.fit(11, e_rel=1e-3)
        torch.cuda.synchronize()
        return [t.cpu().numpy() for t in (b.morph_current, b.sed_current, b.mse_buf[:, :11], b.centers, b.shifts,
                                          b.flags, b.it, b.lipschitz)]
    for kw in ({}, dict(fix_morph=fix), dict(fix_sed=fix)):
        exact, generic = run(False, **kw), run(True, **kw)
        for a, c in zip(exact, generic):
            np.testing.assert_array_equal(a, c)
    # the Hankel-vector cache of the k-space symmetry (workspace; vectors are recomputed only when
    # centre or shift changed) returns exactly what a fresh evaluation gives
    from scarlet_amd import _lib
    _lib.set_option("NO_KSCACHE", 1)
    try:
        uncached = run(False)
    finally:
        _lib.set_option("NO_KSCACHE", 0)
    for a, c in zip(uncached, run(False)):
        np.testing.assert_array_equal(a, c)


def test_multi_iteration_launch_is_bit_identical_to_one_launch_per_iteration(BB):
    """k_fit2x (several iterations of a scene per launch, morphologies resident in LDS between them, the loop closed
    by the kernel's re-entry jump) against k_iterate2<4,5,64> launched once per iteration (NO_PERSIST): every output
    bit for bit, through centroid iterations (it % 5 == 0), a ragged e_rel stop, fixed factors, a host check every
    few iterations (several launches of several iterations each) and a second fit() call on the same batch."""
    from scarlet_amd import synth, _lib
    # 160 distinct scenes tiled to 1600: more workgroups than the 512 the chip holds, so that workgroups start beside
    # workgroups that are in a later iteration (a first version handed centres and SEDs from one iteration to the next
    # through global memory and read stale lines of the vector L1 in ~1 % of the scenes -- only in launches like that)
    U, S = 160, 1600
    data = synth.make_batch(1400, U)
    data = {k: np.tile(v, (S // U,) + (1,) * (v.ndim - 1)) for k, v in data.items() if k in ("images", "centers")}
    fix = np.zeros((S, 4), dtype=np.uint8)
    fix[::3, 1] = 1

    def run(per_iteration, e_rel, check_every, n=(23, 9), dbg=0, **kw):
        _lib.set_option("NO_PERSIST", 1 if per_iteration else 0)
        _lib.set_option("PERSIST_DBG", dbg)
        try:
            b = BB(data["images"], data["centers"])
            for name, arr in kw.items():
                setattr(b, name, torch.as_tensor(arr).cuda())
            b._fill_struct()
            b.init_extended(np.ones(5) * 0.1)
            launched = [b.fit(n[0], e_rel=e_rel, check_every=check_every)]
            launched.append(b.fit(n[1], e_rel=e_rel, check_every=check_every))
            torch.cuda.synchronize()
            return launched, [t.cpu().numpy() for t in (b.morph_current, b.sed_current, b.mse_buf[:, :sum(n)], b.centers,
                                                        b.shifts, b.flags, b.it, b.lipschitz, b.active, b.cur, b.status,
                                                        b.morph[0], b.morph[1], b.sed[0], b.sed[1])]
        finally:
            _lib.set_option("NO_PERSIST", 0)
            _lib.set_option("PERSIST_DBG", 0)

    for e_rel, check_every, kw in ((0.0, 0, {}), (1e-3, 0, {}), (1e-3, 4, {}), (2e-3, 7, dict(fix_morph=fix)),
                                   (0.0, 10, dict(fix_sed=fix))):
        (l1, one), (l2, many) = run(True, e_rel, check_every, **kw), run(False, e_rel, check_every, **kw)
        assert l1 == l2
        for a, c in zip(one, many):
            np.testing.assert_array_equal(a, c)
    # PERSIST_DBG 1: every iteration after a launch's first reloads its tiles from memory instead of finding them in
    # LDS (the path a NaN result takes); 2: launches of ONE iteration go through k_fit2x as well (check_every = 1)
    ref = run(True, 1e-3, 0)
    for dbg, check_every in ((1, 0), (2, 1), (3, 3)):
        got = run(False, 1e-3, check_every, dbg=dbg)
        for a, c in zip(ref[1], got[1]):
            np.testing.assert_array_equal(a, c)
    # a single iteration per call never takes the multi-iteration kernel; 1 + 1 + ... must equal one call of n
    (_, single), (_, many) = run(False, 0.0, 0, n=(1, 1)), run(False, 0.0, 0, n=(2, 0))
    for a, c in zip(single[:7], many[:7]):
        np.testing.assert_array_equal(a, c)


@pytest.mark.parametrize("B,K,H,W,path", [
    (3, 2, 32, 48, "k_iterate2<4,5> (8 waves per scene)"),
    (5, 4, 24, 64, "k_iterate2<4,5>, short tile"),
    (6, 4, 64, 64, "k_iterate<4,6>"),
    (8, 3, 40, 40, "k_iterate<4,8>"),
    (5, 6, 48, 64, "general path, wave-level constraints (K > 4)"),
    (7, 7, 32, 32, "general path, K = B = 7"),
    (5, 3, 96, 80, "general path, workgroup-level constraints (H > 64)"),
    (4, 5, 50, 50, "general path (W % 4 != 0)"),
    (6, 12, 64, 64, "bigk.h: K > 8 in chunks of eight, wave-level constraints"),
    (6, 30, 64, 64, "bigk.h: BASELINE config 5's 30 sources and 6 bands on a 64 x 64 frame"),
    (3, 9, 96, 72, "bigk.h + workgroup-level constraints (H > 64)"),
])
def test_other_shapes_vs_oracle(BB, B, K, H, W, path):
    """Every kernel variant behind scarlet_fit (the dispatch on K, B, H, W is in launch_fused /
    scarlet_fit): 4 scenes, device init + 6 iterations against the CPU oracle."""
    from oracle import pgm
    from scarlet_amd import synth
    S, iters = 4, 6
    scenes = [synth.make_scene(900 + i, B=B, H=H, W=W, K=K) for i in range(S)]
    b = BB(np.stack([s["images"] for s in scenes]), np.stack([s["centers"] for s in scenes]))
    b.init_extended(np.ones(B) * 0.1)
    sed0 = b.sed_current.cpu().numpy(); morph0 = b.morph_current.cpu().numpy()
    cen0 = b.centers.cpu().numpy(); sh0 = b.shifts.cpu().numpy()
    b.fit(iters, e_rel=0)
    torch.cuda.synchronize()
    assert int(b.status.abs().sum().item()) == 0
    worst = 0
    for i in range(S):
        sc = pgm.scene_from_state(scenes[i]["images"], sed0[i], morph0[i], cen0[i], sh0[i])
        pgm.fit(sc, iters, e_rel=0)
        np.testing.assert_array_equal(b.centers[i].cpu().numpy(), np.array([s.center for s in sc.sources]))
        worst = max(worst, rel_err(b.morph_current[i].cpu().numpy(), np.array([s.morph for s in sc.sources])),
                    rel_err(b.sed_current[i].cpu().numpy(), np.array([s.sed for s in sc.sources])),
                    rel_err(b.mse(i), sc.mse))
    assert worst < TOL, (path, worst)


@pytest.mark.parametrize("B,K,H,W,l0", [
    (3, 3, 160, 144, None),
    (6, 5, 256, 256, 0.05),
])
def test_large_frames_vs_oracle(BB, B, K, H, W, l0):
    """Frames whose morphology tile does not fit LDS (BASELINE config 5 is 256 x 256 with L0
    sparsity): the constraint operators run in place on the planes in HBM (k_source_update<2>).
    The initial state comes from the CPU oracle (the device initialisation keeps a float64 tile in
    LDS and stops at ~140 x 140); 4 iterations, 2 scenes, against the oracle."""
    import copy
    from oracle import pgm
    from scarlet_amd import synth
    S, iters = 2, 4
    scenes = [synth.make_scene(700 + i, B=B, H=H, W=W, K=K) for i in range(S)]
    init = [pgm.make_extended_scene(s["images"], s["centers"], np.ones(B) * 0.1, l0_thresh=l0) for s in scenes]
    kw = {} if l0 is None else dict(l0_thresh=l0)
    b = BB(np.stack([s["images"] for s in scenes]), np.stack([s["centers"] for s in scenes]), **kw)
    b.set_state(np.array([[c.sed for c in sc.sources] for sc in init]),
                np.array([[c.morph for c in sc.sources] for sc in init]),
                centers=np.array([[c.center for c in sc.sources] for sc in init]),
                shifts=np.array([[c.shift for c in sc.sources] for sc in init]))
    b.fit(iters, e_rel=0)
    torch.cuda.synchronize()
    assert int(b.status.abs().sum().item()) == 0
    worst = 0
    for i, sc in enumerate(init):
        pgm.fit(sc, iters, e_rel=0)
        np.testing.assert_array_equal(b.centers[i].cpu().numpy(), np.array([s.center for s in sc.sources]))
        worst = max(worst, rel_err(b.morph_current[i].cpu().numpy(), np.array([s.morph for s in sc.sources])),
                    rel_err(b.sed_current[i].cpu().numpy(), np.array([s.sed for s in sc.sources])),
                    rel_err(b.mse(i), sc.mse))
    assert worst < TOL, worst


def test_config5_shape_30_sources_256_l0(BB):
    """BASELINE config 5 at one scene: 6 bands, 256 x 256, 30 overlapping sources (>= 3 px apart),
    symmetry + monotonicity + L0.  Device initialisation (float64 tile in HBM), then 3 iterations
    through bigk.h + k_source_update<2>, against the CPU oracle from the same initial state."""
    from oracle import pgm
    from scarlet_amd import synth
    B, K, H, W, l0 = 6, 30, 256, 256, 0.05
    scn = synth.make_scene(5000, B=B, H=H, W=W, K=K, min_sep=3)
    b = BB(scn["images"][None], scn["centers"][None], l0_thresh=l0)
    b.init_extended(np.ones(B) * 0.1)
    assert int(b.status.abs().sum().item()) == 0
    sed0 = b.sed_current.cpu().numpy()[0]; morph0 = b.morph_current.cpu().numpy()[0]
    cen0 = b.centers.cpu().numpy()[0]; sh0 = b.shifts.cpu().numpy()[0]
    # the device initialisation against the oracle's
    ref = pgm.make_extended_scene(scn["images"], scn["centers"], np.ones(B) * 0.1, l0_thresh=l0)
    assert rel_err(morph0, np.array([s.morph for s in ref.sources])) < TOL
    assert rel_err(sed0, np.array([s.sed for s in ref.sources])) < TOL
    np.testing.assert_array_equal(cen0, np.array([s.center for s in ref.sources]))
    iters = 3
    b.fit(iters, e_rel=0)
    torch.cuda.synchronize()
    assert int(b.status.abs().sum().item()) == 0
    sc = pgm.scene_from_state(scn["images"], sed0, morph0, cen0, sh0, l0_thresh=l0)
    pgm.fit(sc, iters, e_rel=0)
    np.testing.assert_array_equal(b.centers[0].cpu().numpy(), np.array([s.center for s in sc.sources]))
    assert rel_err(b.morph_current[0].cpu().numpy(), np.array([s.morph for s in sc.sources])) < TOL
    assert rel_err(b.sed_current[0].cpu().numpy(), np.array([s.sed for s in sc.sources])) < TOL
    assert rel_err(b.mse(0), sc.mse) < TOL


@pytest.mark.parametrize("B,K,H,W,path", [
    (5, 4, 64, 64, "k_iterate2"),
    (6, 4, 48, 48, "k_iterate<4,6>"),
    (5, 3, 96, 80, "general path"),
    (6, 12, 64, 64, "bigk.h"),
])
def test_weights_and_fixed_factors_vs_oracle(BB, B, K, H, W, path):
    """Per-pixel weights (inverse variance, zeros = masked pixels; observation.py:148-151, 222-247) and
    fix_sed / fix_morph (blend.py:91-96) through every gradient kernel: 2 scenes, 5 iterations."""
    from oracle import pgm
    from scarlet_amd import synth
    S, iters = 2, 5
    rng = np.random.RandomState(3)
    scenes = [synth.make_scene(1300 + i, B=B, H=H, W=W, K=K) for i in range(S)]
    weights = rng.uniform(0.5, 1.5, size=(S, B, H, W)).astype(np.float32)
    weights[rng.rand(S, B, H, W) < 0.03] = 0
    b = BB(np.stack([s["images"] for s in scenes]), np.stack([s["centers"] for s in scenes]), weights=weights)
    b.init_extended(np.ones(B) * 0.1)
    fs = np.zeros((S, K), np.uint8); fm = np.zeros((S, K), np.uint8)
    fs[:, 0] = 1; fm[:, 1] = 1
    b.fix_sed = torch.as_tensor(fs).cuda(); b.fix_morph = torch.as_tensor(fm).cuda()
    b._fill_struct()
    sed0 = b.sed_current.cpu().numpy(); morph0 = b.morph_current.cpu().numpy()
    cen0 = b.centers.cpu().numpy(); sh0 = b.shifts.cpu().numpy()
    b.fit(iters, e_rel=0)
    torch.cuda.synchronize()
    assert int(b.status.abs().sum().item()) == 0
    worst = 0
    for i in range(S):
        sc = pgm.scene_from_state(scenes[i]["images"], sed0[i], morph0[i], cen0[i], sh0[i], weights=weights[i])
        sc.sources[0].fix_sed = True; sc.sources[1].fix_morph = True
        pgm.fit(sc, iters, e_rel=0)
        np.testing.assert_array_equal(b.centers[i].cpu().numpy(), np.array([s.center for s in sc.sources]))
        worst = max(worst, rel_err(b.morph_current[i].cpu().numpy(), np.array([s.morph for s in sc.sources])),
                    rel_err(b.sed_current[i].cpu().numpy(), np.array([s.sed for s in sc.sources])),
                    rel_err(b.mse(i), sc.mse))
    assert worst < TOL, (path, worst)


@pytest.mark.parametrize("sym,mono,l0,l1", [
    (False, True, None, None), (True, False, None, None), (False, False, None, None),
    (True, True, 0.3, None), (True, True, None, 0.2), (False, False, 0.3, 0.1),
])
@pytest.mark.parametrize("B,K,H,W", [(5, 4, 64, 64), (6, 3, 48, 64), (4, 3, 80, 72)])
def test_constraint_switches_vs_oracle(BB, B, K, H, W, sym, mono, l0, l1):
    """PointSource(symmetric=, monotonic=) and the sparsity hooks (source.py:402-440, update.py:71-82)
    in every combination the kernels branch on: fused 8-wave, fused 4-wave and general path."""
    from oracle import pgm
    from scarlet_amd import synth
    S, iters = 2, 5
    scenes = [synth.make_scene(1700 + i, B=B, H=H, W=W, K=K) for i in range(S)]
    b = BB(np.stack([s["images"] for s in scenes]), np.stack([s["centers"] for s in scenes]),
           symmetric=sym, monotonic=mono, l0_thresh=l0, l1_thresh=l1)
    b.init_extended(np.ones(B) * 0.1)
    sed0 = b.sed_current.cpu().numpy(); morph0 = b.morph_current.cpu().numpy()
    cen0 = b.centers.cpu().numpy(); sh0 = b.shifts.cpu().numpy()
    b.fit(iters, e_rel=0)
    torch.cuda.synchronize()
    worst = 0
    for i in range(S):
        sc = pgm.scene_from_state(scenes[i]["images"], sed0[i], morph0[i], cen0[i],
                                  sh0[i] if sym else None, l0_thresh=l0)
        for s in sc.sources:
            s.symmetric, s.monotonic, s.l1_thresh = sym, mono, l1
        pgm.fit(sc, iters, e_rel=0)
        np.testing.assert_array_equal(b.centers[i].cpu().numpy(), np.array([s.center for s in sc.sources]))
        worst = max(worst, rel_err(b.morph_current[i].cpu().numpy(), np.array([s.morph for s in sc.sources])),
                    rel_err(b.sed_current[i].cpu().numpy(), np.array([s.sed for s in sc.sources])),
                    rel_err(b.mse(i), sc.mse))
    assert worst < TOL, worst


def test_full_size_batch_properties(BB):
    """BASELINE config 4 size (10 000 scenes of 5 x 64 x 64, K = 4) through size-independent properties:
    256 distinct scenes tiled to 10 000 -- every copy of a scene must come out bit-identical wherever it
    sits in the batch, identical to a 256-scene run, with no status bits and a decreasing loss."""
    from scarlet_amd import synth
    U, S, iters = 256, 10000, 6
    d = synth.make_batch(4000, U)
    reps = (S + U - 1) // U
    images = np.tile(d["images"], (reps, 1, 1, 1))[:S]
    centers = np.tile(d["centers"], (reps, 1, 1))[:S]

    def run(img, cen):
        b = BB(img, cen)
        b.init_extended(np.ones(5) * 0.1)
        b.fit(iters, e_rel=0)
        torch.cuda.synchronize()
        assert int(b.status.abs().sum().item()) == 0 and int(b.it.min().item()) == iters
        return b.morph_current, b.sed_current, b.mse_buf[:, :iters].clone()

    m, s, mse = run(images, centers)
    ms, ss, mses = run(d["images"], d["centers"])
    idx = torch.arange(S, device=m.device) % U
    assert torch.equal(m, ms[idx]) and torch.equal(s, ss[idx]) and torch.equal(mse, mses[idx])
    assert bool((mse[:, -1] < mse[:, 0]).all())


def test_config5_full_size_properties(BB):
    """BASELINE config 5 at its full batch (512 scenes of 6 x 256 x 256, 30 sources, L0) on one GPU:
    4 distinct scenes tiled to 512 -- copies bit-identical and equal to the 4-scene run."""
    from scarlet_amd import synth
    B, K, H, W, U, S, iters = 6, 30, 256, 256, 4, 512, 2
    scenes = [synth.make_scene(5000 + i, B=B, H=H, W=W, K=K, min_sep=3) for i in range(U)]
    ui = np.stack([s["images"] for s in scenes]); uc = np.stack([s["centers"] for s in scenes])

    def run(img, cen):
        b = BB(img, cen, l0_thresh=0.05)
        b.init_extended(np.ones(B) * 0.1)
        b.fit(iters, e_rel=0)
        torch.cuda.synchronize()
        assert int(b.status.abs().sum().item()) == 0
        return b.morph_current, b.sed_current, b.mse_buf[:, :iters].clone()

    reps = S // U
    m, s, mse = run(np.tile(ui, (reps, 1, 1, 1)), np.tile(uc, (reps, 1, 1)))
    ms, ss, mses = run(ui, uc)
    idx = torch.arange(S, device=m.device) % U
    assert torch.equal(m, ms[idx]) and torch.equal(s, ss[idx]) and torch.equal(mse, mses[idx])


@pytest.mark.parametrize("approx", [False, True])
def test_many_component_forms_agree(BB, approx):
    """K > 8: the forms of the gradient step behind the diagnostic switches -- one MFMA pass (k_bigk_fused) or
    residual + chunked step; Gram matrix by MFMA or by chunk pairs; Gram + eigenvalue on the second stream or in
    line -- are the same mathematics in different summation orders: equal iteration counts, values equal to
    2e-6 of the array's maximum after 6 iterations (each form is also compared with the oracle by the tests above
    with its switch at the default)."""
    from scarlet_amd import synth, _lib
    B, K, H, W, S = 6, 12, 64, 64, 3
    scenes = [synth.make_scene(700 + i, B=B, H=H, W=W, K=K, min_sep=3) for i in range(S)]
    rng = np.random.default_rng(5)
    weights = rng.uniform(0.5, 1.5, size=(S, B, H, W)).astype(np.float32)
    fix_morph = np.zeros((S, K), dtype=np.uint8); fix_morph[:, 2] = 1

    def run(**opts):
        for k, v in opts.items():
            _lib.set_option(k, v)
        try:
            b = BB(np.stack([s["images"] for s in scenes]), np.stack([s["centers"] for s in scenes]), weights=weights)
            b.init_extended(np.ones(B) * 0.1)
            b.fix_morph = torch.as_tensor(fix_morph).cuda(); b._fill_struct()
            b.fit(6, e_rel=1e-3, approximate_L=approx)
            torch.cuda.synchronize()
            assert int(b.status.abs().sum().item()) == 0
            return (b.morph_current.cpu().numpy().copy(), b.sed_current.cpu().numpy().copy(),
                    np.array([b.mse(i) for i in range(S)]), b.lipschitz.cpu().numpy().copy()), b.it.cpu().numpy().copy()
        finally:
            for k in opts:
                _lib.set_option(k, 0)

    ref, it_ref = run()
    for opts in (dict(NO_BIGK_FUSED=1), dict(NO_BIGK_FUSED=1, NO_GRAM_MFMA=1), dict(NO_SIDE_STREAM=1),
                 dict(NO_BIGK_FUSED=1, NO_SIDE_STREAM=1, NO_GRAM_MFMA=1)):
        got, it = run(**opts)
        np.testing.assert_array_equal(it, it_ref)
        for x, y in zip(got, ref):
            assert rel_err(x, y) < 2e-6, opts


def test_many_component_raw_gradients_forms_agree(BB):
    """scarlet_backward_gradients (the raw d loss / d sed, d loss / d morph of one backward pass) with K > 8: the
    one-pass MFMA kernel against the chunked passes, to 2e-6 of the arrays' maxima; loss and Lipschitz constants too."""
    import ctypes
    from scarlet_amd import synth, _lib
    B, K, H, W, S = 6, 12, 64, 64, 2
    scenes = [synth.make_scene(720 + i, B=B, H=H, W=W, K=K, min_sep=3) for i in range(S)]
    out = []
    for chunked in (0, 1):
        _lib.set_option("NO_BIGK_FUSED", chunked)
        try:
            b = BB(np.stack([s["images"] for s in scenes]), np.stack([s["centers"] for s in scenes]))
            b.init_extended(np.ones(B) * 0.1)
            _lib.check(_lib.lib.scarlet_backward_gradients(ctypes.byref(b._c), 0, _lib.stream_ptr()))
            torch.cuda.synchronize()
            cur = b.cur.cpu().numpy()
            g_sed = np.stack([b.sed[1 - cur[i]][i].cpu().numpy() for i in range(S)])
            g_morph = np.stack([b.morph[1 - cur[i]][i].cpu().numpy() for i in range(S)])
            out.append((g_sed, g_morph, b.lipschitz.cpu().numpy().copy(), b.mse_buf[:, 0].cpu().numpy().copy()))
        finally:
            _lib.set_option("NO_BIGK_FUSED", 0)
    assert np.abs(out[0][1]).max() > 0
    for x, y in zip(*out):
        assert rel_err(x, y) < 2e-6


@pytest.mark.parametrize("B,K,H,W,P,path", [
    (5, 4, 64, 64, None, "k_grad / k_step"),
    (6, 12, 64, 64, None, "k_bigk_fused + k_bigk_gram_mfma"),
    (3, 3, 32, 40, (9, 9), "k_psf_conv (LDS-resident FFT) + three-pass form"),
])
def test_device_gradients_equal_autograd(BB, B, K, H, W, P, path):
    """Row a5 on the device against automatic differentiation (no oracle in between): scarlet_backward_gradients'
    d loss / d sed and d loss / d morph (float32) against torch.autograd over the float64 forward chain of
    tests/test_oracle_autograd.py, 1e-5 of the arrays' maxima; the loss too."""
    import ctypes
    from scarlet_amd import synth, _lib, fft as fftmod
    from test_oracle_autograd import _render_torch
    S = 2
    kw = {}
    diff = None
    if P is not None:
        obs_psfs = np.array([synth.gaussian_psf(P, 1.2 + 0.15 * b) for b in range(B)])
        model_psf = synth.gaussian_psf(P, 0.9)
        diff = np.asarray(fftmod.match_psfs(fftmod.Fourier(obs_psfs.astype(np.float32)),
                                            fftmod.Fourier(model_psf[None].astype(np.float32))).image, dtype=np.float32)
        scenes = [synth.make_scene(810 + i, B=B, H=H, W=W, K=K, psfs=obs_psfs) for i in range(S)]
        kw = dict(centroid_weight=model_psf.astype(np.float32))
    else:
        scenes = [synth.make_scene(810 + i, B=B, H=H, W=W, K=K, min_sep=3) for i in range(S)]
    rng = np.random.default_rng(3)
    weights = rng.uniform(0.5, 1.5, size=(S, B, H, W)).astype(np.float32)
    b = BB(np.stack([s["images"] for s in scenes]), np.stack([s["centers"] for s in scenes]), weights=weights, **kw)
    if diff is not None:
        b.set_diff_kernel(diff)
    b.init_extended(np.ones(B) * 0.1)
    sed0, morph0 = b.sed_current.cpu().numpy().astype(np.float64), b.morph_current.cpu().numpy().astype(np.float64)
    _lib.check(_lib.lib.scarlet_backward_gradients(ctypes.byref(b._c), 0, _lib.stream_ptr()))
    torch.cuda.synchronize()
    cur = b.cur.cpu().numpy()
    for i in range(S):
        g_sed = b.sed[1 - cur[i]][i].cpu().numpy()
        g_morph = b.morph[1 - cur[i]][i].cpu().numpy()
        ts = torch.tensor(sed0[i], requires_grad=True)
        tm = torch.tensor(morph0[i], requires_grad=True)
        model = torch.einsum("kb,kyx->byx", ts, tm)
        rendered = _render_torch(model, None if diff is None else torch.tensor(diff.astype(np.float64)))
        d = torch.tensor(weights[i].astype(np.float64)) * (rendered - torch.tensor(scenes[i]["images"].astype(np.float64)))
        tl = 0.5 * (d ** 2).sum()
        ag_sed, ag_morph = torch.autograd.grad(tl, (ts, tm))
        assert rel_err(g_sed, ag_sed.numpy()) < TOL, path
        assert rel_err(g_morph, ag_morph.numpy()) < TOL, path
        assert abs(float(b.mse_buf[i, 0].item()) - float(tl)) < TOL * abs(float(tl)), path


@pytest.mark.parametrize("B,K,H,W,l0,mode", [
    (3, 3, 96, 80, -1.0, "tile in LDS (k_source_update<0/1>)"),
    (5, 8, 128, 128, -1.0, "tile in LDS, scratch in HBM (k_source_update<1>)"),
    (6, 10, 256, 256, 0.05, "plane in HBM (k_source_update<2>), L0"),
])
def test_full_frame_constraint_kernels_agree_with_the_box_kernels(BB, B, K, H, W, l0, mode):
    """Frames beyond 64 x 64 run their constraints on the box around each peak (boxupdate.h); the full-frame kernels of
    round 1 remain as the last resort for footprints beyond 127 x 127 and are rarely reached.  Forced here for every
    component (NO_BOX) and compared with the box path: same centres and iteration counts, values to 5e-6 of the
    arrays' maxima (the symmetry sums run in a different order)."""
    from scarlet_amd import synth, _lib
    S = 2
    scenes = [synth.make_scene(640 + i, B=B, H=H, W=W, K=K, min_sep=3) for i in range(S)]
    out = []
    for full in (0, 1):
        _lib.set_option("NO_BOX", full)
        try:
            b = BB(np.stack([s["images"] for s in scenes]), np.stack([s["centers"] for s in scenes]), l0_thresh=l0)
            b.init_extended(np.ones(B) * 0.1)
            b.fit(6, e_rel=1e-3)
            torch.cuda.synchronize()
            assert int(b.status.abs().sum().item()) == 0
            out.append((b.morph_current.cpu().numpy().copy(), b.sed_current.cpu().numpy().copy(),
                        np.array([b.mse(i) for i in range(S)]), b.centers.cpu().numpy().copy(), b.it.cpu().numpy().copy()))
        finally:
            _lib.set_option("NO_BOX", 0)
    np.testing.assert_array_equal(out[0][3], out[1][3])
    np.testing.assert_array_equal(out[0][4], out[1][4])
    for x, y in zip(out[0][:3], out[1][:3]):
        assert rel_err(x, y) < 5e-6, mode


@pytest.mark.parametrize("B,K,N,l0", [(5, 8, 128, -1.0), (6, 10, 256, 0.05)])
def test_exact_shape_box_instances_are_bit_identical_to_the_generic_ones(BB, B, K, N, l0):
    """The box kernels have instances with the BASELINE frame shapes (128 x 128, 256 x 256) as compile-time constants
    (boxupdate.h, XS): the same arithmetic with the address computations folded -- bit-identical to the generic
    instances (NO_EXACT)."""
    from scarlet_amd import synth, _lib
    S = 2
    scenes = [synth.make_scene(660 + i, B=B, H=N, W=N, K=K, min_sep=3) for i in range(S)]
    out = []
    for generic in (0, 1):
        _lib.set_option("NO_EXACT", generic)
        try:
            b = BB(np.stack([s["images"] for s in scenes]), np.stack([s["centers"] for s in scenes]), l0_thresh=l0)
            b.init_extended(np.ones(B) * 0.1)
            b.fit(6, e_rel=1e-3)
            torch.cuda.synchronize()
            out.append((b.morph_current.cpu().numpy().copy(), b.sed_current.cpu().numpy().copy(), b.mse_buf.cpu().numpy().copy(),
                        b.centers.cpu().numpy().copy(), b.it.cpu().numpy().copy(), b.shifts.cpu().numpy().copy()))
        finally:
            _lib.set_option("NO_EXACT", 0)
    for x, y in zip(*out):
        np.testing.assert_array_equal(x, y)


def test_config5_shape_twelve_iterations_vs_oracle(BB):
    """BASELINE config 5's shape (6 x 256 x 256, 30 sources, L0) beyond the first iterations: 12 iterations through the
    one-pass MFMA gradient step, the MFMA Gram matrix on the second stream and the box kernels, against the CPU oracle
    from the same initial state."""
    from oracle import pgm
    from scarlet_amd import synth
    B, K, H, W, l0 = 6, 30, 256, 256, 0.05
    scn = synth.make_scene(5003, B=B, H=H, W=W, K=K, min_sep=3)
    b = BB(scn["images"][None], scn["centers"][None], l0_thresh=l0)
    b.init_extended(np.ones(B) * 0.1)
    sed0 = b.sed_current.cpu().numpy()[0]; morph0 = b.morph_current.cpu().numpy()[0]
    cen0 = b.centers.cpu().numpy()[0]; sh0 = b.shifts.cpu().numpy()[0]
    iters = 12
    b.fit(iters, e_rel=0)
    torch.cuda.synchronize()
    assert int(b.status.abs().sum().item()) == 0
    sc = pgm.scene_from_state(scn["images"], sed0, morph0, cen0, sh0, l0_thresh=l0)
    pgm.fit(sc, iters, e_rel=0)
    np.testing.assert_array_equal(b.centers[0].cpu().numpy(), np.array([s.center for s in sc.sources]))
    assert rel_err(b.morph_current[0].cpu().numpy(), np.array([s.morph for s in sc.sources])) < TOL
    assert rel_err(b.sed_current[0].cpu().numpy(), np.array([s.sed for s in sc.sources])) < TOL
    assert rel_err(b.mse(0), sc.mse) < TOL
