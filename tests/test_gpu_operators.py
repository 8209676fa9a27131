"""-m gpu: HIP operators (through the C ABI) vs the CPU oracle and the golden fixtures.

Tolerances: integer/index outputs bit-exact; float32 arrays <= 1e-5 max-norm relative
(BASELINE.json north_star); tighter where the arithmetic is order-identical."""
import ctypes

import numpy as np
import pytest

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def L():
    from scarlet_amd import _lib
    _lib.require_gpu()
    return _lib


def dev(a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a)).to(device="cuda", dtype=dtype).contiguous()


def run_op(L, name, x, centers, *args):
    xs = dev(x, torch.float32)
    cs = dev(np.asarray(centers).reshape(-1, 2), torch.int32)
    n, H, W = xs.shape
    rc = getattr(L.lib, name)(L.ptr(xs), n, H, W, L.ptr(cs), *args, L.stream_ptr())
    L.check(rc)
    torch.cuda.synchronize()
    return xs.cpu().numpy()


# ------------------------------------------------------------------ monotonicity
def test_weighted_monotonic_reference_5x5(L):
    X = np.arange(25, dtype=np.float32).reshape(1, 5, 5)
    truth = [[0., 1., 2., 3., 4.], [5., 6., 7., 8., 9.],
             [9.74264069, 11., 12., 12., 10.82842712],
             [11.0306277, 11.70710678, 12., 12., 11.77123617],
             [11.55634919, 11.86886724, 11.91421356, 11.98324916, 11.92809042]]
    Y = run_op(L, "scarlet_prox_weighted_monotonic", X, [(2, 2)], ctypes.c_float(0.0))[0]
    assert rel_err(Y, truth) < 2e-7
    truth25 = [[0.000000000, 1.000000000, 2.000000000, 3.000000000, 4.000000000],
               [5.000000000, 6.000000000, 7.000000000, 7.242640687, 5.806841831],
               [5.801461031, 9.000000000, 12.000000000, 9.000000000, 6.074431804],
               [5.895545844, 7.681980515, 9.000000000, 7.681980515, 5.935521488],
               [4.988519641, 5.949655012, 6.170941546, 5.949655012, 4.997301087]]
    Y = run_op(L, "scarlet_prox_weighted_monotonic", X, [(2, 2)], ctypes.c_float(0.25))[0]
    assert rel_err(Y, truth25) < 2e-7


def test_weighted_monotonic_fixture_and_oracle(L):
    from oracle import pgm
    g = load_golden("monotonic")
    for n in range(4):
        c = tuple(int(v) for v in g["center%d" % n])
        for th in (0.0, 0.1):
            X = g["x%d_f32_%g" % (n, th)]
            Y = run_op(L, "scarlet_prox_weighted_monotonic", X[None], [c], ctypes.c_float(th))[0]
            assert rel_err(Y, g["y%d_f32_%g" % (n, th)]) < 1e-6
    rng = np.random.RandomState(1)
    cases = [((64, 64), (32, 32)), ((64, 64), (10, 50)), ((64, 64), (63, 0)), ((58, 48), (33, 14)),
             ((7, 9), (0, 0)), ((128, 128), (70, 61)), ((33, 130), (5, 100))]
    xs, cs = [], []
    for shape, c in cases:
        yy, xx = np.mgrid[:shape[0], :shape[1]]
        X = (np.exp(-((yy - c[0]) ** 2 + (xx - c[1]) ** 2) / 30.) + 0.2 * rng.rand(*shape)).astype(np.float32)
        for th in (0.0, 0.1):
            Y = run_op(L, "scarlet_prox_weighted_monotonic", X[None], [c], ctypes.c_float(th))[0]
            ref = X.copy()
            pgm.prox_weighted_monotonic(ref, c, th)
            assert rel_err(Y, ref) < 1e-6, (shape, c, th)
    # batch of arrays with different centres in one launch
    X = rng.rand(16, 40, 44).astype(np.float32)
    C = np.stack([rng.randint(0, 40, 16), rng.randint(0, 44, 16)], axis=1)
    Y = run_op(L, "scarlet_prox_weighted_monotonic", X, C, ctypes.c_float(0.0))
    for i in range(16):
        ref = X[i].copy()
        pgm.prox_weighted_monotonic(ref, tuple(C[i]), 0.0)
        assert rel_err(Y[i], ref) < 1e-6


def test_nearest_monotonic(L):
    from oracle import pgm
    g = load_golden("monotonic")
    for n in range(3):
        shape = tuple(g["nshape%d" % n]); c = ((shape[0] - 1) // 2, (shape[1] - 1) // 2)
        X = g["nx%d" % n].astype(np.float32)
        Y = run_op(L, "scarlet_prox_nearest_monotonic", X[None], [c], ctypes.c_float(0.0))[0]
        ref = X.astype(np.float64)
        pgm.prox_nearest_monotonic(ref, c)
        np.testing.assert_array_equal(Y, ref.astype(np.float32))
    with pytest.raises(ValueError):
        run_op(L, "scarlet_prox_nearest_monotonic", np.zeros((1, 5, 5), np.float32), [(2, 2)], ctypes.c_float(0.25))


def test_host_dropins_match_oracle_bitwise(L):
    """The pybind11 replacements: same arguments (weights table, offsets, order)."""
    from oracle import pgm, native
    rng = np.random.RandomState(2)
    shape, c = (33, 29), (10, 20)
    w = pgm.radial_weights(shape, c)
    order = pgm.radius_order(shape, c)[1:].astype(np.int32)
    offs = pgm.neighbour_offsets(shape[1]).astype(np.int32)
    for dt, fn, ct in ((np.float32, "scarlet_host_prox_weighted_monotonic_f32", ctypes.c_float),
                       (np.float64, "scarlet_host_prox_weighted_monotonic_f64", ctypes.c_double)):
        X = rng.rand(*shape).astype(dt)
        ref = X.copy().reshape(-1)
        native.prox_weighted_monotonic(ref, w, offs, order, 0.1)
        Y = X.copy().reshape(-1)
        wt = np.ascontiguousarray(w, dtype=dt)
        L.check(getattr(L.lib, fn)(Y.ctypes.data, Y.size, wt.ctypes.data, offs.ctypes.data,
                                  order.ctypes.data, order.size, ct(0.1)))
        np.testing.assert_array_equal(Y, ref)
    X = rng.rand(9, 9)
    refi = pgm.nearest_reference((9, 9), (4, 4)).astype(np.int32)
    order = pgm.radius_order((9, 9), (4, 4)).astype(np.int32)
    ref = X.copy().reshape(-1); native.prox_monotonic(ref, refi, order, 0.0)
    Y = X.copy().reshape(-1)
    L.check(L.lib.scarlet_host_prox_monotonic_f64(Y.ctypes.data, Y.size, refi.ctypes.data, order.ctypes.data,
                                                  order.size, ctypes.c_double(0.0)))
    np.testing.assert_array_equal(Y, ref)
    img = rng.rand(12, 15).astype(np.float32)
    vals = rng.rand(4).astype(np.float32)
    ys = np.array([0, 1, 0, 2], np.int32); ye = np.array([1, 0, 0, 1], np.int32)
    xs = np.array([0, 0, 2, 1], np.int32); xe = np.array([2, 1, 0, 0], np.int32)
    ref = np.zeros_like(img); native.apply_filter(img, vals, ys, ye, xs, xe, ref)
    out = np.zeros_like(img)
    L.check(L.lib.scarlet_host_apply_filter_f32(img.ctypes.data, 12, 15, vals.ctypes.data, ys.ctypes.data,
                                                ye.ctypes.data, xs.ctypes.data, xe.ctypes.data, 4, out.ctypes.data))
    assert rel_err(out, ref) < 1e-6
    # the double overload (operators_pybind11.cc:87-88): same sums in float64, bit for bit (same order of the terms)
    img64, vals64 = rng.rand(12, 15), rng.rand(4)
    ref64 = np.zeros_like(img64); native.apply_filter(img64, vals64, ys, ye, xs, xe, ref64)
    out64 = np.zeros_like(img64)
    L.check(L.lib.scarlet_host_apply_filter_f64(img64.ctypes.data, 12, 15, vals64.ctypes.data, ys.ctypes.data,
                                                ye.ctypes.data, xs.ctypes.data, xe.ctypes.data, 4, out64.ctypes.data))
    assert rel_err(out64, ref64) < 1e-14


# ------------------------------------------------------------------ symmetry
def sym(L, X, c, shift, alg, strength=.5, fill=None):
    xs = dev(np.asarray(X)[None], torch.float32)
    cs = dev(np.asarray(c).reshape(1, 2), torch.int32)
    sh = None if shift is None else dev(np.asarray(shift, dtype=np.float64).reshape(1, 2), torch.float64)
    H, W = xs.shape[1:]
    rc = L.lib.scarlet_prox_symmetry(L.ptr(xs), 1, H, W, L.ptr(cs), L.ptr(sh), alg, ctypes.c_float(strength),
                                     int(fill is not None), ctypes.c_float(0.0 if fill is None else fill),
                                     L.stream_ptr())
    L.check(rc)
    torch.cuda.synchronize()
    return xs.cpu().numpy()[0]


def test_symmetry_reference_vectors(L):
    X = np.arange(25, dtype=np.float32).reshape(5, 5)
    np.testing.assert_array_equal(sym(L, X, (2, 2), None, L.SYM_SOFT, 1.0), np.ones((5, 5)) * 12)
    np.testing.assert_array_equal(sym(L, X, (2, 2), None, L.SYM_SOFT, 0.5), np.arange(6, 18.5, .5).reshape(5, 5))
    t = X.copy(); t[:3, :3] = 6
    np.testing.assert_array_equal(sym(L, X, (1, 1), None, L.SYM_SOFT, 1.0), t)


def test_symmetry_fixture(L):
    g = load_golden("symmetry")
    algs = {"kspace": L.SYM_KSPACE, "soft": L.SYM_SOFT, "sdss": L.SYM_SDSS}
    for n in range(int(g["n"])):
        X = g["x%d" % n].astype(np.float32); c = tuple(int(v) for v in g["center%d" % n]); sh = g["shift%d" % n]
        for name, alg in algs.items():
            for fill, ftag in ((None, "nofill"), (0.0, "fill")):
                Y = sym(L, X, c, sh, alg, .5, fill)
                want = g["y%d_%s_%s" % (n, name, ftag)]
                assert rel_err(Y, want) < 1e-5, (n, name, ftag, rel_err(Y, want))


def test_kspace_symmetry_vs_oracle_many_windows(L):
    from oracle import pgm
    rng = np.random.RandomState(4)
    worst = 0
    for shape in ((64, 64), (58, 48), (31, 33), (128, 128)):
        for _ in range(6):
            c = (rng.randint(2, shape[0] - 2), rng.randint(2, shape[1] - 2))
            sh = rng.uniform(-.5, .5, 2)
            yy, xx = np.mgrid[:shape[0], :shape[1]]
            X = (np.exp(-((yy - c[0]) ** 2 + (xx - c[1]) ** 2) / 12.) + .05 * rng.randn(*shape)).astype(np.float32)
            Y = sym(L, X, c, sh, L.SYM_KSPACE)
            ref = X.astype(np.float64)
            pgm.prox_symmetry(ref, c, "kspace", None, tuple(sh))
            worst = max(worst, rel_err(Y, ref))
    assert worst < 1e-5, worst
    # centred peak: the reference discards the k-space result
    X = rng.rand(64, 64).astype(np.float32)
    np.testing.assert_array_equal(sym(L, X, (32, 32), (0.2, 0.1), L.SYM_KSPACE), X)


# ------------------------------------------------------------------ measurement
def test_max_pixel_and_centroid_fixture(L):
    g = load_golden("measure")
    psf = dev(g["psf"], torch.float64)
    for n in range(int(g["n"])):
        m = g["m%d" % n].astype(np.float32)
        xs = dev(m[None], torch.float32)
        cs = dev(g["c%d" % n].reshape(1, 2), torch.int32)
        st = torch.zeros(1, dtype=torch.int32, device="cuda")
        L.check(L.lib.scarlet_max_pixel(L.ptr(xs), 1, m.shape[0], m.shape[1], L.ptr(cs), L.ptr(st), L.stream_ptr()))
        np.testing.assert_array_equal(cs.cpu().numpy()[0], g["maxpix%d" % n])
        sh = torch.zeros((1, 2), dtype=torch.float64, device="cuda")
        L.check(L.lib.scarlet_psf_weighted_centroid(L.ptr(xs), 1, m.shape[0], m.shape[1], L.ptr(psf), 41,
                                                    L.ptr(cs), L.ptr(sh), L.ptr(st), L.stream_ptr()))
        np.testing.assert_array_equal(cs.cpu().numpy()[0], g["cen32_%d" % n])
        np.testing.assert_allclose(sh.cpu().numpy()[0], g["shift32_%d" % n], rtol=0, atol=1e-6)
        assert int(st.item()) == 0
    # reference vector tests/test_update.py:11-21
    m = np.zeros((15, 15), np.float32); m[4, 7] = 1; m[11, 9] = 2
    xs = dev(m[None], torch.float32); cs = dev(np.array([[5, 5]]), torch.int32)
    L.check(L.lib.scarlet_max_pixel(L.ptr(xs), 1, 15, 15, L.ptr(cs), None, L.stream_ptr()))
    np.testing.assert_array_equal(cs.cpu().numpy()[0], [4, 7])
    # a centre within 2 px of the low edge makes the reference fail -> status bit
    cs = dev(np.array([[1, 5]]), torch.int32); st = torch.zeros(1, dtype=torch.int32, device="cuda")
    L.check(L.lib.scarlet_max_pixel(L.ptr(xs), 1, 15, 15, L.ptr(cs), L.ptr(st), L.stream_ptr()))
    assert int(st.item()) & L.STATUS_CENTER_AT_EDGE


# ------------------------------------------------------------------ elementwise / normalise
def test_prox_and_normalize(L):
    sed = np.array([-.1, .1, 4, -.2, .2, 0], dtype=np.float32)
    x = dev(sed, torch.float32)
    L.check(L.lib.scarlet_prox_plus(L.ptr(x), 6, L.stream_ptr()))
    np.testing.assert_array_equal(x.cpu().numpy(), np.array([0, .1, 4, 0, .2, 0], np.float32))
    morph = np.arange(25, dtype=np.float32)
    x = dev(morph, torch.float32)
    L.check(L.lib.scarlet_prox_hard(L.ptr(x), 25, ctypes.c_float(4.0), L.stream_ptr()))
    t = morph.copy(); t[:4] = 0
    np.testing.assert_array_equal(x.cpu().numpy(), t)
    x = dev(morph, torch.float32)
    L.check(L.lib.scarlet_prox_soft(L.ptr(x), 25, ctypes.c_float(4.0), L.stream_ptr()))
    t = np.zeros(25, np.float32); t[5:] = np.arange(20) + 1
    np.testing.assert_array_equal(x.cpu().numpy(), t)
    s0 = np.arange(6, dtype=np.float32); m0 = np.arange(25, dtype=np.float32)
    for typ, ws, wm in ((L.NORM_SED, s0 / 15, m0 * 15), (L.NORM_MORPH, s0 * 300, m0 / 300),
                        (L.NORM_MORPH_MAX, s0 * 24, m0 / 24)):
        s = dev(s0[None], torch.float32); m = dev(m0[None], torch.float32)
        L.check(L.lib.scarlet_normalize(L.ptr(s), L.ptr(m), 1, 6, 25, typ, L.stream_ptr()))
        np.testing.assert_array_equal(s.cpu().numpy()[0], ws)
        np.testing.assert_array_equal(m.cpu().numpy()[0], wm)
    with pytest.raises(ValueError):
        L.check(L.lib.scarlet_normalize(L.ptr(s), L.ptr(m), 1, 6, 25, 7, L.stream_ptr()))


def test_operators_on_arrays_beyond_the_lds_tile(L):
    """200 x 256 and 256 x 256 arrays do not fit LDS: the standalone operators then work in place in
    HBM (k_operator<true>).  Weighted monotonicity, k-space symmetry, max_pixel and the centroid
    against the CPU oracle."""
    from oracle import pgm
    rng = np.random.RandomState(11)
    for shape in ((200, 256), (256, 256)):
        c = (rng.randint(40, shape[0] - 40), rng.randint(40, shape[1] - 40))
        yy, xx = np.mgrid[:shape[0], :shape[1]]
        X = (np.exp(-((yy - c[0]) ** 2 + (xx - c[1]) ** 2) / 40.) + .02 * rng.randn(*shape)).astype(np.float32)
        Y = run_op(L, "scarlet_prox_weighted_monotonic", X[None], [c], ctypes.c_float(0.0))[0]
        ref = X.astype(np.float64).copy()
        pgm.prox_weighted_monotonic(ref, c, thresh=0.)
        assert rel_err(Y, ref) < 1e-5
        sh = rng.uniform(-.5, .5, 2)
        Y = sym(L, X, c, sh, L.SYM_KSPACE)
        ref = X.astype(np.float64)
        pgm.prox_symmetry(ref, c, "kspace", None, tuple(sh))
        assert rel_err(Y, ref) < 1e-5
        xs = dev(X[None], torch.float32)
        cs = dev(np.array([[c[0] + 1, c[1] - 2]]), torch.int32)
        st = torch.zeros(1, dtype=torch.int32, device="cuda")
        L.check(L.lib.scarlet_max_pixel(L.ptr(xs), 1, shape[0], shape[1], L.ptr(cs), L.ptr(st), L.stream_ptr()))
        np.testing.assert_array_equal(cs.cpu().numpy()[0], pgm.max_pixel(X, (c[0] + 1, c[1] - 2)))
        psf = pgm.default_centroid_weight()
        shf = torch.zeros((1, 2), dtype=torch.float64, device="cuda")
        L.check(L.lib.scarlet_psf_weighted_centroid(L.ptr(xs), 1, shape[0], shape[1], L.ptr(dev(psf, torch.float64)), 41,
                                                    L.ptr(cs), L.ptr(shf), L.ptr(st), L.stream_ptr()))
        cen, shift = pgm.psf_weighted_centroid(X, psf, tuple(int(v) for v in pgm.max_pixel(X, (c[0] + 1, c[1] - 2))))
        np.testing.assert_array_equal(cs.cpu().numpy()[0], cen)
        np.testing.assert_allclose(shf.cpu().numpy()[0], shift, rtol=0, atol=1e-6)
