"""-m gpu: long-run and converged-regime parity for BASELINE configs 3 and 5, the counterparts of
tests/test_gpu_parity_long.py for the two workloads that do not run the fused kernel:

  config 3  5 x 128 x 128, 8 sources, per-band 41 x 41 PSF: `k_psf_conv` + the three-pass PSF iteration + the box
            constraint kernels; 16 scenes x 50 iterations at e_rel = 0, and a converged run (e_rel = 1e-3) through
            scarlet_fit's TWO-PIPELINE loop (batches >= 1024 scenes run as two half-batches on two streams): the 16
            distinct scenes tiled to 1024, every copy compared with the oracle's result of its scene.
  config 5  6 x 256 x 256, 30 overlapping sources, L0: the many-component kernels (`k_bigk_*`) + the box kernels on
            256 x 256 planes; 4 scenes x 30 iterations at e_rel = 0 and one converged run (e_rel = 1e-2).

Iteration counts, flags and centres bit-exact; sed / morph / loss history <= 1e-5 max-norm relative (north_star);
the threshold exemption of tests/parity_common.py with a fixed count of 1 scene per test, logged.
"""
import multiprocessing as mp
import os

import numpy as np
import pytest

from conftest import rel_err
import parity_common as pc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
TOL = pc.TOL


@pytest.fixture(scope="module")
def env():
    import scarlet_amd
    scarlet_amd._lib.require_gpu()
    from oracle import build as obuild
    obuild.build()
    pool = mp.get_context("spawn").Pool(min(16, os.cpu_count() or 1))
    yield scarlet_amd, pool
    pool.close(); pool.join()


def converged_run(scarlet, wl, images, centers, pool, max_iter, e_rel, tile=1):
    """fit to convergence; `tile` > 1: the scenes repeated `tile` times in one batch (scenes are independent, every
    copy must end exactly like the first and like the oracle's run of its scene)"""
    S = len(images)
    st0, g = pc.gpu_fit(scarlet, wl, np.tile(images, (tile, 1, 1, 1)), np.tile(centers, (tile, 1, 1)), max_iter, e_rel)
    ref = pool.map(pc.oracle_fit, [(images[i], st0[0][i], st0[1][i], st0[2][i], st0[3][i], max_iter, e_rel, np.float32,
                                    wl.oracle_kwargs()) for i in range(S)])
    its = np.array([r[4] for r in ref])
    assert int(np.abs(g["status"]).sum()) == 0
    for c in range(tile):
        sl = slice(c * S, (c + 1) * S)
        np.testing.assert_array_equal(g["it"][sl], its)
        if c:
            for key in ("sed", "morph", "cen", "flags"):
                np.testing.assert_array_equal(g[key][sl], g[key][:S])
    for i in range(S):
        np.testing.assert_array_equal(g["cen"][i], ref[i][3])
        np.testing.assert_array_equal(g["flags"][i], ref[i][5])
        assert rel_err(g["sed"][i], ref[i][0]) <= TOL, i
        assert rel_err(g["morph"][i], ref[i][1]) <= TOL, i
        assert rel_err(g["mse"][i][:its[i]], ref[i][2]) <= TOL, i
    return its


# ------------------------------------------------------------------ config 3
@pytest.fixture(scope="module")
def c3(env):
    wl = pc.Workload(B=5, H=128, W=128, K=8, psf=True)
    images, centers = wl.scenes(360, 16)
    return wl, images, centers


def test_config3_fifty_iterations_16_scenes(env, c3):
    scarlet, pool = env
    wl, images, centers = c3
    pc.check_fixed_iterations(scarlet, wl, images, centers, pool, 50, 1, "config 3 shape, 50 iterations x 16 scenes")


def test_config3_converged_through_the_two_pipeline_loop(env, c3):
    scarlet, pool = env
    wl, images, centers = c3
    from scarlet_amd import _lib
    import ctypes
    b = wl.batch(scarlet, np.tile(images, (64, 1, 1, 1)), np.tile(centers, (64, 1, 1)), 4)
    assert int(_lib.lib.scarlet_batch_pipelines(ctypes.byref(b._c))) == 2      # the path under test
    del b
    its = converged_run(scarlet, wl, images, centers, pool, 120, 1e-3, tile=64)
    assert len(np.unique(its)) > 1 and its.max() < 120                          # ragged, and it converged
    print("config 3 converged run: iteration counts", its.tolist())


# ------------------------------------------------------------------ config 5
@pytest.fixture(scope="module")
def c5(env):
    wl = pc.Workload(B=6, H=256, W=256, K=30, l0=0.05, min_sep=3)
    images, centers = wl.scenes(5400, 4)
    return wl, images, centers


def test_config5_thirty_iterations_4_scenes(env, c5):
    scarlet, pool = env
    wl, images, centers = c5
    pc.check_fixed_iterations(scarlet, wl, images, centers, pool, 30, 1, "config 5 shape, 30 iterations x 4 scenes")


def test_config5_converged_run(env, c5):
    scarlet, pool = env
    wl, images, centers = c5
    its = converged_run(scarlet, wl, images, centers, pool, 80, 1e-2)
    assert its.max() < 80
    print("config 5 converged run: iteration counts", its.tolist())
