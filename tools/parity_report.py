"""Parity report for the headline workload (BASELINE metric, second half: "SED/morph rel-err vs ref").

    python tools/parity_report.py --stage gpu [--scenes 256] [--iters 50] && python tools/parity_report.py --stage cpu

Runs N synthetic 5-band 64x64 / 4-source scenes through the HIP engine (one batch) and through the CPU
oracle (oracle/pgm.py, one scene at a time in worker processes, from the SAME initial state the device
initialisation produced), then reports

  * max-norm relative error of the final SED / morphology / loss history per scene (SURVEY.md 8d:
    max|a - b| / max|b|), fixed iteration count, e_rel = 0;
  * bit-exact items: pixel centres after the last iteration;
  * a second run with e_rel = 1e-3: per-scene iteration counts and convergence flags.

TEST INFRASTRUCTURE (imports oracle/): evidence for profiles/, not part of the product.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


def _cpu(args):
    from oracle import pgm
    images, sed0, morph0, cen0, sh0, iters, e_rel = args
    sc = pgm.scene_from_state(images, sed0, morph0, cen0, sh0)
    pgm.fit(sc, iters, e_rel=e_rel)
    flags = [int(s.flags) for s in sc.sources] if hasattr(sc.sources[0], "flags") else None
    return (np.array([s.sed for s in sc.sources]), np.array([s.morph for s in sc.sources]),
            np.array(sc.mse), np.array([s.center for s in sc.sources]), len(sc.mse), flags)


def stage_gpu(a, runs):
    """GPU fits; everything the CPU stage needs goes to an .npz (no worker processes here: a process tree
    with many children does not get the device on the GPU boxes)."""
    import torch
    from scarlet_amd import synth
    from scarlet_amd.batch import BlendBatch
    S = a.scenes
    scenes = [synth.make_scene(a.first + i) for i in range(S)]
    images = np.stack([s["images"] for s in scenes]); centers = np.stack([s["centers"] for s in scenes])
    out = {"images": images}
    for tag, iters, e_rel in runs:
        b = BlendBatch(images, centers, mse_capacity=iters + 1)
        b.init_extended(np.ones(5) * 0.1)
        out[tag + "_sed0"] = b.sed_current.cpu().numpy(); out[tag + "_morph0"] = b.morph_current.cpu().numpy()
        out[tag + "_cen0"] = b.centers.cpu().numpy(); out[tag + "_sh0"] = b.shifts.cpu().numpy()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.fit(iters, e_rel=e_rel)
        torch.cuda.synchronize()
        out[tag + "_t_gpu"] = np.array(time.perf_counter() - t0)
        out[tag + "_sed1"] = b.sed_current.cpu().numpy(); out[tag + "_morph1"] = b.morph_current.cpu().numpy()
        out[tag + "_cen1"] = b.centers.cpu().numpy(); out[tag + "_it"] = b.it.cpu().numpy()
        out[tag + "_flags"] = b.flags.cpu().numpy(); out[tag + "_mse"] = b.mse_buf[:, :iters].cpu().numpy()
    np.savez(a.state, **out)


def stage_cpu(a, runs):
    from oracle import build as obuild
    obuild.build()
    g = np.load(a.state)
    images = g["images"]; S = images.shape[0]
    pool = mp.get_context("fork").Pool(min(16, os.cpu_count() or 1))
    report = {"workload": "%d scenes of 5-band 64x64, 4 sources (synthetic scenes %d..%d)" % (S, a.first, a.first + S - 1),
              "metric": "max|gpu - cpu| / max|cpu| per scene and array (SURVEY.md 8d); CPU = oracle/pgm.py from the device's initial state",
              "tolerance": 1e-5}
    for tag, iters, e_rel in runs:
        t0 = time.perf_counter()
        ref = pool.map(_cpu, [(images[i], g[tag + "_sed0"][i], g[tag + "_morph0"][i], g[tag + "_cen0"][i], g[tag + "_sh0"][i],
                               iters, e_rel) for i in range(S)])
        t_cpu = time.perf_counter() - t0
        its = g[tag + "_it"]; flags = g[tag + "_flags"]; mse = g[tag + "_mse"]
        e_sed = [rel_err(g[tag + "_sed1"][i], ref[i][0]) for i in range(S)]
        e_morph = [rel_err(g[tag + "_morph1"][i], ref[i][1]) for i in range(S)]
        same_it = [int(its[i]) == ref[i][4] for i in range(S)]
        e_mse = [rel_err(mse[i][:ref[i][4]], ref[i][2]) for i in range(S) if same_it[i]]
        r = {"iterations": iters, "e_rel": e_rel,
             "sed_rel_err_max": max(e_sed), "sed_rel_err_median": float(np.median(e_sed)),
             "morph_rel_err_max": max(e_morph), "morph_rel_err_median": float(np.median(e_morph)),
             "loss_history_rel_err_max": max(e_mse) if e_mse else None,
             "centres_bit_exact": bool(all(np.array_equal(g[tag + "_cen1"][i], ref[i][3]) for i in range(S))),
             "iteration_counts_equal": int(sum(same_it)), "iteration_counts_total": S,
             "iterations_min_max": [int(its.min()), int(its.max())],
             "flags_equal": int(sum(list(flags[i]) == ref[i][5] for i in range(S))),
             "gpu_fit_seconds": float(g[tag + "_t_gpu"]), "cpu_oracle_seconds_16proc": t_cpu,
             "morph_rel_err_worst_scenes": [[int(i), e_morph[i]] for i in np.argsort(e_morph)[::-1][:5]],
             "scenes_above_tolerance": int(sum(max(e_sed[i], e_morph[i]) > 1e-5 for i in range(S)))}
        report[tag] = r
    pool.close(); pool.join()
    json.dump(report, open(a.out, "w"), indent=1)
    print(json.dumps(report, indent=1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", choices=("gpu", "cpu"), required=True,
                    help="run `--stage gpu` and then `--stage cpu` (two processes)")
    ap.add_argument("--scenes", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--first", type=int, default=40000, help="index of the first synthetic scene")
    ap.add_argument("--state", default=os.path.join(ROOT, "gpurun_out", "parity_state.npz"))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_report.json"))
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    runs = (("fixed", a.iters, 0.0), ("ragged", 200, 1e-3))
    (stage_gpu if a.stage == "gpu" else stage_cpu)(a, runs)


if __name__ == "__main__":
    main()
