#!/usr/bin/env python
"""Benchmark of the Blend.fit() hot path: proximal-gradient iterations/s on synthetic scenes.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c5] [--strong]

A "step" is one PGM iteration (Blend.fit inner loop: render [+ PSF convolution], loss gradient,
Lipschitz step, constraint pipeline, convergence flags) over the whole resident batch of scenes.

Workloads (`--config`, BASELINE.json `configs`):
  c2 (default) 10 000 scenes per GPU of 5 bands x 64 x 64, 4 sources, no PSF -- the configuration
               BASELINE.json's metric is quoted on (configs[3] on one GPU; configs[1] is the same
               scene shape at batch 1024).
  c3           4096 scenes per GPU of 5 x 128 x 128, 8 sources, per-band 41 x 41 PSF (FFT-convolution
               path, configs[2]).
  c5           64 scenes per GPU (= 512 over 8 GPUs) of 6 x 256 x 256, 30 overlapping sources,
               symmetry + monotonicity + L0 (configs[4]).
Multi-GPU: one process per GPU.  Run under torchrun (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in the
environment), or plainly as `python bench.py --gpus N`: with WORLD_SIZE unset and N > 1 this
process starts the N rank processes itself (before it touches the GPU) and relays rank 0's line.
Default = weak scaling: every rank owns its own `scenes` scenes, no collective in the timed
region.  `--strong` = BASELINE configs[3] literally: rank 0 generates `scenes` scenes in total,
scatters them over RCCL (timed separately), every rank fits its shard, results are gathered.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
"roofline" (dominant kernel class, HIP-event timed on the launch stream) and "cpu_baseline" (the
CPU oracle timed on this host's cores -- one core and all cores -- rank 0 at N=1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CLASS_NAMES = ("k_grad", "k_step", "k_source_update", "k_converge", "k_iterate", "psf_chain", "-", "-")

CONFIGS = {
    # scenes per GPU, bands, size, sources, psf, l0, min_sep, first scene index, distinct scenes generated
    "c2": dict(S=10000, B=5, H=64, W=64, K=4, psf=False, l0=None, min_sep=4, first=0, unique=None,
               cpu_scenes=96, cpu_iters=50,
               label="%d scenes/GPU of 5-band 64x64, 4 sources/scene, no PSF (BASELINE configs[3] shape; "
                     "configs[1] is the same at batch 1024)"),
    "c3": dict(S=4096, B=5, H=128, W=128, K=8, psf=True, l0=None, min_sep=4, first=300, unique=64,
               cpu_scenes=8, cpu_iters=25,
               label="%d scenes/GPU of 5-band 128x128, 8 sources/scene, per-band 41x41 PSF, FFT-convolution "
                     "render (BASELINE configs[2])"),
    "c5": dict(S=64, B=6, H=256, W=256, K=30, psf=False, l0=0.05, min_sep=3, first=5000, unique=8,
               cpu_scenes=4, cpu_iters=10,
               label="%d scenes/GPU of 6-band 256x256, 30 overlapping sources/scene, symmetry + monotonicity + "
                     "L0 (BASELINE configs[4]: 512 scenes over 8 GPUs)"),
}


def algorithmic_bytes_per_scene_iteration(B, K, H, W):
    """images read once, K morphs and SEDs read+written once."""
    return 4 * H * W * (B + 2 * K) + 8 * K * B


def shard_range(n_scenes, rank, world):
    """scarlet_amd.distributed.shard_range (not imported here: the package loads the HIP library, and
    this process forks its host workers first)"""
    base, rem = divmod(int(n_scenes), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def psf_setup(B):
    """SURVEY.md 8d PSF configs: observed PSF_b = integrated Gaussian sigma 1.2 + 0.15 b on 41 x 41,
    model PSF sigma 0.9; returns (obs_psfs, model_psf).  Runs in a worker process."""
    from scarlet_amd import synth
    obs = np.array([synth.gaussian_psf((41, 41), 1.2 + 0.15 * b) for b in range(B)])
    model = synth.gaussian_psf((41, 41), 0.9)
    return obs, model


# ----------------------------------------------------------------------------- host cores
def host_cores():
    """What this process may use of the host: CPUs in its affinity mask, physical cores among them (one per
    (package, core id): SMT siblings counted once), and the cgroup CPU quota if one is set.  The CPU leg runs one
    process per physical core, capped by the quota (processes beyond the quota would only time-slice)."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    cores, topo = set(), True
    for c in cpus:
        try:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % c
            cores.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        except OSError:
            topo = False
            break
    physical = len(cores) if topo and cores else len(cpus)
    quota = None
    try:                                           # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                       # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    use = physical if quota is None else max(1, min(physical, int(quota)))
    return dict(os_cpu_count=os.cpu_count(), affinity_cpus=len(cpus), physical_cores=physical,
                smt_detected=bool(topo and physical < len(cpus)), cgroup_cpu_quota=quota, cores_used=use)


# ----------------------------------------------------------------------------- host workers
def _gen_chunk(args):
    from scarlet_amd import synth
    start, count, kw = args
    d = synth.make_batch(start, count, **kw)
    return d["images"], d["centers"]


def generate_scenes(start, count, pool, workers, **kw):
    chunk = max(1, (count + 4 * workers - 1) // (4 * workers))
    jobs = [(start + o, min(chunk, count - o), kw) for o in range(0, count, chunk)]
    parts = pool.map(_gen_chunk, jobs) if pool is not None else [_gen_chunk(j) for j in jobs]
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])


def _cpu_worker(args):
    """CPU oracle on a few scenes: returns (scene_iterations, seconds spent in fit)."""
    from oracle import pgm
    from scarlet_amd import synth
    start, count, iters, kw, psf, l0 = args
    diff = obs = model = None
    if psf:
        obs, model = psf_setup(kw["B"])
        diff = pgm.match_psfs(obs, model[None])
    total, spent = 0, 0.0
    for i in range(count):
        scn = synth.make_scene(start + i, psfs=obs, **kw)
        extra = {}
        if psf:
            extra = dict(obs_psfs=obs, frame_psf=model[None])
        sc = pgm.make_extended_scene(scn["images"], scn["centers"], np.ones(kw["B"]) * 0.1, **extra)
        if psf:
            sc.diff_kernel = diff.astype(np.float32)
        if l0 is not None:
            for s in sc.sources:
                s.l0_thresh = l0
        t0 = time.perf_counter()
        pgm.fit(sc, iters, e_rel=0)
        spent += time.perf_counter() - t0
        total += iters
    return total, spent


def cpu_baseline(pool, workers, scenes_per_worker, iters, kw, psf, l0, first):
    """The CPU oracle (numpy + C sweep, one scene at a time exactly like the reference) on a bounded
    sample of the same workload: once on ONE core, once with one process per core."""
    from oracle import build as obuild
    obuild.build()
    t0 = time.perf_counter()
    # one worker busy, the others idle (this process itself must not load the HIP library before its workers exit)
    n1, s1 = pool.apply(_cpu_worker, ((first + 700000, scenes_per_worker, iters, kw, psf, l0),))
    jobs = [(first + 700100 + w * scenes_per_worker, scenes_per_worker, iters, kw, psf, l0) for w in range(workers)]
    res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    n = sum(r[0] for r in res)
    busy = max(r[1] for r in res)      # aggregate rate over the cores actually used
    return dict(value=n / busy, unit="scene-iterations/s", cores=workers, kind="port",
                one_core=dict(value=n1 / s1, unit="scene-iterations/s", cores=1),
                os_cpu_count=os.cpu_count(), host=host_cores(),
                sample="%d scenes x %d iterations of the same workload per process, CPU oracle (numpy + C sweep), "
                       "%d processes = one per physical core this job may use (`value`) and 1 process (one_core), "
                       "fit() time only (wall %.1f s)" % (scenes_per_worker, iters, workers, wall))


# ----------------------------------------------------------------------------- launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n, argv, limit_s=None):
    """`python bench.py --gpus N` without torchrun: start N fresh rank processes (this process has not touched the
    GPU and never will), watch ALL of them, relay rank 0's output.  The first rank that exits non-zero (or the
    time limit, SCARLET_BENCH_LIMIT_S, default 1500 s) ends the run: the other ranks -- which would otherwise block
    in the rendezvous or in a barrier until the job's own limit -- are terminated (fresh processes, never
    re-launched), the failing rank's stderr tail is printed, and the exit code is non-zero."""
    import tempfile
    if limit_s is None:
        limit_s = float(os.environ.get("SCARLET_BENCH_LIMIT_S", "1500"))
    port = _free_port()
    procs, logs = [], []
    tmp = tempfile.mkdtemp(prefix="scarlet_bench_")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = open(os.path.join(tmp, "rank%d.out" % r), "w+b")
        err = open(os.path.join(tmp, "rank%d.err" % r), "w+b")
        logs.append((out, err))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out, stderr=err))
    t0 = time.time()
    failed, why = None, ""
    while True:
        rcs = [p.poll() for p in procs]
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed, why = bad[0], "rank %d exited with code %d" % (bad[0], rcs[bad[0]])
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.time() - t0 > limit_s:
            failed = next(r for r, rc in enumerate(rcs) if rc is None)
            why = "time limit of %.0f s reached, rank %d still running" % (limit_s, failed)
            break
        time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t1 = time.time()
        for p in procs:
            try:
                p.wait(timeout=max(0.1, 10 - (time.time() - t1)))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    def tail(f, nbytes=4000):
        f.flush(); f.seek(0, 2); size = f.tell(); f.seek(max(0, size - nbytes))
        return f.read().decode(errors="replace")
    sys.stdout.write(tail(logs[0][0], 1 << 20))
    sys.stdout.flush()
    if failed is not None:
        sys.stderr.write("bench.py: %s; the other ranks were terminated.  stderr tail of rank %d:\n%s\n"
                         % (why, failed, tail(logs[failed][1])))
    else:
        for r in range(n):
            sys.stderr.write(tail(logs[r][1]))
    for out, err in logs:
        out.close(); err.close()
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return 1 if failed is not None else 0


# ----------------------------------------------------------------------------- the other BASELINE configs
def run_other_configs(args):
    """The default single-GPU run also times BASELINE configs[2] (c3) and configs[4] (c5), each in a fresh child
    process started BEFORE this process touches the GPU (same script, `--config c3|c5`, a few seconds each), and
    embeds their JSON lines under "other_configs".  A failure is recorded, it does not fail the headline."""
    out = {}
    for cfg, steps, warm in (("c3", min(args.steps, 10), min(args.warmup, 3)), ("c5", min(args.steps, 10), min(args.warmup, 3))):
        cmd = [sys.executable, os.path.abspath(__file__), "--config", cfg, "--steps", str(steps), "--warmup", str(warm)]
        if args.no_cpu:
            cmd.append("--no-cpu")
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, capture_output=True, timeout=900)
            line = [l for l in p.stdout.decode().splitlines() if l.startswith('{"metric"')]
            if p.returncode == 0 and line:
                d = json.loads(line[-1])
                d["bench_wall_s"] = time.perf_counter() - t0
                out[cfg] = d
            else:
                out[cfg] = {"error": "exit code %d: %s" % (p.returncode, p.stderr.decode()[-400:])}
        except subprocess.TimeoutExpired:
            out[cfg] = {"error": "timed out after 900 s"}
    return out


# ----------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--scenes", type=int, default=None, help="scenes per GPU (total scenes with --strong)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: `scenes` in total, generated on rank 0 and scattered over RCCL")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--clock-warmup-ms", type=float, default=60.0,
                    help="untimed device clock warm-up on a scratch copy of the batch before the W warmup steps (0: none)")
    ap.add_argument("--no-other", action="store_true",
                    help="headline workload only (the default single-GPU c2 run also times c3 and c5 in child processes)")
    ap.add_argument("--cpu-scenes", type=int, default=None, help="oracle scenes per host process")
    ap.add_argument("--cpu-iters", type=int, default=None)
    ap.add_argument("--no-symmetric", action="store_true", help="ablation: drop the symmetry constraint")
    ap.add_argument("--no-monotonic", action="store_true", help="ablation: drop the monotonicity constraint")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (profiles/)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    # fault injection for tests/test_dist_gloo.py (the launcher's watchdog): a rank that dies / hangs before the rendezvous
    if os.environ.get("SCARLET_BENCH_FAIL_RANK") == str(rank) and world > 1:
        raise RuntimeError("injected failure on rank %d (SCARLET_BENCH_FAIL_RANK)" % rank)
    if os.environ.get("SCARLET_BENCH_HANG_RANK") in (str(rank), "all") and world > 1:
        time.sleep(3600)
    if world != args.gpus:
        print("error: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus), file=sys.stderr)
        sys.exit(2)
    other = None
    if (args.config == "c2" and world == 1 and not args.no_other and not args.strong and args.scenes is None
            and not args.no_symmetric and not args.no_monotonic):
        other = run_other_configs(args)
    cfg = CONFIGS[args.config]
    B, H, W, K = cfg["B"], cfg["H"], cfg["W"], cfg["K"]
    S_arg = args.scenes if args.scenes is not None else cfg["S"]
    kw = dict(B=B, H=H, W=W, K=K, min_sep=cfg["min_sep"])
    obs_psfs = model_psf = None

    # ---- host-side work first, in forked workers, BEFORE this process loads the HIP library
    import multiprocessing as mp
    # one worker process per physical core this job may use (affinity mask, SMT siblings once, cgroup quota),
    # shared between the ranks of a multi-GPU run; SCARLET_BENCH_WORKERS overrides
    hc = host_cores()
    workers = max(1, hc["cores_used"] // max(1, world))
    if os.environ.get("SCARLET_BENCH_WORKERS"):
        workers = max(1, int(os.environ["SCARLET_BENCH_WORKERS"]))
    pool = mp.get_context("fork").Pool(workers)
    cpu = None
    images = centers = None
    try:
        if cfg["psf"]:
            obs_psfs, model_psf = pool.apply(psf_setup, (B,))
        if rank == 0 and world == 1 and not args.no_cpu:
            cpu = cpu_baseline(pool, workers, args.cpu_scenes or cfg["cpu_scenes"], args.cpu_iters or cfg["cpu_iters"],
                               kw, cfg["psf"], cfg["l0"], cfg["first"])
        t0 = time.perf_counter()
        if args.strong:
            S_total = S_arg
            lo, hi = shard_range(S_total, rank, world)
            S = hi - lo
            n_gen, start = (S_total, cfg["first"]) if rank == 0 else (0, 0)
        else:
            S, S_total = S_arg, S_arg * world
            n_gen, start = S, cfg["first"] + rank * S
        if n_gen:
            uniq = min(n_gen, cfg["unique"] or n_gen)
            gkw = dict(kw, psfs=obs_psfs) if cfg["psf"] else kw
            images, centers = generate_scenes(start, uniq, pool, workers, **gkw)
            if uniq < n_gen:                 # large frames: a few distinct scenes, tiled (scenes are independent)
                reps = (n_gen + uniq - 1) // uniq
                images = np.tile(images, (reps, 1, 1, 1))[:n_gen]
                centers = np.tile(centers, (reps, 1, 1))[:n_gen]
        t_gen = time.perf_counter() - t0
    finally:
        pool.close()
        pool.join()

    # ---- device
    import torch
    from scarlet_amd import _lib, distributed
    from scarlet_amd.batch import BlendBatch
    _lib.require_gpu()
    # SCARLET_BENCH_REHEARSE=1: all ranks share GPU 0 and talk over gloo -- a single-GPU rehearsal of
    # the N > 1 control flow (the real N > 1 runs use one GPU per rank and RCCL)
    rehearse = bool(os.environ.get("SCARLET_BENCH_REHEARSE"))
    torch.cuda.set_device(local if world > 1 and not rehearse else 0)
    backend = ("gloo" if rehearse else "nccl") if world > 1 else None
    distributed.init_from_env(backend)
    dev = torch.device("cuda", torch.cuda.current_device())

    t_scatter = 0.0
    if args.strong and world > 1:
        # configs[3]: rank 0 holds all scenes; one scatter of images + centres over RCCL (xGMI)
        if rank == 0:
            tens = [torch.as_tensor(images).to(dev), torch.as_tensor(centers).to(dev)]
            if rehearse:
                tens = [t.cpu() for t in tens]
        else:
            tens = None
        torch.cuda.synchronize()
        distributed.barrier()
        t0 = time.perf_counter()
        images_t, centers_t = distributed.scatter_scenes(tens, S_total)
        torch.cuda.synchronize()
        distributed.barrier()
        t_scatter = distributed.max_over_ranks(time.perf_counter() - t0)
        images, centers = images_t.to(dev), centers_t.to(dev)

    bkw = dict(mse_capacity=args.steps + args.warmup + 1, symmetric=not args.no_symmetric,
               monotonic=not args.no_monotonic, l0_thresh=cfg["l0"])
    if cfg["psf"]:
        bkw["centroid_weight"] = model_psf.astype(np.float32)
    sed_scale = diff = None
    if cfg["psf"]:
        from scarlet_amd import fft as fftmod
        diff = np.asarray(fftmod.match_psfs(fftmod.Fourier(obs_psfs.astype(np.float32)),
                                            fftmod.Fourier(model_psf[None].astype(np.float32))).image, dtype=np.float32)
        sed_scale = (model_psf.max() / obs_psfs.max(axis=(1, 2))).astype(np.float32)

    def make_batch():
        bb = BlendBatch(images, centers, **bkw)
        if diff is not None:
            bb.set_diff_kernel(diff)
        bb.init_extended(np.ones(B, dtype=np.float32) * 0.1, sed_scale=sed_scale)
        return bb

    # Device clock warm-up, NOT part of the W warmup steps and not timed: the GPU has idled through the host-side legs
    # (scene generation, CPU baseline) and W = 5 iterations are ~3 ms -- the timed region would start at idle clocks
    # (measured: 0.66 - 0.68 ms per step after 5 warmup iterations, 0.64 after 50, 0.63 in a 200-step run).  The same
    # iterations run on a SCRATCH copy of the batch for >= clock_warmup_ms, the copy is freed, then the contract's
    # W + K steps run on the real batch.
    clock_warm = {"iterations": 0, "ms": 0.0}
    if args.clock_warmup_ms > 0:
        scratch = make_batch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < args.clock_warmup_ms and clock_warm["iterations"] < 400:
            scratch.fit(10, e_rel=0, check_every=0)
            torch.cuda.synchronize()
            clock_warm["iterations"] += 10
        clock_warm["ms"] = (time.perf_counter() - t0) * 1e3
        del scratch
        torch.cuda.empty_cache()
    batch = make_batch()
    torch.cuda.synchronize()
    if args.warmup > 0:
        batch.fit(args.warmup, e_rel=0, check_every=0)
    torch.cuda.synchronize()
    distributed.barrier()
    _lib.check(_lib.lib.scarlet_profile_begin(args.steps))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    launched = batch.fit(args.steps, e_rel=0, check_every=0)
    torch.cuda.synchronize()
    distributed.barrier()
    elapsed = time.perf_counter() - t0
    import ctypes
    ms = (ctypes.c_double * 8)()
    cnt = (ctypes.c_int64 * 8)()            # iterations covered by the class's launches
    raw = (ctypes.c_int64 * 8)()            # kernel launches (k_fit2x: one launch covers several iterations)
    _lib.check(_lib.lib.scarlet_profile_end_ex(ms, cnt, raw))
    assert launched == args.steps
    elapsed = distributed.max_over_ranks(elapsed)
    n_active = int(batch.active.sum().item())
    status_bad = int((batch.status != 0).sum().item())

    # results travel back to rank 0 over RCCL (outside the fit-only timed region; timed on its own)
    torch.cuda.synchronize()
    distributed.barrier()
    t0 = time.perf_counter()
    payload = [batch.sed_current, batch.mse_buf[:, :args.steps + args.warmup]]
    if args.strong:
        payload.append(batch.morph_current)
    gathered = distributed.gather_scenes(payload, S_total if args.strong else S * world)
    torch.cuda.synchronize()
    distributed.barrier()
    t_gather = distributed.max_over_ranks(time.perf_counter() - t0)
    if rank != 0:
        return
    total_scene_iters = float(S_total) * args.steps
    value = total_scene_iters / elapsed
    bytes_unit = algorithmic_bytes_per_scene_iteration(B, K, H, W)
    S0 = S                                               # scenes per launch on rank 0
    per_class = {CLASS_NAMES[i]: ms[i] / cnt[i] for i in range(8) if cnt[i]}
    dom = int(np.argmax([ms[i] for i in range(8)]))
    launches_per_iter = max(1, int(round(cnt[dom] / float(args.steps))))
    # scarlet_fit runs a large PSF batch as two half-batch pipelines on two streams: a launch then covers half the
    # scenes, and its duration overlaps the other stream's kernels
    pipelines = int(_lib.lib.scarlet_batch_pipelines(ctypes.byref(batch._c)))
    S_launch = S0 // pipelines if launches_per_iter % pipelines == 0 else S0
    avg_ms = ms[dom] / max(1, cnt[dom]) * (launches_per_iter // pipelines if S_launch != S0 else launches_per_iter)
    achieved = bytes_unit * S_launch / (avg_ms * 1e-3) / 1e9
    it_ms = 1e3 * elapsed / args.steps
    mse = gathered[1].cpu().numpy()
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so
    # the figure is the one measured with rocprofv3 --pmc passes of this same command and committed
    # under profiles/ (per launch of the same kernel at the same scenes-per-launch), else null.
    kernel_label = CLASS_NAMES[dom]
    if dom == 4:
        # the fused iteration has two variants (scarlet_hip.hip launch_fused); name the one that ran
        kernel_label = "k_iterate2<4,5,0>" if (K <= 4 and B <= 5 and not os.environ.get("SCARLET_FUSED_V1")) else "k_iterate"
        if kernel_label.startswith("k_iterate2") and (K, B, H, W) == (4, 5, 64, 64) and not os.environ.get("SCARLET_NO_EXACT"):
            kernel_label = "k_iterate2<4,5,64>"        # the exact-shape instance (default pipeline, unit weights)
            if raw[dom] < cnt[dom]:
                kernel_label = "k_fit2x"               # the same iteration, several per launch (fused2.h)
    elif dom == 5:
        kernel_label = "k_psf_conv (render + adjoint, LDS-resident FFT)" if not os.environ.get("SCARLET_PSF_HIPFFT") else "psf_chain (hipFFT)"
    elif dom == 2:
        # frames beyond 64 x 64: the constraint pipeline on the box around each peak (boxupdate.h)
        kernel_label = "k_source_update_box (constraints on the 63 x 63 box; class time includes the listed 127 x 127 pass)" \
            if (H > 64 or W > 64) and not os.environ.get("SCARLET_NO_BOX") else "k_source_update"
    traffic, traffic_src = args.traffic_bytes, "--traffic-bytes" if args.traffic_bytes else None
    if traffic is None:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
            try:
                pm = json.load(open(f))
            except Exception:
                continue
            if kernel_label.split(" ")[0].replace(",", ", ") in pm.get("kernel", "") and pm.get("scenes_per_launch") == S_launch:
                if "hbm_bytes_per_launch_and_iteration" in pm:      # k_fit2x: measured on a launch of several iterations
                    traffic = pm["hbm_bytes_per_launch_and_iteration"] * (cnt[dom] // max(1, raw[dom]))
                else:
                    traffic = pm["hbm_bytes_per_launch"]
                traffic_src = os.path.relpath(f, ROOT)
                break
    metric = "PGM iters/sec on 10k 5-band 64x64 scenes" if args.config == "c2" else \
        "PGM iters/sec (%s)" % args.config
    out = {
        "metric": metric,
        "value": value,
        "unit": "scene-iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": it_ms,
        "higher_is_better": True,
        "scaling": "strong" if args.strong else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": cfg["label"] % S0 + (" -- STRONG scaling: %d scenes in total" % S_total if args.strong else ""),
                   "config": args.config, "scenes_per_gpu": S0, "scenes_total": S_total,
                   "bands": B, "height": H, "width": W, "sources": K,
                   "unique_scenes": int(min(S0, cfg["unique"] or S0)),
                   "parallelism": "scenes sharded, %d rank(s), no collective in the iteration" % world,
                   "batch_iterations_per_s": args.steps / elapsed,
                   "active_scenes_after_timed_region": n_active, "scenes_with_status": status_bad,
                   "mean_loss_first_last": [float(mse[:, 0].mean()), float(mse[:, -1].mean())],
                   "device_clock_warmup": dict(clock_warm, on="a scratch copy of the batch, freed before the W warmup steps; "
                                                               "untimed, not counted in `warmup`"),
                   "host_scene_generation_s": t_gen,
                   "scatter_s": t_scatter, "gather_s": t_gather,
                   "end_to_end_scene_iterations_per_s": total_scene_iters / (elapsed + t_scatter + t_gather)},
        "roofline": {"bound": "hbm", "kernel": kernel_label, "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": bytes_unit * S_launch * (cnt[dom] // max(1, raw[dom])),
                     "algorithmic_bytes_per_launch_and_iteration": bytes_unit * S_launch,
                     "avg_launch_ms": avg_ms * (cnt[dom] // max(1, raw[dom])), "avg_ms_per_iteration_in_launch": avg_ms,
                     "iterations_per_launch": cnt[dom] // max(1, raw[dom]), "launches": int(raw[dom]),
                     "launches_per_iteration_in_class": launches_per_iter, "scenes_per_launch": S_launch,
                     "pipelines": pipelines,
                     "per_class_avg_ms": per_class,
                     "whole_iteration_frac": bytes_unit * S0 / (it_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
    }
    if dom != 4:
        # An iteration of several kernels (configs 3 and 5): the algorithmic bytes belong to the WHOLE iteration, so the
        # roofline figure is the whole-iteration one; the dominant kernel class is reported beside it with its own
        # launch time (with two pipelines that time overlaps the other stream's kernels) and, when profiles/ holds its
        # PMC traffic, the bandwidth of ITS OWN bytes.
        rf = out["roofline"]
        whole = rf["whole_iteration_frac"]
        rf["dominant_kernel"] = {"kernel": kernel_label, "avg_launch_ms": rf["avg_launch_ms"], "scenes_per_launch": S_launch,
                                 "launches_per_iteration_in_class": launches_per_iter,
                                 "own_hbm_bytes_per_launch": traffic, "traffic_source": traffic_src,
                                 "own_bandwidth_GBs": (traffic / (rf["avg_launch_ms"] * 1e-3) / 1e9) if traffic else None,
                                 "algorithmic_bytes_over_launch_time_frac": rf["frac"]}
        rf["kernel"] = "whole iteration (all kernel classes: %s); dominant class: %s" % (
            ", ".join(sorted(per_class)), kernel_label.split(" ")[0])
        rf["achieved"] = whole * HBM_PEAK_GBS
        rf["frac"] = whole
        rf["traffic"] = None
        rf["traffic_source"] = None
        rf["algorithmic_bytes_per_launch"] = bytes_unit * S0
        rf["avg_launch_ms"] = it_ms
    if other is not None:
        out["other_configs"] = other
    if cpu is not None:
        out["cpu_baseline"] = cpu
        out["config"]["gpu_over_cpu_%d_cores" % cpu["cores"]] = value / cpu["value"]
        out["config"]["gpu_over_cpu_one_core"] = value / cpu["one_core"]["value"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
