"""CPU, world_size 2, gloo: the N>1 scene scatter / gather path (no GPU needed)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, n_scenes, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from scarlet_amd import distributed
    distributed.init_from_env("gloo")
    if rank == 0:
        full = [torch.arange(n_scenes * 3 * 4, dtype=torch.float32).reshape(n_scenes, 3, 4),
                torch.arange(n_scenes * 2, dtype=torch.int32).reshape(n_scenes, 2)]
    else:
        full = None
    mine = distributed.scatter_scenes(full, n_scenes)
    lo, hi = distributed.shard_range(n_scenes, rank, world)
    ok = mine[0].shape[0] == hi - lo and float(mine[0][0, 0, 0]) == lo * 12 and mine[1].dtype == torch.int32
    # "fit": a per-scene function of the inputs, then gather back in global order
    out = [mine[0].sum(dim=(1, 2)), mine[1][:, 0].to(torch.float32)]
    g = distributed.gather_scenes(out, n_scenes)
    t = distributed.max_over_ranks(float(rank + 1))
    distributed.barrier()
    if rank == 0:
        ref = torch.arange(n_scenes * 12, dtype=torch.float32).reshape(n_scenes, 12).sum(dim=1)
        ok = ok and bool(torch.equal(g[0], ref)) and bool(torch.equal(g[1], torch.arange(n_scenes) * 2.0))
        ok = ok and t == float(world)
    else:
        ok = ok and g is None and t == float(world)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_scenes", [7, 10])
def test_scatter_gather_world2(n_scenes):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_scenes, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert results == {0: True, 1: True}


def test_launcher_ends_the_run_when_a_rank_dies_before_the_rendezvous():
    """bench.py's own launcher (`python bench.py --gpus 2` without torchrun): rank 1 raises before
    init_process_group, rank 0 would wait for it for ever.  The launcher must notice, terminate rank 0, print the
    failing rank's stderr and return non-zero within seconds -- not at the job's time limit."""
    import subprocess
    import time
    env = dict(os.environ, SCARLET_BENCH_FAIL_RANK="1", SCARLET_BENCH_HANG_RANK="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--scenes", "4", "--no-cpu"], env=env, capture_output=True, timeout=120)
    took = time.time() - t0
    assert p.returncode != 0
    assert took < 60, took
    err = p.stderr.decode()
    assert "rank 1 exited with code" in err and "injected failure on rank 1" in err


def test_launcher_time_limit():
    """ranks that never finish: the launcher's own limit ends the run and says which rank was still running"""
    import subprocess
    env = dict(os.environ, SCARLET_BENCH_HANG_RANK="all")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    code = "import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.self_launch(2, ['--gpus', '2', '--scenes', '4', '--no-cpu'], limit_s=3))" % ROOT
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, timeout=120)
    assert p.returncode != 0 and "time limit" in p.stderr.decode(), p.stderr.decode()[-500:]
