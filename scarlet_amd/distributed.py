"""Scene sharding across the GPUs of one node.

Blend scenes are fully independent (no term of loss, Lipschitz constant, prox or
convergence test couples two scenes), so the only communication is moving data:
one scatter of the inputs from rank 0 and one gather of the fitted factors back --
no collective inside the iteration.  One process per GPU; `torch.distributed` with
the "nccl" backend (= RCCL over xGMI on ROCm) on GPUs, "gloo" in the CPU tests.
"""
import os

import numpy as np


def shard_range(n_scenes, rank, world):
    """Contiguous block of scene indices owned by `rank`: sizes differ by at most one."""
    base, rem = divmod(int(n_scenes), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def init_from_env(backend=None, timeout_s=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns
    (rank, world, local_rank).  world == 1 without env -> no process group.  `timeout_s` (default: environment
    SCARLET_DIST_TIMEOUT_S, else 300): rendezvous and collective timeout -- a rank that dies before it joins makes
    the others fail after this long instead of blocking until the job's time limit."""
    import datetime
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if timeout_s is None:
            timeout_s = float(os.environ.get("SCARLET_DIST_TIMEOUT_S", "300"))
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=float(timeout_s)))
    return rank, world, local


def _run_p2p(ops):
    """Post every transfer of a scatter / gather at once and wait for all of them: on RCCL one grouped call, so
    that the transfers to / from the seven peers of a node run concurrently on their seven xGMI links instead of
    one link at a time."""
    import torch.distributed as dist
    if not ops:
        return
    for req in dist.batch_isend_irecv(ops):
        req.wait()


def scatter_scenes(tensors, n_scenes, src=0):
    """Scatter scene-major tensors from `src` to all ranks.

    `tensors`: on `src` a list of tensors whose first axis is the global scene index
    (length n_scenes); on other ranks a list of (shape_without_scene_axis, dtype) is not
    needed -- shapes/dtypes are broadcast first.  Returns this rank's shards.
    Implemented with point-to-point sends (uneven shards allowed), ALL of them -- every tensor to every peer --
    posted together: on RCCL they run concurrently over the direct xGMI links from rank 0 to each peer."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return list(tensors)
    rank, world = dist.get_rank(), dist.get_world_size()
    meta = [[tuple(t.shape[1:]), str(t.dtype).replace("torch.", "")] for t in tensors] if rank == src else None
    box = [meta]
    dist.broadcast_object_list(box, src=src)
    meta = box[0]
    dev = tensors[0].device if rank == src else (
        torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu"))
    out, ops, keep = [], [], []
    lo, hi = shard_range(n_scenes, rank, world)
    for i, (shape, dtype) in enumerate(meta):
        dt = getattr(torch, dtype)
        if rank == src:
            for r in range(world):
                a, b = shard_range(n_scenes, r, world)
                if r == src:
                    out.append(tensors[i][a:b].clone())
                elif b > a:
                    part = tensors[i][a:b].contiguous()
                    keep.append(part)
                    ops.append(dist.P2POp(dist.isend, part, r))
        else:
            buf = torch.empty((hi - lo,) + tuple(shape), dtype=dt, device=dev)
            if hi > lo:
                ops.append(dist.P2POp(dist.irecv, buf, src))
            out.append(buf)
    _run_p2p(ops)
    return out


def gather_scenes(tensors, n_scenes, dst=0):
    """Gather per-rank scene-major tensors back to `dst` in global scene order (every tensor from every peer
    posted together, see _run_p2p).  Returns the concatenated tensors on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return list(tensors)
    rank, world = dist.get_rank(), dist.get_world_size()
    ops, parts_all, keep = [], [], []
    for t in tensors:
        t = t.contiguous()
        if dist.get_backend() == "gloo" and t.is_cuda:
            t = t.cpu()                    # gloo moves host memory only (CPU tests, single-GPU rehearsal)
        if rank == dst:
            parts = []
            for r in range(world):
                a, b = shard_range(n_scenes, r, world)
                if r == dst:
                    parts.append(t)
                else:
                    buf = torch.empty((b - a,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                    if b > a:
                        ops.append(dist.P2POp(dist.irecv, buf, r))
                    parts.append(buf)
            parts_all.append(parts)
        elif t.shape[0] > 0:
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, dst))
    _run_p2p(ops)
    return [torch.cat(parts, dim=0) for parts in parts_all] if rank == dst else None


def max_over_ranks(value):
    """MAX-reduce a python float over all ranks (the timing rule of bench.py)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return float(value)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
