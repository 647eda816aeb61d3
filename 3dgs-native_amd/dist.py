"""
View-parallel training support: one process per GPU, every rank holds the
full Gaussian set and renders its own camera view; the only exchange is ONE all-reduce per step over
the flat float32 gradient arena that backward() returns (`grads["_arena"]`, 59 floats per Gaussian:
mean3D | scale | rot | opacity | shs -- the five arrays the reference hands its optimizer,
train.py:1047-1051).  The reference itself is single-view, single-device (train.py:928).

backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist

ARENA_FLOATS = 59


def init_from_env(backend=None, device=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend=backend, **kw)
    return dist.get_rank(), dist.get_world_size()


def views_for_rank(num_views, rank, world_size):
    """Indices of the camera views rank `rank` renders: round-robin, so any world size divides the batch
    as evenly as possible and no view is rendered twice."""
    return list(range(rank, num_views, world_size))


def arena_views(arena, n):
    """Split a flat [59*n] arena into the five gradient arrays (views, no copies)."""
    o = [0, 3 * n, 6 * n, 10 * n, 11 * n, 59 * n]
    return {"dL_dmean3D": arena[o[0]:o[1]].view(n, 3), "dL_dscale": arena[o[1]:o[2]].view(n, 3),
            "dL_drot": arena[o[2]:o[3]].view(n, 4), "dL_dopacity": arena[o[3]:o[4]],
            "dL_dshs": arena[o[4]:o[5]].view(n * 16, 3)}


def reduce_gradients(arena, world_size=None, average=True):
    """In-place sum (or mean) of the gradient arena over all ranks: a single collective."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return arena
    ws = dist.get_world_size() if world_size is None else world_size
    if average and arena.is_cuda and dist.get_backend() == "nccl":
        # RCCL scales inside the collective (ncclAvg): no second 236 B/Gaussian pass over the arena
        dist.all_reduce(arena, op=dist.ReduceOp.AVG)
        return arena
    dist.all_reduce(arena, op=dist.ReduceOp.SUM)
    if average:
        arena.mul_(1.0 / ws)   # mean keeps densify_grad_threshold semantics (reference config.py:54)
    return arena
