#!/usr/bin/env python3
"""One rank of the multi-process tests, started by 3dgs-native_amd/launch.py::launch_ranks (the same launcher that
`bench.py --gpus N` and `examples/train.py --gpus N` use).  usage: dist_worker.py MODE OUTDIR

  cpu   gloo on CPU tensors: dist.reduce_gradients + dist.exchange_factored, result per rank -> OUTDIR/rank<r>.npz
  fail  rank 1 exits with code 3 before joining the group (the launcher must stop rank 0 and report 3)
  gpu   config #4 in miniature on ONE GPU (every rank on cuda:0, gloo through host memory): rank r renders Lego train
        frame r of a small seeded scene, backward(sh_gradient="factored", on_payload=FactoredExchange.start_gather),
        FactoredExchange.finish -> the view-averaged optimizer gradients, saved per rank
  nccl  the same step with one GPU per rank (cuda:LOCAL_RANK) over RCCL, the factored exchange AND the dense 59-float
        all-reduce of the same views (keys prefixed "dense_"); needs torch.cuda.device_count() >= WORLD_SIZE
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GPU_CASE = dict(n=6000, scale_median=0.04, scale_sigma=0.6, seed=11, size=208, degree=3)


def gpu_case_scene(gsr):
    c = GPU_CASE
    sc = gsr.scenes.synthetic_scene(c["n"], c["scale_median"], c["scale_sigma"], c["seed"])
    sc["shs"][::3] *= 4.0      # push some colours into the clamp: dL_drgb != dL_dcolor there
    return sc


def gpu_case_view(gsr, scene, frame):
    """(forward kwargs, backward kwargs, camera) of Lego train frame `frame` for the gpu case; needs cuda."""
    import torch
    from conftest import backward_kwargs, lego_camera, render_kwargs
    size = GPU_CASE["size"]
    cam = lego_camera(gsr.cameras, frame=frame, width=size, height=size)
    fkw = render_kwargs(scene, cam, degree=GPU_CASE["degree"])
    img, depth, buf = gsr.render_gaussians(**fkw)
    dpix = torch.as_tensor(np.random.default_rng(100 + frame).normal(0, 1, (size, size, 3)).astype(np.float32) / (size * size * 3)).cuda()
    return fkw, backward_kwargs(scene, cam, fkw, buf, dpix), cam


def main():
    mode, outdir = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if mode == "fail":
        if rank == 1:
            sys.exit(3)
        import time
        time.sleep(60)          # rank 0 would wait for its peer for ever: the launcher has to end it
        sys.exit(0)
    import torch
    gsr = importlib.import_module("3dgs-native_amd")
    d = gsr.dist
    if mode == "nccl":
        dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
        torch.cuda.set_device(dev)
        d.init_from_env(backend="nccl", device=dev)
    else:
        d.init_from_env(backend="gloo")
    if mode == "cpu":
        n = 500
        arena = torch.full((d.ARENA_FLOATS * n,), float(rank + 1))
        d.reduce_gradients(arena, world, average=True)
        small = torch.full((d.SMALL_ARENA_FLOATS * n,), float(rank + 1))
        payload = torch.arange(3 * n + 4, dtype=torch.float32) + 1000.0 * rank
        gathered = d.exchange_factored(small, payload, average=True)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), arena=arena.numpy(), small=small.numpy(), gathered=gathered.numpy())
        if rank == 0:
            print(json.dumps({"rank0": "done", "world": world}), flush=True)
    elif mode == "gpu":
        torch.cuda.set_device(0)
        scene = gpu_case_scene(gsr)
        fkw, bkw, _ = gpu_case_view(gsr, scene, rank)
        ex = d.FactoredExchange()
        g = gsr.backward(**bkw, sh_gradient="factored", on_payload=ex.start_gather)
        means = torch.as_tensor(scene["means"]).cuda().contiguous()
        res = ex.finish(g, means, GPU_CASE["degree"], average=True)
        torch.cuda.synchronize()
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), **{k: v.cpu().numpy() for k, v in res.items()})
    elif mode == "nccl":
        scene = gpu_case_scene(gsr)
        fkw, bkw, _ = gpu_case_view(gsr, scene, rank)
        ex = d.FactoredExchange(timing=True)
        g = gsr.backward(**bkw, sh_gradient="factored", on_payload=ex.start_gather)
        means = torch.as_tensor(scene["means"]).cuda().contiguous()
        res = {k: v.clone() for k, v in ex.finish(g, means, GPU_CASE["degree"], average=True).items()}
        g_dense = gsr.backward(**bkw)                      # the same view again, dense: one all-reduce of the 59-float arena
        d.reduce_gradients(g_dense["_arena"], world, average=True)
        dense = d.arena_views(g_dense["_arena"], GPU_CASE["n"])
        torch.cuda.synchronize()
        out = {k: v.cpu().numpy() for k, v in res.items()}
        out.update({"dense_" + k: v.cpu().numpy() for k, v in dense.items()})
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
        if rank == 0:
            print(json.dumps({"exchange_ms": ex.timings_ms(), "bytes_per_rank": d.FactoredExchange.bytes_per_rank(GPU_CASE["n"], world)}), flush=True)
    else:
        raise SystemExit(f"unknown mode {mode}")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
