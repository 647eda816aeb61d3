#!/usr/bin/env python3
"""RCCL smoke of the exact collective calls of dist.py (all_reduce AVG, async all_gather_into_tensor, barrier) on a real NCCL=RCCL
process group.

    python tools/rccl_smoke.py            world size 1 on cuda:0: proves the ops exist in this torch/RCCL build
    python tools/rccl_smoke.py --gpus N   N ranks, one GPU each, started by 3dgs-native_amd/launch.py (the launcher of
                                          bench.py / examples/train.py); every rank checks the reduced and gathered values
Under `python -m torch.distributed.run --nproc-per-node N tools/rccl_smoke.py` it is simply one of the ranks."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (this pool's driver); before torch loads the HIP runtime

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
args = ap.parse_args()
if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    from importlib import util as _ilu
    spec = _ilu.spec_from_file_location("gsr_launch", os.path.join(ROOT, "3dgs-native_amd", "launch.py"))
    launch = _ilu.module_from_spec(spec)
    spec.loader.exec_module(launch)
    rc = launch.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, timeout=300)
    sys.exit(rc if rc >= 0 else 128 - rc)

import torch                          # noqa: E402
import torch.distributed as dist      # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if torch.cuda.device_count() < world:
    raise SystemExit(f"rccl_smoke: {world} ranks need {world} GPUs, this node shows {torch.cuda.device_count()}")
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", device_id=dev)
n = 1000
a = torch.arange(11 * n, dtype=torch.float32, device=dev) + rank
p = torch.arange(3 * n + 4, dtype=torch.float32, device=dev) + 1000.0 * rank
g = torch.empty((world, p.numel()), device=dev)
w = dist.all_gather_into_tensor(g.view(-1), p, async_op=True)          # FactoredExchange.start_gather
r = dist.all_reduce(a, op=dist.ReduceOp.AVG, async_op=True)             # FactoredExchange.finish
w.wait()
r.wait()
torch.cuda.synchronize()
for v in range(world):
    assert torch.equal(g[v], torch.arange(3 * n + 4, dtype=torch.float32, device=dev) + 1000.0 * v), f"gathered row {v}"
assert abs(float(a[5]) - (5.0 + (world - 1) / 2.0)) < 1e-5, float(a[5])
dist.barrier()
if rank == 0:
    print(f"rccl smoke ok: {world} rank(s), backend {dist.get_backend()}", flush=True)
dist.destroy_process_group()
