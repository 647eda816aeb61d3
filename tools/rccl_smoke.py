#!/usr/bin/env python3
"""World-size-1 RCCL smoke: the exact collective calls of dist.py (all_reduce AVG, async all_gather_into_tensor) on a real
NCCL=RCCL process group.  One GPU is enough to prove the ops exist in this torch/RCCL build; scaling needs the 8-GPU node."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", device_id=dev)
a = torch.arange(11 * 1000, dtype=torch.float32, device=dev)
p = torch.arange(3 * 1000 + 4, dtype=torch.float32, device=dev)
g = torch.empty((1, p.numel()), device=dev)
w = dist.all_gather_into_tensor(g.view(-1), p, async_op=True)          # FactoredExchange.start_gather
r = dist.all_reduce(a, op=dist.ReduceOp.AVG, async_op=True)             # FactoredExchange.finish
w.wait()
r.wait()
torch.cuda.synchronize()
assert torch.equal(g[0], p) and float(a[5]) == 5.0
dist.barrier()
print("rccl smoke ok", dist.get_backend())
dist.destroy_process_group()
