#!/usr/bin/env python3
"""Row f4 microbenchmark: milliseconds and effective HBM bandwidth of each density-control kernel at N rows.
Algorithmic bytes per row: 236 B of parameters read and written per moved row, 4 B per mask / prefix entry."""
import argparse
import importlib
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gsr = importlib.import_module("3dgs-native_amd")
dz = gsr.densify


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    args = ap.parse_args()
    n = args.rows
    rng = np.random.default_rng(0)
    P = dz.alloc_params(n, "cuda")
    for k in P:
        P[k].copy_(torch.as_tensor(rng.uniform(0.001, 1.0, tuple(P[k].shape)).astype(np.float32)))
    P["scales"].mul_(0.02)
    g = torch.as_tensor((rng.normal(0, 1, (n, 3)) * 3e-4).astype(np.float32)).cuda()
    L, C, H = gsr._lib.lib(), __import__("ctypes"), gsr._host
    out = {}
    mask = dz.mark_candidates(P, g, 2e-4, 1.0, 0.01, False)
    prefix, total = dz.exclusive_scan(mask)
    out["flagged_fraction"] = total / n
    out["mark_ms"] = timed(lambda: dz.mark_candidates(P, g, 2e-4, 1.0, 0.01, False))
    out["scan_ms_incl_readback"] = timed(lambda: dz.exclusive_scan(mask))
    # the movers are timed on preallocated outputs (the wrappers allocate + zero the new arrays first)
    dst = dz.alloc_params(n + 2 * total, "cuda")
    pin, pout = dz._params_struct(P, n), dz._params_struct({k: v[: (n + total) * w] for (k, v), w in zip(dst.items(), (1, 1, 1, 1, 16))}, n + total)
    s = H.stream_ptr(torch.device("cuda", 0))
    out["clone_ms"] = timed(lambda: L.gsr_clone_gaussians(C.byref(pin), H.ptr(mask), H.ptr(prefix), 0.01, C.byref(pout), s))
    out["clone_GBps"] = (236 * (2 * n + 2 * total) + 8 * n) / out["clone_ms"] / 1e6
    pout2 = dz._params_struct(dst, n + 2 * total)
    out["split_ms"] = timed(lambda: L.gsr_split_gaussians(C.byref(pin), H.ptr(mask), H.ptr(prefix), 2, 0.8, C.byref(pout2), s))
    out["split_GBps"] = (236 * (2 * n + 3 * total) + 8 * n) / out["split_ms"] / 1e6
    valid = dz.prune_mask(P, 0.1)
    vp, vc = dz.exclusive_scan(valid)
    cdst = dz.alloc_params(vc, "cuda")
    pc = dz._params_struct(cdst, vc)
    out["compact_keep_fraction"] = vc / n
    out["compact_ms"] = timed(lambda: L.gsr_compact_gaussians(C.byref(pin), H.ptr(valid), H.ptr(vp), C.byref(pc), s))
    out["compact_GBps"] = (236 * 2 * vc + 8 * n) / out["compact_ms"] / 1e6
    model = dz.GaussianModel(P, config={"max_allowed_prune_ratio": 1.0}, scene_extent=1.0)
    def whole():
        model.params, model.num_points = P, n
        model.grads = {"positions": g}
        model.create_gradient_arrays = lambda: {}                 # time the density control itself, not three rounds of zero fills
        model.densification_and_pruning(600)
    out["densification_and_pruning_ms"] = timed(whole, reps=5)
    out["rows"] = n
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}))


if __name__ == "__main__":
    main()
