#!/usr/bin/env python3
"""One trainer iteration (reference train.py:920-1066 minus density control): forward -> L1 loss + pixel gradient -> backward ->
Adam, everything resident on the GPU, timed with HIP events over --steps iterations after --warmup, two ways:

  dense     backward() returns the dense 48-float SH gradient (its default, the reference's dict), gsr_adam_update reads it;
  factored  backward(sh_gradient="factored") returns the 3-float view payload instead and gsr_adam_update_views forms
            basis x payload inside the SH update (examples/train.py's default): the 192 bytes per Gaussian of SH gradient are
            neither written by geom_backward_kernel nor read by Adam.

usage: python tools/train_iter_bench.py [C2|C3|C5|C0|C2i] [--steps 30] [--warmup 5]   -> one JSON line per mode"""
import argparse
import importlib
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gsr = importlib.import_module("3dgs-native_amd")

ap = argparse.ArgumentParser()
ap.add_argument("config", nargs="?", default="C3")
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--warmup", type=int, default=5)
args = ap.parse_args()
cfg = gsr.scenes.CONFIGS[args.config]
W, H, N = cfg["width"], cfg["height"], cfg["n"]
dev = torch.device("cuda", 0)
if "init_scale" in cfg:
    P0 = gsr.densify.init_gaussian_params(N, cfg["init_scale"], dev)
else:
    sc = gsr.scenes.synthetic_scene(N, cfg["scale_median"], cfg["scale_sigma"], cfg["seed"])
    t = lambda a, shape: torch.as_tensor(np.ascontiguousarray(a, np.float32)).reshape(shape).to(dev)
    P0 = {"positions": t(sc["means"], (N, 3)), "scales": t(sc["scales"], (N, 3)), "rotations": t(sc["rotations"], (N, 4)),
          "opacities": t(sc["opacities"], (N,)), "shs": t(sc["shs"], (N * 16, 3))}
cam = gsr.cameras.nerf_camera(gsr.scenes.LEGO_FRAME0, W, H, gsr.scenes.LEGO_CAMERA_ANGLE_X)
bg = np.zeros(3, np.float32)
target = torch.as_tensor(np.random.default_rng(7).uniform(0, 1, (H, W, 3)).astype(np.float32)).to(dev)
lrs = {k: 0.0 for k in gsr.optimizer.DEFAULT_LR}     # learning rate 0: the same scene every iteration, the same work


def iteration(P, M, V, it, factored):
    kw = dict(background=bg, means3D=P["positions"], opacity=P["opacities"], scales=P["scales"], rotations=P["rotations"],
              viewmatrix=cam["world_to_camera"], projmatrix=cam["full_proj_matrix"], tan_fovx=cam["tan_fovx"], tan_fovy=cam["tan_fovy"],
              image_height=H, image_width=W, sh=P["shs"], degree=3, campos=cam["camera_center"])
    img, _, buf = gsr.render_gaussians(**kw)
    _, dpix = gsr.loss.l1_loss_and_gradients(img, target)
    g = gsr.backward(background=bg, means3D=P["positions"], dL_dpixels=dpix, opacity=P["opacities"], shs=P["shs"], scales=P["scales"],
                     rotations=P["rotations"], viewmatrix=kw["viewmatrix"], projmatrix=kw["projmatrix"], tan_fovx=kw["tan_fovx"],
                     tan_fovy=kw["tan_fovy"], image_height=H, image_width=W, campos=kw["campos"], radii=buf["radii"],
                     means2D=buf["points_xy_image"], conic_opacity=buf["conic_opacity"], rgb=buf["colors"], cov3Ds=buf["cov3Ds"],
                     clamped=buf["clamped_state"], binning_buffer={"point_list": buf["point_list"]},
                     img_buffer={"ranges": buf["ranges"], "final_Ts": buf["final_Ts"], "n_contrib": buf["n_contrib"]},
                     sh_gradient="factored" if factored else "dense")
    grads = gsr.optimizer.grads_from_backward(g)
    if factored:
        gsr.optimizer.adam_update(P, grads, M, V, lrs, iteration=it, sh_views=[g["_view_payload"]], sh_degree=3, sh_scale=1.0)
    else:
        gsr.optimizer.adam_update(P, grads, M, V, lrs, iteration=it)
    return buf


for mode in ("dense", "factored", "dense", "factored"):
    P = {k: v.clone() for k, v in P0.items()}
    M, V = gsr.optimizer.make_state(P)
    for it in range(args.warmup):
        buf = iteration(P, M, V, it, mode == "factored")
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    marks[0].record()
    for it in range(args.steps):
        iteration(P, M, V, it, mode == "factored")
        marks[it + 1].record()
    torch.cuda.synchronize()
    per = np.array([marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps)])
    print(json.dumps({"what": "trainer iteration: forward + L1 + backward + Adam (no density control), one view, one MI355X", "config": args.config,
                      "mode": mode, "gaussians": N, "tile_pairs_D": int(buf["point_list"].shape[0]), "steps": args.steps,
                      "ms_per_iteration": {"median": round(float(np.median(per)), 4), "p10": round(float(np.percentile(per, 10)), 4),
                                           "p90": round(float(np.percentile(per, 90)), 4)}}), flush=True)
