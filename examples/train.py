#!/usr/bin/env python3
"""
Counterpart of the reference's training iteration (train.py:920-1066) on the
MI355X: per step each rank renders its views, forms the L1 loss and pixel gradient, runs backward, all-reduces
the gradient (11 floats all-reduced + 3 floats all-gathered per Gaussian, the SH gradient rebuilt per rank: dist.py), and applies the fused Adam update -- everything on the GPU,
parameters resident, no per-iteration host upload.  Then the reference's adaptive density control (row f4,
train.py:351-713: clone / split / prune / opacity reset) runs on the replicated parameters -- it is deterministic,
so every rank reaches the same point set without a collective -- and rank 0 writes PLY checkpoints (train.py:796-803).

Targets: with --dataset <NeRF-synthetic dir> the train split (transforms_train.json + PNGs, alpha dropped as
train.py:323-334) is used; without it, targets are renders of a hidden seeded scene from orbiting cameras.

    python examples/train.py --iterations 200 --gaussians 20000
    python examples/train.py --gpus 8 --views-per-step 8        (launches its own 8 ranks; or under torch.distributed.run)
"""
import argparse
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# before torch / the HIP runtime load: dmabuf IPC is what this pool's driver supports, and a rank started by torch.distributed.run
# (not by launch.py) would otherwise never get it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _self_launch():
    """`python examples/train.py --gpus N` (no WORLD_SIZE): start the N ranks before anything here touches the GPU."""
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    gpus = ap.parse_known_args()[0].gpus
    if gpus > 1 and "WORLD_SIZE" not in os.environ:
        from importlib import util as _ilu
        spec = _ilu.spec_from_file_location("gsr_launch", os.path.join(ROOT, "3dgs-native_amd", "launch.py"))
        launch = _ilu.module_from_spec(spec)
        spec.loader.exec_module(launch)
        rc = launch.launch_ranks(os.path.abspath(__file__), sys.argv[1:], gpus)
        sys.exit(rc if rc >= 0 else 128 - rc)


if __name__ == "__main__":
    _self_launch()

import numpy as np  # noqa: E402
import torch  # noqa: E402

gsr = importlib.import_module("3dgs-native_amd")


def load_nerf(path, max_views, split="train"):
    from PIL import Image
    with open(os.path.join(path, f"transforms_{split}.json")) as f:
        tf = json.load(f)
    cams, targets = [], []
    for fr in tf["frames"][:max_views]:
        img = np.asarray(Image.open(os.path.join(path, fr["file_path"] + ".png")), dtype=np.float32) / 255.0
        targets.append(img[:, :, :3].copy())
        cams.append(gsr.cameras.nerf_camera(fr["transform_matrix"], img.shape[1], img.shape[0], tf["camera_angle_x"]))
    return cams, targets


def render_view(P, c, bg):
    return gsr.render_gaussians(background=bg, means3D=P["positions"], opacity=P["opacities"], scales=P["scales"], rotations=P["rotations"],
                                viewmatrix=c["world_to_camera"], projmatrix=c["full_proj_matrix"], tan_fovx=c["tan_fovx"], tan_fovy=c["tan_fovy"],
                                image_height=c["height"], image_width=c["width"], sh=P["shs"], degree=3, campos=c["camera_center"])[0]


def score_views(P, cams, targets, bg):
    """Per-view mean L1 and PSNR (10 log10(1 / MSE), colours in [0, 1]) of the current model; the images come back too."""
    rows, images = [], []
    for c, t in zip(cams, targets):
        img = render_view(P, c, bg).reshape(t.shape)
        d = img - t
        mse = float((d * d).mean().item())
        rows.append({"l1": float(d.abs().mean().item()), "psnr": (10.0 * np.log10(1.0 / mse)) if mse > 0 else float("inf")})
        images.append(img)
    return rows, images


def finish(args, model, cams, targets, bg, loss_hist, density_log, wall, dev):
    """The run record: parameters finite, loss curve, point count after every density-control call, timing, per-view scores, PNGs."""
    from PIL import Image
    P = model.params
    finite = {k: bool(torch.isfinite(P[k]).all().item()) for k in gsr.optimizer.GROUPS}
    rows, images = score_views(P, cams, targets, bg)
    summary = {"iterations": args.iterations, "wall_s": round(wall, 3), "iterations_per_s": round(args.iterations / wall, 1) if wall > 0 else None,
               "points_start": density_log[0]["points"], "points_final": model.num_points, "density_control_calls": len(density_log) - 1,
               "parameters_finite": finite, "train_views": rows, "train_l1_mean": float(np.mean([r["l1"] for r in rows])),
               "train_psnr_mean": float(np.mean([r["psnr"] for r in rows]))}
    if args.holdout and args.dataset:
        split, _, k = args.holdout.partition(":")
        hc, ht = load_nerf(args.dataset, int(k or 8), split)
        hrows, _ = score_views(P, hc, [torch.as_tensor(t).to(dev) for t in ht], bg)
        summary.update({"holdout": args.holdout, "holdout_views": hrows, "holdout_l1_mean": float(np.mean([r["l1"] for r in hrows])),
                        "holdout_psnr_mean": float(np.mean([r["psnr"] for r in hrows]))})
    print(f"trained {args.iterations} iterations in {wall:.2f} s ({summary['iterations_per_s']} it/s); {summary['points_start']} -> "
          f"{model.num_points} points; train L1 {summary['train_l1_mean']:.5f} PSNR {summary['train_psnr_mean']:.2f} dB"
          + (f"; holdout PSNR {summary['holdout_psnr_mean']:.2f} dB" if "holdout_psnr_mean" in summary else "")
          + f"; parameters finite: {all(finite.values())}", flush=True)
    if args.log:
        os.makedirs(os.path.dirname(os.path.abspath(args.log)), exist_ok=True)
        curve = loss_hist[:args.iterations].cpu().numpy()
        with open(args.log, "w") as f:
            f.write(json.dumps({"record": "arguments", **{k: v for k, v in vars(args).items()}}) + "\n")
            for d in density_log:
                f.write(json.dumps({"record": "density_control", **d}) + "\n")
            for i in range(0, len(curve), 100):               # every iteration's loss, 100 per line
                f.write(json.dumps({"record": "loss", "from_iteration": i, "l1": [round(float(x), 6) for x in curve[i:i + 100]]}) + "\n")
            f.write(json.dumps({"record": "summary", **summary}) + "\n")
    if args.eval_dir:
        os.makedirs(args.eval_dir, exist_ok=True)
        to8 = lambda x: (x.clamp(0, 1) * 255.0 + 0.5).to(torch.uint8).cpu().numpy()
        for k in range(min(args.eval_views, len(images))):
            r8, t8 = to8(images[k]), to8(targets[k].reshape(images[k].shape))
            Image.fromarray(r8).save(os.path.join(args.eval_dir, f"render_{k}.png"))
            Image.fromarray(t8).save(os.path.join(args.eval_dir, f"target_{k}.png"))
            Image.fromarray(np.concatenate([r8, t8], axis=1)).save(os.path.join(args.eval_dir, f"pair_{k}.png"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks (one per GPU); > 1 without WORLD_SIZE launches them itself")
    ap.add_argument("--dataset", default=None)
    ap.add_argument("--iterations", type=int, default=100)
    ap.add_argument("--gaussians", type=int, default=5000)      # reference default (config.py:31)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--views-per-step", type=int, default=1)
    ap.add_argument("--view-streams", type=int, default=0, help="when a rank has several views per step (--views-per-step beyond the "
                    "number of GPUs), render up to this many at once on separate HIP streams (1 = one after the other; 0 = decide by "
                    "scene size each step: 3 from 2^17 Gaussians up -- +9 %% at 1 M Gaussians, profiles/r04_c_views_per_gpu.txt -- and 1 "
                    "below, where a step is host-bound and the extra stream hand-overs cost more than they hide: 872 against 750 "
                    "iterations/s at 6 000 Gaussians)")
    ap.add_argument("--size", type=int, default=400)
    ap.add_argument("--densify-from", type=int, default=500)    # train.py:391-394 defaults
    ap.add_argument("--densify-until", type=int, default=15000)
    ap.add_argument("--densify-interval", type=int, default=100)
    ap.add_argument("--opacity-reset-interval", type=int, default=3000)
    ap.add_argument("--save-interval", type=int, default=500)   # config.py:32
    ap.add_argument("--init", default="reference", choices=["reference", "random"], help="initial Gaussians: the reference's "
                    "init_gaussian_params, or a seeded random scene")
    ap.add_argument("--backend", default=None, help="collective backend (default: nccl = RCCL); gloo + --single-device rehearses N ranks on one GPU")
    ap.add_argument("--single-device", action="store_true")
    ap.add_argument("--dense-sh", action="store_true", help="materialise the 48-float SH gradient (backward's dense return, one "
                    "59-float all-reduce when N > 1) instead of forming it inside the Adam update from the view payloads")
    ap.add_argument("--output", default=None, help="directory for point_cloud/iteration_N/point_cloud.ply")
    ap.add_argument("--print-interval", type=int, default=10, help="iterations between loss lines (each one reads the loss back: a sync)")
    ap.add_argument("--log", default=None, help="JSONL run record: one line per density-control call, the per-iteration loss curve, "
                    "and a closing summary (wall time, iterations/s, per-view L1 / PSNR)")
    ap.add_argument("--eval-dir", default=None, help="after training, write render_<k>.png / target_<k>.png / pair_<k>.png for --eval-views")
    ap.add_argument("--eval-views", type=int, default=1, help="how many of the training views --eval-dir renders to PNG")
    ap.add_argument("--holdout", default=None, help="NeRF-synthetic split to score after training without training on it, e.g. "
                    "'test:8' = the first 8 frames of transforms_test.json (needs --dataset)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not 1 <= args.views_per_step <= args.views:
        raise SystemExit(f"--views-per-step {args.views_per_step} must be in 1..--views ({args.views}): a step draws its views without replacement")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local = 0 if args.single_device else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rank = 0
    if world > 1:
        rank, world = gsr.dist.init_from_env(backend=args.backend, device=dev)

    bg = np.zeros(3, np.float32)
    if args.dataset:
        cams, targets = load_nerf(args.dataset, args.views)
        targets = [torch.as_tensor(t).to(dev) for t in targets]
    else:
        cams = [gsr.cameras.nerf_camera(gsr.scenes.orbit_pose(k, args.views), args.size, args.size, gsr.scenes.LEGO_CAMERA_ANGLE_X)
                for k in range(args.views)]
        hidden = gsr.scenes.synthetic_scene(args.gaussians, 0.05, 0.5, seed=7)
        targets = []
        for c in cams:
            img, _, _ = gsr.render_gaussians(background=bg, means3D=hidden["means"], opacity=hidden["opacities"], scales=hidden["scales"],
                                             rotations=hidden["rotations"], viewmatrix=c["world_to_camera"], projmatrix=c["full_proj_matrix"],
                                             tan_fovx=c["tan_fovx"], tan_fovy=c["tan_fovy"], image_height=c["height"], image_width=c["width"],
                                             sh=hidden["shs"], degree=3, campos=c["camera_center"])
            targets.append(img)

    n = args.gaussians
    if args.init == "reference":
        # the reference trainer's start (train.py:37-92, 193-214): randf-hashed positions in (-1.3, 1.3)^3, scale 0.1, opacity 0.1
        P = gsr.densify.init_gaussian_params(n, 0.1, dev)
    else:
        init = gsr.scenes.synthetic_scene(n, 0.05, 0.5, seed=8)             # same on every rank (replicated parameters)
        t = lambda a, shape: torch.as_tensor(np.ascontiguousarray(a, np.float32)).reshape(shape).to(dev)
        P = {"positions": t(init["means"], (n, 3)), "scales": t(init["scales"], (n, 3)), "rotations": t(init["rotations"], (n, 4)),
             "opacities": t(init["opacities"], (n,)), "shs": t(init["shs"], (n * 16, 3))}
    model = gsr.densify.GaussianModel(
        P, scene_extent=gsr.densify.calculate_scene_extent([c["camera_center"] for c in cams]),
        config={"densify_from_iter": args.densify_from, "densify_until_iter": args.densify_until, "densification_interval": args.densify_interval,
                "opacity_reset_interval": args.opacity_reset_interval, "max_allowed_prune_ratio": 1.0, "background_color": [0.0, 0.0, 0.0]})
    sched = {k: gsr.scheduler.LRScheduler(lr) for k, lr in gsr.optimizer.DEFAULT_LR.items()}
    rng = np.random.default_rng(0)                                          # same stream on every rank -> same view batch
    loss_hist = torch.zeros(max(1, args.iterations), device=dev)            # the loss curve stays on the device until the end
    per_rank_views = -(-args.views_per_step // world)
    sum_scale = float(cams[0]["height"] * cams[0]["width"] * 3)             # an L1 sum -> mean (all views of a dataset share one size)
    streams_few, streams_many = gsr.dist.ViewStreams(1, dev), gsr.dist.ViewStreams(min(per_rank_views, args.view_streams or 3), dev)
    density_log = [{"iteration": -1, "points": model.num_points}]
    import time
    torch.cuda.synchronize(dev)
    t_start = time.perf_counter()
    for it in range(args.iterations):
        P, M, V, n = model.params, model.adam_m, model.adam_v, model.num_points
        batch = rng.choice(len(cams), size=args.views_per_step, replace=False)
        mine = [int(batch[i]) for i in gsr.dist.views_for_rank(len(batch), rank, world)]
        arena, loss_acc, payloads = None, (torch.zeros(1, device=dev) if len(mine) > 1 else None), []
        # The SH gradient is never materialised: backward() returns the 3-float view payload it is an outer product of, the
        # ranks exchange that (11 + 3 floats per Gaussian instead of 59, dist.py) and the Adam kernel forms basis x payload
        # inside the SH update (optimizer.adam_update(sh_views=...)) -- also with one rank.  --dense-sh keeps the 48-float path.
        factored = not args.dense_sh
        def one_view(v):
            c = cams[v]
            kw = dict(background=bg, means3D=P["positions"], opacity=P["opacities"], scales=P["scales"], rotations=P["rotations"],
                      viewmatrix=c["world_to_camera"], projmatrix=c["full_proj_matrix"], tan_fovx=c["tan_fovx"], tan_fovy=c["tan_fovy"],
                      image_height=c["height"], image_width=c["width"], sh=P["shs"], degree=3, campos=c["camera_center"])
            img, _, buf = gsr.render_gaussians(**kw)
            # (one view per rank and step: the L1 sum goes straight into this iteration's slot of the loss curve)
            loss_sum, dpix = gsr.loss.l1_loss_and_gradients(img, targets[v], loss_out=loss_hist[it:it + 1] if len(mine) == 1 else None)
            g = gsr.backward(background=bg, means3D=P["positions"], dL_dpixels=dpix, opacity=P["opacities"], shs=P["shs"], scales=P["scales"],
                             rotations=P["rotations"], viewmatrix=kw["viewmatrix"], projmatrix=kw["projmatrix"], tan_fovx=kw["tan_fovx"],
                             tan_fovy=kw["tan_fovy"], image_height=c["height"], image_width=c["width"], campos=kw["campos"],
                             radii=buf["radii"], means2D=buf["points_xy_image"], conic_opacity=buf["conic_opacity"], rgb=buf["colors"],
                             cov3Ds=buf["cov3Ds"], clamped=buf["clamped_state"], binning_buffer={"point_list": buf["point_list"]},
                             img_buffer={"ranges": buf["ranges"], "final_Ts": buf["final_Ts"], "n_contrib": buf["n_contrib"]},
                             sh_gradient="factored" if factored else "dense")
            return (loss_sum if len(mine) == 1 else loss_sum / (c["height"] * c["width"] * 3)), g["_arena"], g["_view_payload"]

        # a rank with several views renders them on separate streams (one view's sort chain under another's blend kernels) and
        # then sums them in view order, exactly as the serial loop does
        view_streams = streams_many if (args.view_streams > 1 or (args.view_streams == 0 and n >= (1 << 17))) else streams_few
        for l_v, a_v, p_v in view_streams.map(one_view, mine):
            if len(mine) > 1:
                loss_acc += l_v
            arena = a_v if arena is None else arena.add_(a_v)
            if factored:
                payloads.append(p_v)
        per_rank = -(-len(batch) // world)                                  # views per rank, rounded up
        if factored:
            if arena is None:
                arena = torch.zeros(gsr.dist.arena_size(n, small=True), device=dev)
            while len(payloads) < per_rank:                                 # ranks with a view less gather a zero payload
                payloads.append(torch.zeros(3 * n + 4, device=dev))
            if world != len(batch):
                arena.mul_(world / len(batch))                              # mean over the batch after the /world of the average
            if world == 1:
                sh_views = payloads                                         # nothing to exchange: the Adam kernel takes the list as it is
            else:
                sh_views = gsr.dist.exchange_factored(arena, torch.stack(payloads).view(-1), average=True).view(world * per_rank, 3 * n + 4)
            grads = gsr.dist.small_arena_views(arena, n)
            grads["dL_dshs"] = None
            sh_scale = 1.0 / len(batch)
            if len(sh_views) > gsr.dist.MAX_VIEWS_PER_CALL:                 # more views than one kernel call takes: rebuild in chunks
                grads["dL_dshs"] = gsr.dist.sh_gradients_from_views(P["positions"], torch.stack(list(sh_views)), 3, scale=sh_scale)
                sh_views, sh_scale = None, None
        else:
            if arena is None:
                arena = torch.zeros(gsr.dist.arena_size(n), device=dev)
            if world != len(batch):
                arena.mul_(world / len(batch))                              # mean over the batch after the /world of the average
            if world > 1:
                gsr.dist.reduce_gradients(arena, world, average=True)       # (this all-reduce was missing before round 4: --dense-sh
            grads = gsr.dist.arena_views(arena, n)                          #  with several ranks trained every rank on its own views only)
            sh_views, sh_scale = None, None
        lrs = {k: s.get_lr(it, args.iterations) for k, s in sched.items()}
        model.grads = gsr.optimizer.grads_from_backward(grads)              # train.py:1047-1051
        gsr.optimizer.adam_update(P, model.grads, M, V, lrs, iteration=it, sh_views=sh_views, sh_degree=3, sh_scale=sh_scale)
        if len(mine) > 1:
            loss_hist[it] = loss_acc[0] / len(mine)
        elif len(mine) == 1:
            pass                                                            # written by the loss kernel as a SUM: scaled once, at the end
        log = model.densification_and_pruning(it)                           # train.py:1060
        if log["cloned"] or log["split"] or log["pruned"] or log["opacity_reset"] or log["prune_skipped"]:
            density_log.append({"iteration": it, "cloned": log["cloned"], "split": log["split"], "split_removed": log["split_removed"],
                                "pruned": log["pruned"], "prune_skipped": log["prune_skipped"], "opacity_reset": log["opacity_reset"],
                                "points": model.num_points})
        if rank == 0 and (log["cloned"] or log["split"] or log["pruned"] or log["opacity_reset"]):
            print(f"iter {it:5d}  densify: +{log['cloned']} cloned, {log['split']} split, -{log['pruned']} pruned"
                  f"{', opacity reset' if log['opacity_reset'] else ''} -> {model.num_points} points")
        if rank == 0 and args.output and (it % args.save_interval == 0 or it == args.iterations - 1):
            gsr.point_cloud.save_ply(model.params, os.path.join(args.output, "point_cloud", f"iteration_{it}", "point_cloud.ply"), model.num_points)
        if rank == 0 and (it % args.print_interval == 0 or it == args.iterations - 1):
            shown = float(loss_hist[it].item()) / (sum_scale if len(mine) == 1 else 1.0)
            print(f"iter {it:5d}  loss {shown:.6f}", flush=True)
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t_start
    if per_rank_views == 1:
        loss_hist /= sum_scale                                              # slots hold sums of |difference|: one division for the whole curve
    if rank == 0:
        finish(args, model, cams, targets, bg, loss_hist, density_log, wall, dev)


if __name__ == "__main__":
    main()
