#!/usr/bin/env python3
"""Compute-side cost of the two gradient-exchange schemes on ONE GPU (the collectives themselves need N GPUs):
dense backward vs factored backward + the V-view SH rebuild.  C3 workload."""
import importlib, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gsr = importlib.import_module("3dgs-native_amd")
cfg = gsr.scenes.CONFIGS["C3"]
W, H, N = cfg["width"], cfg["height"], cfg["n"]
sc = gsr.scenes.synthetic_scene(N, cfg["scale_median"], cfg["scale_sigma"], cfg["seed"])
cam = gsr.cameras.nerf_camera(gsr.scenes.LEGO_FRAME0, W, H, gsr.scenes.LEGO_CAMERA_ANGLE_X)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()
means, shs, opac, scales, rots = t(sc["means"]), t(sc["shs"]), t(sc["opacities"]), t(sc["scales"]), t(sc["rotations"])
bg = np.zeros(3, np.float32)
dpix = t(np.random.default_rng(99).normal(0.0, 1.0, (H, W, 3)) / (H * W * 3))
fkw = dict(background=bg, means3D=means, opacity=opac, scales=scales, rotations=rots, viewmatrix=cam["world_to_camera"],
           projmatrix=cam["full_proj_matrix"], tan_fovx=cam["tan_fovx"], tan_fovy=cam["tan_fovy"], image_height=H, image_width=W, sh=shs,
           degree=3, campos=cam["camera_center"])
img, depth, buf = gsr.render_gaussians(**fkw)
bkw = dict(background=bg, means3D=means, dL_dpixels=dpix, opacity=opac, shs=shs, scales=scales, rotations=rots, viewmatrix=fkw["viewmatrix"],
           projmatrix=fkw["projmatrix"], tan_fovx=fkw["tan_fovx"], tan_fovy=fkw["tan_fovy"], image_height=H, image_width=W, campos=fkw["campos"],
           radii=buf["radii"], means2D=buf["points_xy_image"], conic_opacity=buf["conic_opacity"], rgb=buf["colors"], cov3Ds=buf["cov3Ds"],
           clamped=buf["clamped_state"], binning_buffer={"point_list": buf["point_list"]},
           img_buffer={"ranges": buf["ranges"], "final_Ts": buf["final_Ts"], "n_contrib": buf["n_contrib"]}, degree=3)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


out = {"backward_dense_ms": timed(lambda: gsr.backward(**bkw)), "backward_factored_ms": timed(lambda: gsr.backward(**bkw, sh_gradient="factored"))}
pay = gsr.backward(**bkw, sh_gradient="factored")["_view_payload"]
sh_out = torch.empty((N * 16, 3), device="cuda")
for V in (2, 4, 8):
    g = torch.stack([pay] * V)
    out[f"rebuild_V{V}_ms"] = timed(lambda: gsr.dist.sh_gradients_from_views(means, g, 3, out=sh_out))
print(json.dumps({k: round(v, 4) for k, v in out.items()}))
