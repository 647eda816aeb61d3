#!/usr/bin/env python3
"""Where does the first step after a torch.cuda.synchronize() lose its 0.1-0.17 ms (bench.py's timed region starts that way)?
The library's per-stage events on exactly that step, against a step in the middle of a run, five times each."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
gsr = importlib.import_module("3dgs-native_amd")
dev = torch.device("cuda", 0)
cfg = gsr.scenes.CONFIGS["C3"]
W, H, N = cfg["width"], cfg["height"], cfg["n"]
sc = gsr.scenes.synthetic_scene(N, cfg["scale_median"], cfg["scale_sigma"], cfg["seed"])
cam = gsr.cameras.nerf_camera(gsr.scenes.LEGO_FRAME0, W, H, gsr.scenes.LEGO_CAMERA_ANGLE_X)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
means, shs, opac, scales, rots = t(sc["means"]), t(sc["shs"]), t(sc["opacities"]), t(sc["scales"]), t(sc["rotations"])
dpix = t(np.random.default_rng(99).normal(0.0, 1.0, (H, W, 3)) / (H * W * 3))
bg = np.zeros(3, np.float32)
fkw = dict(background=bg, means3D=means, opacity=opac, scales=scales, rotations=rots, viewmatrix=cam["world_to_camera"],
           projmatrix=cam["full_proj_matrix"], tan_fovx=cam["tan_fovx"], tan_fovy=cam["tan_fovy"], image_height=H, image_width=W,
           sh=shs, degree=3, campos=cam["camera_center"])


def step():
    img, depth, buf = gsr.render_gaussians(**fkw)
    gsr.backward(background=bg, means3D=means, dL_dpixels=dpix, opacity=opac, shs=shs, scales=scales, rotations=rots,
                 viewmatrix=fkw["viewmatrix"], projmatrix=fkw["projmatrix"], tan_fovx=fkw["tan_fovx"], tan_fovy=fkw["tan_fovy"],
                 image_height=H, image_width=W, campos=fkw["campos"], radii=buf["radii"], means2D=buf["points_xy_image"],
                 conic_opacity=buf["conic_opacity"], rgb=buf["colors"], cov3Ds=buf["cov3Ds"], clamped=buf["clamped_state"],
                 binning_buffer={"point_list": buf["point_list"]},
                 img_buffer={"ranges": buf["ranges"], "final_Ts": buf["final_Ts"], "n_contrib": buf["n_contrib"]}, degree=3)


e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(150):
    step()
for rep in range(5):
    for which in ("first after synchronize", "mid-run"):
        for _ in range(40):
            step()
        if which.startswith("first"):
            torch.cuda.synchronize()
        gsr._lib.stage_timing(True, 1, every=1)
        t0 = time.perf_counter()
        e0.record()
        step()
        e1.record()
        host = time.perf_counter() - t0
        gsr._lib.stage_sampling(0)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        st, n = gsr._lib.stage_times()
        gsr._lib.stage_timing(False)
        print(f"{which:24s} step {e0.elapsed_time(e1):.4f} ms  host {host * 1e3:.4f} ms  stages sum {sum(st.values()):.4f}  " +
              " ".join(f"{k} {v * 1e3:.0f}" for k, v in st.items()), flush=True)
