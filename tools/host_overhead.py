#!/usr/bin/env python3
"""Host time of one render_gaussians() + backward() pair: a scene so small (64 Gaussians, 32 x 32 pixels) that the GPU is never
the limit, so the wall clock per step is what the Python / ctypes / launch path costs; then cProfile's top entries."""
import cProfile, importlib, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gsr = importlib.import_module("3dgs-native_amd")
W = H = 32
cam = gsr.cameras.nerf_camera(gsr.scenes.LEGO_FRAME0, W, H, gsr.scenes.LEGO_CAMERA_ANGLE_X)
bg = np.zeros(3, np.float32)
dev = torch.device("cuda", 0)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
sc = gsr.scenes.synthetic_scene(64, 0.05, 0.5, 3)
P = dict(means3D=t(sc["means"]), opacity=t(sc["opacities"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
shs = t(sc["shs"])
dpix = t(np.random.default_rng(9).normal(0.0, 1.0, (H, W, 3)))
kw = dict(background=bg, **P, viewmatrix=cam["world_to_camera"], projmatrix=cam["full_proj_matrix"], tan_fovx=cam["tan_fovx"], tan_fovy=cam["tan_fovy"],
          image_height=H, image_width=W, sh=shs, degree=3, campos=cam["camera_center"])
tf = tb = 0.0


def step():
    global tf, tb
    t0 = time.perf_counter()
    img, depth, buf = gsr.render_gaussians(**kw)
    t1 = time.perf_counter()
    gsr.backward(background=bg, dL_dpixels=dpix, shs=shs, **P, viewmatrix=kw["viewmatrix"], projmatrix=kw["projmatrix"], tan_fovx=kw["tan_fovx"],
                 tan_fovy=kw["tan_fovy"], image_height=H, image_width=W, campos=kw["campos"], radii=buf["radii"], means2D=buf["points_xy_image"],
                 conic_opacity=buf["conic_opacity"], rgb=buf["colors"], cov3Ds=buf["cov3Ds"], clamped=buf["clamped_state"],
                 binning_buffer={"point_list": buf["point_list"]}, img_buffer={"ranges": buf["ranges"], "final_Ts": buf["final_Ts"], "n_contrib": buf["n_contrib"]})
    tf += t1 - t0
    tb += time.perf_counter() - t1


for _ in range(200):
    step()
torch.cuda.synchronize()
tf = tb = 0.0
n = 2000
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"wall {wall / n * 1e6:.1f} us per step: render_gaussians {tf / n * 1e6:.1f} us (includes the wait for D), backward {tb / n * 1e6:.1f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(500):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
