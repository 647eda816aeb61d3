#!/usr/bin/env python3
"""Backward blend time against the scene's tile pairs per Gaussian (D / N), for the block size given in GSR_BWD_BLOCK (read once per
process: run once per setting and compare).  usage: bwd_block_sweep.py [WIDTH HEIGHT [N,SCALE,SEED ...]]; by default 100 000
Gaussians at 800 x 800 with scale medians from 0.004 to 0.1."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gsr = importlib.import_module("3dgs-native_amd")
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (800, 800)
CASES = [(int(a), float(b), int(c)) for a, b, c in (x.split(",") for x in sys.argv[3:])] or \
    [(100000, m, 11) for m in (0.004, 0.008, 0.012, 0.016, 0.024, 0.035, 0.05, 0.1)] + [(20000, 0.03, 11), (20000, 0.06, 11)]
cam = gsr.cameras.nerf_camera(gsr.scenes.LEGO_FRAME0, W, H, gsr.scenes.LEGO_CAMERA_ANGLE_X)
bg = np.zeros(3, np.float32)
dev = torch.device("cuda", 0)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
dpix = t(np.random.default_rng(99).normal(0.0, 1.0, (H, W, 3)) / (H * W * 3))
for n, med, seed in CASES:
    sc = gsr.scenes.synthetic_scene(n, med, 0.5, seed)
    P = dict(means3D=t(sc["means"]), opacity=t(sc["opacities"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
    shs = t(sc["shs"])
    kw = dict(background=bg, **P, viewmatrix=cam["world_to_camera"], projmatrix=cam["full_proj_matrix"], tan_fovx=cam["tan_fovx"], tan_fovy=cam["tan_fovy"],
              image_height=H, image_width=W, sh=shs, degree=3, campos=cam["camera_center"])

    def step():
        img, depth, buf = gsr.render_gaussians(**kw)
        gsr.backward(background=bg, dL_dpixels=dpix, shs=shs, **P, viewmatrix=kw["viewmatrix"], projmatrix=kw["projmatrix"], tan_fovx=kw["tan_fovx"],
                     tan_fovy=kw["tan_fovy"], image_height=H, image_width=W, campos=kw["campos"], radii=buf["radii"], means2D=buf["points_xy_image"],
                     conic_opacity=buf["conic_opacity"], rgb=buf["colors"], cov3Ds=buf["cov3Ds"], clamped=buf["clamped_state"],
                     binning_buffer={"point_list": buf["point_list"]}, img_buffer={"ranges": buf["ranges"], "final_Ts": buf["final_Ts"], "n_contrib": buf["n_contrib"]})
        return buf
    for _ in range(5):
        buf = step()
    torch.cuda.synchronize()
    gsr._lib.stage_timing(True, 20, every=1)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    st, nrec = gsr._lib.stage_times()
    gsr._lib.stage_timing(False)
    D = int(buf["point_list"].shape[0])
    print(f"block {os.environ.get('GSR_BWD_BLOCK', 'auto'):>4s}  {W}x{H} N {n:7d} scale {med:6.4f} seed {seed}  D {D:9d}  D/N {D / n:7.1f}  blend_bwd {st['blend_bwd'] * 1000:7.1f} us  blend_fwd {st['blend_fwd'] * 1000:6.1f} us", flush=True)
