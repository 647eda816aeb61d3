#!/usr/bin/env python3
"""How skewed is the blend's work?  Per 8x8 pixel block: list entries traversed (max n_contrib over its pixels)
against the tile's list length, at a bench workload (default C3)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gsr = importlib.import_module("3dgs-native_amd")
cfg = gsr.scenes.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
W, H, N = cfg["width"], cfg["height"], cfg["n"]
sc = gsr.scenes.synthetic_scene(N, cfg["scale_median"], cfg["scale_sigma"], cfg["seed"])
cam = gsr.cameras.nerf_camera(gsr.scenes.LEGO_FRAME0, W, H, gsr.scenes.LEGO_CAMERA_ANGLE_X)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()
img, depth, buf = gsr.render_gaussians(background=np.zeros(3, np.float32), means3D=t(sc["means"]), colors=None, opacity=t(sc["opacities"]),
                                       scales=t(sc["scales"]), rotations=t(sc["rotations"]), scale_modifier=1.0, viewmatrix=cam["world_to_camera"],
                                       projmatrix=cam["full_proj_matrix"], tan_fovx=cam["tan_fovx"], tan_fovy=cam["tan_fovy"], image_height=H,
                                       image_width=W, sh=t(sc["shs"]), degree=3, campos=cam["camera_center"])
nc = buf["n_contrib"].cpu().numpy().reshape(H, W)
rg = buf["ranges"].cpu().numpy().reshape(-1, 2)
gx, gy = (W + 15) // 16, (H + 15) // 16
ln = (rg[:, 1] - rg[:, 0]).reshape(gy, gx)
pad = np.zeros((gy * 16, gx * 16), np.int64); pad[:H, :W] = nc
blk = pad.reshape(gy * 2, 8, gx * 2, 8).max(axis=(1, 3))          # entries traversed per 8x8 block
tile = pad.reshape(gy, 16, gx, 16).max(axis=(1, 3))
q = lambda a: [int(np.percentile(a, p)) for p in (50, 90, 99, 100)]
print("tile list length   mean %.0f  p50/p90/p99/max %s" % (ln.mean(), q(ln)))
print("tile traversed     mean %.0f  p50/p90/p99/max %s   (sum %.2fM of D=%.2fM)" % (tile.mean(), q(tile), tile.sum() / 1e6, ln.sum() / 1e6))
print("8x8 block traversed mean %.0f  p50/p90/p99/max %s" % (blk.mean(), q(blk)))
print("pixel n_contrib    mean %.0f  p50/p90/p99/max %s" % (nc.mean(), q(nc)))
print("corr(list length, traversed) = %.3f" % np.corrcoef(ln.reshape(-1), tile.reshape(-1))[0, 1])
np.set_printoptions(linewidth=200)
cs = 5
print("traversed, %dx%d-tile means:" % (cs, cs)); print(tile[: gy // cs * cs, : gx // cs * cs].reshape(gy // cs, cs, gx // cs, cs).mean(axis=(1, 3)).astype(int))
print("list length, %dx%d-tile means:" % (cs, cs)); print(ln[: gy // cs * cs, : gx // cs * cs].reshape(gy // cs, cs, gx // cs, cs).mean(axis=(1, 3)).astype(int))
