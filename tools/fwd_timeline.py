#!/usr/bin/env python3
"""Diagnostic (GSR_TIMELINE build of blend_fwd.hip only): per-phase shader-cycle totals of the forward blend over all waves.
usage: make -C 3dgs-native_amd/csrc timeline; GSR_LIB=$PWD/3dgs-native_amd/libgsr_hip_timeline.so python tools/fwd_timeline.py [C3]"""
import ctypes as C, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gsr = importlib.import_module("3dgs-native_amd")
cfg = gsr.scenes.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
sc = gsr.scenes.synthetic_scene(cfg["n"], cfg["scale_median"], cfg["scale_sigma"], cfg["seed"])
cam = gsr.cameras.nerf_camera(gsr.scenes.LEGO_FRAME0, cfg["width"], cfg["height"], gsr.scenes.LEGO_CAMERA_ANGLE_X)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).cuda()
kw = dict(background=np.zeros(3, np.float32), means3D=t(sc["means"]), opacity=t(sc["opacities"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]),
          viewmatrix=cam["world_to_camera"], projmatrix=cam["full_proj_matrix"], tan_fovx=cam["tan_fovx"], tan_fovy=cam["tan_fovy"],
          image_height=cfg["height"], image_width=cfg["width"], sh=t(sc["shs"]), degree=3, campos=cam["camera_center"])
L = gsr._lib.lib()
for _ in range(3):
    gsr.render_gaussians(**kw)
torch.cuda.synchronize()
tiles = ((cfg["width"] + 15) // 16) * ((cfg["height"] + 15) // 16)
waves = tiles * 4
arr = np.zeros((waves, 8), np.uint64)
assert L.gsr_debug_fwd_phases(arr.ctypes.data_as(C.c_void_p), waves) == 0
names = ["top barrier (incl. launch->first)", "wait gathered records", "staging (LDS image + masks)", "staging barrier", "list build", "pair loop", "epilogue"]
tot = arr[:, :7].sum(axis=1).astype(np.float64)
print(f"waves {waves}, pair iterations {int(arr[:, 7].sum())}, cycles per wave: mean {tot.mean():.0f}, median {np.median(tot):.0f}, max {tot.max():.0f}")
for k, nme in enumerate(names):
    col = arr[:, k].astype(np.float64)
    print(f"  {nme:36s} mean {col.mean():9.0f}  p50 {np.median(col):9.0f}  p95 {np.percentile(col, 95):9.0f} cycles/wave  {100.0 * col.sum() / tot.sum():5.1f} %")
pl = arr[:, 5].astype(np.float64) / np.maximum(1, arr[:, 7].astype(np.float64))
print(f"  cycles per pair iteration (wave mean): {pl[arr[:, 7] > 0].mean():.0f}")
