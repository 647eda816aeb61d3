#!/usr/bin/env python3
"""Diagnostic (GSR_TIMELINE build of blend_bwd_splat.hip only): per-phase shader cycles of the backward blend's waves.
usage: make -C 3dgs-native_amd/csrc timeline; GSR_LIB=$PWD/3dgs-native_amd/libgsr_hip_timeline.so python tools/bwd_timeline.py [C3]"""
import ctypes as C, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gsr = importlib.import_module("3dgs-native_amd")
cfg = gsr.scenes.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
sc = gsr.scenes.synthetic_scene(cfg["n"], cfg["scale_median"], cfg["scale_sigma"], cfg["seed"])
W, H = cfg["width"], cfg["height"]
cam = gsr.cameras.nerf_camera(gsr.scenes.LEGO_FRAME0, W, H, gsr.scenes.LEGO_CAMERA_ANGLE_X)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).cuda()
bg = np.zeros(3, np.float32)
P = dict(means3D=t(sc["means"]), opacity=t(sc["opacities"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
shs = t(sc["shs"])
kw = dict(background=bg, **P, viewmatrix=cam["world_to_camera"], projmatrix=cam["full_proj_matrix"], tan_fovx=cam["tan_fovx"], tan_fovy=cam["tan_fovy"],
          image_height=H, image_width=W, sh=shs, degree=3, campos=cam["camera_center"])
dpix = t(np.random.default_rng(99).normal(0.0, 1.0, (H, W, 3)) / (H * W * 3))
L = gsr._lib.lib()
for _ in range(3):
    img, depth, buf = gsr.render_gaussians(**kw)
    gsr.backward(background=bg, dL_dpixels=dpix, shs=shs, **P, viewmatrix=kw["viewmatrix"], projmatrix=kw["projmatrix"], tan_fovx=kw["tan_fovx"],
                 tan_fovy=kw["tan_fovy"], image_height=H, image_width=W, campos=kw["campos"], radii=buf["radii"], means2D=buf["points_xy_image"],
                 conic_opacity=buf["conic_opacity"], rgb=buf["colors"], cov3Ds=buf["cov3Ds"], clamped=buf["clamped_state"],
                 binning_buffer={"point_list": buf["point_list"]}, img_buffer={"ranges": buf["ranges"], "final_Ts": buf["final_Ts"], "n_contrib": buf["n_contrib"]})
torch.cuda.synchronize()
waves = ((W + 15) // 16) * ((H + 15) // 16) * 8
arr = np.zeros((waves, 12), np.uint64)
assert L.gsr_debug_bwd_phases(arr.ctypes.data_as(C.c_void_p), waves) == 0
names = ["prologue", "fill (compaction)", "record gather", "pixel loop", "flush (transpose + atomics)"]
tot = arr[:, :5].sum(axis=1).astype(np.float64)
print(f"waves {waves}, buckets {int(arr[:, 6].sum())}, (bucket, pixel) steps {int(arr[:, 7].sum())}, cycles per wave: mean {tot.mean():.0f}, median {np.median(tot):.0f}, max {tot.max():.0f}")
for k, nme in enumerate(names):
    col = arr[:, k].astype(np.float64)
    print(f"  {nme:30s} mean {col.mean():9.0f}  p50 {np.median(col):9.0f}  p95 {np.percentile(col, 95):9.0f} cycles/wave  {100.0 * col.sum() / tot.sum():5.1f} %")
steps = arr[:, 7].astype(np.float64)
print(f"  cycles per (bucket, pixel) step (wave mean): {(arr[:, 3].astype(np.float64)[steps > 0] / steps[steps > 0]).mean():.0f}")

# Residency per physical CU, from each wave's HW_ID / XCC_ID ([8]) and its s_memrealtime start / end ([9], [10]: 100 MHz, one
# counter for the whole chip).  HW_ID (gfx9): wave_id [3:0], simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13].
ran = arr[:, 9] > 0
hw = (arr[ran, 8] & np.uint64(0xFFFFFFFF)).astype(np.int64)
xcc = ((arr[ran, 8] >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
cu_key = (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
simd_key = (cu_key << 2) | ((hw >> 4) & 3)
r0, r1 = arr[ran, 9].astype(np.int64), arr[ran, 10].astype(np.int64)
t_first, t_last = r0.min(), r1.max()
span = float(t_last - t_first)
print(f"kernel span {span / 100:.1f} us (s_memrealtime); {np.unique(cu_key).size} CUs, {np.unique(simd_key).size} SIMDs seen; "
      f"mean resident waves per SIMD over the span {(r1 - r0).sum() / span / np.unique(simd_key).size:.2f}")
def peak_and_curve(keys):
    peaks, curves = [], []
    grid = t_first + (np.linspace(0.025, 0.975, 20) * span).astype(np.int64)
    for k in np.unique(keys):
        m = keys == k
        ev = np.concatenate([np.stack([r0[m], np.ones(m.sum(), np.int64)], 1), np.stack([r1[m], -np.ones(m.sum(), np.int64)], 1)])
        ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
        conc = np.cumsum(ev[:, 1])
        peaks.append(conc.max())
        curves.append([conc[max(0, np.searchsorted(ev[:, 0], g, side="right") - 1)] for g in grid])
    return np.array(peaks), np.array(curves, np.float64)
pk, cv = peak_and_curve(cu_key)
print(f"peak resident waves per CU: min {pk.min()} median {int(np.median(pk))} max {pk.max()}  (32 = 8 per SIMD)")
print("mean resident waves per CU at 5 % steps of the span: " + " ".join(f"{v:.1f}" for v in cv.mean(axis=0)))
pk, cv = peak_and_curve(simd_key)
print(f"peak resident waves per SIMD: min {pk.min()} median {int(np.median(pk))} max {pk.max()}")
life = tot[ran]
print(f"wave life (shader cycles): p10 {np.percentile(life, 10):.0f} p50 {np.percentile(life, 50):.0f} p90 {np.percentile(life, 90):.0f} p99 {np.percentile(life, 99):.0f} max {life.max():.0f}")
