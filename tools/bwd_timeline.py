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
arr = np.zeros((waves, 8), np.uint64)
assert L.gsr_debug_bwd_phases(arr.ctypes.data_as(C.c_void_p), waves) == 0
names = ["prologue", "fill (compaction)", "record gather", "pixel loop", "flush (transpose + atomics)"]
tot = arr[:, :5].sum(axis=1).astype(np.float64)
print(f"waves {waves}, buckets {int(arr[:, 6].sum())}, (bucket, pixel) steps {int(arr[:, 7].sum())}, cycles per wave: mean {tot.mean():.0f}, median {np.median(tot):.0f}, max {tot.max():.0f}")
for k, nme in enumerate(names):
    col = arr[:, k].astype(np.float64)
    print(f"  {nme:30s} mean {col.mean():9.0f}  p50 {np.median(col):9.0f}  p95 {np.percentile(col, 95):9.0f} cycles/wave  {100.0 * col.sum() / tot.sum():5.1f} %")
steps = arr[:, 7].astype(np.float64)
print(f"  cycles per (bucket, pixel) step (wave mean): {(arr[:, 3].astype(np.float64)[steps > 0] / steps[steps > 0]).mean():.0f}")

# occupancy over the kernel's life from the waves' start ([5]) and end (start + phases) stamps.  Every XCD has its own counter
# (offsets of 1e11 cycles): the waves are grouped by counter domain, one curve per XCD, 1024 wave slots each at 8 waves per SIMD.
start_all = arr[:, 5].astype(np.float64)
life_all = tot
print("per XCD: span (cycles), waves, mean resident waves (of 1024 slots), then resident waves at 5 % steps of the span")
order = np.argsort(start_all)
cuts = np.flatnonzero(np.diff(start_all[order]) > 2e6) + 1 # a new counter domain wherever consecutive starts are > 2e6 cycles apart
for x, grp in enumerate(np.split(order, cuts)):
    st, lf = start_all[grp], life_all[grp]
    ok = st > 0
    st, lf = st[ok], lf[ok]
    if st.size == 0:
        continue
    en = st + lf
    t0, t1 = st.min(), en.max()
    ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([en, -np.ones_like(en)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    conc = np.cumsum(ev[:, 1])
    pts = [int(conc[np.searchsorted(ev[:, 0], t0 + f * (t1 - t0), side="right") - 1]) for f in np.linspace(0.025, 0.975, 20)]
    print(f"  XCD {x}: span {t1 - t0:8.0f}, {ok.sum()} waves, mean {lf.sum() / (t1 - t0):6.0f} | " + " ".join(f"{v:4d}" for v in pts))
life = tot[start_all > 0]
print(f"wave life: p10 {np.percentile(life, 10):.0f} p50 {np.percentile(life, 50):.0f} p90 {np.percentile(life, 90):.0f} p99 {np.percentile(life, 99):.0f} max {life.max():.0f}")
