#!/usr/bin/env python3
"""Residency of the two PRODUCT blend kernels (GSR_CENSUS build: same registers / LDS as libgsr_hip.so, three scalar stamps per wave).
usage: make -C 3dgs-native_amd/csrc census; GSR_LIB=$PWD/3dgs-native_amd/libgsr_hip_census.so python tools/residency.py [C3]
Prints, per kernel: span, waves per SIMD averaged over the span, the peak per CU / per SIMD, the mean residency curve along the
span, wave-life percentiles and how unevenly the CUs finish."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gsr = importlib.import_module("3dgs-native_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = gsr.scenes.CONFIGS[name]
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
tiles = ((W + 15) // 16) * ((H + 15) // 16)
nw = {"fwd": tiles * 4, "bwd": tiles * 8}
fn = {"fwd": L.gsr_debug_fwd_census, "bwd": L.gsr_debug_bwd_census}


def step():
    img, depth, buf = gsr.render_gaussians(**kw)
    gsr.backward(background=bg, dL_dpixels=dpix, shs=shs, **P, viewmatrix=kw["viewmatrix"], projmatrix=kw["projmatrix"], tan_fovx=kw["tan_fovx"],
                 tan_fovy=kw["tan_fovy"], image_height=H, image_width=W, campos=kw["campos"], radii=buf["radii"], means2D=buf["points_xy_image"],
                 conic_opacity=buf["conic_opacity"], rgb=buf["colors"], cov3Ds=buf["cov3Ds"], clamped=buf["clamped_state"],
                 binning_buffer={"point_list": buf["point_list"]}, img_buffer={"ranges": buf["ranges"], "final_Ts": buf["final_Ts"], "n_contrib": buf["n_contrib"]})
    return buf


for _ in range(3):
    buf = step()
torch.cuda.synchronize()
for k in fn:
    assert fn[k](None, nw[k], 1) == 0
buf = step()
torch.cuda.synchronize()
ranges = buf["ranges"].cpu().numpy().reshape(-1, 2)
list_len = (ranges[:, 1] - ranges[:, 0]).astype(np.int64)


def curve(keys, r0, r1, grid):
    peaks, curves = [], []
    for k in np.unique(keys):
        m = keys == k
        ev = np.concatenate([np.stack([r0[m], np.ones(m.sum(), np.int64)], 1), np.stack([r1[m], -np.ones(m.sum(), np.int64)], 1)])
        ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
        conc = np.cumsum(ev[:, 1])
        peaks.append(conc.max())
        curves.append([conc[max(0, np.searchsorted(ev[:, 0], g, side="right") - 1)] for g in grid])
    return np.array(peaks), np.array(curves, np.float64)


for k in ("fwd", "bwd"):
    arr = np.zeros((nw[k], 4), np.uint64)
    assert fn[k](arr.ctypes.data_as(C.c_void_p), nw[k], 0) == 0
    ran = arr[:, 1] > 0
    hw = (arr[ran, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    xcc = ((arr[ran, 0] >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
    # HW_ID (gfx9): wave_id [3:0], simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]
    cu_key = (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
    simd_key = (cu_key << 2) | ((hw >> 4) & 3)
    r0, r1 = arr[ran, 1].astype(np.int64), arr[ran, 2].astype(np.int64)
    t0, t1 = r0.min(), r1.max()
    span = float(t1 - t0)
    n_simd = np.unique(simd_key).size
    grid = t0 + (np.linspace(0.025, 0.975, 20) * span).astype(np.int64)
    print(f"== blend_{k} ({name}): {int(ran.sum())} waves ran of {nw[k]}; span {span / 100:.1f} us; {np.unique(cu_key).size} CUs / {n_simd} SIMDs seen")
    print(f"   mean resident waves per SIMD over the span: {(r1 - r0).sum() / span / n_simd:.2f}")
    pk, cv = curve(cu_key, r0, r1, grid)
    print(f"   peak resident waves per CU: min {pk.min()} median {int(np.median(pk))} max {pk.max()}  (32 = 8 per SIMD)")
    print("   mean resident waves per CU at 5 % steps of the span: " + " ".join(f"{v:.1f}" for v in cv.mean(axis=0)))
    pk, _ = curve(simd_key, r0, r1, grid[:1])
    print(f"   peak resident waves per SIMD: min {pk.min()} median {int(np.median(pk))} max {pk.max()}")
    life = (r1 - r0) / 100.0
    print(f"   wave life (us): p10 {np.percentile(life, 10):.1f} p50 {np.percentile(life, 50):.1f} p90 {np.percentile(life, 90):.1f} p99 {np.percentile(life, 99):.1f} max {life.max():.1f}")
    # when does each CU run dry?
    last = np.array([r1[cu_key == c].max() for c in np.unique(cu_key)], np.float64)
    print(f"   CU finish time as a fraction of the span: p10 {np.percentile((last - t0) / span, 10):.2f} p50 {np.percentile((last - t0) / span, 50):.2f} p90 {np.percentile((last - t0) / span, 90):.2f}")
    start_frac = (r0 - t0) / span
    print(f"   wave start as a fraction of the span: p50 {np.percentile(start_frac, 50):.2f} p90 {np.percentile(start_frac, 90):.2f} p99 {np.percentile(start_frac, 99):.2f} max {start_frac.max():.2f}")
    # life against the tile's list length (is the tail made of deep tiles?)
    per_tile = 4 if k == "fwd" else 8
    wave_tile = np.flatnonzero(ran) // per_tile
    ll = list_len[wave_tile]
    late = r1 > t0 + 0.85 * span
    print(f"   list length of the tile: all waves mean {ll.mean():.0f}; waves still alive after 85 % of the span: {int(late.sum())}, mean list {ll[late].mean() if late.any() else 0:.0f}, mean life {life[late].mean() if late.any() else 0:.1f} us, mean start {start_frac[late].mean() if late.any() else 0:.2f}")
    print(f"   correlation(life, list length) = {np.corrcoef(life, ll)[0, 1]:.2f}")
