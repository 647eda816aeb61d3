#!/usr/bin/env python3
"""Why are the first ~20 steps after a torch.cuda.synchronize() 3-6 % slower than the rest (gpurun_out/r04_a/steps_200_*.txt)?
Per-step HIP-event times of the C3 step under four starts: (a) right after a synchronize, (b) after a synchronize and 20 ms of
host spinning, (c) after a synchronize while a long dummy kernel keeps the GPU busy until the first launch, (d) continuing
without any synchronize.  Also the library's per-stage events on steps 0-9 and 50-59 of an (a)-start: kernels slower, or gaps?"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
gsr = importlib.import_module("3dgs-native_amd")

def main():
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

    K = 60
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    for m in marks:
        m.record()
    for _ in range(150):
        step()
    big = torch.empty(1 << 28, dtype=torch.float32, device=dev)

    def run(label, before):
        torch.cuda.synchronize()
        before()
        marks[0].record()
        for k in range(K):
            step()
            marks[k + 1].record()
        torch.cuda.synchronize()
        a = np.array([marks[k].elapsed_time(marks[k + 1]) for k in range(K)])
        print(f"{label:34s} step0 {a[0]:.3f}  steps1-10 {a[1:11].mean():.4f}  11-20 {a[11:21].mean():.4f}  21-40 {a[21:41].mean():.4f}  41-59 {a[41:].mean():.4f}", flush=True)

    def spin_host():
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.02:
            pass

    def busy_gpu():
        for _ in range(6):
            big.mul_(1.0)      # ~6 x 0.5 ms of streaming kernels queued in front of the first step

    def nothing():
        pass

    import gc
    gc.collect(); gc.disable()
    for rep in range(2):
        run("(a) after synchronize", nothing)
        run("(b) synchronize + 20 ms host spin", spin_host)
        run("(c) synchronize + GPU kept busy", busy_gpu)
    # (d) no synchronize between two runs of K steps
    torch.cuda.synchronize()
    m2 = [torch.cuda.Event(enable_timing=True) for _ in range(2 * K + 1)]
    m2[0].record()
    for k in range(2 * K):
        step()
        m2[k + 1].record()
    torch.cuda.synchronize()
    a = np.array([m2[k].elapsed_time(m2[k + 1]) for k in range(2 * K)])
    print("(d) 120 steps in one go: tens", " ".join(f"{a[i:i + 10].mean():.4f}" for i in range(0, 2 * K, 10)), flush=True)
    # stage events on steps 0-9 and 50-59 after a synchronize
    for first in (0, 50):
        torch.cuda.synchronize()
        gsr._lib.stage_timing(True, 16, every=0)
        torch.cuda.synchronize()
        for k in range(K):
            if k == first:
                gsr._lib.stage_sampling(1)
            if k == first + 10:
                gsr._lib.stage_sampling(0)
            step()
        torch.cuda.synchronize()
        st, n = gsr._lib.stage_times()
        gsr._lib.stage_timing(False)
        print(f"stages of steps {first}-{first + 9} after a synchronize (n={n}): sum {sum(st.values()):.4f} ms ", {k: round(v, 4) for k, v in st.items()}, flush=True)

if __name__ == "__main__":
    main()
