import sys, os, importlib, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import conftest, parity
from oracle import oracle
import test_gpu_fuzz as F
gsr = importlib.import_module("3dgs-native_amd")
cameras, scenes = gsr.cameras, gsr.scenes
seed = int(sys.argv[1])
sc, cam, W, H, degree, tc, bg = F._case(scenes, cameras, seed)
kw = conftest.render_kwargs(sc, cam, width=W, height=H, degree=degree, train_convention=tc, bg=bg)
ref = oracle.render_gaussians(**kw)
dpix = (np.random.default_rng(0).normal(0, 1, (H, W, 3)) / (H * W * 3)).astype(np.float32)
import inspect
src = inspect.getsource(importlib.import_module("test_gpu_parity")._fwd_bwd)
print("N", sc["means"].shape[0], W, H, "degree", degree, "D", len(ref[2]["point_list"]))
g32 = oracle.backward(**conftest.backward_kwargs(sc, cam, kw, ref[2], dpix))
g64 = oracle.backward(**conftest.backward_kwargs(sc, cam, kw, ref[2], dpix), accumulate="f64")
runs = []
for r in range(3):
    got = gsr.render_gaussians(**kw)
    runs.append({k: parity.to_np(v) for k, v in gsr.backward(**conftest.backward_kwargs(sc, cam, kw, got[2], dpix)).items() if k.startswith("dL_") and v is not None})
for k in ("dL_dmean3D", "dL_dscale", "dL_drot", "dL_dconic", "dL_dmean2D"):
    m = np.abs(g32[k]).max()
    f = lambda a, b: (parity.grad_margin(a, b)[0], float(np.abs(a.astype(np.float64) - b).max() / m))
    print(k, "gpu-vs-f32", f(runs[0][k], g32[k]), "gpu-vs-f64acc", f(runs[0][k], g64[k]), "f32-vs-f64acc", f(g32[k], g64[k]), "gpu run-to-run", f(runs[0][k], runs[1][k]), f(runs[1][k], runs[2][k]))
