"""Seeded sweep of small random configurations through the whole forward + backward parity comparison (HIP vs oracle):
ragged sizes, every SH degree, both matrix conventions, tiny and huge splats, near-plane crossings, duplicate depths,
transparent and opaque Gaussians.  GSR_FUZZ_CASES sets the number of cases (default 32 -- seed 27, N = 1 with one big off-centre splat, once read SH rows past the array; the last round-1 sweep ran 20 000 cases, all passing)."""
import os

import numpy as np
import pytest

from conftest import lego_camera
from test_gpu_parity import _fwd_bwd

pytestmark = pytest.mark.gpu
CASES = int(os.environ.get("GSR_FUZZ_CASES", "32"))


def _case(scenes, cameras, seed):
    rng = np.random.default_rng(1000 + seed)
    edge = [16, 17, 31, 32, 33, 48, 255, 256, 257]                          # tile-grid edges
    W = int(rng.choice(edge)) if rng.integers(0, 3) == 0 else int(rng.integers(17, 260))
    H = int(rng.choice(edge[:6])) if rng.integers(0, 3) == 0 else int(rng.integers(17, 200))
    n = int(rng.choice([1, 2, 7, 63, 64, 65, 255, 256, 257, 300, 1023, 1024, 1025, 1500, 4000, 4097]))   # wave / block / chunk edges
    sc = scenes.synthetic_scene(n, float(rng.choice([0.005, 0.03, 0.12, 0.5])), float(rng.uniform(0.1, 1.2)), seed=5000 + seed,
                                extent=float(rng.choice([0.6, 1.3, 3.5])))    # 3.5 reaches behind the camera / the near plane
    kind = seed % 6
    if kind == 1:      # duplicated depths and positions: ties must keep id order
        sc["means"][n // 2:] = sc["means"][: n - n // 2]
    elif kind == 2:    # nearly transparent / nearly opaque
        sc["opacities"][::2] = 0.003
        sc["opacities"][1::2] = 0.999
    elif kind == 3:    # anisotropic (8:1 on top of the lognormal spread; every case caps the longest axis at 1.5 scene units).  Beyond
        # ~100:1 with splats over 1000 px the comparison stops measuring the kernels: the conic is near-singular, the
        # oracle's and the GPU's float32 sums both lose digits, and the GPU's own run-to-run spread (atomic order) reaches the
        # GPU-oracle difference (seeds 129 / 208 of the first 300-case sweep, scales 2.48 : 0.024 and 7.3 : 0.019; seed 63 of a
        # later one, 1.5 : 0.0038 over 1000 px, 4 of 3075 dL_dmean3D components at the level of the run-to-run spread).
        sc["scales"][:, 0] *= 8.0
    elif kind == 4:    # bright SH: colours clamp on both sides
        sc["shs"] *= 6.0
    sc["scales"] = np.minimum(sc["scales"], np.float32(1.5))   # see kind 3: no axis longer than 1.5 scene units ...
    sc["scales"] = np.maximum(sc["scales"], sc["scales"].max(axis=1, keepdims=True) / np.float32(50.0))   # ... nor thinner than 50:1
    cam = lego_camera(cameras, frame=int(rng.integers(0, 8)), width=W, height=H)
    return sc, cam, W, H, int(rng.integers(0, 4)), bool(rng.integers(0, 2)), tuple(rng.uniform(0, 1, 3).round(2))


@pytest.mark.parametrize("seed", range(CASES))
def test_random_configuration(oracle, cameras, scenes, seed):
    sc, cam, W, H, degree, train_convention, bg = _case(scenes, cameras, seed)
    _fwd_bwd(oracle, sc, cam, W, H, degree=degree, bg=bg, train_convention=train_convention)
