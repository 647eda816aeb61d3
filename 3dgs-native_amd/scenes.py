"""
Scene inputs for the rasterizer: the reference's 3-Gaussian demo scene and the seeded synthetic scenes
the BASELINE.json configs are measured on.  Everything is float32 numpy in the
layouts `render_gaussians` takes: means (N,3), scales (N,3), rotations (N,4) as (x,y,z,w),
opacities (N,1), SH (N,16,3).
"""
import numpy as np

# SH table of the reference demo (input data of render.py:52-67; the same 16x3 block for every point).
_TOY_SH = np.array([
    [0.71734341, 0.91905449, 0.49961076], [0.08068483, 0.82132256, 0.01301602],
    [0.8335743, 0.31798138, 0.19709007], [0.82589597, 0.28206231, 0.790489],
    [0.24008527, 0.21312673, 0.53132892], [0.19493135, 0.37989934, 0.61886235],
    [0.98106522, 0.28960672, 0.57313965], [0.92623716, 0.46034381, 0.5485369],
    [0.81660616, 0.7801104, 0.27813915], [0.96114063, 0.69872817, 0.68313804],
    [0.95464185, 0.21984855, 0.92912192], [0.23503135, 0.29786121, 0.24999751],
    [0.29844887, 0.6327788, 0.05423596], [0.08934335, 0.11851827, 0.04186001],
    [0.59331831, 0.919777, 0.71364335], [0.83377388, 0.40242542, 0.8792624]])


def toy_scene():
    """The 3 Gaussians of the reference demo (render.py:52-80): unit scale, opacity 1, rot (0,0,0,1)."""
    pts = np.array([[-5, 0, -10], [0, 0, -10], [5, 0, -10]], dtype=np.float32)
    n = len(pts)
    shs = np.tile(_TOY_SH[None], (n, 1, 1))          # float64, as the reference builds it
    rot = np.zeros((n, 4), np.float32)
    rot[:, 3] = 1.0
    return {"means": pts, "shs": shs, "scales": np.ones((n, 3), np.float32), "rotations": rot,
            "opacities": np.ones((n, 1), np.float32), "colors": np.ones((n, 3), np.float32)}


def synthetic_scene(n, scale_median, scale_sigma, seed, extent=1.3):
    """Seeded random scene: positions U(-extent,extent)^3 (the reference's init range, train.py:52-56),
    scales lognormal clipped >= 1e-3, unit quaternions, opacity U(0.05,0.95), SH DC N(0,0.5), rest N(0,0.1)."""
    rng = np.random.default_rng(seed)
    means = rng.uniform(-extent, extent, (n, 3)).astype(np.float32)
    scales = np.exp(rng.normal(np.log(scale_median), scale_sigma, (n, 3))).astype(np.float32)
    scales = np.maximum(scales, np.float32(1e-3))
    q = rng.normal(0.0, 1.0, (n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = rng.uniform(0.05, 0.95, (n, 1)).astype(np.float32)
    shs = np.empty((n, 16, 3), np.float32)
    shs[:, 0] = rng.normal(0.0, 0.5, (n, 3))
    shs[:, 1:] = rng.normal(0.0, 0.1, (n, 15, 3))
    return {"means": means, "shs": shs, "scales": scales, "rotations": q.astype(np.float32), "opacities": opac}


# BASELINE.json configs: name -> generator arguments and image size.
CONFIGS = {
    "C2": dict(n=100_000, scale_median=0.02, scale_sigma=0.5, seed=1234, width=800, height=800),
    "C3": dict(n=1_000_000, scale_median=0.01, scale_sigma=0.6, seed=2025, width=800, height=800),
    "C5": dict(n=5_000_000, scale_median=0.005, scale_sigma=0.6, seed=5, width=1920, height=1080),
    # The regime the reference trainer STARTS in (train.py:37-92, config.py:30-31,62): its own initial point set -- positions
    # randf-hashed in (-1.3, 1.3)^3, constant scale 0.1, opacity 0.1, quaternion (1,0,0,0) as stored -- at the reference's
    # default 5 000 points (C0) and at 100 000 (C2i; SURVEY.md section 6 "#2-like, reference init": D = 12.1 M, 4 854 entries
    # per tile).  Built on the device by gsr_init_gaussians (densify.init_gaussian_params), not by synthetic_scene.
    "C0": dict(n=5_000, init_scale=0.1, width=800, height=800, seed=0),
    "C2i": dict(n=100_000, init_scale=0.1, width=800, height=800, seed=0),
}

# Lego train frame 0 (reference data/lego/transforms_train.json), used as the benchmark pose so
# bench.py needs no data file; the first 8 frames are also in tests/golden/lego_train_poses.json.
LEGO_CAMERA_ANGLE_X = 0.6911112070083618
LEGO_FRAME0 = [[-0.9999021887779236, 0.004192245192825794, -0.013345719315111637, -0.05379832163453102],
               [-0.013988681137561798, -0.2996590733528137, 0.95394366979599, 3.845470428466797],
               [-4.656612873077393e-10, 0.9540371894836426, 0.29968830943107605, 1.2080823183059692],
               [0.0, 0.0, 0.0, 1.0]]


def orbit_pose(k, n_views=8):
    """Pose k of n_views: LEGO_FRAME0 rotated about the world z axis by 2*pi*k/n_views (multi-view bench)."""
    a = 2.0 * np.pi * k / n_views
    Rz = np.array([[np.cos(a), -np.sin(a), 0, 0], [np.sin(a), np.cos(a), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    return (Rz @ np.asarray(LEGO_FRAME0, dtype=np.float64)).tolist()
