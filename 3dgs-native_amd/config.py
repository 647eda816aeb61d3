"""Constants of the rasterizer path (reference config.py:17-23): 16x16 tiles, VEC6 = 6 packed floats
in upper-triangle order (xx,xy,xz,yy,yz,zz).  The device is whatever torch device the inputs live on;
there is no DEVICE global and no CPU fallback."""
TILE_M = 16
TILE_N = 16
TILE_THREADS = 256
VEC6_LEN = 6
SH_STRIDE = 16
MAX_RENDERED = 1 << 30


class GaussianParams:
    """Training hyper-parameters under the reference's class name and access pattern (reference config.py:26-112:
    class attributes, `update(**kw)`, `get_config_dict()`); the values are the reference's defaults."""

    _DEFAULTS = dict(
        # training
        num_iterations=7000, num_points=5000, save_interval=500,
        # learning-rate schedule (scheduler.LRScheduler)
        use_lr_scheduler=True,
        lr_scheduler_config={"lr_pos": 1e-2, "lr_scale": 5e-3, "lr_rot": 5e-3, "lr_sh": 2e-3, "lr_opac": 5e-3, "final_lr_factor": 0.01},
        # Adam
        adam_beta1=0.9, adam_beta2=0.999, adam_epsilon=1e-8,
        # density control (densify.GaussianModel)
        densification_interval=100, pruning_interval=100, opacity_reset_interval=3000, densify_grad_threshold=0.0002,
        cull_opacity_threshold=0.005, start_prune_iter=500, end_prune_iter=15000, percent_dense=0.01, max_allowed_prune_ratio=1.0,
        # Gaussians
        initial_scale=0.1, scale_modifier=1.0, sh_degree=3,
        # scene / rendering
        scene_scale=1.0, background_color=[0.0, 0.0, 0.0], near=0.01, far=100.0,
        # loss
        lambda_dssim=0.0)

    @classmethod
    def update(cls, **kwargs):
        for key, value in kwargs.items():
            if key not in cls._DEFAULTS:
                raise ValueError(f"Unknown parameter: {key}")
            setattr(cls, key, value)

    @classmethod
    def get_config_dict(cls):
        return {key: getattr(cls, key) for key in cls._DEFAULTS}


for _k, _v in GaussianParams._DEFAULTS.items():
    setattr(GaussianParams, _k, _v)
