import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "3dgs-native_amd"   # not a Python identifier: import with importlib


def pkg():
    """The product package (host-side mirror of the reference's forward()/backward() surface)."""
    return importlib.import_module(PKG_NAME)


def sub(name):
    return importlib.import_module(f"{PKG_NAME}.{name}")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def cameras():
    return sub("cameras")


@pytest.fixture(scope="session")
def scenes():
    return sub("scenes")


def lego_camera(cameras_mod, frame=0, width=800, height=800):
    import json
    with open(os.path.join(ROOT, "tests", "golden", "lego_train_poses.json")) as f:
        d = json.load(f)
    return cameras_mod.nerf_camera(d["frames"][frame]["transform_matrix"], width, height, d["camera_angle_x"])


def render_kwargs(scene, cam, width=None, height=None, degree=3, train_convention=True, bg=(0.0, 0.0, 0.0)):
    """kwargs for render_gaussians as the reference's callers pass them (train.py:935-955 passes
    `world_to_camera`; render.py:104-125 passes `view_matrix`)."""
    return dict(
        background=np.asarray(bg, dtype=np.float32), means3D=scene["means"], colors=None,
        opacity=scene["opacities"], scales=scene["scales"], rotations=scene["rotations"], scale_modifier=1.0,
        viewmatrix=cam["world_to_camera"] if train_convention else cam["view_matrix"],
        projmatrix=cam["full_proj_matrix"], tan_fovx=cam["tan_fovx"], tan_fovy=cam["tan_fovy"],
        image_height=height or cam["height"], image_width=width or cam["width"], sh=scene["shs"],
        degree=degree, campos=cam["camera_center"], prefiltered=False, antialiasing=False, clamped=True)


def backward_kwargs(scene, cam, fwd_kwargs, buffers, dL_dpixels):
    """kwargs for backward() as train.py:1006-1044 builds them from the forward's buffer dict."""
    geom = {"radii": buffers["radii"], "means2D": buffers["points_xy_image"],
            "conic_opacity": buffers["conic_opacity"], "rgb": buffers["colors"],
            "clamped": buffers["clamped_state"]}
    return dict(
        background=fwd_kwargs["background"], means3D=scene["means"], dL_dpixels=dL_dpixels,
        opacity=scene["opacities"], shs=scene["shs"], scales=scene["scales"], rotations=scene["rotations"],
        scale_modifier=fwd_kwargs["scale_modifier"], viewmatrix=fwd_kwargs["viewmatrix"],
        projmatrix=fwd_kwargs["projmatrix"], tan_fovx=fwd_kwargs["tan_fovx"], tan_fovy=fwd_kwargs["tan_fovy"],
        image_height=fwd_kwargs["image_height"], image_width=fwd_kwargs["image_width"],
        campos=fwd_kwargs["campos"], radii=buffers["radii"], means2D=buffers["points_xy_image"],
        conic_opacity=buffers["conic_opacity"], rgb=buffers["colors"], cov3Ds=buffers["cov3Ds"],
        clamped=buffers["clamped_state"], geom_buffer=geom,
        binning_buffer={"point_list": buffers["point_list"]},
        img_buffer={"ranges": buffers["ranges"], "final_Ts": buffers["final_Ts"], "n_contrib": buffers["n_contrib"]},
        degree=fwd_kwargs["degree"], debug=False)
