"""
3dgs-native_amd: MI355X (gfx950) 3D Gaussian Splatting rasterizer behind the forward()/backward()
call surface of zhujinchong/3DGS-native.

    import importlib
    gsr = importlib.import_module("3dgs-native_amd")
    image, depth, buffers = gsr.render_gaussians(...)     # reference forward.py:629
    grads = gsr.backward(...)                             # reference backward.py:955

Compute lives in libgsr_hip.so (hand-written HIP, C ABI in include/gsr.h); Python only marshals
pointers.  There is no fallback implementation.
"""
from .forward import render_gaussians  # noqa: F401
from .backward import backward  # noqa: F401
from . import cameras, scenes, config, dist, loss, optimizer, scheduler, densify, point_cloud  # noqa: F401

__all__ = ["render_gaussians", "backward", "cameras", "scenes", "config", "dist", "loss", "optimizer", "scheduler", "densify", "point_cloud"]
