"""
Host-side camera math for the rasterizer's callers.

The kernels consume 16-float row-major matrices under the row-vector convention (p' = p @ M).  This
module builds those matrices the way the reference's callers do, so scenes set up here reproduce the
reference's inputs bit-for-meaning:

* `nerf_camera`  -- NeRF-synthetic `transform_matrix` -> the dict train.py feeds to the rasterizer
                    (reference utils/camera_utils.py:8-89; utils/math_utils.py:8-41).
* `toy_camera`   -- the hard-coded camera of the 3-Gaussian demo (reference render.py:11-50), including
                    its quirk of passing the un-transposed `world_to_view` matrix as the view matrix
                     and degrees-as-radians FoV (tan(22.5 rad)).
"""
import math

import numpy as np


def _world_to_view(R, t):
    """[[R^T, t],[0,1]] as float32 (reference utils/math_utils.py:8-19 with translate=0, scale=1)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R).T
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    return np.float32(np.linalg.inv(np.linalg.inv(Rt)))


def _projection(fovx, fovy, znear, zfar):
    """OpenGL-style perspective matrix, z_sign=+1 (reference utils/math_utils.py:21-41)."""
    ty, tx = math.tan(fovy / 2), math.tan(fovx / 2)
    top, right = ty * znear, tx * znear
    P = np.zeros((4, 4))
    P[0, 0] = 2.0 * znear / (2.0 * right)
    P[1, 1] = 2.0 * znear / (2.0 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def nerf_camera(transform_matrix, width, height, camera_angle_x, znear=0.01, zfar=100.0):
    """Camera dict for one NeRF-synthetic frame (keys as reference utils/camera_utils.py:67-86)."""
    c2w = np.asarray(transform_matrix, dtype=np.float64).copy()
    c2w[:3, 1:3] *= -1                       # OpenGL/Blender axes -> COLMAP (Y down, Z forward)
    w2c = np.linalg.inv(c2w).astype(np.float32)
    R = w2c[:3, :3].copy()
    T = w2c[:3, 3].copy()
    w2c[3, 3] = 1.0
    world_to_camera = w2c.T.copy()           # row-vector form: translation in row 3
    focal = 0.5 * width / np.tan(0.5 * camera_angle_x)   # reference train.py:296
    fovx = 2 * np.arctan(width / (2 * focal))
    fovy = 2 * np.arctan(height / (2 * focal))
    proj = _projection(fovx, fovy, znear, zfar).T
    full_proj = world_to_camera @ proj       # float32 @ float64 -> float64
    return {
        "R": R, "T": T,
        "world_to_camera": world_to_camera,
        "view_matrix": _world_to_view(R, T),
        "proj_matrix": proj,
        "full_proj_matrix": full_proj,
        "tan_fovx": np.tan(fovx * 0.5), "tan_fovy": np.tan(fovy * 0.5),
        "camera_center": np.linalg.inv(world_to_camera)[3, :3],
        "width": int(width), "height": int(height), "fx": focal, "fy": focal,
    }


def toy_camera(image_width=1800, image_height=1800, fovx=45.0, fovy=45.0, znear=0.01, zfar=100.0):
    """Camera of the reference's 3-Gaussian demo (render.py:11-50)."""
    T = np.array([0, 0, 5], dtype=np.float32)
    R = np.array([[1, 0, 0], [0, 1, 0], [0, 0, -1]], dtype=np.float32)
    w2c = np.eye(4, dtype=np.float32)
    w2c[:3, :3] = R
    w2c[:3, 3] = T
    world_to_camera = w2c.T.copy()
    proj = _projection(fovx, fovy, znear, zfar).T
    tan_fovx, tan_fovy = math.tan(fovx * 0.5), math.tan(fovy * 0.5)
    return {
        "R": R, "T": T,
        "world_to_camera": world_to_camera,
        "view_matrix": _world_to_view(R, T),          # what render.py:112 passes as `viewmatrix`
        "proj_matrix": proj,
        "full_proj_matrix": world_to_camera @ proj,
        "tan_fovx": tan_fovx, "tan_fovy": tan_fovy,
        "camera_center": np.linalg.inv(world_to_camera)[3, :3],
        "width": image_width, "height": image_height,
        "focal_x": image_width / (2 * tan_fovx), "focal_y": image_height / (2 * tan_fovy),
    }
