"""
Host-side camera math for the rasterizer's callers.

The kernels consume 16-float row-major matrices under the row-vector convention (p' = p @ M).  This
module builds those matrices the way the reference's callers do, so scenes set up here reproduce the
reference's inputs bit-for-meaning:

* `nerf_camera`  -- NeRF-synthetic `transform_matrix` -> the dict train.py feeds to the rasterizer
                    (reference utils/camera_utils.py:8-89; utils/math_utils.py:8-41).
* `load_camera` / `world_to_view` / `projection_matrix` / `matrix_to_quaternion`
                 -- the reference's own helper names (utils/camera_utils.py:8-106, utils/math_utils.py:8-95) over the
                    same math, for callers written against them.
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


# ---- the reference's helper names ------------------------------------------------------------------------------------
def world_to_view(R, t, translate=np.array([0.0, 0.0, 0.0]), scale=1.0):
    """World-to-view matrix [[R^T, t], [0, 1]] with the camera centre moved by `translate` and scaled by `scale`
    (reference utils/math_utils.py:8-19), float32."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R).T
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    c2w = np.linalg.inv(Rt)
    c2w[:3, 3] = (c2w[:3, 3] + np.asarray(translate)) * scale
    return np.float32(np.linalg.inv(c2w))


def projection_matrix(fovx, fovy, znear, zfar):
    """Perspective matrix, z_sign = +1, float64 (reference utils/math_utils.py:21-41)."""
    return _projection(fovx, fovy, znear, zfar)


def matrix_to_quaternion(matrix):
    """3x3 rotation matrix -> quaternion (x, y, z, w) float32, largest-pivot form (reference utils/math_utils.py:43-95)."""
    m = np.asarray(matrix, dtype=np.float64)
    if abs(np.linalg.det(m) - 1.0) > 1e-5:
        print(f"Warning: Input matrix determinant is not 1: {np.linalg.det(m)}")
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    if tr > 0:
        s4 = 2.0 * np.sqrt(tr + 1.0)                         # 4w
        q = [(m[2, 1] - m[1, 2]) / s4, (m[0, 2] - m[2, 0]) / s4, (m[1, 0] - m[0, 1]) / s4, 0.25 * s4]
    else:
        a = 0 if (m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]) else (1 if m[1, 1] > m[2, 2] else 2)   # largest diagonal entry
        b, c = (a + 1) % 3, (a + 2) % 3
        s4 = 2.0 * np.sqrt(1.0 + m[a, a] - m[b, b] - m[c, c])  # 4 * q[a]
        q = [0.0, 0.0, 0.0, (m[c, b] - m[b, c]) / s4]
        q[a] = 0.25 * s4
        q[b] = (m[a, b] + m[b, a]) / s4
        q[c] = (m[a, c] + m[c, a]) / s4
    return np.array(q, dtype=np.float32)


_DISTORTION_KEYS = ("k1", "k2", "p1", "p2", "k3", "k4")
_CAMERA_TYPES = {"OPENCV": 0, None: 0, "OPENCV_FISHEYE": 1}


def load_camera(camera_info):
    """Camera dict from a `camera_info` record -- `camera_id`, `camera_to_world` (OpenGL/Blender axes), `width`, `height`,
    `focal`, optional `camera_model` and distortion coefficients -- with the 18 keys of reference
    utils/camera_utils.py:8-89."""
    _ = camera_info["camera_id"]
    width, height, focal = camera_info.get("width"), camera_info.get("height"), camera_info.get("focal")
    model = camera_info.get("camera_model", "OPENCV")
    if model not in _CAMERA_TYPES:
        raise ValueError(f"Unsupported camera_model '{model}'")
    # pose-dependent entries (R, T, world_to_camera, view_matrix, camera_center) come from the same code as nerf_camera; the
    # record carries the focal length itself, so everything that depends on it is formed from `focal` directly
    cam = nerf_camera(camera_info["camera_to_world"], width, height, 2.0 * math.atan(width / (2.0 * focal)))
    fovx, fovy = 2 * np.arctan(width / (2 * focal)), 2 * np.arctan(height / (2 * focal))
    proj = _projection(fovx, fovy, 0.01, 100.0).T
    c2w = np.asarray(camera_info["camera_to_world"], dtype=np.float64).copy()
    c2w[:3, 1:3] *= -1
    cam.update({"proj_matrix": proj, "full_proj_matrix": cam["world_to_camera"] @ proj, "tan_fovx": np.tan(fovx * 0.5),
                "tan_fovy": np.tan(fovy * 0.5), "fx": focal, "fy": focal, "cx": width / 2, "cy": height / 2, "camera_to_world": c2w,
                "camera_type": _CAMERA_TYPES[model],
                "distortion_params": np.array([camera_info.get(k, 0.0) for k in _DISTORTION_KEYS], dtype=np.float32)})
    return cam
