"""
Checkpoint writer: `save_ply` of reference utils/point_cloud_utils.py:10-98,
called from train.py:796-803.  Same vertex layout, field names and byte order as the file the reference writes through
`plyfile` (binary little-endian, packed records): x y z, scale_0..2, opacity, rot_x..w, red green blue (uchar),
f_dc_0..2, f_rest_0..44.  Host I/O only -- the arrays are copied off the device once and packed with numpy, there is
no per-vertex Python loop.  `load_ply` reads such a file back (the reference has no reader; this one exists for
round-trip tests and for resuming from a checkpoint).
"""
import os

import numpy as np

VERTEX_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("scale_0", "<f4"), ("scale_1", "<f4"), ("scale_2", "<f4"), ("opacity", "<f4"),
     ("rot_x", "<f4"), ("rot_y", "<f4"), ("rot_z", "<f4"), ("rot_w", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1"),
     ("f_dc_0", "<f4"), ("f_dc_1", "<f4"), ("f_dc_2", "<f4")] + [(f"f_rest_{i}", "<f4") for i in range(45)])
_PLY_TYPE = {"f4": "float", "u1": "uchar"}


def _host(x, shape):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    elif hasattr(x, "numpy"):
        x = x.numpy()
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32)).reshape(shape)


def vertex_records(params, num_points, colors=None):
    """The packed vertex table of point_cloud_utils.py:36-91 as one structured array."""
    n = int(num_points)
    pos = _host(params["positions"], (-1, 3))[:n]
    scl = _host(params["scales"], (-1, 3))[:n]
    rot = _host(params["rotations"], (-1, 4))[:n]
    opa = _host(params["opacities"], (-1,))[:n]
    shs = _host(params["shs"], (-1, 16, 3))[:n]
    if colors is not None:
        col = _host(colors, (-1, 3))[:n]
    else:
        # DC term only: clip(sh_dc + 0.5, 0, 1) in float32 (point_cloud_utils.py:28-34)
        col = np.clip(shs[:, 0, :] + np.float32(0.5), np.float32(0.0), np.float32(1.0)).astype(np.float32)
    v = np.zeros(n, dtype=VERTEX_DTYPE)
    v["x"], v["y"], v["z"] = pos[:, 0], pos[:, 1], pos[:, 2]
    v["scale_0"], v["scale_1"], v["scale_2"] = scl[:, 0], scl[:, 1], scl[:, 2]
    v["opacity"] = opa
    v["rot_x"], v["rot_y"], v["rot_z"], v["rot_w"] = rot[:, 0], rot[:, 1], rot[:, 2], rot[:, 3]   # stored order, labelled x y z w (:49-50)
    # int(np.clip(c * 255, 0, 255)): float32 product, truncation toward zero (:53-58)
    rgb = np.clip(col * np.float32(255), np.float32(0), np.float32(255)).astype(np.int64)
    v["red"], v["green"], v["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    v["f_dc_0"], v["f_dc_1"], v["f_dc_2"] = shs[:, 0, 0], shs[:, 0, 1], shs[:, 0, 2]
    rest = shs[:, 1:, :].reshape(n, 45)          # coefficient-major, channel-minor (:65-69)
    for i in range(45):
        v[f"f_rest_{i}"] = rest[:, i]
    return v


def ply_header(n):
    lines = ["ply", "format binary_little_endian 1.0", f"element vertex {n}"]
    lines += [f"property {_PLY_TYPE[VERTEX_DTYPE[name].str[1:]]} {name}" for name in VERTEX_DTYPE.names]
    lines.append("end_header")
    return ("\n".join(lines) + "\n").encode("ascii")


def save_ply(params, filepath, num_points, colors=None):
    """reference utils/point_cloud_utils.py:10 -- same signature."""
    v = vertex_records(params, num_points, colors)
    d = os.path.dirname(str(filepath))
    if d:
        os.makedirs(d, exist_ok=True)
    with open(filepath, "wb") as f:
        f.write(ply_header(len(v)))
        v.tofile(f)


def load_ply(filepath):
    """Read a file written by save_ply (or by the reference) back into the trainer's five arrays (+ the uchar colours)."""
    with open(filepath, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        n, fields, fmt = None, [], None
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PLY header not terminated")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if tok[1] != "vertex" or n is not None:
                    raise ValueError("only a single vertex element is supported")
                n = int(tok[2])
            elif tok[0] == "property":
                fields.append((tok[2], {"float": "<f4", "float32": "<f4", "uchar": "u1", "uint8": "u1"}[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt != "binary_little_endian" or n is None:
            raise ValueError("expected a binary_little_endian PLY with a vertex element")
        v = np.fromfile(f, dtype=np.dtype(fields), count=n)
    if len(v) != n:
        raise ValueError("PLY file is truncated")
    shs = np.zeros((n, 16, 3), dtype=np.float32)
    for c in range(3):
        shs[:, 0, c] = v[f"f_dc_{c}"]
    for i in range(45):
        shs[:, 1 + i // 3, i % 3] = v[f"f_rest_{i}"]
    return {"positions": np.stack([v["x"], v["y"], v["z"]], axis=1), "scales": np.stack([v["scale_0"], v["scale_1"], v["scale_2"]], axis=1),
            "rotations": np.stack([v["rot_x"], v["rot_y"], v["rot_z"], v["rot_w"]], axis=1), "opacities": np.array(v["opacity"]),
            "shs": shs.reshape(n * 16, 3), "colors": np.stack([v["red"], v["green"], v["blue"]], axis=1)}
