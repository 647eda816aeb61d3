"""
densify.py -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's adaptive density control (row f4 of SURVEY.md section 8(f)): one function per
Warp kernel, in float32 with the reference's operation order, and the trainer's `densification_and_pruning`
sequence (train.py:351-713) over them.  The product path (3dgs-native_amd/) never imports this module.

Parity pinning: UNPINNED.  The reference ships no densification output, checkpoint or PLY, and its `wp.randf` comes
from NVIDIA Warp (dependency `warp-lang`, version not pinned by the reference's readme), which is not installed here:
`rand_pcg` / `randf` below restate the algorithm of Warp's published native/rand.h (one PCG hash round; float from
the top 24 bits).  Where the reference's behaviour is undefined (reads / writes one row past an array) the defined
part is kept and the stray access dropped; each such place is marked "UB" below.
"""
import struct

import numpy as np

GROUPS = ("positions", "scales", "rotations", "opacities", "shs")
F = np.float32


# ---- wp.randf(wp.uint32(x)) -----------------------------------------------------------------------------------------
def rand_pcg(state):
    state = np.asarray(state).astype(np.uint32)
    with np.errstate(over="ignore"):
        b = state * np.uint32(747796405) + np.uint32(2891336453)
        c = ((b >> ((b >> np.uint32(28)) + np.uint32(4))) ^ b) * np.uint32(277803737)
    return (c >> np.uint32(22)) ^ c


def randf(state):
    return (rand_pcg(state) >> np.uint32(8)).astype(F) * F(1.0 / 16777216.0)


def _rows(params):
    return int(np.asarray(params["opacities"]).size)


def _shape(params):
    n = _rows(params)
    return {"positions": np.asarray(params["positions"], dtype=F).reshape(n, 3), "scales": np.asarray(params["scales"], dtype=F).reshape(n, 3),
            "rotations": np.asarray(params["rotations"], dtype=F).reshape(n, 4), "opacities": np.asarray(params["opacities"], dtype=F).reshape(n),
            "shs": np.asarray(params["shs"], dtype=F).reshape(n, 48)}


def alloc(n):
    return {"positions": np.zeros((n, 3), F), "scales": np.zeros((n, 3), F), "rotations": np.zeros((n, 4), F), "opacities": np.zeros(n, F),
            "shs": np.zeros((n, 48), F)}


# ---- kernels -------------------------------------------------------------------------------------------------------
def compute_grad_norms(pos_grad, n):
    """train.py:398-406 (wp.length = sqrt(x*x + y*y + z*z)); rows past the gradient array read as 0 (UB in the reference)."""
    g = np.zeros((n, 3), F)
    pg = np.asarray(pos_grad, dtype=F).reshape(-1, 3)
    m = min(n, len(pg))
    g[:m] = pg[:m]
    return np.sqrt((g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1]) + g[:, 2] * g[:, 2]).astype(F)


def mark_candidates(grad_norms, scales, grad_threshold, scene_extent, percent_dense, split):
    """optimizer.py:180-239"""
    s = np.asarray(scales, dtype=F).reshape(-1, 3)
    high = grad_norms >= F(grad_threshold)
    max_scale = np.maximum(np.maximum(s[:, 0], s[:, 1]), s[:, 2])
    thr = F(percent_dense) * F(scene_extent)
    size = (max_scale > thr) if split else (max_scale <= thr)
    return (high & size).astype(np.int32)


def exclusive_scan(mask):
    """wp.utils.array_scan(inclusive=False) and `int(prefix.numpy()[-1])` (train.py:432-433)"""
    mask = np.asarray(mask, dtype=np.int32)
    prefix = (np.cumsum(mask, dtype=np.int64) - mask).astype(np.int32)
    return prefix, (int(prefix[-1]) if len(prefix) else 0)


def clone_gaussians(params, mask, prefix, total, noise_scale=0.01):
    """optimizer.py:312-365"""
    p = _shape(params)
    n = _rows(params)
    out = alloc(n + total)
    for k in GROUPS:
        out[k][:n] = p[k]
    idx = np.nonzero(np.asarray(mask) == 1)[0]
    dst = np.asarray(prefix)[idx].astype(np.int64) + n
    keep = dst < n + total                     # UB: the reference writes row n+total when the last row is flagged
    idx, dst = idx[keep], dst[keep]
    i3 = (idx * 3).astype(np.int32)
    noise = np.stack([randf(i3) * F(noise_scale), randf(i3 + 1) * F(noise_scale), randf(i3 + 2) * F(noise_scale)], axis=1).astype(F)
    out["positions"][dst] = p["positions"][idx] + noise
    for k in ("scales", "rotations", "opacities", "shs"):
        out[k][dst] = p[k][idx]
    return out


def split_gaussians(params, mask, prefix, total, n_split=2, scale_factor=0.8):
    """optimizer.py:242-309"""
    p = _shape(params)
    n = _rows(params)
    n_out = n + total * n_split
    out = alloc(n_out)
    for k in GROUPS:
        out[k][:n] = p[k]
    idx = np.nonzero(np.asarray(mask) == 1)[0]
    for j in range(n_split):
        dst = n + np.asarray(prefix)[idx].astype(np.int64) * n_split + j
        keep = dst < n_out                     # optimizer.py:288
        ii, dd = idx[keep], dst[keep]
        d3 = (dd * 3).astype(np.int32)
        off = np.stack([(randf(d3) * F(2.0) - F(1.0)) * F(0.01), (randf(d3 + 1) * F(2.0) - F(1.0)) * F(0.01),
                        (randf(d3 + 2) * F(2.0) - F(1.0)) * F(0.01)], axis=1).astype(F)
        out["positions"][dd] = p["positions"][ii] + off
        out["scales"][dd] = p["scales"][ii] * F(scale_factor)
        for k in ("rotations", "opacities", "shs"):
            out[k][dd] = p[k][ii]
    return out


def split_removal_mask(split_mask, n_total):
    """train.py:547-576: valid = 1 - (i < offset and split_mask[i] == 1)"""
    valid = np.ones(n_total, np.int32)
    sm = np.asarray(split_mask)
    valid[:len(sm)][sm == 1] = 0
    return valid


def prune_mask(opacities, threshold):
    """optimizer.py:367-385"""
    return (np.asarray(opacities, dtype=F).reshape(-1) > F(threshold)).astype(np.int32)


def compact_gaussians(params, valid, prefix, count):
    """optimizer.py:387-416"""
    p = _shape(params)
    out = alloc(count)
    idx = np.nonzero(np.asarray(valid) != 0)[0]
    dst = np.asarray(prefix)[idx].astype(np.int64)
    keep = dst < count                         # UB: the last valid row lands one past the reference's output array
    for k in GROUPS:
        out[k][dst[keep]] = p[k][idx[keep]]
    return out


def init_gaussian_params(num_points, init_scale=0.1):
    """train.py:37-92"""
    n = int(num_points)
    out = alloc(n)
    i3 = (np.arange(n, dtype=np.int64) * 3).astype(np.int32)
    out["positions"] = np.stack([randf(i3) * F(2.6) - F(1.3), randf(i3 + 1) * F(2.6) - F(1.3), randf(i3 + 2) * F(2.6) - F(1.3)], axis=1).astype(F)
    out["scales"][:] = F(init_scale)
    out["rotations"][:, 0] = F(1.0)
    out["opacities"][:] = F(0.1)
    out["shs"][:, :3] = F(-0.007)
    return out


# ---- the trainer sequence -------------------------------------------------------------------------------------------
def densification_and_pruning(params, pos_grad, iteration, config, scene_extent):
    """train.py:351-713 on host arrays.  Returns (params, log); Adam state / gradients are the caller's to zero."""
    cfg = dict(config)
    params = _shape(params)
    densify_from_iter = cfg.get("densify_from_iter", 500)
    densify_until_iter = cfg.get("densify_until_iter", 15000)
    densification_interval = cfg.get("densification_interval", 100)
    opacity_reset_interval = cfg.get("opacity_reset_interval", 3000)
    log = {"cloned": 0, "split": 0, "split_removed": 0, "pruned": 0, "prune_skipped": False, "opacity_reset": False}
    if iteration > densify_from_iter and iteration < densify_until_iter and iteration % densification_interval == 0:
        grad_threshold = cfg.get("densify_grad_threshold", 0.0002)
        percent_dense = cfg.get("percent_dense", 0.01)
        norms = compute_grad_norms(pos_grad, _rows(params))
        clone_mask = mark_candidates(norms, params["scales"], grad_threshold, scene_extent, percent_dense, split=False)
        prefix, total = exclusive_scan(clone_mask)
        if total > 0:
            params = clone_gaussians(params, clone_mask, prefix, total, 0.01)
            log["cloned"] = total
        norms = compute_grad_norms(pos_grad, _rows(params))
        split_mask = mark_candidates(norms, params["scales"], grad_threshold, scene_extent, percent_dense, split=True)
        prefix, total = exclusive_scan(split_mask)
        if total > 0:
            params = split_gaussians(params, split_mask, prefix, total, 2, 0.8)
            log["split"] = total
            valid = split_removal_mask(split_mask, _rows(params))
            prefix, count = exclusive_scan(valid)
            if count < _rows(params):
                log["split_removed"] = _rows(params) - count
                params = compact_gaussians(params, valid, prefix, count)
        valid = prune_mask(params["opacities"], cfg.get("cull_opacity_threshold", 0.005))
        prefix, count = exclusive_scan(valid)
        n = _rows(params)
        ratio = (n - count) / n if n > 0 else 0
        if (count >= cfg.get("min_valid_points", 1000) and count <= cfg.get("max_valid_points", 1000000)
                and ratio <= cfg.get("max_allowed_prune_ratio", 0.5) and count < n):
            log["pruned"] = n - count
            params = compact_gaussians(params, valid, prefix, count)
        else:
            log["prune_skipped"] = True
    white = all(c == 1.0 for c in cfg.get("background_color", [0.0, 0.0, 0.0]))
    if iteration % opacity_reset_interval == 0 or (white and iteration == densify_from_iter):
        params["opacities"] = np.full_like(params["opacities"], F(0.01))
        log["opacity_reset"] = True
    return params, log


# ---- save_ply -------------------------------------------------------------------------------------------------------
def ply_bytes(params, num_points):
    """utils/point_cloud_utils.py:10-98 with plyfile's binary little-endian writer, vertex by vertex (small cases only)."""
    p = _shape(params)
    names = (["x", "y", "z", "scale_0", "scale_1", "scale_2", "opacity", "rot_x", "rot_y", "rot_z", "rot_w"], ["red", "green", "blue"],
             ["f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)])
    head = ["ply", "format binary_little_endian 1.0", f"element vertex {num_points}"]
    head += [f"property float {s}" for s in names[0]] + [f"property uchar {s}" for s in names[1]] + [f"property float {s}" for s in names[2]]
    head.append("end_header")
    blob = [("\n".join(head) + "\n").encode("ascii")]
    for i in range(num_points):
        sh = p["shs"][i].reshape(16, 3)
        col = np.clip(sh[0] + F(0.5), F(0.0), F(1.0))
        rgb = [int(np.clip(col[c] * 255, 0, 255)) for c in range(3)]
        blob.append(struct.pack("<11f", *p["positions"][i], *p["scales"][i], p["opacities"][i], *p["rotations"][i]))
        blob.append(struct.pack("<3B", *rgb))
        blob.append(struct.pack("<48f", *sh[0], *[sh[j][c] for j in range(1, 16) for c in range(3)]))
    return b"".join(blob)
