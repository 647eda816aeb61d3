"""
Adaptive density control: the trainer's `densification_and_pruning`
(reference train.py:351-713) over the HIP row movers of csrc/densify.hip.

`GaussianModel` carries what the reference keeps on its trainer object -- `params`, `grads`, `adam_m`,
`adam_v`, `num_points`, `scene_extent`, `config` (train.py:119-173) -- and `densification_and_pruning(it)` is the
same sequence: clone -> split (+ removal of the split originals) -> opacity prune -> opacity reset, with the
optimizer state re-created as zeros after every change of the point count (train.py:470-475).  Behaviours of the
reference that this keeps, because a drop-in must:

* every count is the LAST ENTRY of an exclusive scan (`int(prefix_sum.numpy()[-1])`, train.py:433/497/581/641), so the
  last row's own flag is never counted: a flagged last row is not cloned / split, and a compaction keeps one row
  fewer than the mask says whenever the last row is valid (the reference writes that row past the end of its
  output array; here it is dropped);
* the split pass runs over the post-clone arrays but the gradient norms were taken before the clone
  (train.py:408 vs 478-494): rows added by the clone have no gradient (the reference reads past `avg_grads`
  there; here the norm is 0, so a fresh clone is never split in the same call);
* position noise comes from Warp's stateless `randf(uint32)` hash, seeded by the row index (optimizer.py:297-299,
  353-355), so the same call always yields the same offsets.
"""
import ctypes as C

import numpy as np
import torch

from . import _host, _lib
from .optimizer import GROUPS

_WIDTH = {"positions": 3, "scales": 3, "rotations": 4, "opacities": 1, "shs": 48}


def _params_struct(params, n):
    for k in GROUPS:
        t = params[k]
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n * _WIDTH[k]):
            raise ValueError(f"densify: '{k}' must be a contiguous float32 device tensor with {n * _WIDTH[k]} elements")
    return _lib.GsrParams(n, *[_host.ptr(params[k]) for k in GROUPS])


def alloc_params(n, dev):
    """Zeroed parameter arrays in the trainer's shapes (train.py:446-452)."""
    return {"positions": torch.zeros(n, 3, dtype=torch.float32, device=dev), "scales": torch.zeros(n, 3, dtype=torch.float32, device=dev),
            "rotations": torch.zeros(n, 4, dtype=torch.float32, device=dev), "opacities": torch.zeros(n, dtype=torch.float32, device=dev),
            "shs": torch.zeros(n * 16, 3, dtype=torch.float32, device=dev)}


def _num_points(params):
    return int(params["opacities"].numel())


# ---- one wrapper per reference kernel -----------------------------------------------------------------------------
def mark_candidates(params, pos_grad, grad_threshold, scene_extent, percent_dense, split):
    """mark_clone_candidates / mark_split_candidates (optimizer.py:180-239) with compute_grad_norms (train.py:398-406)
    folded in.  `pos_grad` may have fewer rows than `params` (rows past it have norm 0)."""
    n = _num_points(params)
    dev = params["positions"].device
    p = _params_struct(params, n)
    n_grad = 0 if pos_grad is None else int(pos_grad.numel() // 3)
    if n_grad and not (pos_grad.is_cuda and pos_grad.dtype == torch.float32 and pos_grad.is_contiguous()):
        raise ValueError("densify: pos_grad must be a contiguous float32 device tensor")
    mask = torch.empty(n, dtype=torch.int32, device=dev)
    with _host.on_device(dev):
        _lib.check(_lib.lib().gsr_densify_mark(C.byref(p), _host.ptr(pos_grad) if n_grad else None, min(n_grad, n), float(grad_threshold),
                                               float(scene_extent), float(percent_dense), _lib.MARK_SPLIT if split else _lib.MARK_CLONE,
                                               _host.ptr(mask), _host.stream_ptr(dev)))
    return mask


def prune_mask(params, opacity_threshold):
    """prune_gaussians (optimizer.py:367-385): 1 = keep."""
    n = _num_points(params)
    dev = params["positions"].device
    p = _params_struct(params, n)
    valid = torch.empty(n, dtype=torch.int32, device=dev)
    with _host.on_device(dev):
        _lib.check(_lib.lib().gsr_prune_mark(C.byref(p), float(opacity_threshold), _host.ptr(valid), _host.stream_ptr(dev)))
    return valid


def split_removal_mask(split_mask, n_total):
    """mark_split_originals_for_removal + invert_mask (train.py:547-576): 1 = keep."""
    dev = split_mask.device
    valid = torch.empty(n_total, dtype=torch.int32, device=dev)
    with _host.on_device(dev):
        _lib.check(_lib.lib().gsr_split_removal_mask(n_total, int(split_mask.numel()), _host.ptr(split_mask), _host.ptr(valid),
                                                     _host.stream_ptr(dev)))
    return valid


def exclusive_scan(mask):
    """wp.utils.array_scan(mask, out, inclusive=False) and the count the reference reads from it (its last entry)."""
    n = int(mask.numel())
    dev = mask.device
    L = _lib.lib()
    prefix = torch.empty(n, dtype=torch.int32, device=dev)
    nbytes = int(L.gsr_mask_scan_workspace_bytes(n))
    scratch = _host.workspace("mask_scan", nbytes, dev)
    count = C.c_int32(0)
    with _host.on_device(dev):
        _lib.check(L.gsr_mask_scan(n, _host.ptr(mask), _host.ptr(prefix), C.byref(count), _host.ptr(scratch), nbytes, _host.stream_ptr(dev)))
    return prefix, int(count.value)


def clone_gaussians(params, mask, prefix, total, noise_scale=0.01):
    """clone_gaussians (optimizer.py:312-365) into freshly allocated arrays of N + total rows."""
    n = _num_points(params)
    dev = params["positions"].device
    out = alloc_params(n + total, dev)
    pin, pout = _params_struct(params, n), _params_struct(out, n + total)
    with _host.on_device(dev):
        _lib.check(_lib.lib().gsr_clone_gaussians(C.byref(pin), _host.ptr(mask), _host.ptr(prefix), float(noise_scale), C.byref(pout),
                                                  _host.stream_ptr(dev)))
    return out


def split_gaussians(params, mask, prefix, total, n_split=2, scale_factor=0.8):
    """split_gaussians (optimizer.py:242-309) into freshly allocated arrays of N + total*n_split rows."""
    n = _num_points(params)
    dev = params["positions"].device
    out = alloc_params(n + total * n_split, dev)
    pin, pout = _params_struct(params, n), _params_struct(out, n + total * n_split)
    with _host.on_device(dev):
        _lib.check(_lib.lib().gsr_split_gaussians(C.byref(pin), _host.ptr(mask), _host.ptr(prefix), int(n_split), float(scale_factor),
                                                  C.byref(pout), _host.stream_ptr(dev)))
    return out


def compact_gaussians(params, valid, prefix, count):
    """compact_gaussians (optimizer.py:387-416) into freshly allocated arrays of `count` rows."""
    n = _num_points(params)
    dev = params["positions"].device
    out = alloc_params(count, dev)
    pin, pout = _params_struct(params, n), _params_struct(out, count)
    with _host.on_device(dev):
        _lib.check(_lib.lib().gsr_compact_gaussians(C.byref(pin), _host.ptr(valid), _host.ptr(prefix), C.byref(pout), _host.stream_ptr(dev)))
    return out


def reset_opacities(opacities, max_opacity=0.01):
    """reset_opacities (optimizer.py:141-156), in place."""
    dev = opacities.device
    with _host.on_device(dev):
        _lib.check(_lib.lib().gsr_reset_opacities(int(opacities.numel()), float(max_opacity), _host.ptr(opacities), _host.stream_ptr(dev)))
    _host.written_in_place(opacities)


def init_gaussian_params(num_points, init_scale=0.1, device="cuda"):
    """The trainer's initial point set (train.py:37-92, 193-214): seeded by the row index alone, so every rank and every run
    starts from the same Gaussians."""
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    out = alloc_params(int(num_points), dev)
    p = _params_struct(out, int(num_points))
    with _host.on_device(dev):
        _lib.check(_lib.lib().gsr_init_gaussians(C.byref(p), float(init_scale), _host.stream_ptr(dev)))
    return out


def calculate_scene_extent(camera_centers, camera_extent_factor=1.0):
    """Radius of the camera positions around their centroid, at least 1 (train.py:233-257)."""
    c = np.asarray(camera_centers)
    if c.size == 0:
        return 1.0
    c = c.reshape(-1, 3)
    centre = np.mean(c, axis=0)
    radius = 0.0
    for pos in c:
        radius = max(radius, float(np.linalg.norm(pos - centre)))
    return max(radius * camera_extent_factor, 1.0)


class GaussianModel:
    """The part of NeRFGaussianSplattingTrainer (train.py:119-173) that densification touches."""

    def __init__(self, params, config=None, scene_extent=1.0):
        self.params = params
        self.num_points = _num_points(params)
        self.config = {"background_color": [0.0, 0.0, 0.0]}
        self.config.update(config or {})
        self.scene_extent = float(scene_extent)
        self.grads = self.create_gradient_arrays()
        self.adam_m = self.create_gradient_arrays()
        self.adam_v = self.create_gradient_arrays()

    def create_gradient_arrays(self):
        """train.py:216-231"""
        return alloc_params(self.num_points, self.params["positions"].device)

    def _replace(self, params):
        """train.py:470-475: new arrays in, gradients and both Adam moments back to zero."""
        self.params = params
        self.num_points = _num_points(params)
        self.grads = self.create_gradient_arrays()
        self.adam_m = self.create_gradient_arrays()
        self.adam_v = self.create_gradient_arrays()

    def densification_and_pruning(self, iteration):
        """train.py:351-713.  Returns a dict of what happened (the reference prints it)."""
        cfg = self.config
        densify_from_iter = cfg.get("densify_from_iter", 500)
        densify_until_iter = cfg.get("densify_until_iter", 15000)
        densification_interval = cfg.get("densification_interval", 100)
        opacity_reset_interval = cfg.get("opacity_reset_interval", 3000)
        log = {"cloned": 0, "split": 0, "split_removed": 0, "pruned": 0, "prune_skipped": False, "opacity_reset": False}

        if iteration > densify_from_iter and iteration < densify_until_iter and iteration % densification_interval == 0:
            pos_grads = self.grads["positions"]          # the norms of train.py:408 are taken from this snapshot
            grad_threshold = cfg.get("densify_grad_threshold", 0.0002)
            percent_dense = cfg.get("percent_dense", 0.01)

            # --- step 1: clone small Gaussians with high gradients (train.py:414-475)
            clone_mask = mark_candidates(self.params, pos_grads, grad_threshold, self.scene_extent, percent_dense, split=False)
            clone_prefix, total_to_clone = exclusive_scan(clone_mask)
            if total_to_clone > 0:
                self._replace(clone_gaussians(self.params, clone_mask, clone_prefix, total_to_clone, 0.01))
                log["cloned"] = total_to_clone

            # --- step 2: split large Gaussians with high gradients (train.py:477-626)
            split_mask = mark_candidates(self.params, pos_grads, grad_threshold, self.scene_extent, percent_dense, split=True)
            split_prefix, total_to_split = exclusive_scan(split_mask)
            if total_to_split > 0:
                self._replace(split_gaussians(self.params, split_mask, split_prefix, total_to_split, 2, 0.8))
                log["split"] = total_to_split
                valid = split_removal_mask(split_mask, self.num_points)
                prefix, valid_count = exclusive_scan(valid)
                if valid_count < self.num_points:
                    log["split_removed"] = self.num_points - valid_count
                    self._replace(compact_gaussians(self.params, valid, prefix, valid_count))

            # --- step 3: opacity pruning (train.py:628-692)
            valid = prune_mask(self.params, cfg.get("cull_opacity_threshold", 0.005))
            prefix, valid_count = exclusive_scan(valid)
            min_valid_points = cfg.get("min_valid_points", 1000)
            max_valid_points = cfg.get("max_valid_points", 1000000)
            max_prune_ratio = cfg.get("max_allowed_prune_ratio", 0.5)
            prune_count = self.num_points - valid_count
            prune_ratio = prune_count / self.num_points if self.num_points > 0 else 0
            if (valid_count >= min_valid_points and valid_count <= max_valid_points and prune_ratio <= max_prune_ratio
                    and valid_count < self.num_points):
                log["pruned"] = prune_count
                self._replace(compact_gaussians(self.params, valid, prefix, valid_count))
            else:
                log["prune_skipped"] = True

        # --- opacity reset (train.py:695-713)
        background_is_white = all(c == 1.0 for c in cfg["background_color"])
        if iteration % opacity_reset_interval == 0 or (background_is_white and iteration == densify_from_iter):
            reset_opacities(self.params["opacities"], 0.01)
            log["opacity_reset"] = True
        return log
