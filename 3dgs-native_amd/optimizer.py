"""
Fused Adam step: the reference's `adam_update` kernel (reference optimizer.py:7-139,
launched from train.py:716-794) with its clamps -- scale >= 1e-3, quaternion renormalised, opacity in [0,1].
Parameters, moments and gradients are device tensors updated in place; gradients may be the views of the
arena `backward()` returns (after `dist.reduce_gradients`).
"""
import ctypes as C

import torch

from . import _host, _lib

GROUPS = ("positions", "scales", "rotations", "opacities", "shs")   # the reference's dict keys (train.py:129-143)
DEFAULT_LR = {"positions": 1e-2, "scales": 5e-3, "rotations": 5e-3, "shs": 2e-3, "opacities": 5e-3}   # config.py:36-42


def make_state(params):
    """Zero first/second moments shaped like the parameters."""
    return ({k: torch.zeros_like(params[k]) for k in GROUPS}, {k: torch.zeros_like(params[k]) for k in GROUPS})


def grads_from_backward(grads):
    """Map backward()'s dict onto the optimizer's group names (what train.py:1047-1051 copies)."""
    return {"positions": grads["dL_dmean3D"], "scales": grads["dL_dscale"], "rotations": grads["dL_drot"],
            "opacities": grads["dL_dopacity"], "shs": grads["dL_dshs"]}


def adam_update(params, grads, m, v, lrs=None, beta1=0.9, beta2=0.999, epsilon=1e-8, iteration=0, sh_views=None, sh_degree=3, sh_scale=None):
    """In-place Adam step over the five parameter groups.  `iteration` is 0-based (bias correction uses +1).

    `sh_views`: instead of a dense `grads["shs"]` (which may then be None), the view payloads of
    backward(..., sh_gradient="factored") -- a [V, 3N + 4] tensor or a list of V [3N + 4] tensors (one view on one GPU, or all
    views after dist's all-gather).  The SH group is then updated with sh_scale * sum_v basis(dir_v) x dL_drgb_v formed inside
    the update kernel (gsr_adam_update_views): bit for bit the step dist.sh_gradients_from_views + this function would take,
    without the 192-byte-per-Gaussian gradient being written by the backward and read back here.  `sh_scale` defaults to 1/V."""
    L = _lib.lib()
    lrs = DEFAULT_LR if lrs is None else lrs
    dev = params["positions"].device
    n = int(params["positions"].shape[0])
    expect = {"positions": 3 * n, "scales": 3 * n, "rotations": 4 * n, "opacities": n, "shs": 48 * n}
    rows = None
    if sh_views is not None:
        rows = [sh_views[i] for i in range(len(sh_views))]
        if not 1 <= len(rows) <= 16:
            raise ValueError("adam_update: sh_views must hold 1..16 view payloads (GSR_MAX_VIEWS)")
        for r in rows:
            if not (isinstance(r, torch.Tensor) and r.is_cuda and r.dtype == torch.float32 and r.is_contiguous() and r.numel() == 3 * n + 4):
                raise ValueError(f"adam_update: each view payload must be a contiguous float32 device tensor of {3 * n + 4} elements")
    groups = []
    for k in GROUPS:
        for d in (params, grads, m, v):
            if d is grads and k == "shs" and rows is not None:
                continue
            t = d[k]
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == expect[k]):
                raise ValueError(f"adam_update: '{k}' must be a contiguous float32 device tensor with {expect[k]} elements")
        g = None if (k == "shs" and rows is not None) else grads[k]
        groups.append(_lib.GsrAdamGroup(_host.ptr(params[k]), _host.ptr(g), _host.ptr(m[k]), _host.ptr(v[k]), float(lrs[k])))
    a = _lib.GsrAdam(n, groups[0], groups[1], groups[2], groups[3], groups[4], float(beta1), float(beta2), float(epsilon), int(iteration))
    with _host.on_device(dev):
        if rows is None:
            _lib.check(L.gsr_adam_update(C.byref(a), _host.stream_ptr(dev)))
        else:
            ptrs = (C.c_void_p * len(rows))(*[r.data_ptr() for r in rows])
            scale = float(sh_scale) if sh_scale is not None else 1.0 / len(rows)
            _lib.check(L.gsr_adam_update_views(C.byref(a), int(sh_degree), len(rows), ptrs, scale, _host.stream_ptr(dev)))
    # the kernel wrote through raw pointers: let torch's version counters say so (a backward() that is handed these tensors with a
    # forward's buffers from BEFORE this step must not reuse what that forward derived from the old values)
    _host.written_in_place(*[d[k] for k in GROUPS for d in (params, m, v)])
