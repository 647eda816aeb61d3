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


def adam_update(params, grads, m, v, lrs=None, beta1=0.9, beta2=0.999, epsilon=1e-8, iteration=0):
    """In-place Adam step over the five parameter groups.  `iteration` is 0-based (bias correction uses +1)."""
    L = _lib.lib()
    lrs = DEFAULT_LR if lrs is None else lrs
    dev = params["positions"].device
    n = int(params["positions"].shape[0])
    expect = {"positions": 3 * n, "scales": 3 * n, "rotations": 4 * n, "opacities": n, "shs": 48 * n}
    groups = []
    for k in GROUPS:
        for d in (params, grads, m, v):
            t = d[k]
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == expect[k]):
                raise ValueError(f"adam_update: '{k}' must be a contiguous float32 device tensor with {expect[k]} elements")
        groups.append(_lib.GsrAdamGroup(_host.ptr(params[k]), _host.ptr(grads[k]), _host.ptr(m[k]), _host.ptr(v[k]), float(lrs[k])))
    a = _lib.GsrAdam(n, groups[0], groups[1], groups[2], groups[3], groups[4], float(beta1), float(beta2), float(epsilon), int(iteration))
    with torch.cuda.device(dev):
        _lib.check(L.gsr_adam_update(C.byref(a), _host.stream_ptr(dev)))
