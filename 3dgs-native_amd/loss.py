"""
Loss step between forward and backward: the reference's `l1_loss` and
`compute_image_gradients` (reference loss.py:148-176, :217-244) on the GPU, in one pass over the image.
As in the reference, `lambda_dssim` only scales the L1 gradient; the SSIM gradient is a TODO there
(loss.py:243) and is not invented here.
"""
import torch

from . import _host, _lib


def l1_loss_and_gradients(rendered, target, lambda_dssim=0.0, want_grad=True, loss_out=None):
    """One kernel: returns (loss_sum device tensor [1] = sum |rendered - target|, pixel_grad (H,W,3) or None).
    mean L1 = loss_sum / (H*W*3); pixel_grad = (1-lambda_dssim)/(H*W*3) * sign(rendered - target).
    `loss_out`: a 1-element float32 device tensor (e.g. a slot of a trainer's loss curve) to receive the sum instead of a fresh one."""
    L = _lib.lib()
    dev = _host.device_of(rendered, target)
    r = _host.to_dev(rendered, torch.float32, dev)
    H, W = int(r.shape[0]), int(r.shape[1])
    r = r.reshape(H, W, 3)
    t = _host.to_dev(target, torch.float32, dev, (H, W, 3))   # alpha already dropped by the caller (train.py:323-334)
    grad = torch.empty((H, W, 3), dtype=torch.float32, device=dev) if want_grad else None
    if loss_out is not None and not (isinstance(loss_out, torch.Tensor) and loss_out.is_cuda and loss_out.dtype == torch.float32 and loss_out.numel() == 1):
        raise ValueError("l1_loss_and_gradients: loss_out must be a 1-element float32 device tensor")
    loss_sum = torch.empty(1, dtype=torch.float32, device=dev) if loss_out is None else loss_out
    l1_weight = (1.0 - float(lambda_dssim)) / (H * W * 3.0)
    with _host.on_device(dev):
        _lib.check(L.gsr_l1_loss_grad(_host.ptr(r), _host.ptr(t), _host.ptr(grad), _host.ptr(loss_sum), W, H, l1_weight,
                                      _host.stream_ptr(dev)))
    return loss_sum, grad


def l1_loss(rendered, target):
    """Mean absolute error as a Python float (reference loss.py:148-176; synchronises, as the reference does)."""
    r = rendered
    s, _ = l1_loss_and_gradients(rendered, target, want_grad=False)
    return float(s.item()) / (int(r.shape[0]) * int(r.shape[1]) * 3)


def compute_image_gradients(rendered, target, lambda_dssim=0.2):
    """dL/dpixels for backward() (reference loss.py:217-244)."""
    return l1_loss_and_gradients(rendered, target, lambda_dssim)[1]


def ssim(rendered, target):
    """Mean SSIM as a Python float (reference loss.py:178-215: 11x11 window, sigma 1.5, weights indexed by distance as the
    reference does -- see gsr.h).  An evaluation helper: the reference's training loop has its SSIM term commented out."""
    L = _lib.lib()
    dev = _host.device_of(rendered, target)
    r = _host.to_dev(rendered, torch.float32, dev)
    H, W = int(r.shape[0]), int(r.shape[1])
    r = r.reshape(H, W, 3)
    t = _host.to_dev(target, torch.float32, dev, (H, W, 3))
    acc = torch.empty(1, dtype=torch.float32, device=dev)
    with _host.on_device(dev):
        _lib.check(L.gsr_ssim(_host.ptr(r), _host.ptr(t), _host.ptr(acc), W, H, _host.stream_ptr(dev)))
    return float(acc.item()) / (W * H)


def depth_loss(rendered_depth, target_depth, depth_mask):
    """Masked mean L1 between inverse-depth images as a Python float (reference loss.py:271-303)."""
    L = _lib.lib()
    dev = _host.device_of(rendered_depth, target_depth, depth_mask)
    r = _host.to_dev(rendered_depth, torch.float32, dev)
    H, W = int(r.shape[0]), int(r.shape[1])
    r = r.reshape(H, W)
    t = _host.to_dev(target_depth, torch.float32, dev, (H, W))
    m = _host.to_dev(depth_mask, torch.float32, dev, (H, W))
    acc = torch.empty(1, dtype=torch.float32, device=dev)
    with _host.on_device(dev):
        _lib.check(L.gsr_depth_loss(_host.ptr(r), _host.ptr(t), _host.ptr(m), _host.ptr(acc), W, H, _host.stream_ptr(dev)))
    return float(acc.item()) / (W * H)
