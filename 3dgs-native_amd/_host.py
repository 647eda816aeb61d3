"""Host-side plumbing shared by forward.py and backward.py: tensor conversion, camera struct packing,
workspace caching.  PyTorch is only the owner of device memory and streams here."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def device_of(*xs):
    for x in xs:
        if isinstance(x, torch.Tensor) and x.is_cuda:
            return x.device
    if not torch.cuda.is_available():
        raise RuntimeError("3dgs-native_amd needs a ROCm GPU (MI355X); no CPU path exists")
    return torch.device("cuda", torch.cuda.current_device())


def to_dev(x, dtype, dev, shape=None):
    """numpy / torch / sequence -> contiguous device tensor (the reference's to_warp_array,
    utils/wp_utils.py:34-44, minus the forced re-upload when the data is already resident)."""
    if isinstance(x, torch.Tensor):
        # the common case on the trainer's path -- a resident, packed tensor of the right type -- costs no torch call at all
        # (twenty conversions per iteration were 60 us of `.to()` that changed nothing)
        if x.dtype == dtype and x.is_cuda and x.device == dev and x.is_contiguous() and x.data_ptr() % 16 == 0:
            return x if shape is None else x.view(shape)
        t = x.to(device=dev, dtype=dtype)
    else:
        t = torch.as_tensor(np.ascontiguousarray(np.asarray(x)), device="cpu").to(dtype).to(dev)
    if shape is not None:
        t = t.reshape(shape)
    t = t.contiguous()
    if t.data_ptr() % 16:
        # include/gsr.h asks for 16-byte aligned arrays (GSR_E_ALIGN); an offset view such as means[1:] is contiguous but
        # not aligned, and the reference (which copies every input, wp_utils.py:34-44) accepts it: copy it too
        t = t.clone()
    return t


def host_f32(x, n):
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    a = np.asarray(x, dtype=np.float64).reshape(-1)[:n]
    return a.astype(np.float32)


_cams = {}


def make_camera(viewmatrix, projmatrix, campos, background, tan_fovx, tan_fovy, W, H):
    """The GsrCamera of a call.  A trainer comes back to its views: the packed struct is kept under the BYTES of its inputs (numpy
    arrays have no version counter, so identity would not do), which costs a quarter of packing it again."""
    key = None
    if type(viewmatrix) is np.ndarray and type(projmatrix) is np.ndarray and type(campos) is np.ndarray and type(background) is np.ndarray:
        key = (viewmatrix.tobytes(), projmatrix.tobytes(), campos.tobytes(), background.tobytes(), viewmatrix.dtype.num, projmatrix.dtype.num,
               campos.dtype.num, background.dtype.num, float(tan_fovx), float(tan_fovy), int(W), int(H))
        cam = _cams.get(key)
        if cam is not None:
            return cam
    cam = _pack_camera(viewmatrix, projmatrix, campos, background, tan_fovx, tan_fovy, W, H)
    if key is not None:
        if len(_cams) >= 4096:
            _cams.clear()
        _cams[key] = cam
    return cam


def _pack_camera(viewmatrix, projmatrix, campos, background, tan_fovx, tan_fovy, W, H):
    cam = _lib.GsrCamera()
    cam.view[:] = host_f32(viewmatrix, 16).tolist()    # float64 -> float32 once (reference forward.py:694-695)
    cam.proj[:] = host_f32(projmatrix, 16).tolist()
    cam.campos[:] = host_f32(campos, 3).tolist()
    cam.bg[:] = host_f32(background, 3).tolist()
    cam.tan_fovx, cam.tan_fovy = float(tan_fovx), float(tan_fovy)
    cam.focal_x = W / (2.0 * float(tan_fovx))          # float64, rounded by the c_float store (quirk Q8)
    cam.focal_y = H / (2.0 * float(tan_fovy))
    cam.W, cam.H = int(W), int(H)
    return cam


_ws = {}


def workspace(kind, nbytes, dev, stream=None):
    """Grow-only scratch buffer per (kind, device, stream): calls issued on different streams (several views in flight on
    one GPU) never share scratch.  `stream`: raw_stream(dev), when the caller has it already."""
    key = (kind, dev.index, raw_stream(dev) if stream is None else stream)
    t = _ws.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=dev)
        _ws[key] = t
    return t


def ptr(t):
    """Device address for a `void *` struct field or argument (ctypes takes an int or None there): NULL for None or an empty tensor."""
    return t.data_ptr() if t is not None and t.numel() > 0 else None


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream    # what torch.cuda.current_stream(dev).cuda_stream returns, without the Stream object
except AttributeError:                                  # pragma: no cover
    _raw_stream = None


def raw_stream(dev):
    """torch's current stream on `dev` as the integer a hipStream_t argument takes."""
    if _raw_stream is not None:
        return _raw_stream(dev.index)
    return torch.cuda.current_stream(dev).cuda_stream


def stream_ptr(dev):
    return raw_stream(dev)


class on_device:
    """`with torch.cuda.device(dev)` for the common case that `dev` is current already (then it costs one call, not a context switch)."""
    __slots__ = ("ctx",)

    def __init__(self, dev):
        self.ctx = None if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


def version_of(t):
    """torch's in-place write counter of a tensor (None for anything else): tags that let backward() reuse state the forward
    derived from an array record it, so a caller's in-place write between the two calls is seen."""
    return t._version if isinstance(t, torch.Tensor) else None


def written_in_place(*tensors):
    """Tell torch that the library has written into these tensors through raw pointers (Adam, opacity reset ...): bumps their
    version counters exactly as a torch in-place op would, so stale forward state tagged with the old versions is dropped."""
    for t in tensors:
        if isinstance(t, torch.Tensor):
            torch.autograd.graph.increment_version(t)
