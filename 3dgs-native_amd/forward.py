"""
render_gaussians(): the reference's forward call surface (reference forward.py:629-894) over the
MI355X library.  Same keyword arguments, same (image, depth, dict) return with the same dict keys;
arrays come back as torch tensors on the GPU instead of wp.array (callers do `.cpu().numpy()`).
"""
import ctypes as C
import os
import weakref

import torch

from . import _host, _lib
from .config import TILE_M, TILE_N

# Training use (the default): the forward blend kernel's spare workgroups clear the accumulators of the backward workspace while
# that kernel drains, so the backward() that follows does not start with a 64-byte-per-Gaussian clear of its own.  A render-only
# user can switch it off (it allocates the backward's scratch): forward.PRECLEAR_BACKWARD = False.
# It is lazy: nothing is allocated or cleared for the backward until this process has called backward() once (a render-only
# process never pays for it); the first training step's backward clears for itself.
PRECLEAR_BACKWARD = not bool(int(os.environ.get("GSR_NO_PRECLEAR", "0")))
_backward_seen = False          # set by backward.backward()
_NO_SH_DIR = bool(int(os.environ.get("GSR_NO_SH_DIR", "0")))
_NO_RECORD_VIEWS = bool(int(os.environ.get("GSR_NO_RECORD_VIEWS", "0")))
_NO_COV_RECOMPUTE = bool(int(os.environ.get("GSR_NO_COV_RECOMPUTE", "0")))   # A/B switch: backward() always reads cov3Ds back
_NO_BLOCK_ORDER = bool(int(os.environ.get("GSR_NO_BLOCK_ORDER", "0")))     # A/B switch: the forward does not file the backward's blocks by cost   # A/B switch: packed xy / conic_opacity / colors arrays beside the records   # A/B switch: backward reads the SH rows itself (same results)


def render_gaussians(background, means3D, colors=None, opacity=None, scales=None, rotations=None, scale_modifier=1.0,
                     viewmatrix=None, projmatrix=None, tan_fovx=0.5, tan_fovy=0.5, image_height=256, image_width=256,
                     sh=None, degree=3, campos=None, prefiltered=False, antialiasing=False, clamped=True, debug=False):
    """Render 3D Gaussians.  `colors`, `prefiltered`, `antialiasing` are accepted and ignored exactly as in
    the reference.  Returns (image (H,W,3) f32, inverse-depth (H,W) f32, buffers)."""
    L = _lib.lib()
    dev = _host.device_of(means3D, sh, opacity, scales, rotations)
    H, W = int(image_height), int(image_width)
    f32, i32 = torch.float32, torch.int32
    means = _host.to_dev(means3D, f32, dev, (-1, 3))
    N = means.shape[0]
    shs = _host.to_dev(sh, f32, dev, (-1, 3))                 # reference forward.py:687
    if shs.shape[0] != N * 16:
        raise ValueError(f"sh must hold 16 coefficients per Gaussian (got {shs.shape[0]} rows for N={N})")
    op = _host.to_dev(opacity, f32, dev, (-1,))               # (N,1) -> (N,)  utils/wp_utils.py:42-43
    sc = _host.to_dev(scales, f32, dev, (-1, 3))
    rot = _host.to_dev(rotations, f32, dev, (-1, 4))
    if not (op.shape[0] == sc.shape[0] == rot.shape[0] == N):
        raise ValueError("means3D, opacity, scales and rotations disagree on the number of Gaussians")
    cam = _host.make_camera(viewmatrix, projmatrix, campos, background, tan_fovx, tan_fovy, W, H)
    gx, gy = (W + TILE_M - 1) // TILE_M, (H + TILE_N - 1) // TILE_N

    scene = _lib.GsrScene(N, _host.ptr(means), _host.ptr(sc), _host.ptr(rot), _host.ptr(op), _host.ptr(shs),
                          int(degree), float(scale_modifier), 1 if clamped else 0)
    e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
    radii, tiles_touched, point_offsets = e((N,), i32), e((N,), i32), e((N,), i32)
    # points_xy_image, conic_opacity and colors are COLUMNS of this call's blend records (one 64-byte row per Gaussian: x, y, conic
    # a b c, opacity, r g b, 1/depth -- GsrGeom.blend_records): strided views, so the forward writes those 36 bytes per Gaussian once
    # instead of twice.  `.cpu().numpy()`, indexing and arithmetic work on them as on any tensor; `.contiguous()` gives a packed copy.
    records = e((N, 16), f32)
    xy, conic_opacity, rgb = records[:, 0:2], records[:, 2:6], records[:, 6:9]
    if _NO_RECORD_VIEWS:
        xy, conic_opacity, rgb = e((N, 2), f32), e((N, 4), f32), e((N, 3), f32)
    depths, cov3Ds, clamped_state = e((N,), f32), e((N, 6), f32), e((N, 3), f32)
    # d(colour)/d(direction), nine floats per Gaussian: what the SH backward needs of the 48 coefficients (GsrGeom.sh_dir_grad).
    # Only worth its 36 bytes per Gaussian when the caller's SH / position tensors can be recognised again by backward(), i.e.
    # when they are device tensors used in place.
    in_place = lambda given, used: isinstance(given, torch.Tensor) and given.is_cuda and given.data_ptr() == used.data_ptr()
    sh_dir = e((N, 9), f32) if (N > 0 and in_place(sh, shs) and in_place(means3D, means) and not _NO_SH_DIR) else None
    arr = (lambda t: _host.ptr(t)) if _NO_RECORD_VIEWS else (lambda t: None)
    geom = _lib.GsrGeom(_host.ptr(radii), _host.ptr(tiles_touched), _host.ptr(point_offsets), arr(xy), _host.ptr(depths),
                        _host.ptr(cov3Ds), arr(rgb), arr(conic_opacity), _host.ptr(clamped_state), _host.ptr(records), _host.ptr(sh_dir))
    image, depth_image = e((H, W, 3), f32), e((H, W), f32)
    final_Ts, n_contrib = e((H, W), f32), e((H, W), i32)
    img = _lib.GsrImage(_host.ptr(image), _host.ptr(depth_image), _host.ptr(final_Ts), _host.ptr(n_contrib))
    ranges = e((gx * gy, 2), i32)
    stream = _host.raw_stream(dev)

    with _host.on_device(dev):
        gws = _host.workspace("geom", L.gsr_geom_workspace_bytes(N), dev, stream)
        D = C.c_int64(0)
        _lib.check(L.gsr_forward_count(C.byref(scene), C.byref(cam), C.byref(geom), _host.ptr(gws), gws.numel(),
                                       C.byref(D), stream))
        D = D.value
        if debug:
            print(f"gsr: {W}x{H}, N={N}, D={D}, SH degree {degree}")
        point_list = e((D,), i32)
        # per-entry 8x4-block hit masks: written by the forward blend, read by backward() (allocated with 16 spare bytes: the
        # backward reads them 16 at a time)
        block_masks = e((D + 16,), torch.uint8)[:D]
        # the backward blend's blocks filed by cost, heaviest first (GsrBinning.block_order): filled by the forward blend from the masks
        block_order = None if _NO_BLOCK_ORDER else e((int(L.gsr_block_order_ints(W, H)),), i32)
        # The backward's workspace, one per call: its accumulator records are what backward() returns dL_dcolor / dL_dmean2D /
        # dL_dconic as views of, so it must not be shared between calls.  Handed to the forward, its records are cleared by the
        # blend kernel's spare workgroups.
        bwd_ws = None
        if PRECLEAR_BACKWARD and _backward_seen and N > 0 and D > 0:
            bwd_ws = e((int(L.gsr_backward_workspace_bytes(N, D, W, H)),), torch.uint8)
        binning = _lib.GsrBinning(D, _host.ptr(point_list), _host.ptr(ranges), _host.ptr(block_masks), _host.ptr(block_order),
                                  _host.ptr(bwd_ws), 0)
        bws = _host.workspace("bin", L.gsr_binning_workspace_bytes(N, D, W, H), dev, stream)
        _lib.check(L.gsr_forward_render(C.byref(scene), C.byref(cam), C.byref(geom), C.byref(binning), C.byref(img),
                                        _host.ptr(gws), gws.numel(), _host.ptr(bws), bws.numel(), stream))
    # Let a following backward() use the records as they are: the tag rides on the means2D view (the reference's callers re-pack
    # the dicts by hand, train.py:986-1000) and is honoured when backward() is handed these very three views, unwritten since
    # (views share their base's version counter, so a write through any of them, or into the records, is seen).
    if N > 0:
        xy._gsr_records = (records, records._version, {"means2D": weakref.ref(xy), "conic_opacity": weakref.ref(conic_opacity), "rgb": weakref.ref(rgb)})
        # likewise the block masks ride on the point_list tensor (their own allocation, alive as long as it is): a caller
        # that hands backward() this very tensor gets the mask-driven compaction, anyone else the self-contained one
        # -- and only together with the other buffers of this call (backward() checks identity): the masks describe these
        # records, up to these n_contrib
        if sh_dir is not None:
            # the direction derivatives ride on clamped_state (which backward() receives as `clamped`), valid for these very
            # sh / means3D tensors, this camera position and this degree
            # -- and only while neither has been written in place since (torch's version counters; the library's own in-place
            # writers, Adam and the opacity reset, bump them too: _host.written_in_place)
            clamped_state._gsr_sh_dir = (sh_dir, weakref.ref(sh), weakref.ref(means3D), tuple(cam.campos), int(degree),
                                         sh._version, means3D._version)
        if in_place(scales, sc) and in_place(rotations, rot) and not _NO_COV_RECOMPUTE:
            # backward() need not read cov3Ds back when it is handed this very tensor, unwritten, with these very scales / rotations
            # (unwritten too) and the same scale_modifier: the kernel recomputes Sigma3D with the forward's instructions (gsr.h GsrGeom.cov3D)
            cov3Ds._gsr_sigma_of = (weakref.ref(scales), weakref.ref(rotations), scales._version, rotations._version, float(scale_modifier), cov3Ds._version)
        owners = {"ranges": ranges, "n_contrib": n_contrib, "final_Ts": final_Ts, "means2D": xy, "conic_opacity": conic_opacity}
        point_list._gsr_block_masks = (block_masks, {k: (weakref.ref(v), v._version) for k, v in owners.items()}, block_order)
        if bwd_ws is not None:      # "a backward workspace with clean accumulators": the first backward() handed this point_list takes it
            point_list._gsr_cleared_ws = [bwd_ws, N]
    return image, depth_image, {
        "radii": radii, "point_offsets": point_offsets, "points_xy_image": xy, "depths": depths, "colors": rgb,
        "cov3Ds": cov3Ds, "conic_opacity": conic_opacity, "point_list": point_list, "ranges": ranges,
        "final_Ts": final_Ts, "n_contrib": n_contrib, "clamped_state": clamped_state,
    }
