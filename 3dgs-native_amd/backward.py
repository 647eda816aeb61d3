"""
backward(): the reference's gradient call surface (reference backward.py:955-1196) over the MI355X
library.  Same keyword arguments and the same nine-key return dict (`dL_dcov3D` is all zeros, as in
the reference, whose real dL/dSigma3D is a local that never leaves backward_preprocess, :812/:1119).
The five optimizer gradients are views into one flat float32 arena (`_arena`, 59 floats per
Gaussian: mean3D | scale | rot | opacity | shs, each segment padded to a multiple of 4 floats, dist.arena_offsets) so
data-parallel training reduces them with a single RCCL all-reduce.

One keyword beyond the reference's: `sh_gradient="factored"` (view-parallel training, dist.py) leaves the 48-float SH
gradient unwritten and returns instead the 3-float colour gradient it is an outer product of, as a self-contained view
payload (`_view_payload`: N rows, then the camera position); `_arena` is then the 11 floats mean3D | scale | rot | opacity
and `dL_dshs` is None until `dist.sh_gradients_from_views` rebuilds it from all views' payloads ("both" returns the dense
arena and the payload of the same call).  `on_payload(payload)` is called as soon as the payload is complete -- after the
blend half, before the per-Gaussian half -- so an exchange can start early (dist.FactoredExchange).
"""
import ctypes as C

import torch

from . import _host, _lib


_ZERO = {}
import os
_NO_GRAD_VIEWS = bool(int(os.environ.get("GSR_NO_GRAD_VIEWS", "0")))   # A/B switch: packed dL_dcolor / dL_dmean2D / dL_dconic arrays


def _zeros_cov3d(n, dev):
    """The reference returns an all-zero dL_dcov3D (backward.py:1119 is allocated, never filled).  Here it is a real, dense
    (N, 6) zero tensor (`.view(-1)`, `.numpy()`, strides as the reference's array) that is cleared ONCE per (N, device) and
    then SHARED by every later call with that N: it is meant to be read.  A caller that writes into it changes what later
    calls return; `dL_dcov3D.clone()` gives a private copy."""
    z = _ZERO.get(dev.index)               # one entry per device, replaced when that device's N changes (a trainer's N changes
    if z is None or z.shape[0] != n:       # only at densification; a process that alternates two GPUs keeps both)
        z = _ZERO[dev.index] = torch.zeros((n, 6), dtype=torch.float32, device=dev)
    return z


def _get(buf, key):
    if buf is None:
        raise NameError(f"backward() needs the forward buffer holding '{key}' (the reference fails the same way, "
                        "backward.py:1084-1090)")
    return buf.get(key)


def backward(background, means3D, dL_dpixels, opacity=None, shs=None, scales=None, rotations=None, scale_modifier=1.0,
             viewmatrix=None, projmatrix=None, tan_fovx=0.5, tan_fovy=0.5, image_height=256, image_width=256, campos=None,
             radii=None, means2D=None, conic_opacity=None, rgb=None, clamped=None, cov3Ds=None, geom_buffer=None,
             binning_buffer=None, img_buffer=None, degree=3, debug=False, *, sh_gradient="dense", on_payload=None):
    if sh_gradient not in ("dense", "factored", "both"):
        raise ValueError("sh_gradient must be 'dense', 'factored' or 'both'")
    factored = sh_gradient == "factored"
    from . import forward as _forward
    _forward._backward_seen = True     # from now on this process's forwards pre-clear the backward workspace (forward.PRECLEAR_BACKWARD)
    L = _lib.lib()
    dev = _host.device_of(means3D, dL_dpixels, shs, radii)
    H, W = int(image_height), int(image_width)
    f32, i32 = torch.float32, torch.int32
    means = _host.to_dev(means3D, f32, dev, (-1, 3))
    N = means.shape[0]
    dpix = _host.to_dev(dL_dpixels, f32, dev, (H, W, 3))
    sh = _host.to_dev(shs, f32, dev, (-1, 3))
    sc = _host.to_dev(scales, f32, dev, (-1, 3))
    rot = _host.to_dev(rotations, f32, dev, (-1, 4))
    op = _host.to_dev(opacity, f32, dev, (-1,)) if opacity is not None else means.new_zeros((N,))  # unused (quirk Q7)
    ranges = _get(img_buffer, "ranges")                       # reference backward.py:1084-1090
    final_Ts = _get(img_buffer, "final_Ts")
    n_contrib = _get(img_buffer, "n_contrib")
    point_list = _get(binning_buffer, "point_list")
    if geom_buffer is not None:                               # reference backward.py:1092-1103
        radii = geom_buffer.get("radii") if radii is None else radii
        means2D = geom_buffer.get("means2D") if means2D is None else means2D
        conic_opacity = geom_buffer.get("conic_opacity") if conic_opacity is None else conic_opacity
        rgb = geom_buffer.get("rgb") if rgb is None else rgb
        clamped = geom_buffer.get("clamped_state") if clamped is None else clamped
    # The forward's blend records (forward.py: means2D / conic_opacity / rgb are columns of one (N, 16) tensor) stand in for the
    # three arrays when the caller hands back those very views, unwritten since: no re-pack, no packed copies of the views.
    records = None
    rec_tag = getattr(means2D, "_gsr_records", None)     # set by render_gaussians on its points_xy_image view
    if rec_tag is not None:
        r_t, r_ver, src = rec_tag
        given3 = {"means2D": means2D, "conic_opacity": conic_opacity, "rgb": rgb}
        if all(src[k]() is given3[k] for k in given3) and r_t._version == r_ver and r_t.device == dev and r_t.shape[0] == N:
            records = r_t
    backward.last_call_used_forward_records = records is not None   # for tests and debugging
    # The forward's per-entry block masks ride on its point_list tensor.  They are conservative only for THAT forward's records
    # and written only up to each tile's saturation batch, so they are honoured only when every buffer they were derived from
    # or are read against is the forward's own tensor (identity, not equality): a caller who mixes in perturbed means2D /
    # conic_opacity or another run's ranges / n_contrib gets the self-contained block test instead.
    masks, order = None, None
    cleared_tag = getattr(point_list, "_gsr_cleared_ws", None)     # "the forward cleared the backward workspace's accumulators"
    mask_tag = getattr(point_list, "_gsr_block_masks", None)
    if mask_tag is not None:
        m_t, owners, o_t = mask_tag
        given = {"ranges": ranges, "n_contrib": n_contrib, "final_Ts": final_Ts, "means2D": means2D, "conic_opacity": conic_opacity}
        if all(owners[k][0]() is given[k] and _host.version_of(given[k]) == owners[k][1] for k in given):
            masks, order = m_t, o_t      # (the block order rides with the masks: it was derived from them)
    # likewise the forward's d(colour)/d(direction) sums (GsrGeom.sh_dir_grad) ride on its clamped_state tensor: used when shs
    # and means3D are the very tensors that forward read, with the same camera position and degree -- geom_backward_kernel then
    # reads 36 bytes per Gaussian instead of the 192 bytes of coefficients
    cam = _host.make_camera(viewmatrix, projmatrix, campos, background, tan_fovx, tan_fovy, W, H)
    sh_dir = None
    dir_tag = getattr(clamped, "_gsr_sh_dir", None)
    if dir_tag is not None:
        d_t, sh_ref, means_ref, campos_f, deg_f, sh_ver, means_ver = dir_tag
        if (sh_ref() is shs and means_ref() is means3D and _host.version_of(shs) == sh_ver and _host.version_of(means3D) == means_ver
                and deg_f == int(degree) and d_t.device == dev
                and campos_f == tuple(cam.campos)):
            sh_dir = d_t
    backward.last_call_used_forward_sh_dir = sh_dir is not None     # for tests and debugging
    radii = _host.to_dev(radii, i32, dev, (-1,))
    if records is None:      # (the reference re-reads the three arrays; so does this path -- packed copies if they are strided views)
        m2d = _host.to_dev(means2D, f32, dev, (-1, 2))
        con = _host.to_dev(conic_opacity, f32, dev, (-1, 4))
        col = _host.to_dev(rgb, f32, dev, (-1, 3))
    else:
        m2d = con = col = None
    cl = _host.to_dev(clamped, f32, dev, (-1, 3))
    # Sigma3D is recomputed inside the kernel instead of read back (24 bytes per Gaussian) when `cov3Ds` is the forward's own tensor,
    # unwritten, made from these very scales / rotations (unwritten too) with this scale_modifier (forward.py; gsr.h GsrGeom.cov3D)
    c3 = None
    sig_tag = getattr(cov3Ds, "_gsr_sigma_of", None)
    if sig_tag is not None:
        sc_ref, rot_ref, sc_ver, rot_ver, smod, c_ver = sig_tag
        recompute = (sc_ref() is scales and rot_ref() is rotations and _host.version_of(scales) == sc_ver and _host.version_of(rotations) == rot_ver
                     and smod == float(scale_modifier) and cov3Ds._version == c_ver and cov3Ds.device == dev and cov3Ds.shape[0] == N)
    else:
        recompute = False
    backward.last_call_recomputed_sigma3d = recompute     # for tests and debugging
    if not recompute:
        c3 = _host.to_dev(cov3Ds, f32, dev, (-1, 6))
    ranges = _host.to_dev(ranges, i32, dev, (-1, 2))
    final_Ts = _host.to_dev(final_Ts, f32, dev, (H, W))
    n_contrib = _host.to_dev(n_contrib, i32, dev, (H, W))
    point_list = _host.to_dev(point_list, i32, dev, (-1,))
    D = point_list.shape[0]

    scene = _lib.GsrScene(N, _host.ptr(means), _host.ptr(sc), _host.ptr(rot), _host.ptr(op), _host.ptr(sh), int(degree),
                          float(scale_modifier), 1)
    geom = _lib.GsrGeom(_host.ptr(radii), None, None, _host.ptr(m2d), None, _host.ptr(c3), _host.ptr(col), _host.ptr(con),
                        _host.ptr(cl), _host.ptr(records), _host.ptr(sh_dir))
    if masks is not None and not (isinstance(masks, torch.Tensor) and masks.dtype == torch.uint8 and masks.device == dev
                                  and masks.numel() == D and masks.is_contiguous()):
        masks = order = None
    backward.last_call_used_forward_masks = masks is not None      # for tests and debugging
    img = _lib.GsrImage(None, None, _host.ptr(final_Ts), _host.ptr(n_contrib))

    from . import dist as _dist
    o = _dist.arena_offsets(N, small=factored)     # every segment starts on a multiple of 4 floats (16-byte vector stores)
    payload = None
    arena = torch.empty(o[-1], dtype=f32, device=dev)
    if N % 4:                                      # the <= 3 padding floats behind a segment: defined (zero), never garbage
        for k, sz in enumerate([3 * N, 3 * N, 4 * N, N][:len(o) - 2]):
            arena[o[k] + sz:o[k + 1]].zero_()
    if factored:
        payload = torch.empty(N * 3 + 4, dtype=f32, device=dev)   # its own allocation: aligned for the collective
        dL_dsh = None
    else:
        dL_dsh = arena[o[4]:o[4] + 48 * N].view(N * 16, 3)   # always N*16 rows: the reference under-allocates for degree < 3 (quirk Q6)
        if sh_gradient == "both":                   # dense gradient AND the payload it factors into (tests, debugging)
            payload = torch.empty(N * 3 + 4, dtype=f32, device=dev)
    dL_dmean3D = arena[o[0]:o[0] + 3 * N].view(N, 3)
    dL_dscale = arena[o[1]:o[1] + 3 * N].view(N, 3)
    dL_drot = arena[o[2]:o[2] + 4 * N].view(N, 4)
    dL_dopacity = arena[o[3]:o[3] + N]
    stream = _host.raw_stream(dev)
    with _host.on_device(dev):
        # The workspace belongs to THIS call (the returned blend-stage gradients are views of it).  If the forward made one and
        # cleared its accumulators in its blend kernel -- and no backward has taken it yet -- it is that one; else a fresh one,
        # which the library clears itself.
        need = int(L.gsr_backward_workspace_bytes(N, D, W, H))
        cleared = (cleared_tag is not None and cleared_tag[0] is not None and cleared_tag[1] == N and cleared_tag[0].device == dev
                   and cleared_tag[0].numel() >= need)
        if cleared:
            ws, cleared_tag[0] = cleared_tag[0], None   # taken: a second backward() on the same forward gets a fresh one
        else:
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
        # dL_dcolor / dL_dmean2D / dL_dconic are columns 0-2 / 3-5 / 6-9 of the accumulator records the blend backward sums into
        # (gsr.h GsrGrads): strided views, so the per-Gaussian kernel does not write 40 bytes per Gaussian of copies
        off = int(L.gsr_backward_accumulators_offset(N))
        acc = ws[off:off + 64 * N].view(f32).view(N, 16)
        dL_dcolor, dL_dmean2D, dL_dconic = acc[:, 0:3], acc[:, 3:6], acc[:, 6:10]
        packed = (lambda t: None)
        if _NO_GRAD_VIEWS:
            dL_dcolor, dL_dmean2D, dL_dconic = (torch.empty((N, k), dtype=f32, device=dev) for k in (3, 3, 4))
            packed = _host.ptr
        grads = _lib.GsrGrads(_host.ptr(dL_dmean3D), _host.ptr(dL_dscale), _host.ptr(dL_drot), _host.ptr(dL_dopacity),
                              _host.ptr(dL_dsh), packed(dL_dcolor), packed(dL_dmean2D), packed(dL_dconic), _host.ptr(payload))
        binning = _lib.GsrBinning(D, _host.ptr(point_list), _host.ptr(ranges), _host.ptr(masks), _host.ptr(order),
                                  _host.ptr(ws) if cleared else None, 1 if cleared else 0)
        backward.last_call_skipped_the_clear = cleared     # for tests and debugging
        if on_payload is not None and payload is not None:
            # two halves: the view payload is complete after the blend half, so the caller's hook can start its exchange
            # (an asynchronous all-gather) while the per-Gaussian half still runs
            _lib.check(L.gsr_backward_blend(C.byref(scene), C.byref(cam), C.byref(geom), C.byref(binning), C.byref(img), _host.ptr(dpix),
                                            _host.ptr(payload), _host.ptr(ws), ws.numel(), stream))
            on_payload(payload)
            grads.dL_drgb = None
            _lib.check(L.gsr_backward_geom(C.byref(scene), C.byref(cam), C.byref(geom), C.byref(grads), _host.ptr(ws), ws.numel(), stream))
        else:
            _lib.check(L.gsr_backward(C.byref(scene), C.byref(cam), C.byref(geom), C.byref(binning), C.byref(img), _host.ptr(dpix),
                                      C.byref(grads), _host.ptr(ws), ws.numel(), stream))
    return {
        "dL_dmean3D": dL_dmean3D, "dL_dcolor": dL_dcolor, "dL_dshs": dL_dsh, "dL_dopacity": dL_dopacity,
        "dL_dscale": dL_dscale, "dL_drot": dL_drot, "dL_dmean2D": dL_dmean2D, "dL_dconic": dL_dconic,
        "dL_dcov3D": _zeros_cov3d(N, dev),
        "_arena": arena,
        "_view_payload": payload,
    }
