"""
oracle.py -- TEST INFRASTRUCTURE ONLY.

numpy/ctypes driver for libgsr_oracle.so (gsr_oracle.c).  `render_gaussians` and `backward` restate
the orchestration of the reference (forward.py:629-894, backward.py:955-1196) on host numpy arrays,
calling one C function per reference kernel in the reference's launch order.  The product path
(3dgs-native_amd/) never imports this module.

Parity pinning: forward pinned by the reference's assets/example_render.png
(tests/test_oracle_golden.py); backward "parity unpinned" (the reference holds no backward output).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)


def build():
    """Compile libgsr_oracle.so with gcc (oracle/Makefile)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgsr_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        for name in ("gsro_preprocess", "gsro_prefix_sum", "gsro_duplicate_with_keys", "gsro_sort_pairs",
                     "gsro_identify_tile_ranges", "gsro_render_rows", "gsro_render_backward_rows", "gsro_render_backward_rows_acc64",
                     "gsro_cov2d_backward", "gsro_projection_backward", "gsro_sh_backward",
                     "gsro_cov3d_backward", "gsro_l1_pixel_grad", "gsro_adam_update"):
            getattr(_LIB, name).restype = None
        _LIB.gsro_l1_loss_sum.restype = C.c_float
        _LIB.gsro_ssim_sum.restype = C.c_float
        _LIB.gsro_depth_loss_sum.restype = C.c_float
    return _LIB


def _f(a):
    return a.ctypes.data_as(f32p)


def _i(a):
    return a.ctypes.data_as(i32p)


def _l(a):
    return a.ctypes.data_as(i64p)


def _f32(x, shape=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    return a.reshape(shape) if shape is not None else a


def _flat_opacity(op):
    op = np.asarray(op, dtype=np.float32)
    if op.ndim == 2 and op.shape[1] == 1:  # utils/wp_utils.py:42-43
        op = op.flatten()
    return np.ascontiguousarray(op)


TILE = 16


def _bands(rows, threads):
    """`rows` tile rows in at most `threads` contiguous bands."""
    t = max(1, min(int(threads), rows))
    cuts = [rows * k // t for k in range(t + 1)]
    return [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]


def _in_threads(jobs):
    """Run the ctypes calls (they release the GIL) of `jobs` concurrently, one thread each."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
        for f in [ex.submit(j) for j in jobs]:
            f.result()


def render_gaussians(background, means3D, colors=None, opacity=None, scales=None, rotations=None,
                     scale_modifier=1.0, viewmatrix=None, projmatrix=None, tan_fovx=0.5, tan_fovy=0.5,
                     image_height=256, image_width=256, sh=None, degree=3, campos=None,
                     prefiltered=False, antialiasing=False, clamped=True, debug=False,
                     tile_rows=None, keep_keys=False, threads=1):
    """Restates forward.py:629-894.  `tile_rows=(y0,y1)` blends only those tile rows (bench sample).  `threads` > 1 (bench.py's
    all-cores courtesy figure) blends bands of tile rows concurrently: pixels are independent, so the
    result is the same bit for bit, but it is not how the reference's CPU path runs (one serial loop)."""
    L = lib()
    H, W = int(image_height), int(image_width)
    means = _f32(means3D).reshape(-1, 3)
    N = means.shape[0]
    shs = _f32(sh).reshape(-1, 3)                                   # forward.py:687
    op = _flat_opacity(opacity)
    sc = _f32(scales).reshape(-1, 3)
    rot = _f32(rotations).reshape(-1, 4)
    view = _f32(np.asarray(viewmatrix, dtype=np.float64).flatten())  # forward.py:694 (f64 -> f32)
    proj = _f32(np.asarray(projmatrix, dtype=np.float64).flatten())
    cam = _f32(np.asarray(campos, dtype=np.float64)[:3])
    bg = _f32(np.asarray(background, dtype=np.float64)[:3])
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE

    image = np.zeros((H, W, 3), np.float32)
    depth_image = np.zeros((H, W), np.float32)
    final_Ts = np.zeros((H, W), np.float32)
    n_contrib = np.zeros((H, W), np.int32)
    radii = np.zeros(N, np.int32)
    xy = np.zeros((N, 2), np.float32)
    depths = np.zeros(N, np.float32)
    cov3Ds = np.zeros((N, 6), np.float32)
    rgb = np.zeros((N, 3), np.float32)
    conic_opacity = np.zeros((N, 4), np.float32)
    tiles_touched = np.zeros(N, np.int32)
    clamped_state = np.zeros((N, 3), np.float32)

    def pre(a, b):      # Gaussians [a, b): every array is per Gaussian, so a slice is a self-contained call
        L.gsro_preprocess(C.c_int(b - a), _f(means[a:b]), _f(sc[a:b]), C.c_float(scale_modifier), _f(rot[a:b]), _f(op[a:b]), _f(shs[16 * a:16 * b]),
                          C.c_int(int(degree)), C.c_int(1 if clamped else 0), _f(view), _f(proj), _f(cam),
                          C.c_int(W), C.c_int(H), C.c_float(tan_fovx), C.c_float(tan_fovy), _i(radii[a:b]), _f(xy[a:b]),
                          _f(depths[a:b]), _f(cov3Ds[a:b]), _f(rgb[a:b]), _f(conic_opacity[a:b]), _i(tiles_touched[a:b]), _f(clamped_state[a:b]))
    if threads > 1 and N > 0:
        _in_threads([(lambda a=a, b=b: pre(a, b)) for a, b in _bands(N, threads)])
    else:
        pre(0, N)
    point_offsets = np.zeros(N, np.int32)
    L.gsro_prefix_sum(C.c_int(N), _i(tiles_touched), _i(point_offsets))
    num_rendered = int(point_offsets[-1]) if N > 0 else 0            # forward.py:764 (N==0: Q10)
    if num_rendered > (1 << 30):
        raise ValueError("Number of rendered points exceeds the maximum supported by Warp.")
    keys = np.zeros(num_rendered, np.int64)
    point_list = np.zeros(num_rendered, np.int32)
    L.gsro_duplicate_with_keys(C.c_int(N), _f(xy), _f(depths), _i(point_offsets), _l(keys), _i(point_list),
                               _i(radii), C.c_int(W), C.c_int(H))
    L.gsro_sort_pairs(C.c_int64(num_rendered), _l(keys), _i(point_list))
    ranges = np.zeros((gx * gy, 2), np.int32)
    if num_rendered > 0:
        L.gsro_identify_tile_ranges(C.c_int64(num_rendered), _l(keys), _i(ranges))
        y0, y1 = (0, gy) if tile_rows is None else tile_rows
        rows = lambda a, b: L.gsro_render_rows(C.c_int(W), C.c_int(H), C.c_int(a), C.c_int(b), _i(ranges), _i(point_list),
                                               _f(xy), _f(rgb), _f(conic_opacity), _f(depths), _f(bg), _f(image), _f(depth_image),
                                               _f(final_Ts), _i(n_contrib))
        if threads > 1:
            _in_threads([(lambda a=y0 + a, b=y0 + b: rows(a, b)) for a, b in _bands(y1 - y0, threads)])
        else:
            rows(y0, y1)
        # track_pixel_stats (forward.py:590-627) is unreachable after a real render (quirk Q9).
    out = {
        "radii": radii, "point_offsets": point_offsets, "points_xy_image": xy, "depths": depths,
        "colors": rgb, "cov3Ds": cov3Ds, "conic_opacity": conic_opacity, "point_list": point_list,
        "ranges": ranges, "final_Ts": final_Ts, "n_contrib": n_contrib, "clamped_state": clamped_state,
    }
    if keep_keys:
        out["_keys"] = keys
        out["_tiles_touched"] = tiles_touched
    return image, depth_image, out


def backward(background, means3D, dL_dpixels, opacity=None, shs=None, scales=None, rotations=None,
             scale_modifier=1.0, viewmatrix=None, projmatrix=None, tan_fovx=0.5, tan_fovy=0.5,
             image_height=256, image_width=256, campos=None, radii=None, means2D=None,
             conic_opacity=None, rgb=None, clamped=None, cov3Ds=None, geom_buffer=None,
             binning_buffer=None, img_buffer=None, degree=3, debug=False, tile_rows=None, accumulate="f32", threads=1):
    """Restates backward.py:955-1196 (backward_render :890, backward_preprocess :770).  accumulate="f64" is NOT the
    reference: it swaps the blend backward for the float64-accumulating second checker (gsro_render_backward_rows_acc64).
    `threads` > 1 is NOT the reference either (bench.py's all-cores courtesy figure): bands of tile rows replay concurrently into
    per-thread accumulators that are summed afterwards, so the float32 sums are formed in another order."""
    L = lib()
    H, W = int(image_height), int(image_width)
    focal_y = H / (2.0 * float(tan_fovy))                            # backward.py:1044-1045 (float64)
    focal_x = W / (2.0 * float(tan_fovx))
    means = _f32(means3D).reshape(-1, 3)
    N = means.shape[0]
    dpix = _f32(dL_dpixels).reshape(H, W, 3)
    sh = _f32(shs).reshape(-1, 3)
    sc = _f32(scales).reshape(-1, 3)
    rot = _f32(rotations).reshape(-1, 4)
    view = _f32(np.asarray(viewmatrix, dtype=np.float64).flatten())
    proj = _f32(np.asarray(projmatrix, dtype=np.float64).flatten())
    cam = _f32(np.asarray(campos, dtype=np.float64)[:3])
    bg = _f32(np.asarray(background, dtype=np.float64)[:3])
    ranges = np.ascontiguousarray(img_buffer["ranges"], dtype=np.int32)          # :1084-1087
    final_Ts = _f32(img_buffer["final_Ts"])
    n_contrib = np.ascontiguousarray(img_buffer["n_contrib"], dtype=np.int32)
    point_list = np.ascontiguousarray(binning_buffer["point_list"], dtype=np.int32)  # :1089-1090
    if geom_buffer is not None:                                                   # :1092-1103
        radii = geom_buffer.get("radii") if radii is None else radii
        means2D = geom_buffer.get("means2D") if means2D is None else means2D
        conic_opacity = geom_buffer.get("conic_opacity") if conic_opacity is None else conic_opacity
        rgb = geom_buffer.get("rgb") if rgb is None else rgb
        clamped = geom_buffer.get("clamped_state") if clamped is None else clamped
    radii = np.ascontiguousarray(radii, dtype=np.int32)
    m2d = _f32(means2D).reshape(-1, 2)
    con = _f32(conic_opacity).reshape(-1, 4)
    col = _f32(rgb).reshape(-1, 3)
    cl = _f32(clamped).reshape(-1, 3)
    c3 = _f32(cov3Ds).reshape(-1, 6)

    dL_dmean2D = np.zeros((N, 3), np.float32)
    dL_dconic = np.zeros((N, 4), np.float32)
    dL_dopacity = np.zeros(N, np.float32)
    dL_dcolor = np.zeros((N, 3), np.float32)
    dL_dmean3D = np.zeros((N, 3), np.float32)
    dL_dcov3D_ret = np.zeros((N, 6), np.float32)    # backward.py:1119 -- returned, never filled
    dL_dsh = np.zeros((N * 16, 3), np.float32)      # always N*16 (quirk Q6)
    dL_dscale = np.zeros((N, 3), np.float32)
    dL_drot = np.zeros((N, 4), np.float32)
    gy = (H + TILE - 1) // TILE
    y0, y1 = (0, gy) if tile_rows is None else tile_rows
    if accumulate == "f64":
        # second checker (not a reference restatement): same float32 terms, per-Gaussian sums kept in float64, rounded once
        acc = [np.zeros((N, 3)), np.zeros((N, 4)), np.zeros(N), np.zeros((N, 3))]
        _d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        L.gsro_render_backward_rows_acc64(C.c_int(W), C.c_int(H), C.c_int(y0), C.c_int(y1), _i(ranges), _i(point_list),
                                          _f(bg), _f(m2d), _f(con), _f(col), _f(final_Ts), _i(n_contrib), _f(dpix),
                                          _d(acc[0]), _d(acc[1]), _d(acc[2]), _d(acc[3]))
        dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor = [np.ascontiguousarray(a, dtype=np.float32) for a in acc]
    elif threads > 1:
        bands = _bands(y1 - y0, threads)
        priv = [(np.zeros((N, 3), np.float32), np.zeros((N, 4), np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)) for _ in bands]
        _in_threads([(lambda a=y0 + a, b=y0 + b, q=q: L.gsro_render_backward_rows(
            C.c_int(W), C.c_int(H), C.c_int(a), C.c_int(b), _i(ranges), _i(point_list), _f(bg), _f(m2d), _f(con), _f(col), _f(final_Ts),
            _i(n_contrib), _f(dpix), _f(q[0]), _f(q[1]), _f(q[2]), _f(q[3]))) for (a, b), q in zip(bands, priv)])
        for q in priv:
            dL_dmean2D += q[0]; dL_dconic += q[1]; dL_dopacity += q[2]; dL_dcolor += q[3]
    else:
        L.gsro_render_backward_rows(C.c_int(W), C.c_int(H), C.c_int(y0), C.c_int(y1), _i(ranges), _i(point_list),
                                    _f(bg), _f(m2d), _f(con), _f(col), _f(final_Ts), _i(n_contrib), _f(dpix),
                                    _f(dL_dmean2D), _f(dL_dconic), _f(dL_dopacity), _f(dL_dcolor))
    dL_dcov3D = np.zeros((N, 6), np.float32)        # backward.py:812 (local)

    def geom(a, b):     # the four per-Gaussian kernels over Gaussians [a, b), in the reference's launch order
        n = C.c_int(b - a)
        L.gsro_cov2d_backward(n, _f(means[a:b]), _f(c3[a:b]), _i(radii[a:b]), C.c_float(focal_x), C.c_float(focal_y),
                              C.c_float(tan_fovx), C.c_float(tan_fovy), _f(view), _f(dL_dconic[a:b]), _f(dL_dmean3D[a:b]),
                              _f(dL_dcov3D[a:b]))
        L.gsro_projection_backward(n, _f(means[a:b]), _i(radii[a:b]), _f(proj), _f(dL_dmean2D[a:b]), _f(dL_dmean3D[a:b]))
        L.gsro_sh_backward(n, C.c_int(int(degree)), _f(means[a:b]), _f(sh[16 * a:16 * b]), _i(radii[a:b]), _f(cam), _f(cl[a:b]),
                           _f(dL_dcolor[a:b]), _f(dL_dmean3D[a:b]), _f(dL_dsh[16 * a:16 * b]))
        # backward() never forwards scale_modifier to backward_preprocess (backward.py:1155-1182), whose
        # default is 1.0 (:805): the cov3d backward always runs with scale_modifier = 1.0 (quirk Q16).
        L.gsro_cov3d_backward(n, _f(sc[a:b]), _f(rot[a:b]), _i(radii[a:b]), C.c_float(1.0), _f(dL_dcov3D[a:b]),
                              _f(dL_dscale[a:b]), _f(dL_drot[a:b]))
    if threads > 1 and N > 0:
        _in_threads([(lambda a=a, b=b: geom(a, b)) for a, b in _bands(N, threads)])
    else:
        geom(0, N)
    return {
        "dL_dmean3D": dL_dmean3D, "dL_dcolor": dL_dcolor, "dL_dshs": dL_dsh, "dL_dopacity": dL_dopacity,
        "dL_dscale": dL_dscale, "dL_drot": dL_drot, "dL_dmean2D": dL_dmean2D, "dL_dconic": dL_dconic,
        "dL_dcov3D": dL_dcov3D_ret, "_dL_dcov3D_local": dL_dcov3D,
    }


# ---- "next" rows f2 / f3 ----
def l1_loss(rendered, target):
    """Restates loss.py:148-176: serial float32 sum (x outer, y fastest) / (W*H*3)."""
    r, t = _f32(rendered), _f32(target)
    H, W = r.shape[0], r.shape[1]
    s = lib().gsro_l1_loss_sum(C.c_int(W), C.c_int(H), _f(r), _f(t))
    return float(s) / (W * H * 3)


def ssim(rendered, target):
    """Restates loss.py:178-215 (window weights indexed by distance, quirk Q21): serial float32 sum / (W*H)."""
    r, t = _f32(rendered), _f32(target)
    H, W = r.shape[0], r.shape[1]
    return float(lib().gsro_ssim_sum(C.c_int(W), C.c_int(H), _f(r), _f(t))) / (W * H)


def depth_loss(rendered_depth, target_depth, depth_mask):
    """Restates loss.py:271-303."""
    r, t, m = _f32(rendered_depth), _f32(target_depth), _f32(depth_mask)
    H, W = r.shape[0], r.shape[1]
    return float(lib().gsro_depth_loss_sum(C.c_int(W), C.c_int(H), _f(r), _f(t), _f(m))) / (W * H)


def compute_image_gradients(rendered, target, lambda_dssim=0.2):
    """Restates loss.py:217-244."""
    r, t = _f32(rendered), _f32(target)
    H, W = r.shape[0], r.shape[1]
    g = np.zeros((H, W, 3), np.float32)
    l1_weight = (1.0 - lambda_dssim) / (H * W * 3.0)
    lib().gsro_l1_pixel_grad(C.c_int(W), C.c_int(H), _f(r), _f(t), C.c_float(l1_weight), _f(g))
    return g


def adam_update(params, grads, m, v, lrs, beta1=0.9, beta2=0.999, epsilon=1e-8, iteration=0):
    """Restates optimizer.py:7-139 on numpy float32 arrays (updated in place). Keys: positions, scales,
    rotations, opacities, shs."""
    n = params["positions"].shape[0]
    K = ("positions", "scales", "rotations", "opacities", "shs")
    for d in (params, grads, m, v):
        for k in K:
            assert d[k].dtype == np.float32 and d[k].flags["C_CONTIGUOUS"]
    lib().gsro_adam_update(C.c_int(n), *[_f(grads[k]) for k in K],
                           C.c_float(lrs["positions"]), C.c_float(lrs["scales"]), C.c_float(lrs["rotations"]),
                           C.c_float(lrs["opacities"]), C.c_float(lrs["shs"]), C.c_float(beta1), C.c_float(beta2),
                           C.c_float(epsilon), C.c_int(iteration), *[_f(params[k]) for k in K], *[_f(m[k]) for k in K],
                           *[_f(v[k]) for k in K])
