"""
ctypes binding of libgsr_hip.so (include/gsr.h).  The library is the only compute path: if it is
missing this module raises -- there is no Python, torch or CPU fallback behind it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSR_LIB", os.path.join(_HERE, "libgsr_hip.so"))   # GSR_LIB: A/B runs of two builds on one box

GSR_OK, GSR_E_NULL, GSR_E_DIMS, GSR_E_OVERFLOW, GSR_E_WORKSPACE, GSR_E_HIP, GSR_E_CAPACITY, GSR_E_ALIGN = 0, -1, -2, -3, -4, -5, -6, -7

vp = C.c_void_p


class GsrScene(C.Structure):
    _fields_ = [("N", C.c_int64), ("means", vp), ("scales", vp), ("rotations", vp), ("opacity", vp), ("sh", vp),
                ("sh_degree", C.c_int32), ("scale_modifier", C.c_float), ("clamped", C.c_int32)]


class GsrCamera(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("proj", C.c_float * 16), ("campos", C.c_float * 3), ("bg", C.c_float * 3),
                ("tan_fovx", C.c_float), ("tan_fovy", C.c_float), ("focal_x", C.c_float), ("focal_y", C.c_float),
                ("W", C.c_int32), ("H", C.c_int32)]


class GsrGeom(C.Structure):
    _fields_ = [("radii", vp), ("tiles_touched", vp), ("point_offsets", vp), ("xy", vp), ("depths", vp), ("cov3D", vp),
                ("rgb", vp), ("conic_opacity", vp), ("clamped_state", vp), ("blend_records", vp), ("sh_dir_grad", vp)]


class GsrBinning(C.Structure):
    _fields_ = [("D", C.c_int64), ("point_list", vp), ("ranges", vp), ("block_masks", vp), ("block_order", vp), ("backward_ws", vp),
                ("backward_ws_cleared", C.c_int32)]


class GsrImage(C.Structure):
    _fields_ = [("image", vp), ("inv_depth", vp), ("final_T", vp), ("n_contrib", vp)]


class GsrGrads(C.Structure):
    _fields_ = [("dL_dmean3D", vp), ("dL_dscale", vp), ("dL_drot", vp), ("dL_dopacity", vp), ("dL_dshs", vp),
                ("dL_dcolor", vp), ("dL_dmean2D", vp), ("dL_dconic", vp), ("dL_drgb", vp)]


class GsrAdamGroup(C.Structure):
    _fields_ = [("param", vp), ("grad", vp), ("m", vp), ("v", vp), ("lr", C.c_float)]


class GsrAdam(C.Structure):
    _fields_ = [("N", C.c_int64), ("pos", GsrAdamGroup), ("scale", GsrAdamGroup), ("rot", GsrAdamGroup),
                ("opacity", GsrAdamGroup), ("sh", GsrAdamGroup), ("beta1", C.c_float), ("beta2", C.c_float),
                ("epsilon", C.c_float), ("iteration", C.c_int32)]


class GsrParams(C.Structure):
    _fields_ = [("N", C.c_int64), ("positions", vp), ("scales", vp), ("rotations", vp), ("opacities", vp), ("shs", vp)]


MARK_CLONE, MARK_SPLIT = 0, 1

EXPORTS = {
    "gsr_abi_version": (C.c_int, []),
    "gsr_strerror": (C.c_char_p, [C.c_int]),
    "gsr_build_flags": (C.c_int, []),
    "gsr_geom_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "gsr_binning_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "gsr_backward_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "gsr_backward_accumulators_offset": (C.c_size_t, [C.c_int64]),
    "gsr_block_order_ints": (C.c_size_t, [C.c_int32, C.c_int32]),
    "gsr_forward_count": (C.c_int, [C.POINTER(GsrScene), C.POINTER(GsrCamera), C.POINTER(GsrGeom), vp, C.c_size_t,
                                    C.POINTER(C.c_int64), vp]),
    "gsr_forward_render": (C.c_int, [C.POINTER(GsrScene), C.POINTER(GsrCamera), C.POINTER(GsrGeom), C.POINTER(GsrBinning),
                                     C.POINTER(GsrImage), vp, C.c_size_t, vp, C.c_size_t, vp]),
    "gsr_backward": (C.c_int, [C.POINTER(GsrScene), C.POINTER(GsrCamera), C.POINTER(GsrGeom), C.POINTER(GsrBinning),
                               C.POINTER(GsrImage), vp, C.POINTER(GsrGrads), vp, C.c_size_t, vp]),
    "gsr_backward_blend": (C.c_int, [C.POINTER(GsrScene), C.POINTER(GsrCamera), C.POINTER(GsrGeom), C.POINTER(GsrBinning),
                                     C.POINTER(GsrImage), vp, vp, vp, C.c_size_t, vp]),
    "gsr_backward_geom": (C.c_int, [C.POINTER(GsrScene), C.POINTER(GsrCamera), C.POINTER(GsrGeom), C.POINTER(GsrGrads), vp, C.c_size_t, vp]),
    "gsr_l1_loss_grad": (C.c_int, [vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_float, vp]),
    "gsr_ssim": (C.c_int, [vp, vp, vp, C.c_int32, C.c_int32, vp]),
    "gsr_depth_loss": (C.c_int, [vp, vp, vp, vp, C.c_int32, C.c_int32, vp]),
    "gsr_adam_update": (C.c_int, [C.POINTER(GsrAdam), vp]),
    "gsr_adam_update_views": (C.c_int, [C.POINTER(GsrAdam), C.c_int32, C.c_int32, C.POINTER(vp), C.c_float, vp]),
    "gsr_sh_grad_from_views": (C.c_int, [C.c_int64, vp, C.c_int32, C.c_int32, C.POINTER(vp), C.c_float, vp, vp]),
    "gsr_densify_mark": (C.c_int, [C.POINTER(GsrParams), vp, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int, vp, vp]),
    "gsr_prune_mark": (C.c_int, [C.POINTER(GsrParams), C.c_float, vp, vp]),
    "gsr_split_removal_mask": (C.c_int, [C.c_int64, C.c_int64, vp, vp, vp]),
    "gsr_mask_scan_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "gsr_mask_scan": (C.c_int, [C.c_int64, vp, vp, C.POINTER(C.c_int32), vp, C.c_size_t, vp]),
    "gsr_clone_gaussians": (C.c_int, [C.POINTER(GsrParams), vp, vp, C.c_float, C.POINTER(GsrParams), vp]),
    "gsr_split_gaussians": (C.c_int, [C.POINTER(GsrParams), vp, vp, C.c_int32, C.c_float, C.POINTER(GsrParams), vp]),
    "gsr_compact_gaussians": (C.c_int, [C.POINTER(GsrParams), vp, vp, C.POINTER(GsrParams), vp]),
    "gsr_init_gaussians": (C.c_int, [C.POINTER(GsrParams), C.c_float, vp]),
    "gsr_reset_opacities": (C.c_int, [C.c_int64, C.c_float, vp, vp]),
    "gsr_stage_timing": (C.c_int, [C.c_int, C.c_int]),
    "gsr_stage_sampling": (C.c_int, [C.c_int]),
    "gsr_stage_times": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_int)]),
}

STAGES = ["preprocess", "scan", "depth_sort", "host_gap", "depth_scan", "expand", "tile_sort", "ranges", "blend_fwd",
          "bwd_prep", "blend_bwd", "geom_bwd"]

_lib = None


def lib():
    """Load libgsr_hip.so once.  Raises if it has not been built (`__graft_entry__.build()`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950).  There is no fallback path.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        if h.gsr_abi_version() != 7:
            raise RuntimeError("libgsr_hip.so ABI version mismatch")
        _lib = h
    return _lib


def build_hash():
    """First 16 hex digits of the SHA-256 of the loaded library file: the key under which profiles/ stores per-build counters."""
    import hashlib
    with open(LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def strerror(code):
    return lib().gsr_strerror(code).decode()


def check(code):
    """Map a GSR_E_* return code to the exception the reference would raise."""
    if code == GSR_OK:
        return
    if code == GSR_E_OVERFLOW:   # reference forward.py:765-767 raises ValueError
        raise ValueError("Number of rendered points exceeds the maximum supported (2^30).")
    raise RuntimeError(f"libgsr_hip: {strerror(code)} (code {code})")


def stage_timing(enable, max_steps=256, every=1):
    """Record per-stage HIP events on one forward/backward pair in `every` (event records are not free).  every = 0 creates the
    events but records nothing until stage_sampling(k >= 1): a benchmark allocates them before its warm-up and switches them on
    only for steps it does not time."""
    check(lib().gsr_stage_timing(1 if enable else 0, int(max_steps)))
    if enable:
        check(lib().gsr_stage_sampling(int(every)))


def stage_sampling(every):
    """0 = pause the recording (events stay allocated), k >= 1 = record one forward/backward pair in k."""
    check(lib().gsr_stage_sampling(int(every)))


def stage_times():
    """Average ms per stage over the steps recorded since stage_timing(True); synchronise first."""
    arr = (C.c_float * len(STAGES))()
    n = C.c_int(0)
    check(lib().gsr_stage_times(arr, C.byref(n)))
    return {k: float(arr[i]) for i, k in enumerate(STAGES)}, n.value
