"""CPU-side checks: the C-ABI library loads and exports every symbol include/gsr.h declares (no compute
calls without a GPU), ctypes struct layouts match the header, host-side camera math follows the
reference's conventions, error codes map to the reference's exception types."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, PKG_NAME, lego_camera, sub


@pytest.fixture(scope="module")
def libpath():
    path = os.path.join(ROOT, PKG_NAME, "libgsr_hip.so")
    if not os.path.exists(path):   # hipcc cross-compiles gfx950 without a GPU
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, PKG_NAME, "csrc")])
    return path


def test_every_declared_symbol_is_exported(libpath):
    hdr = open(os.path.join(ROOT, "include", "gsr.h")).read()
    declared = set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", hdr))
    assert {"gsr_forward_count", "gsr_forward_render", "gsr_backward", "gsr_strerror"} <= declared
    lib = C.CDLL(libpath)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in gsr.h but not exported"
    _lib = sub("_lib")
    assert set(_lib.EXPORTS) == declared, "ctypes binding and header disagree"


def test_host_only_entry_points(libpath):
    _lib = sub("_lib")
    L = _lib.lib()
    assert L.gsr_abi_version() == 7
    assert _lib.strerror(0) == "ok" and "2^30" in _lib.strerror(_lib.GSR_E_OVERFLOW)
    a, b = L.gsr_geom_workspace_bytes(1000), L.gsr_geom_workspace_bytes(2000)
    assert 0 < a < b and b >= 2000 * (64 + 8 + 8)
    assert L.gsr_binning_workspace_bytes(1000, 5000, 800, 800) >= 2 * 5000 * 8
    assert L.gsr_backward_workspace_bytes(1000, 5000, 800, 800) >= 1000 * 128
    with pytest.raises(ValueError):
        _lib.check(_lib.GSR_E_OVERFLOW)       # reference raises ValueError (forward.py:765-767)
    with pytest.raises(RuntimeError):
        _lib.check(_lib.GSR_E_WORKSPACE)


def test_struct_layouts_match_header():
    _lib = sub("_lib")
    assert C.sizeof(_lib.GsrCamera) == (16 + 16 + 3 + 3 + 2 + 2) * 4 + 2 * 4
    assert C.sizeof(_lib.GsrScene) == 8 + 5 * 8 + 3 * 4 + 4   # trailing pad to 8
    assert C.sizeof(_lib.GsrGeom) == 11 * 8 and C.sizeof(_lib.GsrGrads) == 9 * 8 and C.sizeof(_lib.GsrParams) == 6 * 8
    assert C.sizeof(_lib.GsrBinning) == 56 and C.sizeof(_lib.GsrImage) == 32
    assert _lib.GsrCamera.focal_x.offset == (16 + 16 + 3 + 3 + 2) * 4


def test_missing_library_fails_loudly(monkeypatch):
    _lib = sub("_lib")
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgsr_hip.so")
    with pytest.raises(RuntimeError, match="no fallback"):
        _lib.lib()


def test_product_never_imports_oracle():
    pkg_dir = os.path.join(ROOT, PKG_NAME)
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "gsro_" not in text, f


def test_nerf_camera_conventions(cameras):
    cam = lego_camera(cameras, 0, 800, 800)
    w2c = cam["world_to_camera"]
    assert w2c.dtype == np.float32
    np.testing.assert_allclose(w2c[:3, 3], 0.0, atol=1e-7)            # row-vector form: translation in row 3
    assert abs(w2c[3, 3] - 1.0) < 1e-7 and np.abs(w2c[3, :3]).max() > 1.0
    R = w2c[:3, :3]
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-5)
    # camera centre maps to the view-space origin
    c = np.append(cam["camera_center"], 1.0) @ w2c
    np.testing.assert_allclose(c[:3], 0.0, atol=1e-5)
    # full projection = view @ proj, tan_fov from camera_angle_x
    np.testing.assert_allclose(cam["full_proj_matrix"], w2c @ cam["proj_matrix"], rtol=1e-6)
    assert abs(cam["tan_fovx"] - np.tan(0.5 * 0.6911112070083618)) < 1e-9
    # a point 4 units in front of the camera projects to the image centre with w = depth
    p_world = np.append(cam["camera_center"], 1.0) + np.append(4.0 * np.linalg.inv(w2c)[2, :3], 0.0)
    hom = p_world @ cam["full_proj_matrix"]
    np.testing.assert_allclose(hom[:2] / hom[3], 0.0, atol=1e-5)
    assert abs(hom[3] - 4.0) < 1e-4


def test_toy_camera_quirk_q3(cameras):
    cam = cameras.toy_camera()
    v = cam["view_matrix"]
    # render.py passes the un-transposed matrix: translation sits in COLUMN 3, so p*V has no translation
    np.testing.assert_allclose(v[:3, 3], [0, 0, 5])
    np.testing.assert_allclose(v[3, :3], 0.0)
    assert abs(cam["tan_fovx"] - 0.5578517) < 1e-6
    np.testing.assert_allclose(cam["camera_center"], [0, 0, 5], atol=1e-6)   # 'camera_center = (0,0,5)' SURVEY section 4


def test_synthetic_scene_is_seeded(scenes):
    a, b = scenes.synthetic_scene(100, 0.02, 0.5, 7), scenes.synthetic_scene(100, 0.02, 0.5, 7)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    assert a["shs"].shape == (100, 16, 3) and a["rotations"].shape == (100, 4)
    np.testing.assert_allclose(np.linalg.norm(a["rotations"], axis=1), 1.0, atol=1e-6)
    assert a["scales"].min() >= 1e-3


def test_argument_validation_needs_no_gpu(libpath):
    """Bad arguments are rejected before any HIP call, so this runs on a CPU-only box."""
    _lib = sub("_lib")
    L = _lib.lib()
    scene, cam, geom = _lib.GsrScene(), _lib.GsrCamera(), _lib.GsrGeom()
    D = C.c_int64(-1)
    assert L.gsr_forward_count(None, C.byref(cam), C.byref(geom), None, 0, C.byref(D), None) == _lib.GSR_E_NULL
    scene.N, cam.W, cam.H, scene.sh_degree = 10, 0, 16, 3
    assert L.gsr_forward_count(C.byref(scene), C.byref(cam), C.byref(geom), None, 0, C.byref(D), None) == _lib.GSR_E_DIMS
    cam.W, scene.sh_degree = 16, 4
    assert L.gsr_forward_count(C.byref(scene), C.byref(cam), C.byref(geom), None, 0, C.byref(D), None) == _lib.GSR_E_DIMS
    scene.sh_degree = 3     # N > 0 with null arrays
    assert L.gsr_forward_count(C.byref(scene), C.byref(cam), C.byref(geom), None, 0, C.byref(D), None) == _lib.GSR_E_NULL
    assert L.gsr_forward_count(C.byref(scene), C.byref(cam), C.byref(geom), None, 0, None, None) == _lib.GSR_E_NULL
    scene.N = 0             # N == 0 is accepted and yields D = 0 without touching the device
    assert L.gsr_forward_count(C.byref(scene), C.byref(cam), C.byref(geom), None, 0, C.byref(D), None) == _lib.GSR_OK and D.value == 0
    assert L.gsr_adam_update(None, None) == _lib.GSR_E_NULL
    assert L.gsr_l1_loss_grad(None, None, None, None, 4, 4, C.c_float(1.0), None) == _lib.GSR_E_NULL
    a = _lib.GsrAdam()
    a.N = 0
    assert L.gsr_adam_update(C.byref(a), None) == _lib.GSR_OK


def test_lr_scheduler_matches_the_reference_formula():
    sch = sub("scheduler").LRScheduler(1e-2, 0.01)
    assert sch.get_lr(0, 7000) == 1e-2
    assert abs(sch.get_lr(6999, 7000) - 1e-4) < 1e-12
    assert abs(sch.get_lr(3500, 7001) - 1e-3) < 1e-9        # geometric midpoint
    assert sch.get_lr(10 ** 6, 7000) == sch.get_lr(6999, 7000) and sch.get_lr(5, 1) == 1e-2


def test_header_is_plain_c_and_client_links(tmp_path):
    """include/gsr.h must compile as C11 (it is the boundary a non-C++ host binds), and the plain-C client of
    tests/c_abi must link against the library with nothing but the HIP runtime."""
    import subprocess
    probe = tmp_path / "probe.c"
    probe.write_text('#include "gsr.h"\nint main(void) { GsrScene s; GsrParams p; (void)s; (void)p; return GSR_ABI_VERSION == 7 ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c11", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(probe)])
    subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(ROOT, "tests", "c_abi")])
    assert os.path.exists(os.path.join(ROOT, "tests", "c_abi", "gsr_client"))


def test_reference_camera_helper_names(cameras):
    """load_camera / world_to_view / projection_matrix / matrix_to_quaternion (reference utils/camera_utils.py, math_utils.py)."""
    import json
    from scipy.spatial.transform import Rotation
    with open(os.path.join(ROOT, "tests", "golden", "lego_train_poses.json")) as f:
        d = json.load(f)
    W = H = 800
    focal = 0.5 * W / np.tan(0.5 * d["camera_angle_x"])                     # train.py:296
    info = {"camera_id": 3, "camera_to_world": d["frames"][3]["transform_matrix"], "width": W, "height": H, "focal": focal}
    cam = cameras.load_camera(info)
    ref = cameras.nerf_camera(d["frames"][3]["transform_matrix"], W, H, d["camera_angle_x"])
    assert set(cam) == {"R", "T", "camera_center", "view_matrix", "proj_matrix", "full_proj_matrix", "tan_fovx", "tan_fovy", "fx", "fy", "cx",
                        "cy", "width", "height", "camera_to_world", "world_to_camera", "camera_type", "distortion_params"}
    for k in ("R", "T", "world_to_camera", "view_matrix", "camera_center"):
        np.testing.assert_array_equal(cam[k], ref[k])
    np.testing.assert_allclose(cam["full_proj_matrix"], ref["full_proj_matrix"], rtol=1e-12)
    assert cam["camera_type"] == 0 and cam["distortion_params"].shape == (6,) and cam["cx"] == 400
    assert cameras.load_camera(dict(info, camera_model="OPENCV_FISHEYE", k1=0.1))["distortion_params"][0] == np.float32(0.1)
    with pytest.raises(ValueError):
        cameras.load_camera(dict(info, camera_model="PINHOLE_X"))
    # world_to_view: translate / scale act on the camera centre
    R, T = cam["R"], cam["T"]
    v0 = cameras.world_to_view(R, T)
    np.testing.assert_array_equal(v0, cam["view_matrix"])
    v1 = cameras.world_to_view(R, T, translate=np.array([1.0, 2.0, 3.0]), scale=2.0)
    c0, c1 = np.linalg.inv(v0)[:3, 3], np.linalg.inv(v1)[:3, 3]
    np.testing.assert_allclose(c1, (c0 + [1.0, 2.0, 3.0]) * 2.0, rtol=1e-5)
    P = cameras.projection_matrix(0.7, 0.5, 0.01, 100.0)
    assert P[3, 2] == 1.0 and abs(P[0, 0] - 1.0 / np.tan(0.35)) < 1e-12 and abs(P[2, 2] - 100.0 / 99.99) < 1e-12
    # matrix_to_quaternion: (x, y, z, w), every branch (trace > 0 and each largest-diagonal case), against scipy up to sign
    rots = [Rotation.from_euler("xyz", e).as_matrix() for e in ([0.1, 0.2, 0.3], [3.0, 0.1, 0.1], [0.1, 3.0, 0.1], [0.1, 0.1, 3.0])]
    rots += list(Rotation.random(20, random_state=5).as_matrix())
    for m in rots:
        q = cameras.matrix_to_quaternion(m)
        r = Rotation.from_matrix(m).as_quat()
        assert q.dtype == np.float32 and min(np.abs(q - r).max(), np.abs(q + r).max()) < 1e-6


def test_gaussian_params_config_surface():
    cfg = sub("config")
    d = cfg.GaussianParams.get_config_dict()
    assert d["num_points"] == 5000 and d["densify_grad_threshold"] == 0.0002 and d["lr_scheduler_config"]["lr_sh"] == 2e-3 and len(d) == 25
    cfg.GaussianParams.update(num_points=123)
    try:
        assert cfg.GaussianParams.get_config_dict()["num_points"] == 123 and cfg.GaussianParams.num_points == 123
        with pytest.raises(ValueError):
            cfg.GaussianParams.update(not_a_parameter=1)
    finally:
        cfg.GaussianParams.update(num_points=5000)


def test_every_abi_symbol_is_documented_for_integrators():
    """INTEGRATION.md is the maintainer-facing map from each entry point to the reference code it replaces: no symbol of
    include/gsr.h may be missing from it."""
    import re
    header = open(os.path.join(ROOT, "include", "gsr.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    names = sorted(set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", header)))
    assert len(names) >= 28
    missing = [n for n in names if n not in doc]
    assert not missing, missing


def test_product_build_has_no_ablation_switches(libpath):
    """GSR_DEBUG bits 0-3 (skip atomics / one pixel per bucket / no SH fetch / no stores: wrong results, for timing only) exist
    only in the separate `make ablate` build; the product library reports build flags 0 and its recipe has no -DGSR_ABLATE."""
    _lib = sub("_lib")
    assert _lib.lib().gsr_build_flags() == 0
    mk = open(os.path.join(ROOT, PKG_NAME, "csrc", "Makefile")).read()
    product_flags = re.search(r"^CXXFLAGS = (.*(?:\\\n.*)*)", mk, re.M).group(1)
    assert "GSR_ABLATE" not in product_flags and "-DGSR_ABLATE" in mk
    hdr = open(os.path.join(ROOT, PKG_NAME, "csrc", "gsr_internal.h")).read()
    assert "#define GSR_ABL(flags, bit) false" in hdr and "#define GSR_DEBUG_ALLOWED (32 | 64 | 128 | 256 | 512 | 1024)" in hdr
    for f in ("blend_bwd_splat.hip", "preprocess.hip"):      # every use of the kernels' debug word goes through GSR_ABL
        src = open(os.path.join(ROOT, PKG_NAME, "csrc", f)).read()
        assert not re.search(r"\bdbg\s*&", src), f


def test_alignment_and_capacity_are_checked_before_any_hip_call(libpath):
    """include/gsr.h 'Alignment': array pointers must be 16-byte aligned -> GSR_E_ALIGN; gsr_forward_render refuses a geom
    workspace that gsr_forward_count never counted -> GSR_E_CAPACITY.  Fake pointers: nothing is dereferenced."""
    _lib = sub("_lib")
    L = _lib.lib()
    A = 0x10000         # any 16-byte aligned non-null value
    scene = _lib.GsrScene(8, A, A, A + 4, A, A, 3, 1.0, 1)           # rotations off by 4 bytes
    cam = _lib.GsrCamera()
    cam.W, cam.H, cam.tan_fovx, cam.tan_fovy = 32, 32, 0.5, 0.5
    geom = _lib.GsrGeom(A, A, A, A, A, A, A, A, A, None, None)
    D = C.c_int64(0)
    assert L.gsr_forward_count(C.byref(scene), C.byref(cam), C.byref(geom), A, 1 << 30, C.byref(D), None) == _lib.GSR_E_ALIGN
    scene.rotations = A
    geom.conic_opacity = A + 8
    assert L.gsr_forward_count(C.byref(scene), C.byref(cam), C.byref(geom), A, 1 << 30, C.byref(D), None) == _lib.GSR_E_ALIGN
    geom.conic_opacity = A
    binning, img = _lib.GsrBinning(100, A, A, None), _lib.GsrImage(A, A, A, A)
    rc = L.gsr_forward_render(C.byref(scene), C.byref(cam), C.byref(geom), C.byref(binning), C.byref(img), A, 1 << 30, A, 1 << 30, None)
    assert rc == _lib.GSR_E_CAPACITY and "count" in _lib.strerror(rc)
    grads = _lib.GsrGrads(A, A, A + 4, A, A, A, A, A, None)
    rc = L.gsr_backward_geom(C.byref(scene), C.byref(cam), C.byref(geom), C.byref(grads), A, 1 << 30, None)
    assert rc == _lib.GSR_E_ALIGN
    with pytest.raises(RuntimeError, match="aligned"):
        _lib.check(_lib.GSR_E_ALIGN)


def test_packed_cameras_are_kept_by_the_bytes_of_their_inputs(cameras):
    """_host.make_camera keeps the packed GsrCamera under the bytes of the arrays it was made from (numpy has no version counter):
    the same values give the same struct back, a value written in place gives a new one, and either equals a fresh packing."""
    import importlib
    host = importlib.import_module(PKG_NAME + "._host")
    cam = lego_camera(cameras, frame=2, width=64, height=48)
    view, proj, pos = (np.array(cam[k], copy=True) for k in ("world_to_camera", "full_proj_matrix", "camera_center"))
    bg = np.zeros(3, np.float32)
    args = (view, proj, pos, bg, cam["tan_fovx"], cam["tan_fovy"], 64, 48)
    a = host.make_camera(*args)
    assert host.make_camera(*args) is a                                   # kept
    assert host.make_camera(view.copy(), proj.copy(), pos.copy(), bg.copy(), *args[4:]) is a    # by value, not by identity
    fresh = host._pack_camera(*args)
    assert bytes(a) == bytes(fresh)
    view[3, 0] += 0.25                                                    # the caller moves its camera in place
    b = host.make_camera(*args)
    assert b is not a and bytes(b) == bytes(host._pack_camera(*args)) and bytes(b) != bytes(fresh)
    assert host.make_camera(view, proj, pos, bg, cam["tan_fovx"], cam["tan_fovy"], 64, 64) is not b      # another image size
    assert host.make_camera(view.astype(np.float64), proj, pos, bg, *args[4:]) is not b and \
        bytes(host.make_camera(view.astype(np.float64), proj, pos, bg, *args[4:])) == bytes(b)          # another dtype, same camera
    lists = host.make_camera(view.tolist(), proj, pos, bg, *args[4:])       # not arrays: packed every time, same contents
    assert lists is not b and bytes(lists) == bytes(b)
    assert host.ptr(None) is None


def test_arena_offsets_are_16_byte_aligned():
    d = sub("dist")
    for n in (0, 1, 2, 3, 5, 7, 1000, 100003):
        for small in (False, True):
            o = d.arena_offsets(n, small)
            assert all(x % 4 == 0 for x in o[:-1]) and o[-1] == d.arena_size(n, small)
            sizes = [3 * n, 3 * n, 4 * n, n] + ([] if small else [48 * n])
            assert all(o[k] + sizes[k] <= o[k + 1] for k in range(len(sizes)))
            assert o[-1] - sum(sizes) <= 3 * (len(sizes) - 1)
    assert d.arena_offsets(1000) == [0, 3000, 6000, 10000, 11000, 59000]      # N % 4 == 0: the plain 3N|3N|4N|N|48N layout


def test_cpu_library_has_the_same_abi_and_reproduces_the_oracle(oracle, cameras, scenes, tmp_path):
    """oracle/libgsr_cpu.so (test infrastructure): the CPU oracle behind the product's own C ABI -- SURVEY.md section 8(b)'s
    "identical symbol set".  Every symbol include/gsr.h declares is exported, and the plain-C client of tests/c_abi, built
    against it with host buffers, reproduces oracle.py bit for bit (both run the same gsro_* functions)."""
    import re
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "c_abi"), "gsr_client_cpu"])
    hdr = open(os.path.join(ROOT, "include", "gsr.h")).read()
    declared = set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", hdr))
    cpu = C.CDLL(os.path.join(ROOT, "oracle", "libgsr_cpu.so"))
    for name in sorted(declared):
        assert hasattr(cpu, name), f"{name} declared in gsr.h but missing from libgsr_cpu.so"
    assert cpu.gsr_abi_version() == sub("_lib").lib().gsr_abi_version() and cpu.gsr_build_flags() == 2
    from test_gpu_c_abi import run_client
    from conftest import backward_kwargs, render_kwargs
    W, H, n = 96, 64, 400
    scene = scenes.synthetic_scene(n, 0.06, 0.5, seed=4)
    cam = lego_camera(cameras, frame=1, width=W, height=H)
    kw = render_kwargs(scene, cam, width=W, height=H)
    dpix = (np.random.default_rng(2).normal(0.0, 1.0, (H, W, 3)) / (H * W * 3)).astype(np.float32)
    image, inv_depth, buf, grads = run_client(os.path.join(ROOT, "tests", "c_abi", "gsr_client_cpu"), scene, kw, dpix, W, H, n, tmp_path)
    ref = oracle.render_gaussians(**kw)
    np.testing.assert_array_equal(image, ref[0])
    np.testing.assert_array_equal(inv_depth, ref[1])
    for k in ("radii", "point_offsets", "point_list", "ranges", "n_contrib", "final_Ts", "colors", "conic_opacity", "cov3Ds"):
        np.testing.assert_array_equal(buf[k], np.asarray(ref[2][k]).reshape(buf[k].shape), err_msg=k)
    g_ref = oracle.backward(**backward_kwargs(scene, cam, kw, ref[2], dpix))
    for k in ("dL_dmean3D", "dL_dscale", "dL_drot", "dL_dopacity", "dL_dshs", "dL_dcolor", "dL_dmean2D", "dL_dconic"):
        np.testing.assert_array_equal(grads[k], np.asarray(g_ref[k]).reshape(grads[k].shape), err_msg=k)
