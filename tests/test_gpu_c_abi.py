"""The drop-in boundary is a C ABI: a plain C program (tests/c_abi/gsr_client.c, built with gcc -- no torch, no Python, no
C++) allocates device memory with hipMalloc, calls gsr_forward_count / gsr_forward_render / gsr_backward through
include/gsr.h, and its outputs must match the CPU oracle under the tolerances of tests/parity.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import parity
from conftest import ROOT, backward_kwargs, lego_camera, render_kwargs, sub

pytestmark = pytest.mark.gpu
HERE = os.path.join(ROOT, "tests", "c_abi")


def _client(name="gsr_client"):
    exe = os.path.join(HERE, name)
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", HERE, name])
    return exe


def run_client(exe, scene, kw, dpix, W, H, n, tmp_path, degree=3, mode=None):
    """Write the flat input file, run the plain-C client, parse its flat output: (image, inv_depth, buffers, grads)."""
    _host = sub("_host")
    cstruct = _host.make_camera(kw["viewmatrix"], kw["projmatrix"], kw["campos"], kw["background"], kw["tan_fovx"], kw["tan_fovy"], W, H)
    f32 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float32)).tobytes()
    tag = os.path.basename(exe) + ("_" + mode if mode else "")
    with open(tmp_path / f"in_{tag}.bin", "wb") as f:
        f.write(np.int64(n).tobytes() + np.array([W, H, degree, 0, 0, 0], np.int32).tobytes() + bytes(cstruct))
        for k in ("means", "scales", "rotations", "opacities", "shs"):
            f.write(f32(scene[k]))
        f.write(f32(dpix))
    r = subprocess.run([exe, str(tmp_path / f"in_{tag}.bin"), str(tmp_path / f"out_{tag}.bin")] + ([mode] if mode else []), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    blob = open(tmp_path / f"out_{tag}.bin", "rb").read()
    D = int(np.frombuffer(blob, np.int64, 1)[0])
    off = 8
    tiles = ((W + 15) // 16) * ((H + 15) // 16)

    def take(dtype, count, shape):
        nonlocal off
        a = np.frombuffer(blob, dtype, count, off).reshape(shape)
        off += a.nbytes
        return a

    i32, f4 = np.int32, np.float32
    buf = {"radii": take(i32, n, (n,)), "point_offsets": take(i32, n, (n,)), "points_xy_image": take(f4, 2 * n, (n, 2)),
           "depths": take(f4, n, (n,)), "cov3Ds": take(f4, 6 * n, (n, 6)), "colors": take(f4, 3 * n, (n, 3)),
           "conic_opacity": take(f4, 4 * n, (n, 4)), "clamped_state": take(f4, 3 * n, (n, 3)), "point_list": take(i32, D, (D,)),
           "ranges": take(i32, 2 * tiles, (tiles, 2))}
    image, inv_depth = take(f4, 3 * W * H, (H, W, 3)), take(f4, W * H, (H, W))
    buf["final_Ts"], buf["n_contrib"] = take(f4, W * H, (H, W)), take(i32, W * H, (H, W))
    grads = {"dL_dmean3D": take(f4, 3 * n, (n, 3)), "dL_dscale": take(f4, 3 * n, (n, 3)), "dL_drot": take(f4, 4 * n, (n, 4)),
             "dL_dopacity": take(f4, n, (n,)), "dL_dshs": take(f4, 48 * n, (16 * n, 3)), "dL_dcolor": take(f4, 3 * n, (n, 3)),
             "dL_dmean2D": take(f4, 3 * n, (n, 3)), "dL_dconic": take(f4, 4 * n, (n, 4)), "dL_dcov3D": np.zeros((n, 6), f4)}
    assert off == len(blob)
    return image, inv_depth, buf, grads


def test_plain_c_client_matches_oracle(oracle, cameras, scenes, tmp_path):
    _lib, _host = sub("_lib"), sub("_host")
    W, H, n, degree = 176, 144, 3000, 3
    scene = scenes.synthetic_scene(n, 0.04, 0.5, seed=21)
    cam = lego_camera(cameras, frame=2, width=W, height=H)
    kw = render_kwargs(scene, cam, width=W, height=H, degree=degree)
    dpix = (np.random.default_rng(7).normal(0.0, 1.0, (H, W, 3)) / (H * W * 3)).astype(np.float32)
    cstruct = _host.make_camera(kw["viewmatrix"], kw["projmatrix"], kw["campos"], kw["background"], kw["tan_fovx"], kw["tan_fovy"], W, H)
    image, inv_depth, buf, grads = run_client(_client(), scene, kw, dpix, W, H, n, tmp_path, degree)
    D = len(buf["point_list"])

    ref = oracle.render_gaussians(**kw)
    assert D == len(ref[2]["point_list"]) > 1000
    parity.compare_forward((image, inv_depth, buf), ref)
    # gradients: the oracle replays the client's own forward buffers (they passed the comparison above), so a forward
    # threshold flip cannot leak into the gradient comparison
    g_ref = oracle.backward(**backward_kwargs(scene, cam, kw, {**buf, "final_Ts": buf["final_Ts"], "n_contrib": buf["n_contrib"]}, dpix))
    parity.compare_backward(grads, g_ref)
    assert C.sizeof(_lib.GsrCamera) == len(bytes(cstruct))


def test_same_client_against_both_libraries(cameras, scenes, tmp_path):
    """libgsr_hip.so and the oracle's libgsr_cpu.so export the same ABI; the SAME plain-C program,
    built once per library, is run on the same input file and the two dumps are diffed -- byte for byte on every integer
    output, under tests/parity.py on the floats."""
    W, H, n = 208, 160, 5000
    scene = scenes.synthetic_scene(n, 0.04, 0.6, seed=33)
    cam = lego_camera(cameras, frame=6, width=W, height=H)
    kw = render_kwargs(scene, cam, width=W, height=H)
    dpix = (np.random.default_rng(8).normal(0.0, 1.0, (H, W, 3)) / (H * W * 3)).astype(np.float32)
    gi, gd, gb, gg = run_client(_client(), scene, kw, dpix, W, H, n, tmp_path)
    ci, cd, cb, cg = run_client(_client("gsr_client_cpu"), scene, kw, dpix, W, H, n, tmp_path)
    for k in ("radii", "point_offsets", "point_list", "ranges"):
        assert gb[k].tobytes() == cb[k].tobytes(), k                      # byte for byte
    parity.compare_forward((gi, gd, gb), (ci, cd, cb))
    parity.compare_backward(gg, cg)


def test_records_only_mode_through_the_c_abi(oracle, cameras, scenes, tmp_path):
    """ABI 7 from plain C: the forward is given a record buffer and NO xy / conic_opacity / rgb arrays, the backward NO dL_dcolor /
    dL_dmean2D / dL_dconic arrays; the client reads all six back as columns of the blend records and of the accumulator records
    (gsr_backward_accumulators_offset) and writes them in the usual dump layout.  Against the same client with packed arrays: every
    forward output byte for byte (the same kernels wrote the same values, only elsewhere), gradients under tests/parity.py (float
    atomics), and both against the oracle."""
    W, H, n = 192, 144, 4000
    scene = scenes.synthetic_scene(n, 0.04, 0.6, seed=57)
    cam = lego_camera(cameras, frame=1, width=W, height=H)
    kw = render_kwargs(scene, cam, width=W, height=H)
    dpix = (np.random.default_rng(9).normal(0.0, 1.0, (H, W, 3)) / (H * W * 3)).astype(np.float32)
    pi, pd, pb, pg = run_client(_client(), scene, kw, dpix, W, H, n, tmp_path)
    ri, rd, rb, rg = run_client(_client(), scene, kw, dpix, W, H, n, tmp_path, mode="records")
    assert pi.tobytes() == ri.tobytes() and pd.tobytes() == rd.tobytes()
    for k in pb:
        assert pb[k].tobytes() == rb[k].tobytes(), k
    parity.compare_backward(rg, pg)
    assert float(np.abs(rg["dL_dmean2D"][:, 2]).max()) == 0.0 and float(np.abs(rg["dL_dconic"][:, 2]).max()) == 0.0
    ref = oracle.render_gaussians(**kw)
    parity.compare_forward((ri, rd, rb), ref)
    parity.compare_backward(rg, oracle.backward(**backward_kwargs(scene, cam, kw, rb, dpix)))
