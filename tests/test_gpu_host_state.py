"""GPU tests of the library's host-side state (include/gsr.h, 'Conventions'): the count check behind GSR_E_CAPACITY, concurrent
host threads over distinct buffers and streams (readback slots are leased per call, not per thread), and the timing-ablation
switches of GSR_DEBUG being inert in the product build."""
import ctypes as C
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import parity
from conftest import ROOT, lego_camera, pkg, render_kwargs, sub

pytestmark = pytest.mark.gpu


def test_forward_render_refuses_a_wrong_count(cameras, scenes):
    """gsr_forward_render used to trust GsrBinning.D -- too large wrote out of bounds on the device, too small
    truncated the list.  Now it must equal the count gsr_forward_count returned for the same geom workspace."""
    import torch
    _lib, _host = sub("_lib"), sub("_host")
    L = _lib.lib()
    W, H, n = 128, 96, 1500
    sc = scenes.synthetic_scene(n, 0.05, 0.5, seed=3)
    cam = lego_camera(cameras, frame=1, width=W, height=H)
    kw = render_kwargs(sc, cam, width=W, height=H)
    dev = torch.device("cuda", 0)
    t = lambda a, shape: torch.as_tensor(np.ascontiguousarray(a, np.float32)).reshape(shape).to(dev)
    means, scales, rots, op, shs = t(sc["means"], (n, 3)), t(sc["scales"], (n, 3)), t(sc["rotations"], (n, 4)), t(sc["opacities"], (n,)), t(sc["shs"], (n * 16, 3))
    scene = _lib.GsrScene(n, _host.ptr(means), _host.ptr(scales), _host.ptr(rots), _host.ptr(op), _host.ptr(shs), 3, 1.0, 1)
    cs = _host.make_camera(kw["viewmatrix"], kw["projmatrix"], kw["campos"], kw["background"], kw["tan_fovx"], kw["tan_fovy"], W, H)
    e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
    i32, f32 = torch.int32, torch.float32
    bufs = [e((n,), i32), e((n,), i32), e((n,), i32), e((n, 2), f32), e((n,), f32), e((n, 6), f32), e((n, 3), f32), e((n, 4), f32), e((n, 3), f32)]
    geom = _lib.GsrGeom(*[_host.ptr(b) for b in bufs], None)
    gws = torch.empty(L.gsr_geom_workspace_bytes(n), dtype=torch.uint8, device=dev)
    gws2 = torch.empty_like(gws)                                            # never counted
    D = C.c_int64(0)
    stream = _host.stream_ptr(dev)
    assert L.gsr_forward_count(C.byref(scene), C.byref(cs), C.byref(geom), _host.ptr(gws), gws.numel(), C.byref(D), stream) == 0
    D = D.value
    assert D > 0
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    img = [e((H, W, 3), f32), e((H, W), f32), e((H, W), f32), e((H, W), i32)]
    image = _lib.GsrImage(*[_host.ptr(b) for b in img])
    ranges = e((tiles, 2), i32)
    point_list = e((D + 64,), i32)
    bws = torch.empty(L.gsr_binning_workspace_bytes(n, D + 64, W, H), dtype=torch.uint8, device=dev)

    def render(d, ws):
        b = _lib.GsrBinning(d, _host.ptr(point_list), _host.ptr(ranges), None)
        return L.gsr_forward_render(C.byref(scene), C.byref(cs), C.byref(geom), C.byref(b), C.byref(image), _host.ptr(ws), ws.numel(),
                                    _host.ptr(bws), bws.numel(), stream)

    assert render(D + 1, gws) == _lib.GSR_E_CAPACITY
    assert render(D - 1, gws) == _lib.GSR_E_CAPACITY
    assert render(D, gws2) == _lib.GSR_E_CAPACITY
    assert render(D, gws) == 0                                              # the right count still renders, and twice
    first = img[0].clone()
    assert render(D, gws) == 0
    torch.cuda.synchronize()
    assert torch.equal(first, img[0]) and bool(torch.isfinite(first).all())
    with pytest.raises(RuntimeError, match="count"):
        _lib.check(_lib.GSR_E_CAPACITY)


def test_two_host_threads_over_distinct_buffers(oracle, cameras, scenes):
    """Two host threads, each with its own HIP stream, scene and (per-stream) workspaces, call render_gaussians + backward
    concurrently (ctypes drops the GIL inside the library): results equal what the same calls give one after the other."""
    import torch
    gsr = pkg()
    cases = []
    for k, (n, seed, frame) in enumerate([(20000, 71, 0), (26000, 72, 4)]):
        sc = scenes.synthetic_scene(n, 0.03, 0.6, seed=seed)
        cam = lego_camera(cameras, frame=frame, width=256, height=192)
        kw = render_kwargs(sc, cam, width=256, height=192)
        for key in ("means3D", "opacity", "scales", "rotations", "sh"):
            kw[key] = torch.as_tensor(np.ascontiguousarray(kw[key], np.float32)).cuda()
        cases.append(kw)

    def run(kw, reps, out):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(reps):
                img, depth, buf = gsr.render_gaussians(**kw)
            st.synchronize()
        out.append((img.cpu().numpy(), buf["point_list"].cpu().numpy(), buf["ranges"].cpu().numpy(), buf["n_contrib"].cpu().numpy()))

    seq = [[], []]
    for k in range(2):
        run(cases[k], 1, seq[k])
    par = [[], []]
    errs = []

    def guarded(k):
        try:
            run(cases[k], 25, par[k])
        except Exception as ex:      # noqa: BLE001 -- surfaced below
            errs.append(ex)

    threads = [threading.Thread(target=guarded, args=(k,)) for k in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errs, errs
    for k in range(2):
        for a, b in zip(seq[k][0], par[k][0]):
            np.testing.assert_array_equal(a, b)
    ref = oracle.render_gaussians(**{**cases[0], **{key: cases[0][key].cpu().numpy() for key in ("means3D", "opacity", "scales", "rotations", "sh")}})
    parity.assert_exact("point_list", par[0][0][1], ref[2]["point_list"])


def test_debug_ablation_bits_are_inert_in_the_product_build():
    """GSR_DEBUG=15 (every timing ablation: no atomics, one pixel per bucket, no SH fetch, no stores) must change nothing in
    libgsr_hip.so: smoke() still matches the oracle.  The library reads the variable once, hence a subprocess."""
    env = dict(os.environ, GSR_DEBUG="15")
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
