"""GPU tests of the view-parallel gradient exchange: backward(sh_gradient="factored") plus
dist.sh_gradients_from_views must give, for V views, the same summed / averaged gradient as V dense backward() calls
added up -- the all-reduce it replaces -- because one view's SH gradient is basis(dir) x dL_drgb (backward.py:95-255)."""
import numpy as np
import pytest

from conftest import backward_kwargs, lego_camera, pkg, render_kwargs

pytestmark = pytest.mark.gpu


def _views(gsr, scene, cams_mod, frames, size, degree):
    import torch
    out = []
    for f in frames:
        cam = lego_camera(cams_mod, frame=f, width=size, height=size)
        fkw = render_kwargs(scene, cam, degree=degree)
        img, depth, buf = gsr.render_gaussians(**fkw)
        dpix = torch.as_tensor(np.random.default_rng(100 + f).normal(0, 1, (size, size, 3)).astype(np.float32) / (size * size * 3)).cuda()
        out.append((cam, fkw, buf, dpix))
    return out


@pytest.mark.parametrize("degree", [3, 1, 0])
def test_factored_exchange_equals_summed_dense_gradients(cameras, scenes, degree):
    import torch
    gsr = pkg()
    scene = scenes.synthetic_scene(20000, 0.03, 0.6, seed=5)
    scene["shs"][::3] *= 4.0                                   # push some colours into the clamp so dL_drgb != dL_dcolor
    views = _views(gsr, scene, cameras, [0, 3, 7], 160, degree)
    dense, payloads, small = [], [], []
    for cam, fkw, buf, dpix in views:
        bkw = backward_kwargs(scene, cam, fkw, buf, dpix)
        g = gsr.backward(**bkw, sh_gradient="both")             # dense SH gradient and its payload from ONE replay
        f = gsr.backward(**bkw, sh_gradient="factored")
        n = g["dL_dmean3D"].shape[0]
        assert f["dL_dshs"] is None and f["_arena"].numel() == 11 * n and f["_view_payload"].numel() == 3 * n + 4
        tol = dict(rtol=2e-3, atol=1e-4 * float(g["dL_dmean3D"].abs().max()))   # two replays differ by float-atomic order only
        for k in ("dL_dmean3D", "dL_dscale", "dL_drot", "dL_dopacity", "dL_dcolor"):
            np.testing.assert_allclose(f[k].cpu().numpy(), g[k].cpu().numpy(), err_msg=k, **tol)
        for src in (g, f):
            pay = src["_view_payload"]
            np.testing.assert_array_equal(pay[3 * n: 3 * n + 3].cpu().numpy(), np.asarray(cam["camera_center"], np.float32))
            assert float(pay[3 * n + 3]) == 0.0
            # dL_drgb = dL_dcolor * (1 - clamped) where the Gaussian is visible (backward.py:88-92), zero elsewhere
            vis = (buf["radii"] > 0).unsqueeze(1)
            expect = torch.where(vis, src["dL_dcolor"] * (1.0 + (-1.0 * buf["clamped_state"])), torch.zeros_like(src["dL_dcolor"]))
            assert torch.equal(pay[: 3 * n].view(n, 3), expect)
        dense.append(g["dL_dshs"].clone())
        payloads.append(g["_view_payload"].clone())
        small.append(f["_arena"].clone())
    means = torch.as_tensor(scene["means"]).cuda().contiguous()
    # one view: the kernel reproduces backward()'s own SH gradient bit for bit
    one = gsr.dist.sh_gradients_from_views(means, payloads[:1], degree, average=False)
    assert torch.equal(one, dense[0])
    # three views, summed in view order and averaged -- what the all-reduce of the dense arenas gives
    got = gsr.dist.sh_gradients_from_views(means, torch.stack(payloads), degree, average=True)
    ref = ((dense[0] + dense[1]) + dense[2]) * (1.0 / 3.0)
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-6, atol=1e-12)
    assert got.shape == (n * 16, 3)
    if degree < 3:
        assert not got.view(n, 16, 3)[:, (degree + 1) ** 2:].any()   # higher coefficients: zero, as in the dense path
    # single process: exchange_factored is the identity on the payload and leaves the arena alone
    a = small[0].clone()
    gathered = gsr.dist.exchange_factored(a, payloads[0])
    assert gathered.shape == (1, 3 * n + 4) and torch.equal(a, small[0])


def test_view_exchange_error_paths(scenes):
    import torch
    gsr = pkg()
    means = torch.zeros((10, 3), device="cuda")
    good = torch.zeros(34, device="cuda")
    with pytest.raises(ValueError):
        gsr.dist.sh_gradients_from_views(means, [good[:-1]], 3)
    with pytest.raises(ValueError):
        gsr.dist.sh_gradients_from_views(means, [], 3)
    # more rows than GSR_MAX_VIEWS (16): rebuilt in chunks of 16 and summed -- 20 copies of one payload, averaged, give it back
    means_r = torch.as_tensor(np.random.default_rng(3).normal(0, 1, (10, 3)).astype(np.float32)).cuda()
    pay = torch.as_tensor(np.random.default_rng(4).normal(0, 1, 34).astype(np.float32)).cuda()
    one = gsr.dist.sh_gradients_from_views(means_r, [pay], 3, average=False)
    many = gsr.dist.sh_gradients_from_views(means_r, [pay] * 20, 3, average=True)
    np.testing.assert_allclose(many.cpu().numpy(), one.cpu().numpy(), rtol=2e-6, atol=1e-7)
    out = gsr.dist.sh_gradients_from_views(means, [good], 3)
    assert out.shape == (160, 3) and not out.any()
    with pytest.raises(ValueError):
        gsr.backward(background=np.zeros(3, np.float32), means3D=means, dL_dpixels=torch.zeros((8, 8, 3), device="cuda"), sh_gradient="sparse")


def test_overlapped_exchange_single_process(cameras, scenes):
    """backward(..., on_payload=hook) takes the payload from the blend half (gsr_backward_blend) and skips the SH outputs of
    the per-Gaussian half (gsr_backward_geom); FactoredExchange.finish then rebuilds dL_dshs.  Single process: V = 1."""
    import torch
    gsr = pkg()
    scene = scenes.synthetic_scene(30000, 0.03, 0.6, seed=6)
    scene["shs"][::2] *= 5.0
    (cam, fkw, buf, dpix), = _views(gsr, scene, cameras, [5], 192, 3)
    bkw = backward_kwargs(scene, cam, fkw, buf, dpix)
    seen = []
    ex = gsr.dist.FactoredExchange()

    def hook(payload):
        seen.append(payload)
        ex.start_gather(payload)

    g = gsr.backward(**bkw, sh_gradient="factored", on_payload=hook)
    n = g["dL_dmean3D"].shape[0]
    assert len(seen) == 1 and seen[0] is g["_view_payload"] and g["dL_dshs"] is None
    pay = g["_view_payload"]
    vis = (buf["radii"] > 0).unsqueeze(1)
    expect = torch.where(vis, g["dL_dcolor"] * (1.0 + (-1.0 * buf["clamped_state"])), torch.zeros_like(g["dL_dcolor"]))
    assert torch.equal(pay[: 3 * n].view(n, 3), expect)               # the early payload == what the geom half derives it from
    np.testing.assert_array_equal(pay[3 * n: 3 * n + 3].cpu().numpy(), np.asarray(cam["camera_center"], np.float32))
    means = torch.as_tensor(scene["means"]).cuda().contiguous()
    res = ex.finish(g, means, 3, average=True)
    assert set(res) == {"dL_dshs", "dL_dmean3D", "dL_dscale", "dL_drot", "dL_dopacity"}
    assert torch.equal(res["dL_dmean3D"], g["dL_dmean3D"])
    dense = gsr.backward(**bkw, sh_gradient="both")                     # a second replay: equal up to float-atomic order
    tol = dict(rtol=2e-3, atol=1e-4 * float(dense["dL_dshs"].abs().max()))
    np.testing.assert_allclose(res["dL_dshs"].cpu().numpy(), dense["dL_dshs"].cpu().numpy(), **tol)
    np.testing.assert_allclose(res["dL_dmean3D"].cpu().numpy(), dense["dL_dmean3D"].cpu().numpy(), rtol=2e-3,
                               atol=1e-4 * float(dense["dL_dmean3D"].abs().max()))
    # and exactly: rebuilt from THIS call's payload == the single-view kernel on the same payload
    assert torch.equal(res["dL_dshs"], gsr.dist.sh_gradients_from_views(means, [pay], 3, average=False))
