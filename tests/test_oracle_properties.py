"""
Oracle self-checks that need no GPU.  The reference holds no backward output, so the backward half of
the oracle is "parity unpinned"; these tests at least prove, by finite differences of the oracle's OWN
forward, that the sub-expressions which are true derivatives are transcribed correctly (SURVEY.md
quirk Q2 lists which ones are: blend-stage dL_dcolor / dL_dopacity / dL_dmean2D / dL_dconic, dL_dshs;
dL_dscale / dL_drot / the covariance part of dL_dmean3D are NOT gradients of the reference's forward
and can only be checked by literal transcription).
"""
import ctypes as C

import numpy as np

from conftest import backward_kwargs, lego_camera, render_kwargs

W, H = 64, 48


def _scene(scenes, cameras, n=40, seed=3):
    sc = scenes.synthetic_scene(n, 0.12, 0.3, seed, extent=0.8)
    sc["opacities"][:] = np.clip(sc["opacities"], 0.2, 0.6)
    cam = lego_camera(cameras, 0, W, H)
    return sc, cam


def _blend(oracle, buf, bg, xy=None, con=None, col=None):
    L = oracle.lib()
    f = lambda a: a.ctypes.data_as(oracle.f32p)
    i = lambda a: a.ctypes.data_as(oracle.i32p)
    xy = np.ascontiguousarray(buf["points_xy_image"] if xy is None else xy, np.float32)
    con = np.ascontiguousarray(buf["conic_opacity"] if con is None else con, np.float32)
    col = np.ascontiguousarray(buf["colors"] if col is None else col, np.float32)
    img = np.zeros((H, W, 3), np.float32)
    dep, fT, nc = np.zeros((H, W), np.float32), np.zeros((H, W), np.float32), np.zeros((H, W), np.int32)
    L.gsro_render_rows(C.c_int(W), C.c_int(H), C.c_int(0), C.c_int((H + 15) // 16), i(buf["ranges"]), i(buf["point_list"]), f(xy),
                       f(col), f(con), f(buf["depths"]), f(bg), f(img), f(dep), f(fT), i(nc))
    return img.astype(np.float64)


def test_blend_stage_gradients_by_finite_differences(oracle, cameras, scenes):
    sc, cam = _scene(scenes, cameras)
    kw = render_kwargs(sc, cam, width=W, height=H, bg=(0.3, 0.2, 0.1))
    img, dep, buf = oracle.render_gaussians(**kw)
    rng = np.random.default_rng(0)
    dpix = rng.normal(0, 1, (H, W, 3)).astype(np.float32)
    g = oracle.backward(**backward_kwargs(sc, cam, kw, buf, dpix))
    bg = kw["background"]
    loss = lambda **k: float((_blend(oracle, buf, bg, **k) * dpix).sum())
    vis = np.where(buf["radii"] > 0)[0]
    assert len(vis) >= 10
    checked = 0
    for gid in vis[:12]:
        # colour (linear -> tight), opacity, conic a/c, conic b (reference stores HALF the derivative), mean2D (x 0.5*W)
        cases = [("colors", 1, 1e-2, g["dL_dcolor"][gid, 1], 1.0, 2e-3)]
        cases += [("op", 3, 2e-3, g["dL_dopacity"][gid], 1.0, 3e-2)]
        cases += [("con", 0, 1e-4, g["dL_dconic"][gid, 0], 1.0, 5e-2), ("con", 2, 1e-4, g["dL_dconic"][gid, 3], 1.0, 5e-2)]
        cases += [("con", 1, 1e-4, g["dL_dconic"][gid, 1], 0.5, 5e-2)]
        cases += [("xy", 0, 2e-3, g["dL_dmean2D"][gid, 0], 0.5 * W, 5e-2), ("xy", 1, 2e-3, g["dL_dmean2D"][gid, 1], 0.5 * H, 5e-2)]
        for what, comp, eps, analytic, factor, tol in cases:
            def perturbed(sign):
                if what == "colors":
                    a = buf["colors"].copy(); a[gid, comp] += sign * eps; return loss(col=a)
                if what in ("op", "con"):
                    a = buf["conic_opacity"].copy(); a[gid, comp] += sign * eps; return loss(con=a)
                a = buf["points_xy_image"].copy(); a[gid, comp] += sign * eps; return loss(xy=a)
            # the forward is discontinuous where alpha crosses 1/255, so a finite difference is polluted
            # whenever a pixel crosses that ring inside +-eps: accept the step size that avoids it
            fds = []
            for shrink in (1.0, 0.25, 0.0625):
                eps_k = eps * shrink
                fds.append((perturbed(+shrink) - perturbed(-shrink)) / (2 * eps_k) * factor)
            ok = [abs(fd - analytic) <= tol * max(abs(fd), abs(analytic), 1e-3) + 1e-2 for fd in fds]
            assert any(ok), (gid, what, comp, fds, analytic)
            checked += 1
    assert checked >= 70
    assert np.all(g["dL_dmean2D"][:, 2] == 0) and np.all(g["dL_dconic"][:, 2] == 0) and np.all(g["dL_dcov3D"] == 0)


def test_sh_gradient_is_the_basis(oracle, cameras, scenes):
    """dL_dshs[k] = basis_k(dir) * dL_dcolor masked by the clamp flag: check against an FD of the oracle's preprocess colour."""
    sc, cam = _scene(scenes, cameras, n=30, seed=5)
    kw = render_kwargs(sc, cam, width=W, height=H)
    _, _, buf = oracle.render_gaussians(**kw)
    dpix = np.random.default_rng(1).normal(0, 1, (H, W, 3)).astype(np.float32)
    g = oracle.backward(**backward_kwargs(sc, cam, kw, buf, dpix))
    vis = np.where((buf["radii"] > 0) & (buf["clamped_state"].sum(1) == 0))[0]
    for gid in vis[:5]:
        for k in (0, 2, 7, 13):
            sh2 = sc["shs"].copy()
            sh2[gid, k, 1] += 0.25
            kw2 = dict(kw); kw2["sh"] = sh2
            _, _, b2 = oracle.render_gaussians(**kw2)
            basis = (b2["colors"][gid, 1] - buf["colors"][gid, 1]) / 0.25
            np.testing.assert_allclose(g["dL_dshs"][gid * 16 + k, 1], basis * g["dL_dcolor"][gid, 1], rtol=2e-3, atol=1e-6)


def test_sort_is_stable_and_ranges_cover_the_list(oracle, cameras, scenes):
    sc = scenes.synthetic_scene(200, 0.1, 0.4, 9)
    sc = {k: np.concatenate([v, v]) for k, v in sc.items()}           # exact depth ties
    cam = lego_camera(cameras, 1, W, H)
    _, _, buf = oracle.render_gaussians(**render_kwargs(sc, cam, width=W, height=H), keep_keys=True)
    keys, pl, r = buf["_keys"], buf["point_list"], buf["ranges"]
    assert np.all(np.diff(keys) >= 0)
    ties = np.where(np.diff(keys) == 0)[0]
    assert len(ties) > 0 and np.all(pl[ties] < pl[ties + 1])           # equal keys keep ascending id (quirk Q13)
    assert int(buf["point_offsets"][-1]) == len(pl) == int(buf["_tiles_touched"].sum())
    nz = r[(r[:, 1] - r[:, 0]) > 0]
    assert nz[0, 0] == 0 and nz[-1, 1] == len(pl) and np.all(nz[1:, 0] == nz[:-1, 1])
    tiles = (keys >> 32).astype(np.int64)
    for t in np.unique(tiles)[:20]:
        assert r[t, 0] == np.searchsorted(tiles, t, "left") and r[t, 1] == np.searchsorted(tiles, t, "right")


def test_zero_rendered_gives_zero_image(oracle, cameras, scenes):
    sc = scenes.synthetic_scene(20, 0.05, 0.5, 2)
    cam = lego_camera(cameras, 0, W, H)
    sc["means"] = (sc["means"] * 0.01 + np.asarray(cam["camera_center"], np.float32) * 3.0).astype(np.float32)
    img, dep, buf = oracle.render_gaussians(**render_kwargs(sc, cam, width=W, height=H, bg=(1, 1, 1)))
    assert buf["point_list"].shape == (0,) and img.max() == 0.0 and buf["final_Ts"].max() == 0.0   # quirk Q10


def test_ssim_and_depth_loss_oracle_properties(oracle):
    """loss.py's two evaluation helpers (restated with the reference's window-weight quirk Q21): identities and bounds."""
    rng = np.random.default_rng(12)
    a = rng.uniform(0, 1, (23, 31, 3)).astype(np.float32)
    b = rng.uniform(0, 1, (23, 31, 3)).astype(np.float32)
    assert abs(oracle.ssim(a, a) - 1.0) < 1e-6
    assert abs(oracle.ssim(a, b) - oracle.ssim(b, a)) < 1e-7           # symmetric
    assert oracle.ssim(a, b) < 0.2 < 0.99 < oracle.ssim(a, np.clip(a + 0.01, 0, 1).astype(np.float32))
    const = np.full((9, 9, 3), 0.5, np.float32)
    assert abs(oracle.ssim(const, const) - 1.0) < 1e-6                  # zero variance: c2/c2
    d1, d2 = a[..., 0], b[..., 0]
    ones = np.ones_like(d1)
    assert abs(oracle.depth_loss(d1, d2, ones) - float(np.abs(d1 - d2).mean())) < 1e-6
    assert oracle.depth_loss(d1, d2, np.zeros_like(d1)) == 0.0
    half = ones.copy(); half[:, 16:] = 0
    assert abs(oracle.depth_loss(d1, d2, half) - float((np.abs(d1 - d2) * half).mean())) < 1e-6


def test_f64_accumulating_checker_agrees_with_the_restatement(oracle, cameras, scenes):
    """oracle.backward(accumulate="f64") (the second checker of tests/test_gpu_fuzz.py's needle cases: same float32 terms,
    per-Gaussian sums kept in float64) differs from the literal restatement only by float32 accumulation rounding: on a
    well-conditioned scene the two agree to a few float32 ulps of the sum, on every gradient array."""
    sc, cam = _scene(scenes, cameras, n=400, seed=9)
    kw = render_kwargs(sc, cam, width=W, height=H, bg=(0.3, 0.2, 0.1))
    img, dep, buf = oracle.render_gaussians(**kw)
    dpix = np.random.default_rng(1).normal(0, 1, (H, W, 3)).astype(np.float32)
    bkw = backward_kwargs(sc, cam, kw, buf, dpix)
    a, b = oracle.backward(**bkw), oracle.backward(**bkw, accumulate="f64")
    assert np.abs(a["dL_dcolor"]).max() > 0
    for k in ("dL_dcolor", "dL_dopacity", "dL_dmean2D", "dL_dconic", "dL_dmean3D", "dL_dscale", "dL_drot", "dL_dshs"):
        m = np.abs(b[k]).max()
        assert np.abs(a[k].astype(np.float64) - b[k]).max() <= 2e-5 * m, k
    assert not b["dL_dmean2D"][:, 2].any() and not b["dL_dconic"][:, 2].any()
