"""
Size-independent properties of the HIP path at BASELINE.json's full sizes (C3: 800x800 / 1M, C5: 1920x1080 /
5M), where the CPU oracle is too slow to be the checker:
  * binning: point_offsets[-1] == D; the non-empty ranges tile [0, D) without gaps; the list is sorted by
    (tile, depth bits, id) -- the exact order of the reference's stable 64-bit sort (quirk Q13);
  * blend: final_T in [1e-4, 1], n_contrib <= list length, image finite and in the convex hull the colours allow;
  * backward: linear in dL_dpixels (g(2x) = 2 g(x)), repeatable up to float-atomic re-association, zero for culled.
"""
import numpy as np
import pytest

from conftest import backward_kwargs, lego_camera, pkg, render_kwargs

pytestmark = pytest.mark.gpu


def _run(cfg_name):
    import torch
    gsr = pkg()
    cfg = gsr.scenes.CONFIGS[cfg_name]
    W, H = cfg["width"], cfg["height"]
    sc = gsr.scenes.synthetic_scene(cfg["n"], cfg["scale_median"], cfg["scale_sigma"], cfg["seed"])
    cam = gsr.cameras.nerf_camera(gsr.scenes.LEGO_FRAME0, W, H, gsr.scenes.LEGO_CAMERA_ANGLE_X)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    tens = {k: t(v) for k, v in sc.items()}
    bg = np.float32([0.1, 0.2, 0.3])
    fkw = dict(background=bg, means3D=tens["means"], opacity=tens["opacities"], scales=tens["scales"], rotations=tens["rotations"],
               viewmatrix=cam["world_to_camera"], projmatrix=cam["full_proj_matrix"], tan_fovx=cam["tan_fovx"], tan_fovy=cam["tan_fovy"],
               image_height=H, image_width=W, sh=tens["shs"], degree=3, campos=cam["camera_center"])
    img, depth, buf = gsr.render_gaussians(**fkw)

    def bwd(dpix):
        return gsr.backward(background=bg, means3D=tens["means"], dL_dpixels=dpix, opacity=tens["opacities"], shs=tens["shs"],
                            scales=tens["scales"], rotations=tens["rotations"], viewmatrix=fkw["viewmatrix"], projmatrix=fkw["projmatrix"],
                            tan_fovx=fkw["tan_fovx"], tan_fovy=fkw["tan_fovy"], image_height=H, image_width=W, campos=fkw["campos"],
                            radii=buf["radii"], means2D=buf["points_xy_image"], conic_opacity=buf["conic_opacity"], rgb=buf["colors"],
                            cov3Ds=buf["cov3Ds"], clamped=buf["clamped_state"], binning_buffer={"point_list": buf["point_list"]},
                            img_buffer={"ranges": buf["ranges"], "final_Ts": buf["final_Ts"], "n_contrib": buf["n_contrib"]})
    return torch, W, H, img, depth, buf, bwd


@pytest.mark.parametrize("cfg_name", ["C3", "C5"])
def test_full_size_properties(cfg_name):
    torch, W, H, img, depth, buf, bwd = _run(cfg_name)
    pl, rng, off = buf["point_list"].long(), buf["ranges"].long(), buf["point_offsets"]
    D = pl.shape[0]
    assert int(off[-1]) == D and D > 0
    # ranges: non-empty ones are contiguous and cover [0, D)
    ne = rng[rng[:, 1] > rng[:, 0]]
    assert int(ne[0, 0]) == 0 and int(ne[-1, 1]) == D and bool((ne[1:, 0] == ne[:-1, 1]).all())
    # per-tile counts equal what the rectangles say
    assert int((rng[:, 1] - rng[:, 0]).sum()) == D
    # sorted by (tile, depth bits, id)
    tile_of = torch.searchsorted(ne[:, 1].contiguous(), torch.arange(D, device=pl.device), right=True)
    dbits = buf["depths"].view(torch.int32)[pl].long()
    key = tile_of * (1 << 32) + dbits
    dk = key[1:] - key[:-1]
    assert bool((dk >= 0).all())
    ties = dk == 0
    assert bool((pl[1:][ties] > pl[:-1][ties]).all())
    # every visible Gaussian appears tiles_touched times
    counts = torch.bincount(pl, minlength=buf["radii"].shape[0])
    tt = torch.diff(off.long(), prepend=torch.zeros(1, dtype=torch.long, device=off.device))
    assert bool((counts == tt).all()) and bool(((buf["radii"] > 0) == (tt > 0)).all())
    # blend
    T, nc = buf["final_Ts"], buf["n_contrib"].long()
    assert bool(torch.isfinite(img).all()) and float(T.min()) >= 1e-4 - 1e-7 and float(T.max()) <= 1.0
    tiles_x = (W + 15) // 16
    ys, xs = torch.meshgrid(torch.arange(H, device=T.device), torch.arange(W, device=T.device), indexing="ij")
    tile_id = (ys // 16) * tiles_x + xs // 16
    assert bool((nc <= (rng[:, 1] - rng[:, 0])[tile_id]).all()) and bool(((nc == 0) == (T == 1.0)).all())
    assert float(depth.min()) >= 0.0
    # backward: linearity and repeatability
    g = torch.Generator(device="cpu").manual_seed(7)
    dpix = (torch.randn((H, W, 3), generator=g) / (H * W * 3)).to(T.device)
    a = bwd(dpix)["_arena"]
    b = bwd(dpix)["_arena"]
    c = bwd(2.0 * dpix)["_arena"]
    assert bool(torch.isfinite(a).all())
    scale = float(a.abs().max())
    assert scale > 0
    assert float((a - b).abs().max()) <= 2e-4 * scale          # float-atomic order only
    assert float((c - 2.0 * a).abs().max()) <= 4e-4 * scale
    n = buf["radii"].shape[0]
    culled = buf["radii"] <= 0
    assert float(a[: 3 * n].view(n, 3)[culled].abs().max()) == 0.0 if bool(culled.any()) else True


def test_views_in_flight_on_separate_streams(cameras, scenes):
    """Several views of one scene issued on separate HIP streams (scratch is per stream) give exactly what the same calls give
    one after the other: the integer outputs and the image bit for bit."""
    import torch
    gsr = pkg()
    sc = scenes.synthetic_scene(60000, 0.02, 0.6, seed=44)
    dev_sc = {k: torch.as_tensor(np.ascontiguousarray(v, np.float32)).cuda() for k, v in sc.items()}
    cams = [lego_camera(cameras, frame=f, width=320, height=240) for f in (0, 3, 5)]
    dpix = torch.full((240, 320, 3), 1e-6, device="cuda")

    def run(cam):
        kw = render_kwargs(dev_sc, cam, width=320, height=240)
        img, depth, buf = gsr.render_gaussians(**kw)
        g = gsr.backward(**backward_kwargs(dev_sc, cam, kw, buf, dpix))
        return img, buf, g

    seq = [run(c) for c in cams]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in cams]
    par = []
    for _ in range(3):                       # a few rounds so the streams really overlap
        par = []
        for c, s in zip(cams, streams):
            with torch.cuda.stream(s):
                par.append(run(c))
    torch.cuda.synchronize()
    for (i0, b0, g0), (i1, b1, g1) in zip(seq, par):
        assert torch.equal(i0, i1)
        for k in ("point_list", "ranges", "radii", "n_contrib"):
            assert torch.equal(b0[k], b1[k]), k
        m = float(g0["dL_dmean3D"].abs().max())
        assert float((g0["dL_dmean3D"] - g1["dL_dmean3D"]).abs().max()) <= 1e-4 * m      # float-atomic order only


def test_forward_tile_order_changes_only_the_dispatch(cameras, scenes):
    """Round 4: the forward blend dispatches its tiles heaviest first by what they cost LAST frame (costs kept in the geom workspace,
    the order made by a spare workgroup of this frame's preprocess).  Whatever the table holds -- a fresh workspace's garbage on the
    first call, another camera's costs, another image size's -- the outputs must be those of the plain row-major dispatch, bit for
    bit: the same scene is rendered from three cameras in turn, twice round, and at a second image size in between."""
    import torch
    gsr = pkg()
    sc = scenes.synthetic_scene(40000, 0.03, 0.6, seed=71)
    dev_sc = {k: torch.as_tensor(np.ascontiguousarray(v, np.float32)).cuda() for k, v in sc.items()}
    seen = {}
    for rnd in range(3):
        for f, (w, h) in ((0, (320, 240)), (3, (320, 240)), (5, (208, 176)), (6, (320, 240))):
            cam = lego_camera(cameras, frame=f, width=w, height=h)
            img, depth, buf = gsr.render_gaussians(**render_kwargs(dev_sc, cam, width=w, height=h))
            got = (img.clone(), buf["final_Ts"].clone(), buf["n_contrib"].clone(), buf["point_list"].clone(), buf["ranges"].clone())
            if (f, w) in seen:
                for a, b in zip(seen[(f, w)], got):
                    assert torch.equal(a, b), (rnd, f)
            seen[(f, w)] = got
