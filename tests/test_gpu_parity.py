"""
GPU parity tests proper: the HIP library, called through the C ABI via the reference-shaped Python
surface, against the CPU oracle on the same seeded inputs.  Tolerances are stated in tests/parity.py.
Run on the GPU box with `pytest -m gpu`.
"""
import numpy as np
import pytest

from conftest import backward_kwargs, lego_camera, pkg, render_kwargs
import parity

pytestmark = pytest.mark.gpu


def _both(oracle, kw):
    gsr = pkg()
    got = gsr.render_gaussians(**kw)
    ref = oracle.render_gaussians(**kw)
    return gsr, got, ref


def _pixel_grad(H, W, seed=99):
    rng = np.random.default_rng(seed)
    return (rng.normal(0.0, 1.0, (H, W, 3)) / (H * W * 3)).astype(np.float32)


def _fwd_bwd(oracle, scene, cam, W, H, degree=3, bg=(0.0, 0.0, 0.0), train_convention=True, report=None):
    kw = render_kwargs(scene, cam, width=W, height=H, degree=degree, train_convention=train_convention, bg=bg)
    gsr, got, ref = _both(oracle, kw)
    parity.compare_forward(got, ref, report)
    dpix = _pixel_grad(H, W)
    # both backward passes start from the ORACLE's forward buffers, so a forward threshold flip cannot
    # leak into the gradient comparison
    bkw = backward_kwargs(scene, cam, kw, ref[2], dpix)
    g_got = gsr.backward(**bkw)
    g_ref = oracle.backward(**bkw)
    parity.compare_backward(g_got, g_ref, report)
    # and the product's own forward -> backward chain must agree with it too
    bkw2 = backward_kwargs(scene, cam, kw, got[2], dpix)
    g_chain = gsr.backward(**bkw2)
    parity.compare_backward(g_chain, g_ref, None)
    return got, ref, g_got, g_ref


def test_toy_scene_forward(oracle, cameras, scenes):
    """BASELINE config #1: the reference's 3-Gaussian demo at 1800x1800 (render.py)."""
    cam, sc = cameras.toy_camera(), scenes.toy_scene()
    kw = render_kwargs(sc, cam, train_convention=False)
    kw["colors"] = sc["colors"]
    _, got, ref = _both(oracle, kw)
    parity.compare_forward(got, ref)
    np.testing.assert_array_equal(parity.to_np(got[2]["radii"]), [542, 485, 542])


def test_toy_scene_matches_reference_png(cameras, scenes):
    """The HIP render of the reference's demo scene against the reference's own assets/example_render.png
    (tests/golden/example_render.png), with no oracle in between: <= 2/255 after the same clip + resample."""
    import os
    from PIL import Image
    from conftest import ROOT
    gsr = pkg()
    cam, sc = cameras.toy_camera(), scenes.toy_scene()
    kw = render_kwargs(sc, cam, train_convention=False)
    img = gsr.render_gaussians(**kw)[0].cpu().numpy()
    png = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "example_render.png")).convert("RGB"))
    crop = png[15:1170, 15:1170].astype(np.float32) / 255.0
    clipped = np.clip(img, 0.0, 1.0)
    chans = [np.asarray(Image.fromarray(clipped[:, :, c]).resize((1155, 1155), Image.BOX)) for c in range(3)]
    err = np.abs(np.round(np.stack(chans, -1) * 255.0) / 255.0 - crop) * 255.0
    assert err.max() <= 2.0 and err.mean() <= 0.5, (err.max(), err.mean())


def test_toy_scene_backward(oracle, cameras, scenes):
    """Same scene through backward(): render.py's view-matrix convention makes view[j][3] non-zero (quirk Q3)."""
    cam, sc = cameras.toy_camera(image_width=320, image_height=240), scenes.toy_scene()
    _fwd_bwd(oracle, sc, cam, 320, 240, train_convention=False, bg=(0.2, 0.1, 0.3))


@pytest.mark.parametrize("W,H,n,seed", [(160, 120, 2000, 1), (200, 136, 5000, 2), (97, 61, 700, 3)])
def test_small_random_scenes(oracle, cameras, scenes, W, H, n, seed):
    """Ragged image sizes (W, H not multiples of 16), non-black background."""
    sc = scenes.synthetic_scene(n, 0.05, 0.6, seed)
    cam = lego_camera(cameras, frame=seed % 8, width=W, height=H)
    _fwd_bwd(oracle, sc, cam, W, H, bg=(0.3, 0.5, 0.7))


@pytest.mark.parametrize("degree", [0, 1, 2])
def test_lower_sh_degrees(oracle, cameras, scenes, degree):
    sc = scenes.synthetic_scene(1500, 0.05, 0.5, 10 + degree)
    cam = lego_camera(cameras, frame=1, width=128, height=96)
    _fwd_bwd(oracle, sc, cam, 128, 96, degree=degree)


def test_c2_lego_100k(oracle, cameras, scenes):
    """BASELINE config #2: 800x800, 100k Gaussians, Lego train pose 0, forward + backward."""
    cfg = scenes.CONFIGS["C2"]
    sc = scenes.synthetic_scene(cfg["n"], cfg["scale_median"], cfg["scale_sigma"], cfg["seed"])
    cam = lego_camera(cameras, frame=0, width=800, height=800)
    report = {}
    _fwd_bwd(oracle, sc, cam, 800, 800, report=report)
    parity.assert_tripwires("C2", report)
    print("\nC2 measured margins (image: frac within 2e-5, max err, flips; gradients: frac within 1e-4 max|g| + 1e-3 |g|, max err / max|g|):")
    print(parity.format_report(report))


@pytest.mark.parametrize("name", ["C3", "C5"])
def test_full_size_configs(oracle, cameras, scenes, name):
    """BASELINE configs #3 (800x800, 1 M Gaussians: the headline workload) and #5 (1920x1080, 5 M Gaussians, 64-bit tile
    items, large-n radix path) against the oracle at FULL size -- the single-thread C oracle needs about 4 s / 25 s for them."""
    cfg = scenes.CONFIGS[name]
    sc = scenes.synthetic_scene(cfg["n"], cfg["scale_median"], cfg["scale_sigma"], cfg["seed"])
    cam = cameras.nerf_camera(scenes.LEGO_FRAME0, cfg["width"], cfg["height"], scenes.LEGO_CAMERA_ANGLE_X)
    report = {}
    _fwd_bwd(oracle, sc, cam, cfg["width"], cfg["height"], report=report)
    parity.assert_tripwires(name, report)
    print(f"\n{name} measured margins (image: frac within 2e-5, max err, flips; gradients: frac within 1e-4 max|g| + 1e-3 |g|, max err / max|g|):")
    print(parity.format_report(report))


def test_empty_and_culled(oracle, cameras, scenes):
    gsr = pkg()
    cam = lego_camera(cameras, frame=0, width=64, height=48)
    # every Gaussian behind the camera: D == 0 -> zeros, not background (quirk Q10)
    sc = scenes.synthetic_scene(100, 0.05, 0.5, 7)
    sc["means"] = (sc["means"] * 0.01 + np.asarray(cam["camera_center"], np.float32) * 3.0).astype(np.float32)
    kw = render_kwargs(sc, cam, bg=(0.4, 0.4, 0.4))
    got, ref = gsr.render_gaussians(**kw), oracle.render_gaussians(**kw)
    assert parity.to_np(got[2]["point_list"]).shape == (0,)
    parity.compare_forward(got, ref)
    assert float(parity.to_np(got[0]).max()) == 0.0
    # N == 0: undefined in the reference; the build returns zeros and empty buffers
    sc0 = {k: v[:0] for k, v in sc.items()}
    img, depth, buf = gsr.render_gaussians(**render_kwargs(sc0, cam))
    assert tuple(img.shape) == (48, 64, 3) and float(parity.to_np(img).max()) == 0.0
    assert parity.to_np(buf["radii"]).shape == (0,)


def test_depth_ties_keep_id_order(oracle, cameras, scenes):
    """Duplicated Gaussians have bit-identical depths: the list must keep ascending id order (quirk Q13)."""
    sc = scenes.synthetic_scene(300, 0.08, 0.4, 21)
    sc = {k: np.concatenate([v, v, v]) for k, v in sc.items()}
    cam = lego_camera(cameras, frame=3, width=96, height=80)
    kw = render_kwargs(sc, cam)
    _, got, ref = _both(oracle, kw)
    parity.compare_forward(got, ref)


@pytest.mark.parametrize("near,far,n", [(0.25, 90.0, 4000), (3.0, 3.0, 700), (2.0, 7.9, 3000), (1.9, 2.1, 1500), (50.0, 60.0, 300)])
def test_depth_ranges_and_the_device_side_pass_plan(oracle, cameras, scenes, near, far, n):
    """The depth sort runs as many 8-bit passes as the frame's depth RANGE needs (decided on the device from the visible
    minimum / maximum: keys are bits - min).  Scenes spanning nine octaves (four passes), a single depth plane (one pass, all
    ties -> id order), ranges straddling a power of two (the exponent bit flips inside the range), and far, thin ranges; some
    Gaussians behind the camera (culled: they sort last) -- point_list must be exact every time."""
    sc = scenes.synthetic_scene(n, 0.03, 0.5, int(near * 100) + n)
    rng = np.random.default_rng(n)
    cam = cameras.nerf_camera(np.eye(4).tolist(), 208, 144, 0.6911112)  # at the origin, looking down -z (NeRF pose convention)
    depth = rng.uniform(near, far, n).astype(np.float32)
    depth[::7] = -depth[::7]                                            # behind the camera: culled
    sc["means"][:, 2] = -depth
    sc["means"][:, :2] = (rng.uniform(-0.4, 0.4, (n, 2)) * np.abs(depth)[:, None]).astype(np.float32)
    sc["scales"] *= np.abs(depth)[:, None] / 3.0                       # similar footprint at every depth
    kw = render_kwargs(sc, cam, width=208, height=144)
    gsr, got, ref = _both(oracle, kw)
    parity.compare_forward(got, ref)
    d = parity.to_np(got[2]["depths"])
    assert (d > 0).sum() > n // 2 and abs(float(d[d > 0].min()) - near) < 0.2 * near + 0.5 and int(parity.to_np(got[2]["point_list"]).size) > n


def test_depth_pass_guess_misses_and_overshoots(oracle, cameras, scenes):
    """The host launches as many depth-sort passes as the PREVIOUS frame in the same workspace needed; the device decides how
    many this frame needs.  A sequence of frames through one workspace whose need goes 1 -> 4 (guess too low: the launched
    passes must leave the data alone and the full sort is launched after the readback) -> 3 -> 1 (guess too high: the extra
    launches exit) -> 4 must give the exact list every time.  N is above the single-workgroup small-scene path."""
    n = 9000
    cam = cameras.nerf_camera(np.eye(4).tolist(), 208, 144, 0.6911112)
    for k, (near, far) in enumerate([(3.0, 3.0), (0.25, 90.0), (2.0, 7.9), (5.0, 5.0), (0.5, 200.0), (0.5, 200.0)]):
        sc = scenes.synthetic_scene(n, 0.02, 0.5, 900 + k)
        rng = np.random.default_rng(k)
        depth = rng.uniform(near, far, n).astype(np.float32)
        depth[::5] = -depth[::5]
        sc["means"][:, 2] = -depth
        sc["means"][:, :2] = (rng.uniform(-0.4, 0.4, (n, 2)) * np.abs(depth)[:, None]).astype(np.float32)
        sc["scales"] *= np.abs(depth)[:, None] / 3.0
        kw = render_kwargs(sc, cam, width=208, height=144)
        _, got, ref = _both(oracle, kw)
        parity.compare_forward(got, ref)
        assert int(parity.to_np(got[2]["point_list"]).size) > n


@pytest.mark.parametrize("case", ["specks_and_giants", "giants_only", "mostly_culled"])
def test_expansion_by_output_block(oracle, cameras, scenes, case):
    """The expansion hands every workgroup one radix block of the OUTPUT (1024 items at these sizes) and lets it find the
    Gaussians that own those items: thousands of one-tile specks per block (many 256-Gaussian batches per workgroup), giants
    that cover every tile (one Gaussian spread over many workgroups; blocks that begin and end inside one Gaussian; D an exact
    multiple of the block when all of them cover all 256 tiles), and a scene whose Gaussians are mostly behind the camera
    (the sorted arrays are only valid up to the visible count).  N is not a multiple of 256."""
    n = {"specks_and_giants": 20_003, "giants_only": 12, "mostly_culled": 9_001}[case]
    sc = scenes.synthetic_scene(n, 0.004, 0.2, 77 + n)
    cam = lego_camera(cameras, frame=1, width=256, height=256)
    if case == "specks_and_giants":
        sc["scales"][::997] = 6.0                        # 21 giants among the specks
    elif case == "giants_only":
        sc["scales"][:] = 8.0
        sc["means"] *= 0.2
    else:
        centre = np.asarray(cam["camera_center"], np.float32)
        away = sc["means"] - centre                      # mirror four of five through the camera: behind it
        keep = np.arange(n) % 5 == 0
        sc["means"][~keep] = centre - away[~keep]
    kw = render_kwargs(sc, cam, width=256, height=256)
    _, got, ref = _both(oracle, kw)
    parity.compare_forward(got, ref)
    D = int(parity.to_np(got[2]["point_list"]).size)
    if case == "giants_only":
        assert D == n * 256                              # every Gaussian covers the whole 16 x 16 tile grid
    if case == "mostly_culled":
        assert (parity.to_np(got[2]["radii"]) > 0).sum() < n // 3 and D > 0


def test_huge_gaussian_covers_all_tiles(oracle, cameras, scenes):
    sc = scenes.synthetic_scene(50, 0.05, 0.5, 33)
    sc["scales"][0] = 5.0
    sc["means"][0] = 0.0
    cam = lego_camera(cameras, frame=2, width=256, height=192)
    got, ref, _, _ = _fwd_bwd(oracle, sc, cam, 256, 192)
    assert int(parity.to_np(got[2]["point_offsets"])[0]) == 16 * 12


def test_torch_inputs_stay_on_device(oracle, cameras, scenes):
    """Device-resident torch tensors are consumed in place (no host round trip) and give the same answer."""
    import torch
    gsr = pkg()
    sc = scenes.synthetic_scene(3000, 0.05, 0.6, 5)
    cam = lego_camera(cameras, frame=4, width=160, height=160)
    kw = render_kwargs(sc, cam)
    ref = gsr.render_gaussians(**kw)
    kw2 = dict(kw)
    for k_np, k_kw in [("means", "means3D"), ("opacities", "opacity"), ("scales", "scales"), ("rotations", "rotations"), ("shs", "sh")]:
        kw2[k_kw] = torch.as_tensor(sc[k_np]).cuda()
    got = gsr.render_gaussians(**kw2)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[2]["point_list"], ref[2]["point_list"])


def test_camera_arrays_written_in_place_between_calls(oracle, cameras, scenes):
    """The host keeps packed camera structs by the BYTES of the arrays they were made from (numpy arrays carry no version counter):
    a caller who moves its camera by writing into the same arrays, or swaps the background, gets the new view -- forward and backward
    against the oracle fed the same arrays at each point."""
    gsr = pkg()
    W = H = 96
    sc = scenes.synthetic_scene(1500, 0.06, 0.5, 21)
    cam = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in lego_camera(cameras, frame=0, width=W, height=H).items()}
    other = lego_camera(cameras, frame=5, width=W, height=H)
    kw = render_kwargs(sc, cam, width=W, height=H, bg=(0.0, 0.0, 0.0))
    dpix = _pixel_grad(H, W)
    first = gsr.render_gaussians(**kw)[0].cpu().numpy()
    for step in range(2):
        if step == 0:      # the same array objects, new contents
            for k in ("world_to_camera", "full_proj_matrix", "camera_center"):
                cam[k][...] = other[k]
        else:              # and a background written in place
            kw["background"][...] = (0.2, 0.5, 0.8)
        got = gsr.render_gaussians(**kw)
        ref = oracle.render_gaussians(**kw)
        parity.compare_forward(got, ref, None)
        assert np.abs(got[0].cpu().numpy() - first).max() > 1e-2      # it IS another picture
        bkw = backward_kwargs(sc, cam, kw, ref[2], dpix)
        parity.compare_backward(gsr.backward(**bkw), oracle.backward(**bkw), None)


def test_workspace_and_capacity_errors(cameras, scenes):
    """Too-small scratch buffers are refused with GSR_E_WORKSPACE (mapped to RuntimeError), not overrun."""
    import ctypes as C
    import torch
    from conftest import sub
    gsr = pkg()
    _lib, _host = sub("_lib"), sub("_host")
    L = _lib.lib()
    sc = scenes.synthetic_scene(500, 0.05, 0.5, 3)
    cam = lego_camera(cameras, 0, 64, 64)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
    means, scl, rot, op, sh = t(sc["means"]), t(sc["scales"]), t(sc["rotations"]), t(sc["opacities"].reshape(-1)), t(sc["shs"].reshape(-1, 3))
    scene = _lib.GsrScene(500, _host.ptr(means), _host.ptr(scl), _host.ptr(rot), _host.ptr(op), _host.ptr(sh), 3, 1.0, 1)
    camera = _host.make_camera(cam["world_to_camera"], cam["full_proj_matrix"], cam["camera_center"], [0, 0, 0], cam["tan_fovx"], cam["tan_fovy"], 64, 64)
    bufs = [torch.empty(500 * 8, dtype=torch.float32, device=dev) for _ in range(9)]
    geom = _lib.GsrGeom(*[_host.ptr(b) for b in bufs], None)
    small = torch.empty(1024, dtype=torch.uint8, device=dev)
    D = C.c_int64(0)
    rc = L.gsr_forward_count(C.byref(scene), C.byref(camera), C.byref(geom), _host.ptr(small), small.numel(), C.byref(D), None)
    assert rc == _lib.GSR_E_WORKSPACE
    with pytest.raises(RuntimeError):
        _lib.check(rc)
    with pytest.raises(ValueError):
        gsr.render_gaussians(**dict(render_kwargs(sc, cam), sh=sc["shs"][:, :4]))   # not 16 coefficients per Gaussian


def test_more_than_2_30_pairs_is_refused_on_the_device_path(cameras, scenes):
    """Reference forward.py:765-767: `if num_rendered > (1 << 30): raise ValueError`.  Here the count really comes out of the
    device scan: 140 000 Gaussians of scale 50 at 1920x1080 each cover all 120 x 68 tiles, D = 1 142 400 000 > 2^30 (and
    < 2^31, so the int32 scan itself does not wrap).  gsr_forward_count must return GSR_E_OVERFLOW -> ValueError before any
    D-sized buffer is allocated or written; a normal call afterwards works."""
    import torch
    gsr = pkg()
    n = 140_000
    sc = scenes.synthetic_scene(n, 50.0, 0.0, seed=8, extent=0.5)
    cam = cameras.nerf_camera(scenes.LEGO_FRAME0, 1920, 1080, scenes.LEGO_CAMERA_ANGLE_X)
    before = torch.cuda.memory_allocated()
    with pytest.raises(ValueError, match="2\\^30"):
        gsr.render_gaussians(**render_kwargs(sc, cam))
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() - before < (1 << 30)      # nothing of size D (4.6 GB of point_list alone) was allocated
    # one Gaussian fewer than the limit allows is accepted by the count: 131 000 x 8160 = 1 068 960 000 <= 2^30
    small = scenes.synthetic_scene(2000, 0.05, 0.5, seed=9)
    img = gsr.render_gaussians(**render_kwargs(small, lego_camera(cameras, 0, 128, 96)))[0]
    assert torch.isfinite(img).all()


def test_offset_views_are_accepted(oracle, cameras, scenes):
    """contiguous but unaligned device views (means[1:], scales[1:] -- 12-byte rows) are what a caller slicing its
    parameter tensors passes; the reference copies every input and accepts them.  They must give the answer of a fresh copy."""
    import torch
    gsr = pkg()
    sc = scenes.synthetic_scene(1201, 0.05, 0.6, 21)
    cam = lego_camera(cameras, frame=3, width=144, height=112)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    full = {k: dev(sc[k]) for k in ("means", "opacities", "scales", "rotations", "shs")}
    cut = {k: v[1:] for k, v in full.items()}
    assert cut["means"].data_ptr() % 16 != 0
    sub_sc = {k: sc[k][1:] for k in full}
    kw_ref = render_kwargs(sub_sc, cam, width=144, height=112)
    kw = dict(kw_ref, means3D=cut["means"], opacity=cut["opacities"], scales=cut["scales"], rotations=cut["rotations"], sh=cut["shs"].reshape(-1, 3))
    got, want = gsr.render_gaussians(**kw), gsr.render_gaussians(**kw_ref)
    assert torch.equal(got[0], want[0]) and torch.equal(got[2]["point_list"], want[2]["point_list"])
    dpix = dev(_pixel_grad(112, 144))
    off_dpix = torch.cat([dpix.reshape(-1)[:1], dpix.reshape(-1)])[1:].reshape(112, 144, 3)   # a 4-byte-offset view of the same values
    bkw = backward_kwargs(sub_sc, cam, kw_ref, got[2], off_dpix)
    assert bkw["dL_dpixels"].data_ptr() % 16 != 0
    bkw.update(means3D=cut["means"], opacity=cut["opacities"], scales=cut["scales"], rotations=cut["rotations"], shs=cut["shs"].reshape(-1, 3))
    g = gsr.backward(**bkw)
    g_ref = oracle.backward(**backward_kwargs(sub_sc, cam, kw_ref, oracle.render_gaussians(**kw_ref)[2], dpix.cpu().numpy()))
    parity.compare_backward(g, g_ref)


def test_forward_masks_are_reused_only_with_the_forwards_own_buffers(oracle, cameras, scenes):
    """the forward's block masks describe ITS records up to ITS n_contrib.  backward() takes the mask-driven
    compaction only when point_list, ranges, n_contrib, final_Ts, means2D and conic_opacity are that forward's own tensors; with
    any of them replaced (here: equal-valued copies, and the oracle's buffers) it must fall back to the self-contained block
    test -- and either way agree with the oracle."""
    gsr = pkg()
    from conftest import sub
    bwd = sub("backward").backward
    sc = scenes.synthetic_scene(4000, 0.05, 0.6, 31)
    cam = lego_camera(cameras, frame=5, width=176, height=128)
    kw = render_kwargs(sc, cam, width=176, height=128)
    got, ref = gsr.render_gaussians(**kw), oracle.render_gaussians(**kw)
    dpix = _pixel_grad(128, 176)
    g_ref = oracle.backward(**backward_kwargs(sc, cam, kw, ref[2], dpix))
    own = backward_kwargs(sc, cam, kw, got[2], dpix)
    parity.compare_backward(gsr.backward(**own), g_ref)
    assert bwd.last_call_used_forward_masks
    for key in ("means2D", "conic_opacity"):
        mixed = dict(own)
        mixed[key] = own[key].clone()
        mixed["geom_buffer"] = dict(own["geom_buffer"], **{key: mixed[key]})
        parity.compare_backward(gsr.backward(**mixed), g_ref)
        assert not bwd.last_call_used_forward_masks, key
    for key in ("ranges", "n_contrib", "final_Ts"):
        mixed = dict(own, img_buffer=dict(own["img_buffer"], **{key: own["img_buffer"][key].clone()}))
        parity.compare_backward(gsr.backward(**mixed), g_ref)
        assert not bwd.last_call_used_forward_masks, key
    parity.compare_backward(gsr.backward(**backward_kwargs(sc, cam, kw, ref[2], dpix)), g_ref)   # the oracle's buffers (numpy)
    assert not bwd.last_call_used_forward_masks


@pytest.mark.parametrize("degree", [0, 1, 2, 3])
def test_sh_direction_sums_handed_from_forward_to_backward(oracle, cameras, scenes, degree):
    """GsrGeom.sh_dir_grad: with device-resident parameter tensors the forward leaves d(colour)/d(direction) (nine floats per
    Gaussian) for the backward, whose per-Gaussian kernel then skips the 192-byte SH rows.  Same float operations either way
    (sh_stage.h sh_direction_sums), so the two paths must agree far inside the parity tolerance (the blend half's float atomics
    are the only run-to-run difference), and each with the oracle; a caller that swaps any of the tensors the sums were formed
    from (sh, means, camera position, degree) must get the self-contained path."""
    import torch
    gsr = pkg()
    from conftest import sub
    bwd = sub("backward").backward
    sc = scenes.synthetic_scene(5000, 0.05, 0.6, 41 + degree)
    sc["shs"][::3] *= 4.0                       # some colours clamp
    cam = lego_camera(cameras, frame=degree, width=208, height=160)
    kw_np = render_kwargs(sc, cam, width=208, height=160, degree=degree)
    dev = {k: torch.as_tensor(np.ascontiguousarray(sc[k])).cuda() for k in ("means", "opacities", "scales", "rotations")}
    dev["shs"] = torch.as_tensor(np.ascontiguousarray(sc["shs"])).cuda().reshape(-1, 3)
    kw = dict(kw_np, means3D=dev["means"], opacity=dev["opacities"], scales=dev["scales"], rotations=dev["rotations"], sh=dev["shs"])
    img, _, buf = gsr.render_gaussians(**kw)
    dpix = torch.as_tensor(_pixel_grad(160, 208)).cuda()
    own = backward_kwargs(sc, cam, kw, buf, dpix)
    own.update(means3D=dev["means"], opacity=dev["opacities"], scales=dev["scales"], rotations=dev["rotations"], shs=dev["shs"])
    g_fast = gsr.backward(**own)
    assert bwd.last_call_used_forward_sh_dir
    ref = oracle.render_gaussians(**kw_np)
    g_ref = oracle.backward(**backward_kwargs(sc, cam, kw_np, ref[2], dpix.cpu().numpy()))
    parity.compare_backward(g_fast, g_ref)
    for swap in ("shs", "means3D", "campos", "degree", "clamped"):
        other = dict(own)
        if swap == "campos":
            other["campos"] = np.asarray(own["campos"], np.float64) + 0.0    # equal values in a new array: still the fast path
        elif swap == "degree":
            if degree == 3:
                continue
            other["degree"] = degree + 1
        elif swap == "clamped":
            other["clamped"] = own["clamped"].clone()
            other["geom_buffer"] = dict(own["geom_buffer"], clamped_state=other["clamped"])
        else:
            other[swap] = own[swap].clone()
        g_slow = gsr.backward(**other)
        assert bwd.last_call_used_forward_sh_dir == (swap == "campos"), swap
        if swap != "degree":
            for k in ("dL_dmean3D", "dL_dshs", "dL_dscale", "dL_drot"):
                a, b = parity.to_np(g_fast[k]), parity.to_np(g_slow[k])
                np.testing.assert_allclose(a, b, rtol=1e-4, atol=2e-5 * float(np.abs(b).max()), err_msg=f"{swap} {k}")


def test_sigma3d_recomputed_only_for_the_forwards_own_array(oracle, cameras, scenes):
    """backward() does not read cov3Ds back when it is the forward's own tensor, unwritten, for the very scales / rotations tensors
    (unwritten too) and scale_modifier of that forward: the kernel recomputes Sigma3D with the forward's instructions (sigma3d.h: one
    definition for both kernels).  Same gradients as the path that reads the array; any other cov3Ds -- a copy, a written one, one behind written scales -- is read,
    as the reference reads it (backward.py:975-980)."""
    import torch
    gsr = pkg()
    bwd = __import__("importlib").import_module(gsr.__name__ + ".backward").backward
    W, H = 208, 160
    sc = scenes.synthetic_scene(6000, 0.05, 0.6, 77)
    cam = lego_camera(cameras, frame=3, width=W, height=H)
    kw = render_kwargs(sc, cam, width=W, height=H)
    kw["scale_modifier"] = 1.3
    dev = {k_kw: torch.as_tensor(sc[k_np]).cuda() for k_np, k_kw in
           [("means", "means3D"), ("opacities", "opacity"), ("scales", "scales"), ("rotations", "rotations"), ("shs", "sh")]}
    kw.update(dev)
    dpix = _pixel_grad(H, W)
    buf = gsr.render_gaussians(**kw)[2]
    scene_t = {"means": dev["means3D"], "opacities": dev["opacity"], "shs": dev["sh"], "scales": dev["scales"], "rotations": dev["rotations"]}
    bkw = backward_kwargs(scene_t, cam, kw, buf, dpix)
    g1 = gsr.backward(**bkw)
    assert bwd.last_call_recomputed_sigma3d
    bkw_copy = dict(bkw, cov3Ds=buf["cov3Ds"].clone())                       # not the forward's tensor: read
    g2 = gsr.backward(**bkw_copy)
    assert not bwd.last_call_recomputed_sigma3d
    for k in ("dL_dmean3D", "dL_dscale", "dL_drot"):       # the same Sigma3D either way: what differs is the float-atomic order of two blend runs
        assert float((g1[k] - g2[k]).abs().max()) <= 2e-5 * float(g2[k].abs().max()), k
    # ... and bit for bit: the per-Gaussian half alone (gsr_backward_geom) over g1's accumulators, once handed the array, once NULL
    import ctypes as C
    pk = gsr.__name__
    _lib, _host = (__import__("importlib").import_module(pk + "." + m) for m in ("_lib", "_host"))
    L, N = _lib.lib(), dev["means3D"].shape[0]
    # the call's backward workspace: dL_dcolor is a view of its accumulator records, which start gsr_backward_accumulators_offset(N) in
    acc_flat = g1["dL_dcolor"]._base
    assert acc_flat is not None and acc_flat.numel() == 16 * N
    ws_ptr = acc_flat.data_ptr() - int(L.gsr_backward_accumulators_offset(N))
    ws_bytes = int(L.gsr_backward_workspace_bytes(N, int(buf["point_list"].shape[0]), W, H))
    scene_c = _lib.GsrScene(N, _host.ptr(dev["means3D"]), _host.ptr(dev["scales"]), _host.ptr(dev["rotations"]), _host.ptr(dev["opacity"]),
                            _host.ptr(dev["sh"].view(-1, 3)), 3, 1.3, 1)
    cam_c = _host.make_camera(kw["viewmatrix"], kw["projmatrix"], kw["campos"], kw["background"], kw["tan_fovx"], kw["tan_fovy"], W, H)
    outs = []
    for cov in (buf["cov3Ds"], None):
        geom_c = _lib.GsrGeom(_host.ptr(buf["radii"]), None, None, None, None, _host.ptr(cov), None, None, _host.ptr(buf["clamped_state"]), None, None)
        o = [torch.empty(n, dtype=torch.float32, device="cuda") for n in (3 * N, 3 * N, 4 * N, N, 48 * N)]
        grads_c = _lib.GsrGrads(*[_host.ptr(t) for t in o], None, None, None, None)
        _lib.check(L.gsr_backward_geom(C.byref(scene_c), C.byref(cam_c), C.byref(geom_c), C.byref(grads_c), ws_ptr, ws_bytes, _host.raw_stream(acc_flat.device)))
        outs.append(o)
    torch.cuda.synchronize()
    for a_, b_, name in zip(outs[0], outs[1], ("dL_dmean3D", "dL_dscale", "dL_drot", "dL_dopacity", "dL_dshs")):
        assert torch.equal(a_, b_), name
    assert float(outs[0][1].abs().max()) > 0.0
    ref_kw = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
    ref = oracle.render_gaussians(**ref_kw)
    rb = backward_kwargs(sc, cam, ref_kw, ref[2], dpix)
    parity.compare_backward(g1, oracle.backward(**rb), None)
    # scales written in place after the forward: the array (made from the OLD scales) is read, as the reference would
    with torch.no_grad():
        dev["scales"].mul_(1.05)
    g3 = gsr.backward(**bkw)
    assert not bwd.last_call_recomputed_sigma3d
    sc2 = dict(sc, scales=dev["scales"].cpu().numpy())
    rb2 = backward_kwargs(sc2, cam, ref_kw, ref[2], dpix)                      # new scales, the old forward's buffers
    parity.compare_backward(g3, oracle.backward(**rb2), None)
    # another scale_modifier than the forward's: read as well
    buf2 = gsr.render_gaussians(**kw)[2]
    bkw4 = backward_kwargs(scene_t, cam, dict(kw, scale_modifier=1.0), buf2, dpix)
    gsr.backward(**bkw4)
    assert not bwd.last_call_recomputed_sigma3d


def test_backward_accumulators_cleared_by_the_forward(oracle, cameras, scenes):
    """GsrBinning.backward_ws: the forward blend kernel's spare workgroups clear the accumulators of a backward workspace made for
    that frame, and the first backward() handed the frame's point_list takes that workspace and skips its own clear -- once: a second
    backward() on the same forward, or one with a foreign point_list, gets a fresh workspace and clears it.  (Workspaces are per
    call since ABI 7 -- the returned blend-stage gradients are views of them -- so an older forward's workspace stays valid while a
    later forward runs.)  Every one of them has to give the oracle's gradients, and earlier returns must not change."""
    import torch
    gsr = pkg()
    from conftest import sub
    bwd = sub("backward").backward
    sub("forward")._backward_seen = True     # (the pre-clear is lazy: it starts with the process's first backward(); see the test below)
    sc = scenes.synthetic_scene(6000, 0.05, 0.6, 77)
    cam = lego_camera(cameras, frame=6, width=192, height=160)
    kw = render_kwargs(sc, cam, width=192, height=160)
    ref = oracle.render_gaussians(**kw)
    dpix = _pixel_grad(160, 192)
    g_ref = oracle.backward(**backward_kwargs(sc, cam, kw, ref[2], dpix))
    buf_a = gsr.render_gaussians(**kw)[2]
    g_a = gsr.backward(**backward_kwargs(sc, cam, kw, buf_a, dpix))
    parity.compare_backward(g_a, g_ref)
    assert bwd.last_call_skipped_the_clear
    keep = {k: g_a[k].clone() for k in ("dL_dcolor", "dL_dmean2D", "dL_dconic")}
    parity.compare_backward(gsr.backward(**backward_kwargs(sc, cam, kw, buf_a, dpix)), g_ref)      # the same forward again
    assert not bwd.last_call_skipped_the_clear
    buf_b = gsr.render_gaussians(**kw)[2]
    buf_c = gsr.render_gaussians(**kw)[2]
    parity.compare_backward(gsr.backward(**backward_kwargs(sc, cam, kw, buf_b, dpix)), g_ref)      # an older forward: its own workspace, still clean
    assert bwd.last_call_skipped_the_clear
    parity.compare_backward(gsr.backward(**backward_kwargs(sc, cam, kw, buf_c, dpix)), g_ref)
    assert bwd.last_call_skipped_the_clear
    for k, v in keep.items():                                                                  # the first call's views were not touched by the later calls
        assert torch.equal(g_a[k], v), k
    assert g_a["dL_dcolor"]._base is not None and g_a["dL_dmean2D"]._base is g_a["dL_dcolor"]._base and tuple(g_a["dL_dconic"].shape) == (6000, 4)
    assert float(g_a["dL_dmean2D"][:, 2].abs().max()) == 0.0 and float(g_a["dL_dconic"][:, 2].abs().max()) == 0.0
    parity.compare_backward(gsr.backward(**backward_kwargs(sc, cam, kw, ref[2], dpix)), g_ref)     # the oracle's (numpy) buffers
    assert not bwd.last_call_skipped_the_clear


def test_preclear_waits_for_the_first_backward(cameras, scenes):
    """forward.PRECLEAR_BACKWARD is lazy: a process that has never called backward() (a render-only user) gets no backward workspace
    allocated or cleared by its forwards; from the first backward() on, every forward pre-clears."""
    gsr = pkg()
    from conftest import sub
    fwd, bwd, host = sub("forward"), sub("backward").backward, sub("_host")
    sc = scenes.synthetic_scene(800, 0.05, 0.6, 12)
    cam = lego_camera(cameras, 2, 96, 64)
    kw = render_kwargs(sc, cam, width=96, height=64)
    seen, fwd._backward_seen = fwd._backward_seen, False
    try:
        buf = gsr.render_gaussians(**kw)[2]
        assert getattr(buf["point_list"], "_gsr_cleared_ws", None) is None
        gsr.backward(**backward_kwargs(sc, cam, kw, buf, _pixel_grad(64, 96)))
        assert not bwd.last_call_skipped_the_clear and fwd._backward_seen
        buf = gsr.render_gaussians(**kw)[2]
        assert getattr(buf["point_list"], "_gsr_cleared_ws", None) is not None
        gsr.backward(**backward_kwargs(sc, cam, kw, buf, _pixel_grad(64, 96)))
        assert bwd.last_call_skipped_the_clear
    finally:
        fwd._backward_seen = seen or fwd._backward_seen


def test_in_place_writes_between_forward_and_backward_are_seen(oracle, cameras, scenes):
    """The reference's backward() re-reads every array it is handed (backward.py:95-255 the SH rows and positions, :559-706 the
    2-D means / conics / colours).  Ours reuses state the forward derived from them -- the d(colour)/d(direction) sums, the packed
    blend records, the block masks -- and must drop it when the caller has written into the arrays IN PLACE since (same tensor
    objects, new contents): torch's version counters are recorded in the tags.  Each case is compared with the oracle fed the
    NEW arrays and the forward's (old) buffers, which is exactly what the reference would compute."""
    import torch
    gsr = pkg()
    from conftest import sub
    bwd = sub("backward").backward
    sc = scenes.synthetic_scene(4000, 0.05, 0.6, 91)
    cam = lego_camera(cameras, frame=3, width=176, height=144)
    kw_np = render_kwargs(sc, cam, width=176, height=144)
    ref = oracle.render_gaussians(**kw_np)
    dpix = _pixel_grad(144, 176)

    def fresh():
        dev = {k: torch.as_tensor(np.ascontiguousarray(sc[k])).cuda() for k in ("means", "opacities", "scales", "rotations")}
        dev["shs"] = torch.as_tensor(np.ascontiguousarray(sc["shs"])).cuda().reshape(-1, 3)
        kw = dict(kw_np, means3D=dev["means"], opacity=dev["opacities"], scales=dev["scales"], rotations=dev["rotations"], sh=dev["shs"])
        buf = gsr.render_gaussians(**kw)[2]
        own = backward_kwargs(sc, cam, kw, buf, torch.as_tensor(dpix).cuda())
        own.update(means3D=dev["means"], opacity=dev["opacities"], scales=dev["scales"], rotations=dev["rotations"], shs=dev["shs"])
        return dev, buf, own

    # untouched: every piece of forward state is reused
    dev, buf, own = fresh()
    parity.compare_backward(gsr.backward(**own), oracle.backward(**backward_kwargs(sc, cam, kw_np, ref[2], dpix)))
    assert bwd.last_call_used_forward_sh_dir and bwd.last_call_used_forward_masks and bwd.last_call_used_forward_records

    # SH coefficients scaled in place, positions nudged in place
    dev, buf, own = fresh()
    dev["shs"].mul_(1.1)
    dev["means"].add_(torch.tensor([0.003, -0.002, 0.001], device="cuda"))
    sc2 = dict(sc, shs=dev["shs"].cpu().numpy().reshape(sc["shs"].shape), means=dev["means"].cpu().numpy())
    g = gsr.backward(**own)
    assert not bwd.last_call_used_forward_sh_dir
    assert bwd.last_call_used_forward_masks and bwd.last_call_used_forward_records      # these do not depend on shs / means3D
    parity.compare_backward(g, oracle.backward(**backward_kwargs(sc2, cam, kw_np, ref[2], dpix)))

    # the library's own in-place writer (Adam through raw pointers) counts as a write too
    dev, buf, own = fresh()
    P = {"positions": dev["means"], "scales": dev["scales"], "rotations": dev["rotations"], "opacities": dev["opacities"].reshape(-1), "shs": dev["shs"]}
    G = {k: torch.full_like(v, 1e-3) for k, v in P.items()}
    M, V = gsr.optimizer.make_state(P)
    gsr.optimizer.adam_update(P, G, M, V, iteration=0)
    sc3 = {"means": dev["means"].cpu().numpy(), "scales": dev["scales"].cpu().numpy(), "rotations": dev["rotations"].cpu().numpy(),
           "opacities": dev["opacities"].cpu().numpy(), "shs": dev["shs"].cpu().numpy().reshape(sc["shs"].shape)}
    g = gsr.backward(**own)
    assert not bwd.last_call_used_forward_sh_dir
    parity.compare_backward(g, oracle.backward(**backward_kwargs(sc3, cam, kw_np, ref[2], dpix)))

    # a forward buffer written in place: conic_opacity (records and masks were derived from it), then the colours (records only --
    # but the colours are a view of the same records tensor as means2D / conic_opacity and share its version counter, so the masks
    # are dropped with them: conservative, the self-contained block test gives the same gradients)
    for key, fwd_key in (("conic_opacity", "conic_opacity"), ("rgb", "colors")):
        dev, buf, own = fresh()
        buf[fwd_key].mul_(0.97)
        ref_buf = dict(ref[2], **{fwd_key: buf[fwd_key].cpu().numpy()})
        g = gsr.backward(**own)
        assert not bwd.last_call_used_forward_records, key
        assert not bwd.last_call_used_forward_masks, key
        parity.compare_backward(g, oracle.backward(**backward_kwargs(sc, cam, kw_np, ref_buf, dpix)))


def test_xy_conic_and_colours_are_columns_of_the_blend_records(oracle, cameras, scenes):
    """ABI 7: the forward writes points_xy_image / conic_opacity / colors ONCE, as columns 0-1 / 2-5 / 6-8 of its 64-byte blend
    records (GsrGeom.blend_records with xy = conic_opacity = rgb = NULL); the dict entries are strided views of that one tensor.
    They must still behave like the reference's arrays for a caller (values, shapes, numpy conversion, arithmetic, packed copies),
    backward() must take the records as they are when handed the views back, and re-read packed copies (or anything else) when not."""
    import torch
    gsr = pkg()
    from conftest import sub
    bwd = sub("backward").backward
    sc = scenes.synthetic_scene(3000, 0.05, 0.6, 19)
    cam = lego_camera(cameras, frame=5, width=160, height=128)
    kw = render_kwargs(sc, cam, width=160, height=128)
    ref = oracle.render_gaussians(**kw)
    img, _, buf = gsr.render_gaussians(**kw)
    parity.compare_forward((img, _, buf), ref)
    xy, con, col = buf["points_xy_image"], buf["conic_opacity"], buf["colors"]
    assert tuple(xy.shape) == (3000, 2) and tuple(con.shape) == (3000, 4) and tuple(col.shape) == (3000, 3)
    base = xy._base
    assert base is not None and con._base is base and col._base is base and tuple(base.shape) == (3000, 16) and base.is_contiguous()
    assert xy.stride() == (16, 1) and con.data_ptr() == base.data_ptr() + 8 and col.data_ptr() == base.data_ptr() + 24
    vis = ref[2]["radii"] > 0
    np.testing.assert_array_equal(base[:, 9].cpu().numpy()[vis], (np.float32(1.0) / ref[2]["depths"][vis]).astype(np.float32))   # 1 / depth rides along
    assert float(base[:, 10:].abs().max()) == 0.0
    for t, k in ((xy, "points_xy_image"), (con, "conic_opacity"), (col, "colors")):
        a = t.cpu().numpy()
        assert a.flags["C_CONTIGUOUS"] and t.contiguous().is_contiguous() and (t * 2.0).shape == t.shape
        np.testing.assert_array_equal(np.asarray(t.cpu()), a)
    dpix = _pixel_grad(128, 160)
    g_ref = oracle.backward(**backward_kwargs(sc, cam, kw, ref[2], dpix))
    parity.compare_backward(gsr.backward(**backward_kwargs(sc, cam, kw, buf, dpix)), g_ref)
    assert bwd.last_call_used_forward_records
    packed = dict(buf, points_xy_image=xy.contiguous(), conic_opacity=con.contiguous(), colors=col.contiguous())
    parity.compare_backward(gsr.backward(**backward_kwargs(sc, cam, kw, packed, dpix)), g_ref)       # packed copies: re-packed by the backward
    assert not bwd.last_call_used_forward_records
    host = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in buf.items()}
    parity.compare_backward(gsr.backward(**backward_kwargs(sc, cam, kw, host, dpix)), g_ref)         # numpy arrays, as the reference's callers hold
    assert not bwd.last_call_used_forward_records


def test_dL_dcov3D_is_a_dense_zero_array(cameras, scenes):
    """backward.py:1119 returns an (N, 6) zero array; ours must behave like one (dense strides, .view(-1), numpy)."""
    gsr = pkg()
    sc = scenes.synthetic_scene(300, 0.05, 0.5, 4)
    cam = lego_camera(cameras, 1, 64, 48)
    kw = render_kwargs(sc, cam, width=64, height=48)
    buf = gsr.render_gaussians(**kw)[2]
    g = gsr.backward(**backward_kwargs(sc, cam, kw, buf, _pixel_grad(48, 64)))
    z = g["dL_dcov3D"]
    assert tuple(z.shape) == (300, 6) and z.is_contiguous() and z.view(-1).numel() == 1800
    assert float(z.abs().max()) == 0.0 and z.cpu().numpy().strides == (24, 4)


@pytest.mark.parametrize("field,val", [("means", np.nan), ("means", np.inf), ("means", 1e30), ("scales", np.nan), ("scales", 0.0),
                                       ("scales", -1.0), ("scales", 1e20), ("rotations", np.nan), ("rotations", 0.0),
                                       ("opacities", np.nan), ("opacities", -5.0), ("shs", np.inf)])
def test_nonfinite_and_degenerate_inputs_do_not_fault(cameras, scenes, field, val):
    """NaN / Inf / zero / negative / huge parameters in a few Gaussians (whole rows and single components): whatever the
    values rendered, every index stays in range -- the list is well-formed and forward + backward complete."""
    import torch
    gsr = pkg()
    rng = np.random.default_rng(17)
    sc = scenes.synthetic_scene(3000, 0.05, 0.6, seed=3)
    idx = rng.choice(3000, 40, replace=False)
    arr = sc[field].reshape(3000, -1)
    arr[idx[:20]] = val
    arr[idx[20:], 0] = val
    cam = lego_camera(cameras, frame=2, width=200, height=152)
    kw = render_kwargs(sc, cam, width=200, height=152)
    img, depth, buf = gsr.render_gaussians(**kw)
    D = int(buf["point_list"].shape[0])
    pl, rg = parity.to_np(buf["point_list"]), parity.to_np(buf["ranges"])
    assert D == int(parity.to_np(buf["point_offsets"])[-1]) and (D == 0 or (0 <= pl.min() and pl.max() < 3000))
    assert rg.min() >= 0 and rg.max() <= D and np.all(rg[:, 0] <= rg[:, 1])
    g = gsr.backward(**backward_kwargs(sc, cam, kw, buf, np.full((152, 200, 3), 1e-5, np.float32)))
    torch.cuda.synchronize()
    assert tuple(g["dL_dshs"].shape) == (48000, 3)
    if field not in ("shs",):     # a non-finite colour spreads through the blend; everything else stays contained
        assert bool(torch.isfinite(img).all())
