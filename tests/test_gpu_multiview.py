"""
BASELINE.json config #4 on ONE GPU: eight Lego training views per iteration, the view-averaged gradient of the
north-star's data-parallel step (replicated Gaussians, one view per GPU, gradient averaged over views -- the reference
itself is single-view: train.py:928, and hands ONE view's gradients to its optimizer at train.py:1047-1051).

  test_c4_eight_lego_views      C2 scene (800x800, 100 k Gaussians), frames 0-7 of tests/golden/lego_train_poses.json
                                rendered sequentially; the 8-view mean of the 59 optimizer floats per Gaussian from
                                (i) the summed dense arenas (what one all-reduce of the arena gives) and (ii) the factored
                                exchange (11-float arenas summed + dist.sh_gradients_from_views over the 8 payloads), each
                                compared with the ORACLE's 8-view mean under tests/parity.py.
  test_two_rank_factored_exchange   two processes (launch.launch_ranks, gloo, both on cuda:0) run FactoredExchange end to
                                end; both ranks must hold the same bits, equal to the single-process rebuild of the same
                                two views and within tolerance of the oracle's two-view mean.
  test_two_gpus_rccl            SKIPPED unless torch.cuda.device_count() >= 2 (the first multi-GPU lease runs it): two ranks,
                                one GPU each, backend nccl = RCCL over xGMI, one view each of the same scene; both ranks hold
                                identical bits, factored == dense all-reduce within the order-of-summation tolerance, both
                                within tests/parity.py of the oracle's two-view mean.
"""
import importlib
import os

import numpy as np
import pytest

from conftest import PKG_NAME, ROOT, backward_kwargs, lego_camera, pkg, render_kwargs
import parity

pytestmark = pytest.mark.gpu

OPT_KEYS = ["dL_dmean3D", "dL_dscale", "dL_drot", "dL_dopacity", "dL_dshs"]   # what train.py:1047-1051 copies to the optimizer


def test_c4_eight_lego_views(oracle, cameras, scenes):
    import torch
    gsr = pkg()
    cfg = scenes.CONFIGS["C2"]
    W, H, n = cfg["width"], cfg["height"], cfg["n"]
    scene = scenes.synthetic_scene(n, cfg["scale_median"], cfg["scale_sigma"], cfg["seed"])
    V = 8
    ref_mean = {k: None for k in OPT_KEYS}
    dense_sum, small_sum, payloads = None, None, []
    for f in range(V):
        cam = lego_camera(cameras, frame=f, width=W, height=H)
        fkw = render_kwargs(scene, cam, degree=3)
        dpix = (np.random.default_rng(500 + f).normal(0.0, 1.0, (H, W, 3)) / (H * W * 3)).astype(np.float32)
        got = gsr.render_gaussians(**fkw)
        ref = oracle.render_gaussians(**fkw)
        parity.assert_exact(f"view {f} point_list", got[2]["point_list"], ref[2]["point_list"])
        parity.assert_image(f"view {f} image", got[0], ref[0])
        g_ref = oracle.backward(**backward_kwargs(scene, cam, fkw, ref[2], dpix))
        for k in OPT_KEYS:
            a = np.asarray(g_ref[k], dtype=np.float64)
            ref_mean[k] = a if ref_mean[k] is None else ref_mean[k] + a
        bkw = backward_kwargs(scene, cam, fkw, got[2], dpix)
        g_dense = gsr.backward(**bkw)                                   # (i) the 59-float arena of this view
        dense_sum = g_dense["_arena"].clone() if dense_sum is None else dense_sum.add_(g_dense["_arena"])
        g_fact = gsr.backward(**bkw, sh_gradient="factored")            # (ii) 11-float arena + 3-float payload
        small_sum = g_fact["_arena"].clone() if small_sum is None else small_sum.add_(g_fact["_arena"])
        payloads.append(g_fact["_view_payload"].clone())
    ref_mean = {k: v / V for k, v in ref_mean.items()}
    report = {}
    dense = gsr.dist.arena_views(dense_sum.mul_(1.0 / V), n)
    for k in OPT_KEYS:
        report["dense " + k] = parity.assert_grad("dense 8-view mean " + k, dense[k], ref_mean[k].astype(np.float32).reshape(dense[k].shape))
    fact = gsr.dist.small_arena_views(small_sum.mul_(1.0 / V), n)
    means = torch.as_tensor(scene["means"]).cuda().contiguous()
    fact["dL_dshs"] = gsr.dist.sh_gradients_from_views(means, payloads, 3, average=True)
    for k in OPT_KEYS:
        report["factored " + k] = parity.assert_grad("factored 8-view mean " + k, fact[k], ref_mean[k].astype(np.float32).reshape(fact[k].shape))
    print("\nC4 (8 Lego views, C2 scene) margins (fraction inside tolerance, max err / max|g|):")
    for k, (ok, rel) in report.items():
        print(f"  {k:28s} {ok:.6f}  {rel:.2e}")


def test_two_rank_factored_exchange(oracle, tmp_path):
    import torch
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker
    gsr = pkg()
    launch = importlib.import_module(f"{PKG_NAME}.launch")
    rc = launch.launch_ranks(os.path.join(ROOT, "tests", "dist_worker.py"), ["gpu", str(tmp_path)], 2, timeout=400)
    assert rc == 0
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert sorted(r0.files) == sorted(OPT_KEYS)
    for k in OPT_KEYS:                       # every rank ends the step with the same bits
        np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)
    # the same two views in ONE process: dist.sh_gradients_from_views over the two payloads, 11-float arenas averaged
    scene = dist_worker.gpu_case_scene(gsr)
    n, deg = dist_worker.GPU_CASE["n"], dist_worker.GPU_CASE["degree"]
    pay, small, ref_mean = [], [], {k: 0.0 for k in OPT_KEYS}
    for f in range(2):
        fkw, bkw, cam = dist_worker.gpu_case_view(gsr, scene, f)
        g = gsr.backward(**bkw, sh_gradient="factored")
        pay.append(g["_view_payload"].clone())
        small.append(g["_arena"].clone())
        ref = oracle.render_gaussians(**fkw)
        okw = backward_kwargs(scene, cam, fkw, ref[2], bkw["dL_dpixels"].cpu().numpy())
        g_ref = oracle.backward(**okw)
        for k in OPT_KEYS:
            ref_mean[k] = ref_mean[k] + np.asarray(g_ref[k], dtype=np.float64) / 2.0
    means = torch.as_tensor(scene["means"]).cuda().contiguous()
    sh_one = gsr.dist.sh_gradients_from_views(means, pay, deg, average=True).cpu().numpy()
    # the SH rebuild is a deterministic kernel over the gathered payloads; the payloads and the 11-float arenas carry
    # float-atomic order from the blend backward, so two replays agree to tolerance, not to the bit
    tol = lambda ref: dict(rtol=2e-3, atol=1e-4 * float(np.abs(ref).max()))
    np.testing.assert_allclose(r0["dL_dshs"], sh_one, **tol(sh_one))
    one = gsr.dist.small_arena_views((small[0] + small[1]) * 0.5, n)
    for k in OPT_KEYS[:4]:
        a = one[k].cpu().numpy()
        np.testing.assert_allclose(r0[k], a, err_msg=k, **tol(a))
    for k in OPT_KEYS:                       # and against the oracle's two-view mean
        parity.assert_grad("2-rank " + k, r0[k], ref_mean[k].astype(np.float32).reshape(r0[k].shape))


def _device_count():
    import torch
    return torch.cuda.device_count()     # counts devices without initialising the GPU runtime in this process


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_two_gpus_rccl(oracle, tmp_path, capfd):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker
    gsr = pkg()
    launch = importlib.import_module(f"{PKG_NAME}.launch")
    rc = launch.launch_ranks(os.path.join(ROOT, "tests", "dist_worker.py"), ["nccl", str(tmp_path)], 2, timeout=400)
    assert rc == 0
    print(capfd.readouterr().out)        # rank 0's exchange timings and byte counts
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k in r0.files:                   # every rank ends the step with the same bits, for both exchanges
        np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)
    scene = dist_worker.gpu_case_scene(gsr)
    ref_mean = {k: 0.0 for k in OPT_KEYS}
    for f in range(2):
        fkw, bkw, cam = dist_worker.gpu_case_view(gsr, scene, f)
        ref = oracle.render_gaussians(**fkw)
        g_ref = oracle.backward(**backward_kwargs(scene, cam, fkw, ref[2], bkw["dL_dpixels"].cpu().numpy()))
        for k in OPT_KEYS:
            ref_mean[k] = ref_mean[k] + np.asarray(g_ref[k], dtype=np.float64) / 2.0
    for k in OPT_KEYS:
        want = ref_mean[k].astype(np.float32).reshape(r0[k].shape)
        # factored vs dense: the same per-view products, summed over views in a different order (and float atomics per replay)
        np.testing.assert_allclose(r0[k], r0["dense_" + k].reshape(r0[k].shape), rtol=2e-3, atol=1e-4 * float(np.abs(want).max()), err_msg=k)
        parity.assert_grad("2-GPU factored " + k, r0[k], want)
        parity.assert_grad("2-GPU dense " + k, r0["dense_" + k].reshape(r0[k].shape), want)
