"""GPU parity of row f4 (adaptive density control, csrc/densify.hip through the C ABI) against the numpy oracle:
bit-exact -- the path is copies, integer scans and a handful of single float32 operations."""
import numpy as np
import pytest

from conftest import sub
from oracle import densify as od
from test_oracle_densify import make_params

pytestmark = pytest.mark.gpu
F = np.float32


def to_dev(p):
    import torch
    return {"positions": torch.as_tensor(p["positions"]).cuda().contiguous(), "scales": torch.as_tensor(p["scales"]).cuda().contiguous(),
            "rotations": torch.as_tensor(p["rotations"]).cuda().contiguous(), "opacities": torch.as_tensor(p["opacities"]).cuda().contiguous(),
            "shs": torch.as_tensor(p["shs"]).reshape(-1, 3).cuda().contiguous()}


def assert_params_equal(dev, ref):
    ref = od._shape(ref)
    n = od._rows(ref)
    assert int(dev["opacities"].numel()) == n
    for k in od.GROUPS:
        np.testing.assert_array_equal(dev[k].cpu().numpy().reshape(ref[k].shape), ref[k], err_msg=k)


@pytest.mark.parametrize("n", [1, 63, 1000, 20011])
def test_kernels_match_oracle(n):
    import torch
    dz = sub("densify")
    p, g = make_params(n, seed=n)
    d = to_dev(p)
    dg = torch.as_tensor(g).cuda()
    norms = od.compute_grad_norms(g, n)
    for split in (False, True):
        ref_mask = od.mark_candidates(norms, p["scales"], 2e-4, 1.3, 0.01, split)
        mask = dz.mark_candidates(d, dg, 2e-4, 1.3, 0.01, split)
        np.testing.assert_array_equal(mask.cpu().numpy(), ref_mask)
        ref_prefix, ref_total = od.exclusive_scan(ref_mask)
        prefix, total = dz.exclusive_scan(mask)
        assert total == ref_total
        np.testing.assert_array_equal(prefix.cpu().numpy(), ref_prefix)
        if split:
            out = dz.split_gaussians(d, mask, prefix, total, 2, 0.8)
            ref = od.split_gaussians(p, ref_mask, ref_prefix, ref_total, 2, 0.8)
            assert_params_equal(out, ref)
            valid = dz.split_removal_mask(mask, n + 2 * total)
            ref_valid = od.split_removal_mask(ref_mask, n + 2 * ref_total)
            np.testing.assert_array_equal(valid.cpu().numpy(), ref_valid)
            vp, vc = dz.exclusive_scan(valid)
            rvp, rvc = od.exclusive_scan(ref_valid)
            assert vc == rvc
            assert_params_equal(dz.compact_gaussians(out, valid, vp, vc), od.compact_gaussians(ref, ref_valid, rvp, rvc))
        else:
            assert_params_equal(dz.clone_gaussians(d, mask, prefix, total, 0.01), od.clone_gaussians(p, ref_mask, ref_prefix, ref_total, 0.01))
    valid = dz.prune_mask(d, 0.005)
    np.testing.assert_array_equal(valid.cpu().numpy(), od.prune_mask(p["opacities"], 0.005))
    dz.reset_opacities(d["opacities"], 0.01)
    assert torch.all(d["opacities"] == 0.01)


def test_gradient_rows_may_be_fewer_than_points():
    """After a clone the split pass marks N+clones rows against N gradient norms (train.py:478-494)."""
    import torch
    dz = sub("densify")
    p, g = make_params(1500, seed=9)
    d = to_dev(p)
    dg = torch.as_tensor(g[:1000]).cuda().contiguous()
    mask = dz.mark_candidates(d, dg, 2e-4, 1.0, 0.01, True)
    ref = od.mark_candidates(od.compute_grad_norms(g[:1000], 1500), p["scales"], 2e-4, 1.0, 0.01, True)
    np.testing.assert_array_equal(mask.cpu().numpy(), ref)
    assert not mask[1000:].any()
    none = dz.mark_candidates(d, None, 2e-4, 1.0, 0.01, False)
    assert not none.any()


def test_flagged_last_row_is_dropped_not_written_out_of_bounds():
    import torch
    dz = sub("densify")
    p, _ = make_params(130, seed=10, big_frac=0.0)
    d = to_dev(p)
    mask = torch.zeros(130, dtype=torch.int32, device="cuda")
    mask[[3, 64, 129]] = 1
    prefix, total = dz.exclusive_scan(mask)
    assert total == 2                                                    # row 129's flag is not counted (train.py:433)
    out = dz.clone_gaussians(d, mask, prefix, total, 0.01)
    ref = od.clone_gaussians(p, mask.cpu().numpy(), prefix.cpu().numpy(), total, 0.01)
    assert_params_equal(out, ref)
    out = dz.split_gaussians(d, mask, prefix, total, 2, 0.8)
    assert_params_equal(out, od.split_gaussians(p, mask.cpu().numpy(), prefix.cpu().numpy(), total, 2, 0.8))


@pytest.mark.parametrize("n,it,cfg", [
    (3000, 600, {"max_allowed_prune_ratio": 1.0}),
    (3000, 600, {}),                                                    # default ratio 0.5
    (3000, 3000, {"max_allowed_prune_ratio": 1.0}),                     # densify + opacity reset
    (3000, 650, {}),                                                    # off the interval
    (400, 600, {"max_allowed_prune_ratio": 1.0}),                       # prune skipped: below min_valid_points
    (3000, 500, {"background_color": [1.0, 1.0, 1.0]}),                 # white background reset
    (50000, 1200, {"max_allowed_prune_ratio": 1.0, "densify_grad_threshold": 0.0004, "percent_dense": 0.02}),
])
def test_trainer_sequence_matches_oracle(n, it, cfg):
    import torch
    dz = sub("densify")
    p, g = make_params(n, seed=n + it)
    model = dz.GaussianModel(to_dev(p), config=cfg, scene_extent=1.0)
    model.grads["positions"].copy_(torch.as_tensor(g))
    model.adam_m["scales"].fill_(1.0)
    log = model.densification_and_pruning(it)
    ref, ref_log = od.densification_and_pruning(p, g, it, dict({"background_color": [0.0, 0.0, 0.0]}, **cfg), 1.0)
    assert log == ref_log
    assert_params_equal(model.params, ref)
    assert model.num_points == od._rows(ref)
    changed = model.num_points != n or log["cloned"] or log["split"] or log["pruned"]
    if changed:                                                          # train.py:470-475: optimizer state restarts from zero
        for state in (model.grads, model.adam_m, model.adam_v):
            for k in od.GROUPS:
                assert state[k].numel() == model.params[k].numel() and not state[k].any()
    else:
        assert torch.all(model.adam_m["scales"] == 1.0)


def test_million_points_properties():
    """Size-independent properties at the C3 point count: row conservation, stable order, exact copies."""
    import torch
    dz = sub("densify")
    n = 1_000_000
    p, g = make_params(n, seed=77)
    d = to_dev(p)
    dg = torch.as_tensor(g).cuda()
    mask = dz.mark_candidates(d, dg, 2e-4, 1.0, 0.01, False)
    prefix, total = dz.exclusive_scan(mask)
    m = mask.cpu().numpy()
    assert total == int(m[:-1].sum())
    assert int(prefix[-1]) == total and torch.all(prefix[1:] - prefix[:-1] == mask[:-1])
    out = dz.clone_gaussians(d, mask, prefix, total, 0.01)
    src = torch.nonzero(mask)[:total, 0]
    assert torch.equal(out["shs"].reshape(-1, 48)[n:], d["shs"].reshape(-1, 48)[src])
    assert torch.equal(out["rotations"][:n], d["rotations"]) and torch.equal(out["opacities"][n:], d["opacities"][src])
    delta = out["positions"][n:] - d["positions"][src]
    assert float(delta.min()) >= -1e-6 and float(delta.max()) <= 0.01 + 1e-6
    valid = dz.prune_mask(out, 0.005)
    vp, vc = dz.exclusive_scan(valid)
    comp = dz.compact_gaussians(out, valid, vp, vc)
    keep = torch.nonzero(valid)[:vc, 0]
    assert torch.equal(comp["positions"], out["positions"][keep]) and torch.equal(comp["shs"].reshape(-1, 48), out["shs"].reshape(-1, 48)[keep])
    assert float(comp["opacities"].min()) > 0.005


def test_error_paths():
    import torch
    dz = sub("densify")
    p, _ = make_params(10, seed=1)
    d = to_dev(p)
    bad = dict(d, shs=d["shs"][:-1])
    with pytest.raises(ValueError):
        dz.prune_mask(bad, 0.005)
    bad = dict(d, scales=d["scales"].double())
    with pytest.raises(ValueError):
        dz.mark_candidates(bad, None, 1.0, 1.0, 0.01, False)
    empty = dz.alloc_params(0, "cuda")
    mask = dz.prune_mask(empty, 0.005)
    prefix, total = dz.exclusive_scan(mask)
    assert total == 0 and prefix.numel() == 0
    model = dz.GaussianModel(empty)
    assert model.densification_and_pruning(600)["prune_skipped"]


def test_trainer_sequence_sweep():
    """Seeded sweep of the whole density-control sequence against the oracle: row counts at wave / scan-unit edges,
    thresholds that flag nothing / everything / the last row, every branch of the prune gate."""
    import torch
    dz = sub("densify")
    rng = np.random.default_rng(2024)
    for case in range(60):
        n = int(rng.choice([1, 2, 63, 64, 65, 1000, 1023, 1024, 1025, 2047, 2049, 5000, 12345]))
        p, g = make_params(n, seed=7000 + case, big_frac=float(rng.choice([0.0, 0.3, 1.0])))
        cfg = {"densify_grad_threshold": float(rng.choice([0.0, 2e-4, 4e-4, 1.0])), "percent_dense": float(rng.choice([0.0, 0.01, 1.0])),
               "cull_opacity_threshold": float(rng.choice([0.0, 0.005, 0.5, 2.0])), "min_valid_points": int(rng.choice([0, 1000])),
               "max_allowed_prune_ratio": float(rng.choice([0.05, 0.5, 1.0])), "max_valid_points": int(rng.choice([1500, 1000000]))}
        if case % 5 == 0:
            g[-1] = 1.0                                            # the last row always flagged: its flag is not counted (Q17)
        it = int(rng.choice([600, 3000, 650]))
        model = dz.GaussianModel(to_dev(p), config=cfg, scene_extent=float(rng.choice([1.0, 4.0])))
        model.grads["positions"].copy_(torch.as_tensor(g))
        log = model.densification_and_pruning(it)
        ref, ref_log = od.densification_and_pruning(p, g, it, dict({"background_color": [0.0, 0.0, 0.0]}, **cfg), model.scene_extent)
        assert log == ref_log, (case, n, cfg, log, ref_log)
        assert_params_equal(model.params, ref)


@pytest.mark.parametrize("n", [0, 1, 5000, 100003])
def test_init_gaussian_params_matches_oracle(n):
    dz = sub("densify")
    got = dz.init_gaussian_params(n, 0.1)
    ref = od.init_gaussian_params(n, 0.1)
    assert_params_equal(got, ref)
    if n >= 5000:
        p = got["positions"]
        assert float(p.min()) >= -1.3 and float(p.max()) < 1.3 and abs(float(p.mean())) < 0.05   # train.py:52-56: U(-1.3, 1.3)
