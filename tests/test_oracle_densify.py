"""CPU tests of row f4 (adaptive density control + PLY checkpoint): the numpy oracle's own invariants, the behaviours
of the reference it must keep (train.py:351-713), and the product's host-only PLY writer against the oracle's
vertex-by-vertex restatement.  The reference holds no densification fixture and Warp is absent: parity unpinned."""
import numpy as np

from conftest import sub
from oracle import densify as od

F = np.float32


def make_params(n, seed=0, big_frac=0.3):
    rng = np.random.default_rng(seed)
    p = {"positions": rng.normal(0, 1, (n, 3)).astype(F), "scales": rng.uniform(0.001, 0.02, (n, 3)).astype(F),
         "rotations": rng.normal(0, 1, (n, 4)).astype(F), "opacities": rng.uniform(0.0, 1.0, n).astype(F),
         "shs": rng.normal(0, 0.5, (n, 48)).astype(F)}
    big = rng.uniform(0, 1, n) < big_frac
    p["scales"][big] *= 10.0
    low = rng.uniform(0, 1, n) < 0.1
    p["opacities"][low] = 0.001
    g = (rng.normal(0, 1, (n, 3)) * 3e-4).astype(F)
    return p, g


def test_randf_is_a_stateless_unit_interval_hash():
    s = np.arange(0, 300000, dtype=np.int32)
    r = od.randf(s)
    assert r.dtype == np.float32 and r.min() >= 0.0 and r.max() < 1.0
    assert abs(float(r.mean()) - 0.5) < 5e-3 and abs(float(r.var()) - 1 / 12) < 5e-3
    np.testing.assert_array_equal(r, od.randf(s))                       # stateless: same seed, same value
    assert len(np.unique(r[:4096])) > 4000
    # scalar restatement of one PCG round, in Python integers
    for x in (0, 1, 2, 12345, 2**31 - 1, -5 & 0xFFFFFFFF):
        b = (x * 747796405 + 2891336453) & 0xFFFFFFFF
        c = (((b >> ((b >> 28) + 4)) ^ b) * 277803737) & 0xFFFFFFFF
        h = (c >> 22) ^ c
        assert int(od.rand_pcg(np.uint32(x))) == h
        assert float(od.randf(np.uint32(x))) == (h >> 8) / 16777216.0


def test_exclusive_scan_count_ignores_the_last_flag():
    m = np.array([1, 0, 1, 1], np.int32)
    p, c = od.exclusive_scan(m)
    assert p.tolist() == [0, 1, 1, 2] and c == 2                        # train.py:433: int(prefix[-1]) -- last flag not counted
    p, c = od.exclusive_scan(np.array([1, 0, 1, 0], np.int32))
    assert c == 2
    assert od.exclusive_scan(np.zeros(0, np.int32))[1] == 0


def test_clone_split_compact_rows():
    p, g = make_params(500, seed=1)
    norms = od.compute_grad_norms(g, 500)
    cm = od.mark_candidates(norms, p["scales"], 2e-4, 1.0, 0.01, split=False)
    sm = od.mark_candidates(norms, p["scales"], 2e-4, 1.0, 0.01, split=True)
    assert cm.sum() > 10 and sm.sum() > 10 and not np.any(cm & sm)
    assert np.array_equal((cm | sm) == 1, norms >= F(2e-4))
    pre, tot = od.exclusive_scan(cm)
    out = od.clone_gaussians(p, cm, pre, tot)
    assert od._rows(out) == 500 + tot
    src = np.nonzero(cm)[0][:tot]
    for k in ("scales", "rotations", "opacities", "shs"):
        np.testing.assert_array_equal(out[k][:500], od._shape(p)[k])
        np.testing.assert_array_equal(out[k][500:], od._shape(p)[k][src])
    d = out["positions"][500:] - p["positions"][src]
    assert d.min() >= -1e-6 and d.max() <= 0.01 + 1e-6                  # noise = randf * 0.01: one-sided (optimizer.py:352-356)
    pre, tot = od.exclusive_scan(sm)
    out = od.split_gaussians(p, sm, pre, tot, 2, 0.8)
    assert od._rows(out) == 500 + 2 * tot
    src = np.repeat(np.nonzero(sm)[0][:tot], 2)
    np.testing.assert_array_equal(out["scales"][500:], p["scales"][src] * F(0.8))
    d = out["positions"][500:] - p["positions"][src]
    assert np.abs(d).max() <= 0.01 + 1e-6 and d.min() < 0
    valid = od.split_removal_mask(sm, od._rows(out))
    assert valid.sum() == od._rows(out) - sm.sum()
    pre, cnt = od.exclusive_scan(valid)
    comp = od.compact_gaussians(out, valid, pre, cnt)
    assert cnt == valid.sum() - 1                                        # the last row is valid and is lost (train.py:581)
    np.testing.assert_array_equal(comp["shs"], out["shs"][valid == 1][:cnt])


def test_sequence_keeps_reference_behaviours():
    p, g = make_params(3000, seed=2)
    cfg = {"max_allowed_prune_ratio": 1.0}
    out, log = od.densification_and_pruning(p, g, 600, cfg, 1.0)
    assert log["cloned"] > 0 and log["split"] > 0 and log["pruned"] > 0 and not log["opacity_reset"]
    assert od._rows(out) == 3000 + log["cloned"] + 2 * log["split"] - log["split_removed"] - log["pruned"]
    assert np.all(out["opacities"] > F(0.005))
    # outside the window / off the interval: nothing but the reset rule runs
    for it in (500, 650, 15000):
        o2, l2 = od.densification_and_pruning(p, g, it, cfg, 1.0)
        assert od._rows(o2) == 3000 and l2["cloned"] == 0 and l2["pruned"] == 0
    o3, l3 = od.densification_and_pruning(p, g, 3000, cfg, 1.0)
    assert l3["opacity_reset"] and np.all(o3["opacities"] == F(0.01))
    o4, l4 = od.densification_and_pruning(p, g, 500, {"background_color": [1.0, 1.0, 1.0]}, 1.0)
    assert l4["opacity_reset"] and od._rows(o4) == 3000                 # white background: reset at densify_from_iter (train.py:697-700)
    # pruning is skipped when fewer than min_valid_points would remain (train.py:652-655)
    small, gs = make_params(400, seed=3)
    o5, l5 = od.densification_and_pruning(small, gs, 600, cfg, 1.0)
    assert l5["prune_skipped"] and np.any(o5["opacities"] <= F(0.005))


def test_ply_writer_matches_oracle_and_round_trips(tmp_path):
    pc = sub("point_cloud")
    p, _ = make_params(37, seed=4)
    p["shs"][:, :3] = np.random.default_rng(5).uniform(-0.7, 0.7, (37, 3)).astype(F)   # colours on both clip edges
    path = tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply"
    pc.save_ply({k: v for k, v in p.items()}, str(path), 37)
    blob = path.read_bytes()
    assert blob == od.ply_bytes(p, 37)
    assert blob.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 37\nproperty float x\n")
    assert len(blob) == len(pc.ply_header(37)) + 37 * (59 * 4 + 3)
    back = pc.load_ply(str(path))
    for k in ("positions", "scales", "rotations", "opacities"):
        np.testing.assert_array_equal(back[k].reshape(-1), p[k].reshape(-1))
    np.testing.assert_array_equal(back["shs"].reshape(37, 48), p["shs"])
    col = np.clip(p["shs"][:, :3] + F(0.5), 0, 1)
    np.testing.assert_array_equal(back["colors"], (col * F(255)).astype(np.int64).astype(np.uint8))
    # fewer points than rows, explicit colours
    pc.save_ply(p, str(path), 5, colors=np.full((37, 3), 0.5, F))
    back = pc.load_ply(str(path))
    assert back["positions"].shape == (5, 3) and np.all(back["colors"] == 127)


def test_scene_extent():
    dz = sub("densify")
    assert dz.calculate_scene_extent([]) == 1.0
    c = np.array([[4.0, 0, 0], [-4.0, 0, 0], [0, 4.0, 0], [0, -4.0, 0]])
    assert abs(dz.calculate_scene_extent(c) - 4.0) < 1e-12
    assert abs(dz.calculate_scene_extent(c, 1.5) - 6.0) < 1e-12
    assert dz.calculate_scene_extent(c * 0.01) == 1.0                   # never below 1 (train.py:257)
