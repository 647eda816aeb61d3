"""The reference documents ONE behaviour of its backward pass: trained on the Lego views its loss falls and the bulldozer
appears (readme.md / assets/example_train_lego.gif; no number, no checkpoint).  This runs the counterpart of that trainer
(examples/train.py: forward -> L1 -> backward -> Adam -> density control, reference train.py:920-1066) for 60 iterations on the
committed real targets (data/lego: r_0..r_7, alpha dropped) from the reference's own initial point set (train.py:37-92:
5 000 Gaussians, scale 0.1, opacity 0.1) and requires the L1 loss to fall."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_sixty_iterations_on_real_lego_targets():
    cmd = [sys.executable, os.path.join(ROOT, "examples", "train.py"), "--dataset", os.path.join(ROOT, "data", "lego"), "--views", "8",
           "--iterations", "60", "--gaussians", "5000"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    losses = [(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"iter\s+(\d+)\s+loss\s+([0-9.eE+-]+)", p.stdout)]
    assert losses[0][0] == 0 and losses[-1][0] == 59, losses
    first, last = losses[0][1], min(l for _, l in losses[-2:])
    print("\nL1 on the real Lego targets:", " ".join(f"{i}:{l:.4f}" for i, l in losses))
    assert all(l == l and l < 1.0 for _, l in losses)          # finite, an L1 of colours in [0, 1]
    assert last < 0.85 * first, (first, last)


def test_reference_schedule_through_ten_density_control_calls(tmp_path):
    """The reference's own schedule (config.py:30-59: 5 000 initial points, densify every 100 iterations after 500, prune below
    opacity 0.005, opacity reset at iteration 0; loop train.py:920-1066, density control train.py:351-713) for 1 600 iterations at
    800 x 800 on the committed Lego views: ten density-control calls (iterations 600 .. 1500), each of which changes N and so
    re-sizes every workspace, the Adam moments, the shared zero dL_dcov3D tensor and the library's per-workspace notes.
    The full 7 000-iteration run of the same command is on file in profiles/r04_train_lego/ (final train L1 0.014, 27.5 dB).

    Bounds: the 7 000-iteration run's loss at iterations 1 500-1 600 averaged 0.043 (it still runs at a high learning rate there);
    this run decays its rate to 1 % by iteration 1 599 and measured a final 8-view L1 of 0.0265 / 22.9 dB (N 5 000 -> 1 110 at the
    first prune -> 7 749; profiles/r04_train_lego/test_1600_iterations.log) -- the bound leaves 1.7x."""
    import json
    import importlib
    import numpy as np
    out = tmp_path / "run"
    log = tmp_path / "train.jsonl"
    cmd = [sys.executable, os.path.join(ROOT, "examples", "train.py"), "--dataset", os.path.join(ROOT, "data", "lego"), "--views", "8",
           "--iterations", "1600", "--gaussians", "5000", "--print-interval", "100", "--log", str(log), "--output", str(out),
           "--save-interval", "800"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2500:])
    recs = [json.loads(l) for l in open(log)]
    calls = [r for r in recs if r["record"] == "density_control" and r["iteration"] > 0]
    summary = [r for r in recs if r["record"] == "summary"][0]
    curve = np.concatenate([np.asarray(r["l1"], np.float64) for r in recs if r["record"] == "loss"])
    print("\npoints after each density-control call:", " ".join(f"{r['iteration']}:{r['points']}" for r in calls))
    print(f"{summary['iterations_per_s']} iterations/s; final 8-view L1 {summary['train_l1_mean']:.5f}, PSNR {summary['train_psnr_mean']:.2f} dB; "
          f"loss first/last hundred {curve[:100].mean():.5f} / {curve[-100:].mean():.5f}")
    assert len(calls) >= 10 and [r["iteration"] for r in calls][:10] == list(range(600, 1600, 100))
    assert all(r["cloned"] + r["split"] > 0 for r in calls), calls                 # every call densified
    assert sum(r["pruned"] > 0 for r in calls) >= 8, calls                         # and nearly every one pruned
    pts = [r["points"] for r in calls]
    assert pts[-1] > 1.5 * min(pts) and sum(b > a for a, b in zip(pts[1:], pts[2:])) >= 7, pts   # N grows after the first big prune
    assert all(summary["parameters_finite"].values()), summary["parameters_finite"]
    assert curve.shape == (1600,) and np.isfinite(curve).all() and (curve < 1.0).all()
    assert curve[-100:].mean() < 0.6 * curve[:100].mean()
    assert summary["train_l1_mean"] < 0.045 and summary["train_psnr_mean"] > 21.0, summary
    # the checkpoints are PLY files a reader can take back in: same count as the model, finite rows
    gsr = importlib.import_module("3dgs-native_amd")
    ply = out / "point_cloud" / "iteration_1599" / "point_cloud.ply"
    assert ply.exists() and (out / "point_cloud" / "iteration_800" / "point_cloud.ply").exists()
    back = gsr.point_cloud.load_ply(str(ply))
    assert int(back["positions"].shape[0]) == summary["points_final"] == pts[-1]
    for k in ("positions", "scales", "rotations", "opacities", "shs"):
        assert np.isfinite(np.asarray(back[k])).all(), k
    assert (np.asarray(back["scales"]) >= 1e-3 - 1e-9).all() and (np.asarray(back["opacities"]) >= 0).all() and (np.asarray(back["opacities"]) <= 1).all()


def test_several_views_per_step_on_streams_match_the_serial_step():
    """A rank that renders several views per step (config #4 on fewer GPUs than views) issues them on separate HIP streams
    (dist.ViewStreams; examples/train.py --view-streams) and sums them in view order.  The same run with one stream must give the
    same training trajectory up to the float-atomic order of the blend backward: 40 iterations of three views per step on the real
    Lego targets, loss lines compared."""
    def run(streams):
        cmd = [sys.executable, os.path.join(ROOT, "examples", "train.py"), "--dataset", os.path.join(ROOT, "data", "lego"), "--views", "8",
               "--iterations", "40", "--gaussians", "5000", "--views-per-step", "3", "--view-streams", str(streams), "--print-interval", "5"]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        return [(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"iter\s+(\d+)\s+loss\s+([0-9.eE+-]+)", p.stdout)]
    serial, streamed = run(1), run(3)
    print("\nserial  :", " ".join(f"{i}:{l:.6f}" for i, l in serial))
    print("streamed:", " ".join(f"{i}:{l:.6f}" for i, l in streamed))
    assert [i for i, _ in serial] == [i for i, _ in streamed] and len(serial) >= 8
    for (i, a), (_, b) in zip(serial, streamed):
        assert abs(a - b) <= 2e-3 * max(a, b), (i, a, b)
    assert serial[-1][1] < 0.6 * serial[0][1]
