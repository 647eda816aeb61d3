"""The code paths that sizes pick -- 64-bit tile items (tile bits + id bits > 32: e.g. 1080p with 5 M Gaussians) and the
4096-item radix chunks of the depth sort (N > 4 M), the scanned super-block rows of radix passes over more than 2048 blocks
(D > 8.4 M) -- only run at sizes the oracle cannot replay.  GSR_DEBUG bits 5, 6 and 7
force them at any size, so the oracle comparison covers them too; bit 8 makes the depth sort run all four of its 8-bit passes
whatever the frame's depth range (by default the device decides: three for a scene within two octaves of depth); bit 9 the
expansion by Gaussian with its device-wide offset scan and separate first histogram (the product path expands by output block);
bit 10 the multi-kernel depth stage for small scenes too.  The library reads GSR_DEBUG once, hence a subprocess."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flags", [32, 64, 96, 128, 224, 256, 512, 1024, 1248, 2016])
def test_parity_with_forced_paths(flags):
    env = dict(os.environ, GSR_DEBUG=str(flags), GSR_FUZZ_CASES="48", GSR_NEEDLE_CASES="4")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_fuzz.py"),
                        "-k", "not png and not c2_lego and not workspace and not full_size"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_parity_with_the_plain_tile_dispatch():
    """GSR_FWD_NO_ORDER=1: the forward blend dispatches its tiles row-major instead of by last frame's cost classes (the product's
    default since round 4).  Same outputs either way; the parity and fuzz suites run once more with the plain order."""
    env = dict(os.environ, GSR_FWD_NO_ORDER="1", GSR_FUZZ_CASES="48", GSR_NEEDLE_CASES="4")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_fuzz.py"), os.path.join(ROOT, "tests", "test_gpu_properties.py"),
                        "-k", "not png and not full_size"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("px", [16, 32, 64])
def test_parity_with_each_backward_block_size(px):
    """The backward blend picks 8x4 or 8x8 pixel blocks per frame from its tile pairs per Gaussian (D >= 24 N: 8x8); GSR_BWD_BLOCK
    forces one size (4x4 too) for the whole process, so each kernel meets every parity and fuzz case, not only the frames that
    would choose it."""
    env = dict(os.environ, GSR_BWD_BLOCK=str(px), GSR_FUZZ_CASES="48", GSR_NEEDLE_CASES="4")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_fuzz.py"),
                        "-k", "not png and not full_size"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("narrowing", [True, False])
def test_wide_tile_items_over_two_passes(narrowing):
    """Tile id bits + Gaussian id bits beyond 32 (1080p with 5 M Gaussians: 13 + 23) make the tile partition run on 64-bit items; with
    two passes the first one then writes 32-bit items and the last recovers the first digit from each item's position
    (GSR_NO_NARROWING=1: it keeps 64-bit items, the path of rounds 1-3).  GSR_DEBUG bit 5 forces the wide items at any size, the
    unusual image shapes (more than 256 tiles: two passes) supply the sizes; forward and backward against the oracle."""
    env = dict(os.environ, GSR_DEBUG="32", GSR_NO_NARROWING="0" if narrowing else "1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_odd_sizes.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_depth_sort_with_plain_items_through_every_pass():
    """A frame whose depth keys need two or three passes (most frames) and whose tile grid fits 6 bits per coordinate gets PACKED depth
    items from the first active pass on (key remainder | rectangle | id: the last pass unpacks instead of gathering rectangles by id).
    GSR_NO_DEPTH_PACK=1 keeps the plain items and the gather (the path of rounds 1-3, still taken by four-pass frames and large
    grids); with GSR_DEBUG bit 10 the multi-kernel depth stage also serves the small fuzz scenes.  Both against the oracle."""
    for pack in ("1", "0"):
        env = dict(os.environ, GSR_NO_DEPTH_PACK=pack, GSR_DEBUG="1024", GSR_FUZZ_CASES="96", GSR_NEEDLE_CASES="4")
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                            os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_fuzz.py"),
                            "-k", "not png and not workspace and not full_size"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, pack + r.stdout[-3000:] + r.stderr[-2000:]
        assert " passed" in r.stdout


def test_two_training_steps_with_the_forward_xcd_map():
    """GSR_FWD_XCD=1 (neighbouring tiles of the forward blend on one XCD) cannot host the spare workgroups that clear the backward's
    accumulators; the forward must then clear them another way, or the SECOND step's gradients would carry the first step's sums
    (ADVICE r3).  forward, backward, forward, backward in a fresh interpreter with the map on; both steps against the oracle."""
    code = r'''
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import conftest, parity
from oracle import oracle
gsr = importlib.import_module("3dgs-native_amd")
bwd = importlib.import_module("3dgs-native_amd.backward").backward
sc = gsr.scenes.synthetic_scene(5000, 0.05, 0.6, 33)
cam = gsr.cameras.nerf_camera(gsr.scenes.LEGO_FRAME0, 176, 144, gsr.scenes.LEGO_CAMERA_ANGLE_X)
kw = conftest.render_kwargs(sc, cam, width=176, height=144)
ref = oracle.render_gaussians(**kw)
skipped = []
for step in range(3):
    dpix = (np.random.default_rng(step).normal(0, 1, (144, 176, 3)) / (144 * 176 * 3)).astype(np.float32)
    buf = gsr.render_gaussians(**kw)[2]
    g = gsr.backward(**conftest.backward_kwargs(sc, cam, kw, buf, dpix))
    skipped.append(bool(bwd.last_call_skipped_the_clear))
    parity.compare_backward(g, oracle.backward(**conftest.backward_kwargs(sc, cam, kw, ref[2], dpix)))
assert skipped == [False, True, True], skipped
print("ok", skipped)
'''
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, GSR_FWD_XCD="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
