"""Image shapes the other parity tests do not reach: beyond 1080p, a 65 536-tile square (more tiles than the block-order limit,
16 tile bits), strips one tile high or three tiles wide, a single pixel, a 512 x 128 tile grid -- forward and backward against
the oracle under the usual contract (tests/parity.py)."""
import numpy as np
import pytest

from conftest import backward_kwargs, lego_camera, pkg, render_kwargs
import parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H,n,scale", [(2560, 1440, 3000, 0.03), (4096, 4096, 600, 0.05), (4095, 17, 400, 0.05), (33, 4000, 400, 0.05),
                                         (1, 1, 50, 0.2), (8192, 2048, 300, 0.02)])
def test_unusual_image_shapes(oracle, cameras, scenes, W, H, n, scale):
    gsr = pkg()
    sc = scenes.synthetic_scene(n, scale, 0.5, W + H)
    cam = lego_camera(cameras, frame=2, width=W, height=H)
    kw = render_kwargs(sc, cam, width=W, height=H)
    got, ref = gsr.render_gaussians(**kw), oracle.render_gaussians(**kw)
    parity.compare_forward(got, ref)
    dpix = (np.random.default_rng(1).normal(0.0, 1.0, (H, W, 3)) / (H * W * 3)).astype(np.float32)
    gb = gsr.backward(**backward_kwargs(sc, cam, kw, got[2], dpix))
    rb = oracle.backward(**backward_kwargs(sc, cam, kw, ref[2], dpix))
    parity.compare_backward(gb, rb)
    assert int(parity.to_np(got[2]["point_list"]).size) > 0
