"""The trainer's real inputs: `examples/train.py::load_nerf` on the committed
Lego views (data/lego, provenance in data/README.md) follows the reference's loader -- train.py:265-321 (poses through
utils/camera_utils.py), train.py:323-334 (imageio.imread -> float32 / 255 -> alpha channel dropped, nothing composited)."""
import importlib.util
import json
import os

import numpy as np

from conftest import ROOT, sub


def _train_module():
    spec = importlib.util.spec_from_file_location("gsr_example_train", os.path.join(ROOT, "examples", "train.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)          # __name__ != "__main__": no self-launch, no GPU call
    return mod


def test_load_nerf_follows_the_reference_loader():
    from PIL import Image
    cams, targets = _train_module().load_nerf(os.path.join(ROOT, "data", "lego"), 8)
    assert len(cams) == len(targets) == 8
    with open(os.path.join(ROOT, "data", "lego", "transforms_train.json")) as f:
        tf = json.load(f)
    cameras = sub("cameras")
    for k, (cam, tgt) in enumerate(zip(cams, targets)):
        png = np.asarray(Image.open(os.path.join(ROOT, "data", "lego", "train", f"r_{k}.png")))
        assert png.shape == (800, 800, 4) and png.dtype == np.uint8
        assert tgt.dtype == np.float32 and tgt.shape == (800, 800, 3) and tgt.flags["C_CONTIGUOUS"]
        # train.py:327-331: astype(float32) / 255.0, then [:, :, :3] -- the alpha channel is dropped, not composited
        np.testing.assert_array_equal(tgt, (png.astype(np.float32) / 255.0)[:, :, :3])
        assert (png[:, :, 3] == 0).any()      # there ARE transparent pixels: their stored RGB is used as it is (near black)
        want = cameras.nerf_camera(tf["frames"][k]["transform_matrix"], 800, 800, tf["camera_angle_x"])
        for key in ("world_to_camera", "full_proj_matrix", "camera_center"):
            np.testing.assert_array_equal(np.asarray(cam[key]), np.asarray(want[key]))
        assert cam["width"] == 800 and cam["height"] == 800
    # the poses are the first eight of the golden pose file the camera tests use
    with open(os.path.join(ROOT, "tests", "golden", "lego_train_poses.json")) as f:
        gold = json.load(f)
    assert tf["camera_angle_x"] == gold["camera_angle_x"]
    assert [fr["transform_matrix"] for fr in tf["frames"]] == [fr["transform_matrix"] for fr in gold["frames"]]
