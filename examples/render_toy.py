#!/usr/bin/env python3
"""
Counterpart of the reference's render.py: render its hard-coded 3-Gaussian
scene at 1800x1800 through render_gaussians() on the MI355X and save the image.  Unlike the reference it
writes the raw pixels (PIL, no matplotlib resampling or margin).

    python examples/render_toy.py [out.png]
"""
import importlib
import os
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gsr = importlib.import_module("3dgs-native_amd")


def main(out="example_render.png"):
    cam, sc = gsr.cameras.toy_camera(), gsr.scenes.toy_scene()
    image, depth, _ = gsr.render_gaussians(
        background=np.zeros(3, np.float32), means3D=sc["means"], colors=sc["colors"], opacity=sc["opacities"], scales=sc["scales"],
        rotations=sc["rotations"], scale_modifier=1.0, viewmatrix=cam["view_matrix"], projmatrix=cam["full_proj_matrix"],
        tan_fovx=cam["tan_fovx"], tan_fovy=cam["tan_fovy"], image_height=cam["height"], image_width=cam["width"], sh=sc["shs"],
        degree=3, campos=cam["camera_center"], prefiltered=False, antialiasing=False, clamped=True)
    arr = np.clip(image.cpu().numpy(), 0.0, 1.0)
    Image.fromarray((arr * 255.0 + 0.5).astype(np.uint8)).save(out)
    print(f"Rendered image saved to {out}")


if __name__ == "__main__":
    main(*sys.argv[1:2])
