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
