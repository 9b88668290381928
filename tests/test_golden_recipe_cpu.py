"""The pin is only as good as the recipe is runnable: where the reference is present (the build container; never the GPU
box), ``python tests/golden/make_golden.py`` with NO arguments must exit 0 and reproduce every committed fixture bit for bit."""
import filecmp
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/COCO"), reason="the reference only exists in the build container")
def test_no_argument_recipe_regenerates_every_fixture_bit_identically(tmp_path):
    env = dict(os.environ, COCODR_GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden.py")], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    committed = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))
    assert len(committed) >= 15
    for f in committed:
        g = os.path.join(str(tmp_path), os.path.basename(f))
        assert os.path.exists(g), f"the recipe did not write {os.path.basename(f)}"
        assert filecmp.cmp(f, g, shallow=False), f"{os.path.basename(f)} differs from what the recipe produces"
