"""A short run of the randomised parity campaign (tests/tools/fuzz.py): awkward shapes and sizes through the chain, the
occupancy step and the outlier removal, on the default paths and on the fallbacks."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env", [{}, {"PCOP_VOXEL_PATH": "sort", "PCOP_CELL_PATH": "hash"}], ids=["default", "fallbacks"])
def test_campaign(env):
    e = dict(os.environ, **env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "fuzz.py"), "400", "11"], env=e,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "mismatches 0" in out.stdout
