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
    e = dict(os.environ, FUZZ_CHECK_CLEAN="1", **env)   # (after every call: occupancy flags zero, cell table empty)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "fuzz.py"), "400", "11"], env=e,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "mismatches 0" in out.stdout and "DIRTY" not in out.stdout


def test_campaign_segment_that_left_occupancy_flags_behind():
    """Cases 400-559 of seed 32: a batch of two where one frame needed more occupancy tiles than the launch had and the other
    lay outside the context's box - two re-plans in one attempt, the second attempt with another cell grid - left seven flag
    bytes set, and 137 cases later a frame whose grid covered them had cells without points (a memory fault in k_cgrid_hook).
    Every re-run now starts on clean flags; the campaign checks the flags and the cell table after every call."""
    e = dict(os.environ, FUZZ_CHECK_CLEAN="1", FUZZ_START="400")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "fuzz.py"), "560", "32"], env=e,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "mismatches 0" in out.stdout and "DIRTY" not in out.stdout


@pytest.mark.parametrize("env", [{}, {"PCOP_VOXEL_BOX": "0"}], ids=["default", "bounds-pass"])
def test_batches_campaign(env):
    """tests/tools/fuzz_batches.py: batches of two to five unlike frames on one long-lived context, parameters changing now and
    then, frames outside the context's box, NaN holes, empty frames; flags and cell tables of every slot checked after every batch."""
    e = dict(os.environ, FUZZ_CHECK_CLEAN="1", **env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "fuzz_batches.py"), "80", "7"], env=e,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "mismatches 0" in out.stdout and "DIRTY" not in out.stdout
