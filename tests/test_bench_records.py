"""The numbers on the bench line that come from tracked files, and the tracked bench lines themselves, checked on every CPU run:
bench.py's parsers find the dominant kernel in the newest tracked PMC passes and kernel-trace summaries (a renamed kernel or a new
column would otherwise drop `traffic` / `frac_solo_tracked` / `frac_shared` from the line without a sound), and every ratio on a tracked
line follows from the figures next to it - what a reader recomputes by hand (algorithmic bytes = 12 N + 28 M a frame)."""
import glob
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(pattern):
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", pattern)):
        m = re.match(r"r(\d+)_", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    return best[1] if best else None


def tracked_lines():
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1*.json"))):
        if int(re.match(r"r(\d+)_", os.path.basename(f)).group(1)) < 4:   # (the line has carried these fields since round 4)
            continue
        rows = [ln for ln in open(f).read().splitlines() if ln.startswith("{")]
        if rows:
            out.append((os.path.basename(f), json.loads(rows[-1])))
    return out


def test_parsers_find_the_dominant_kernel_in_the_tracked_files():
    import bench
    line = json.loads([ln for ln in open(newest("r*_bench_n1.json")).read().splitlines() if ln.startswith("{")][-1])
    kernel, batch = line["roofline"]["kernel"], line["roofline"]["frames_per_launch"]
    traffic, src = bench.pmc_traffic_of(kernel, batch)
    assert traffic and traffic > 0 and "pmc_traffic.json" in src
    solo, src_solo = bench.tracked_launch_us(kernel, batch, True)
    shared, src_shared = bench.tracked_launch_us(kernel, batch, False)
    assert solo and shared and 0 < solo <= shared, (solo, shared)   # (sharing the GPU never shortens a launch)
    assert src_solo.endswith("_solo.txt") and not src_shared.endswith("_solo.txt")
    # the traffic of a kernel that reads and writes the cloud once is at least its own algorithmic bytes
    assert traffic >= 0.9 * line["roofline"]["kernel_own_alg_bytes_per_launch"]


@pytest.mark.parametrize("name,line", tracked_lines())
def test_every_ratio_on_a_tracked_line_follows_from_its_figures(name, line):
    r = line["roofline"]
    n = 1_000_000
    assert line["config"]["workload"].startswith("C3: %d-pt" % n)
    assert r["alg_bytes_per_frame"] == 12 * n + 28 * line["config"]["voxels_per_frame"]
    assert r["alg_bytes_per_launch"] == r["alg_bytes_per_frame"] * r["frames_per_launch"]
    assert r["peak"] == 8000.0 and r["unit"] == "GB/s" and r["bound"] == "hbm"
    assert r["achieved"] == pytest.approx(r["alg_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9, rel=1e-6)
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    assert r["achieved"] <= r["peak"]
    if r.get("traffic"):
        assert r["traffic_over_alg"] == pytest.approx(r["traffic"] / r["alg_bytes_per_launch"], rel=1e-6)
    for key_us, key_frac in (("avg_launch_us_solo_tracked", "frac_solo_tracked"), ("avg_launch_us_shared_tracked", "frac_shared")):
        if r.get(key_us):
            assert r[key_frac] == pytest.approx(r["alg_bytes_per_launch"] / (r[key_us] * 1e-6) / 1e9 / r["peak"], rel=1e-6)
    # the line itself: frames of a step over the time of a step, and the dominant kernel's launches fit inside a step
    assert line["value"] == pytest.approx(line["config"]["frames_per_step"] / (line["ms_per_step"] * 1e-3), rel=1e-6)
    launches_per_step = line["config"]["frames_per_step"] / r["frames_per_launch"]
    assert launches_per_step * r["avg_launch_ms"] / line["config"]["contexts_per_gpu"] <= line["ms_per_step"]
    fr = line["frame_roofline"]
    assert fr["frac_pipelined"] == pytest.approx(r["alg_bytes_per_frame"] * line["value"] / 8e12, rel=1e-6)
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1
    assert line["vs_baseline"] is None and line["dtype"] == "f32" and line["scaling"] == "weak"
