#!/usr/bin/env python3
"""Registers / LDS / occupancy per kernel, from hipcc's -Rpass-analysis=kernel-resource-usage (no GPU needed).
usage: tools/resources.py [substring ...]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "pointcloud_obstacle_processing_amd", "csrc")
flags = "-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt".split()
out = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", "pcop_capi.hip"],
                     cwd=csrc, capture_output=True, text=True).stderr
cur = None
rows = {}
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "").replace("pcop::", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print("%-34s %5s %5s %7s %8s %4s" % ("kernel", "vgpr", "sgpr", "scratch", "lds", "occ"))
for k, v in rows.items():
    if len(sys.argv) > 1 and not any(a in k for a in sys.argv[1:]):
        continue
    print("%-34s %5d %5d %7d %8d %4d" % (k[:34], v.get("VGPRs", -1), v.get("TotalSGPRs", -1), v.get("ScratchSize", -1), v.get("LDS Size", -1), v.get("Occupancy", -1)))
