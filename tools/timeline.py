"""Timeline figures from a rocprofv3 --kernel-trace CSV of a bench run: per queue the busy fraction, per heavy kernel the mean duration and
how much of its run time another heavy kernel was running too, and the gaps between consecutive heavy kernels.
usage: python tools/timeline.py <kernel_trace.csv> [lo_frac hi_frac]"""
import csv
import sys
from collections import defaultdict

HEAVY = ("k_bin_hist", "k_bin_scatter", "k_bucket_reduce", "k_voxel_gather")

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1].replace("void ", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), name))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.9
hs = sorted(r[0] for r in rows if r[3] in HEAVY)   # the window: between two quantiles of the heavy dispatches (the timed region holds most of them)
a, b = hs[int(lo * (len(hs) - 1))], hs[int(hi * (len(hs) - 1))]
win = [r for r in rows if r[0] >= a and r[1] <= b]
span = b - a
print(f"{len(rows)} dispatches over {(t1 - t0) / 1e6:.1f} ms; window {span / 1e6:.1f} ms with {len(win)} dispatches")

def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None or s > ce:
            if cs is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if cs is not None:
        tot += ce - cs
    return tot

byq = defaultdict(list)
for s, e, q, n in win:
    byq[q].append((s, e))
for q, iv in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    print(f"queue {q}: {len(iv)} dispatches, busy {union(iv) / span:.2f} of the window")
print(f"any kernel running: {union([(s, e) for s, e, _, _ in win]) / span:.2f}")
heavy = [(s, e, q, n) for s, e, q, n in win if n in HEAVY]
print(f"a heavy kernel running: {union([(s, e) for s, e, _, _ in heavy]) / span:.2f}; sum of heavy durations / window: {sum(e - s for s, e, _, _ in heavy) / span:.2f}")
light = [(s, e) for s, e, q, n in win if n not in HEAVY]
print(f"a light kernel running: {union(light) / span:.2f}; sum of light durations / window: {sum(e - s for s, e in light) / span:.2f}")
dur = defaultdict(list)
for s, e, q, n in win:
    dur[n].append(e - s)
print("kernel                 launches   mean us   sum/window")
for n, d in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:16]:
    print(f"  {n:22s} {len(d):6d} {sum(d) / len(d) / 1e3:9.1f} {sum(d) / span:9.3f}")
# gaps between consecutive heavy kernels (end of one to the start of the next one that starts after it ended)
heavy.sort()
gaps = []
for i in range(1, len(heavy)):
    g = heavy[i][0] - max(h[1] for h in heavy[max(0, i - 4):i])
    gaps.append(g)
if gaps:
    pos = [g for g in gaps if g > 0]
    print(f"gaps between heavy kernels: {len(pos)} of {len(gaps)} positive, mean {sum(pos) / max(1, len(pos)) / 1e3:.1f} us, total {sum(pos) / span:.2f} of the window; overlapping starts: {len(gaps) - len(pos)}")
