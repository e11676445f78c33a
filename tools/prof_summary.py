"""Per-kernel summary of a rocprofv3 kernel trace of a batched bench run: average launch duration and per-frame share."""
import collections, csv, json, sys
d = sys.argv[1]; batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = list(csv.DictReader(open(d + "/trace_kernel_trace.csv")))
acc = collections.defaultdict(list)
for r in rows:
    if r["Grid_Size_Y"] == str(batch):
        acc[r["Kernel_Name"][:64]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
acc = {k: v for k, v in acc.items() if "pcop::k_" in k and "probe" not in k and "unpack" not in k}
import statistics
nb = int(statistics.median(len(v) for v in acc.values()))   # launches of a kernel that runs once per batch (a kernel seen only while the plan settles counts less)
acc = {k: v for k, v in acc.items() if 2 * len(v) >= nb}   # (a kernel seen only while the plan settles is left out)
items = sorted(((k, sum(v) / len(v), max(1, round(len(v) / nb))) for k, v in acc.items()), key=lambda t: -t[1] * t[2])
tot = 0.0
print("%-66s %10s %4s %12s" % ("kernel (batch of %d frames per launch)" % batch, "avg_us", "x", "us/frame"))
for k, a, c in items:
    print("%-66s %10.2f %4d %12.2f" % (k, a, c, a * c / batch)); tot += a * c
print("total per batch %.1f us, per frame %.1f us" % (tot, tot / batch))
