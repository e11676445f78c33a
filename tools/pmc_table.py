"""Per-kernel averages of the counters of one or more rocprofv3 --pmc runs.  usage: pmc_table.py <dir> [<dir> ...]
Only the batched launches (the most frequent grid of each kernel) are averaged."""
import csv, collections, glob, sys
for d in sys.argv[1:]:
    f = (glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    gmax = collections.defaultdict(int)
    for r in rows:
        gmax[r["Kernel_Name"]] = max(gmax[r["Kernel_Name"]], int(r["Grid_Size"]))
    for r in rows:
        if int(r["Grid_Size"]) != gmax[r["Kernel_Name"]]: continue
        k = r["Kernel_Name"].split("(")[0].replace("void pcop::", "").replace("pcop::", "")
        if not k.startswith("k_"): continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    names = sorted({c for k in agg for c in agg[k]})
    print(d)
    print("%-22s %8s " % ("kernel", "us") + " ".join("%14s" % c[:14] for c in names))
    for k in agg:
        print("%-22s %8.1f " % (k[:22], sum(dur[k]) / len(dur[k])) + " ".join("%14.0f" % (sum(agg[k][c]) / len(agg[k][c])) if c in agg[k] else " " * 14 for c in names))
