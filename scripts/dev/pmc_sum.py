"""scripts/dev/pmc_sum.py <dir> [kernel substring]: mean of every counter per dispatch of the named kernel in a rocprofv3 --pmc output dir"""
import csv, glob, sys, collections
pat = sys.argv[2] if len(sys.argv) > 2 else "k_pure_vle"
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in per.items():
        acc[c].append(v)
for c, v in sorted(acc.items()):
    print(f"{c:28s} n {len(v)} mean {sum(v)/len(v):.6g}")
# kernel durations from the trace of the same run
import re
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    d = [ (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
    if d: print(f"{'duration_ms':28s} n {len(d)} mean {sum(d)/len(d):.4f}")
