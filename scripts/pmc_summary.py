"""Summarise a rocprofv3 --pmc (+ --kernel-trace) csv output directory: per kernel name, launches, mean duration
and the mean of every counter.  usage: python scripts/pmc_summary.py <dir> [name-filter]"""
import csv, glob, os, sys, collections

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r and r.get("End_Timestamp"):
            dur[(r["Kernel_Name"], r["Dispatch_Id"])] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[(r["Kernel_Name"], r["Dispatch_Id"])] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
per = collections.defaultdict(list)
for (k, _), v in dur.items():
    per[k].append(v)
for k in sorted(cnt):
    if flt and flt not in k:
        continue
    n = max(len(v) for v in cnt[k].values())
    line = {"kernel": k[:70], "launches": n}
    if per.get(k):
        line["mean_us"] = round(sum(per[k]) / len(per[k]) / 1e3, 2)
    for c, v in sorted(cnt[k].items()):
        line[c] = round(sum(v) / len(v), 1)
    print(line)
