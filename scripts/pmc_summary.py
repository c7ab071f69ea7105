"""Summarise a rocprofv3 --pmc (+ --kernel-trace) csv output directory: per kernel name, launches, mean duration
and the mean of every counter.  usage: python scripts/pmc_summary.py <dir> [name-filter] [--per-dispatch]"""
import csv, glob, os, sys, collections

args = [a for a in sys.argv[1:] if not a.startswith("--")]
per_dispatch = "--per-dispatch" in sys.argv
d = args[0]
flt = args[1] if len(args) > 1 else ""
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
disp = collections.defaultdict(dict)
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        disp[(r["Kernel_Name"], int(r["Dispatch_Id"]))][r["Counter_Name"]] = float(r["Counter_Value"])
        if "Start_Timestamp" in r and r.get("End_Timestamp"):
            dur[(r["Kernel_Name"], r["Dispatch_Id"])] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[(r["Kernel_Name"], r["Dispatch_Id"])] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
per = collections.defaultdict(list)
for (k, _), v in dur.items():
    per[k].append(v)
if per_dispatch:
    for (k, i) in sorted(disp, key=lambda x: x[1]):
        if flt and flt not in k:
            continue
        line = {"kernel": k[:40], "dispatch": i}
        if (k, str(i)) in dur:
            line["us"] = round(dur[(k, str(i))] / 1e3, 2)
        line.update({c: v for c, v in sorted(disp[(k, i)].items())})
        print(line)
    sys.exit(0)
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
