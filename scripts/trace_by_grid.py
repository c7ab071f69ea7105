"""Summarise a rocprofv3 --kernel-trace csv by (kernel, grid size): calls, mean / median / p95 duration.  The search kernels
run at several sizes in one bench run (C3: 256 slots per group, C5: 1024, C1: 1), so the per-kernel average of --stats mixes
them; this is the table the bench line's tick_kernels are compared with.   usage: python scripts/trace_by_grid.py <dir>"""
import csv, glob, os, re, sys, collections
import numpy as np
d = sys.argv[1]
per = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
        if not name.startswith("k_"):
            continue
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
        per[(name, grid // max(wg, 1), wg)].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
print("%-28s %8s %6s %8s %10s %10s %10s" % ("kernel", "blocks", "wg", "calls", "mean us", "median us", "p95 us"))
for (name, blocks, wg), v in sorted(per.items()):
    v = np.array(v)
    print("%-28s %8d %6d %8d %10.2f %10.2f %10.2f" % (name, blocks, wg, len(v), v.mean(), np.median(v), np.percentile(v, 95)))
