"""Print the kernel timeline of a few search ticks from a rocprofv3 --kernel-trace csv directory.
usage: python scripts/tick_timeline.py <dir>"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Stream_Id", r.get("Queue_Id", ""))))
rows.sort()
sel = [i for i, r in enumerate(rows) if "k_select" in r[2]]
i0 = sel[len(sel) // 2]
t0 = rows[i0][0]
for r in rows[i0:i0 + 14]:
    print("%9.1f us -> %9.1f us  (%6.1f)  q=%s  %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[2]))
