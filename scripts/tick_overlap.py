"""How the kernels of the slot groups' ticks share the chip, from a rocprofv3 --kernel-trace csv directory of
scripts/mcts_c3_ticks.py: over the last 60 % of the trace -- per kernel the mean duration, per stream the gap between the
end of one kernel and the start of the next, and how much of the time k towers / k search kernels were running at once.
usage: python scripts/tick_overlap.py <dir>"""
import csv, glob, os, re, sys, collections
import numpy as np
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
        if name.startswith("k_trunk") or name.startswith("k_expand_select"):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "tower" if name.startswith("k_trunk") else "search",
                         r.get("Queue_Id", r.get("Stream_Id", "0"))))
rows.sort()
t_lo = rows[0][0] + 0.4 * (rows[-1][1] - rows[0][0])
rows = [r for r in rows if r[0] >= t_lo]
span = (rows[-1][1] - rows[0][0]) / 1e3
print("window %.1f ms, %d kernels, %d queues" % (span / 1e3, len(rows), len({r[3] for r in rows})))
for kind in ("tower", "search"):
    d = np.array([(r[1] - r[0]) / 1e3 for r in rows if r[2] == kind])
    print("%-7s launches %6d  mean %6.1f us  median %6.1f  p95 %6.1f   sum / window = %.2f" %
          (kind, len(d), d.mean(), np.median(d), np.percentile(d, 95), d.sum() / span))
per_q = collections.defaultdict(list)
for r in rows:
    per_q[r[3]].append(r)
gaps = {"tower->search": [], "search->tower": []}
for q, rs in per_q.items():
    for a, b in zip(rs, rs[1:]):
        if a[2] != b[2]:
            gaps["%s->%s" % (a[2], b[2])].append((b[0] - a[1]) / 1e3)
for k, g in gaps.items():
    g = np.array(g)
    print("gap %-14s mean %5.1f us  median %5.1f  p95 %5.1f" % (k, g.mean(), np.median(g), np.percentile(g, 95)))
ticks = min(sum(1 for r in rs if r[2] == "search") for rs in per_q.values())
print("round of ticks (window / ticks of one queue): %.1f us" % (span / ticks))
# concurrency histogram
ev = []
for s, e, kind, q in rows:
    ev.append((s, kind, 1)); ev.append((e, kind, -1))
ev.sort()
cur = {"tower": 0, "search": 0}
hist = collections.Counter()
last = ev[0][0]
for t, kind, d in ev:
    hist[(cur["tower"], cur["search"])] += t - last
    last = t
    cur[kind] += d
tot = sum(hist.values())
print("share of the window with (towers, search kernels) running:")
for (a, b), v in sorted(hist.items()):
    if v / tot >= 0.005:
        print("   %d towers, %d search: %5.1f %%" % (a, b, 100.0 * v / tot))
for kind, idx in (("tower", 0), ("search", 1)):
    m = sum(k[idx] * v for k, v in hist.items()) / tot
    print("mean %s kernels in flight: %.2f" % (kind, m))
