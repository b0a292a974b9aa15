"""Overlap structure of a rocprofv3 --kernel-trace CSV: per kernel family the summed duration, and how much wall time had 1, 2, 3+ kernels in flight.
usage: python tools/ktrace_overlap.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
fam = collections.defaultdict(lambda: [0, 0.0])
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    fam[name][0] += 1; fam[name][1] += (e - s) / 1e6
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, 0, collections.defaultdict(float)
for t, d in ev:
    hist[depth] += (t - last) / 1e6
    depth += d; last = t
print("wall %.1f ms" % (last / 1e6))
for k in sorted(hist): print("  %d kernels in flight: %8.1f ms" % (k, hist[k]))
for n, (c, ms) in sorted(fam.items(), key=lambda x: -x[1][1])[:8]: print("  %-50s %6d calls %9.1f ms" % (n, c, ms))
