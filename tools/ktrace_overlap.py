"""Overlap structure of a rocprofv3 --kernel-trace CSV: per kernel family the summed duration, how much wall time had 1, 2, 3+ kernels
in flight, and -- inside the steady-state window (middle 60 % of the trace's busy span) -- the distribution of (tracer kernels,
other kernels) in flight.
usage: python tools/ktrace_overlap.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
fam = collections.defaultdict(lambda: [0, 0.0])
t0 = min(int(r["Start_Timestamp"]) for r in rows)
big = [r for r in rows if "k_extend" in r["Kernel_Name"] or "k_shade" in r["Kernel_Name"]]
lo = min(int(r["Start_Timestamp"]) for r in big) - t0
hi = max(int(r["End_Timestamp"]) for r in big) - t0
# the run's last burst of frames: everything after the longest idle gap
big.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    fam[name][0] += 1; fam[name][1] += (e - s) / 1e6
    cls = 0 if "k_extend" in name else 1
    ev.append((s, 1, cls)); ev.append((e, -1, cls))
ev.sort()
depth, last, hist = 0, 0, collections.defaultdict(float)
cnt = [0, 0]
joint = collections.defaultdict(float)
span = hi - lo
w0, w1 = hi - 0.5 * span, hi - 0.1 * span     # a window inside the timed steps (the trace ends with them)
for t, d, cls in ev:
    hist[depth] += (t - last) / 1e6
    a, b = max(last, w0), min(t, w1)
    if b > a: joint[(min(cnt[0], 4), min(cnt[1], 4))] += (b - a) / 1e6
    depth += d; cnt[cls] += d; last = t
print("wall %.1f ms" % (last / 1e6))
for k in sorted(hist): print("  %d kernels in flight: %8.1f ms" % (k, hist[k]))
tot = sum(joint.values())
print("steady-state window %.1f ms: share of time with (tracer kernels, other kernels) in flight" % tot)
for k in sorted(joint, key=lambda k: -joint[k])[:10]: print("  tracers %d others %d: %5.1f %%" % (k[0], k[1], 100 * joint[k] / tot))
for n, (c, ms) in sorted(fam.items(), key=lambda x: -x[1][1])[:8]: print("  %-50s %6d calls %9.1f ms" % (n, c, ms))
