"""Per-launch durations of one kernel family from a rocprofv3 --kernel-trace CSV, in launch order: python tools/ktrace_dump.py <csv> <name substring> [first] [count]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
count = int(sys.argv[4]) if len(sys.argv) > 4 else len(rows)
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[first:first + count]]
print(len(rows), "launches; us:", " ".join(f"{x:.0f}" for x in d))
print("sum %.1f ms  min %.1f us  median %.1f us" % (sum(d) / 1e3, min(d), sorted(d)[len(d) // 2]))
