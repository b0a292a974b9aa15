#!/bin/bash
# tools/r03_iter.sh <tag>: per-launch durations of the tracing kernels of ONE frame (serial lanes), in launch order -> gpurun_out/<tag>_iter.txt
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export LUPIN_LANES=1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_kt -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary "$@" > gpurun_out/${TAG}_kt.log 2>&1
python3 - gpurun_out/${TAG}_kt/*/*kernel_trace.csv > gpurun_out/${TAG}_iter.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ext = [r for r in rows if "k_extend_persistent" in r["Kernel_Name"]]
# second frame only (the first is the warm-up)
def split(rs, retr):
    return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rs if (("false, true>" in r["Kernel_Name"]) == retr)]
main, retr = split(ext, False), split(ext, True)
h = len(main) // 2
print("main", " ".join(f"{x:.0f}" for x in main[h:]))
if retr: print("retrace", " ".join(f"{x:.0f}" for x in retr[len(retr) // 2:]))
print("sum_main_ms", sum(main[h:]) / 1e3, "sum_retrace_ms", sum(retr[len(retr) // 2:]) / 1e3 if retr else 0)
PY
rm -rf gpurun_out/${TAG}_kt
cat gpurun_out/${TAG}_iter.txt
