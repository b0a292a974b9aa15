#!/bin/bash
# tools/kstats.sh <tag> [bench args]: rocprofv3 kernel stats (serial lanes) of the headline bench under the current environment -> gpurun_out/<tag>_kstats.csv
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export LUPIN_LANES=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_kstats_dir -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary "$@" > gpurun_out/${TAG}_kstats.log 2>&1
cp gpurun_out/${TAG}_kstats_dir/*/*kernel_stats.csv gpurun_out/${TAG}_kstats.csv
rm -rf gpurun_out/${TAG}_kstats_dir
python3 - gpurun_out/${TAG}_kstats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms  {int(r["Calls"]):5d} calls  {float(r["Percentage"]):5.1f} %  {r["Name"][:110]}')
PY
