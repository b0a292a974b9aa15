#!/bin/bash
# rocprofv3 kernel stats of the default headline bench (production launches) -> gpurun_out/<tag>_kstats.csv
TAG=${1:-r03_default}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_dir -- python3 bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-kernel-timing --no-secondary "$@" > gpurun_out/${TAG}_kstats.log 2>&1
cp gpurun_out/${TAG}_dir/*/*kernel_stats.csv gpurun_out/${TAG}_kstats.csv
rm -rf gpurun_out/${TAG}_dir
python3 - gpurun_out/${TAG}_kstats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    print("%9.2f ms %6d calls avg %9.1f us %5.1f %%  %s" % (float(r["TotalDurationNs"])/1e6, int(r["Calls"]), float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot, r["Name"][:70]))
print("total %.1f ms" % (tot/1e6))
PY
grep -h '^{"metric"' gpurun_out/${TAG}_kstats.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
