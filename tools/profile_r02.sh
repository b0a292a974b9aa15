#!/bin/bash
# Round-2 evidence on the GPU box: tools/profile_r02.sh <tag>
#   gpurun_out/<tag>_stats_serial/   rocprofv3 --kernel-trace --stats of the headline bench, LUPIN_LANES=1 (kernels one at a time)
#   gpurun_out/<tag>_pmc{1..5}/      PMC passes of the same command (tools/pmc_passes.sh), folded by tools/pmc_summary.py
#   gpurun_out/<tag>_calib*.{log,d}  FETCH_SIZE calibration on this build's access patterns (tools/calib/fetch_calib)
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary"
export LUPIN_LANES=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats_serial -- $BENCH > gpurun_out/${TAG}_stats_serial.log 2>&1
unset LUPIN_LANES
cp gpurun_out/${TAG}_stats_serial/*/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats_serial.csv
tools/pmc_passes.sh ${TAG}_pmc $BENCH
UNITS=$(grep -h '^{"metric"' gpurun_out/${TAG}_pmc1.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['path_bounces']*1.5)")
python3 tools/pmc_summary.py $UNITS gpurun_out/${TAG}_pmc.json gpurun_out/${TAG}_pmc1 gpurun_out/${TAG}_pmc2 gpurun_out/${TAG}_pmc3 gpurun_out/${TAG}_pmc4 gpurun_out/${TAG}_pmc5 > gpurun_out/${TAG}_pmc_derived.json
for k in k_stream k_gather64 k_gather48; do
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_calib_$k --pmc FETCH_SIZE TCC_HIT -- tools/calib/fetch_calib $k > gpurun_out/${TAG}_calib_$k.log 2>&1
done
python3 - <<PY
import csv, glob, json
for k in ("k_stream", "k_gather64", "k_gather48"):
    req = [json.loads(l) for l in open(f"gpurun_out/${TAG}_calib_{k}.log") if l.startswith("{")][0]
    tot = {}
    for f in glob.glob(f"gpurun_out/${TAG}_calib_{k}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(k):
                tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    print(json.dumps({"kernel": k, "requested_bytes": req["requested_bytes"], "GBps": req["GBps"], "FETCH_SIZE_bytes": tot.get("FETCH_SIZE", 0) * 1024,
                      "ratio_requested_over_fetch": req["requested_bytes"] / max(1.0, tot.get("FETCH_SIZE", 0) * 1024), "TCC_HIT": tot.get("TCC_HIT")}))
PY
