#!/bin/bash
# Round-2 evidence on the GPU box: tools/profile_r02.sh <tag>  ->  gpurun_out/<tag>_*; copy the summaries into profiles/.
#   <tag>_bench.json                 the default bench line (all legs)
#   <tag>_stats_serial/              rocprofv3 --kernel-trace --stats of the headline workload, LUPIN_LANES=1 (kernels one at a
#                                    time: the average durations bench.py's hipEvent pass must agree with)
#   <tag>_pmc{1..5}/                 PMC passes of the same command (tools/pmc_passes.sh)
#   <tag>_calib.jsonl                FETCH_SIZE calibration on this build's access patterns (tools/calib/fetch_calib)
#   profiles/r02_pmc_traffic.json    folded by tools/pmc_traffic.py (bench.py reads it)
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary"
export LUPIN_LANES=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats_serial -- $BENCH > gpurun_out/${TAG}_stats_serial.log 2>&1
unset LUPIN_LANES
cp gpurun_out/${TAG}_stats_serial/*/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats_serial.csv
rm -f gpurun_out/${TAG}_calib.jsonl
for k in k_stream k_gather64 k_gather48; do
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_calib_$k --pmc FETCH_SIZE TCC_HIT -- tools/calib/fetch_calib $k > gpurun_out/${TAG}_calib_$k.log 2>&1
  python3 - $k ${TAG} >> gpurun_out/${TAG}_calib.jsonl <<'PY'
import csv, glob, json, sys
k, tag = sys.argv[1], sys.argv[2]
req = [json.loads(l) for l in open(f"gpurun_out/{tag}_calib_{k}.log") if l.startswith("{")][0]
tot = {}
for f in glob.glob(f"gpurun_out/{tag}_calib_{k}/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith(k):
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print(json.dumps({"kernel": k, "requested_bytes": req["requested_bytes"], "GBps": req["GBps"], "FETCH_SIZE_bytes": tot.get("FETCH_SIZE", 0) * 1024,
                  "ratio_requested_over_fetch": req["requested_bytes"] / max(1.0, tot.get("FETCH_SIZE", 0) * 1024)}))
PY
done
profile_workload() {  # key, bench args...
  KEY=$1; shift
  tools/pmc_passes.sh ${TAG}_pmc_${KEY}_ python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary "$@"
  UNITS=$(grep -h '^{"metric"' gpurun_out/${TAG}_pmc_${KEY}_1.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['path_bounces']*1.5)")
  python3 tools/pmc_traffic.py $KEY $UNITS gpurun_out/${TAG}_calib.jsonl gpurun_out/${TAG}_pmc_${KEY}_1 gpurun_out/${TAG}_pmc_${KEY}_2 gpurun_out/${TAG}_pmc_${KEY}_3 gpurun_out/${TAG}_pmc_${KEY}_4 gpurun_out/${TAG}_pmc_${KEY}_5 > gpurun_out/${TAG}_pmc_${KEY}_derived.json
}
profile_workload bistro_class_3840x2160_b16_spp8_standard
profile_workload cornellbox_1024x1024_b8_spp8_standard --scene cornellbox --width 1024 --height 1024 --bounces 8
profile_workload materials1_1920x1080_b12_spp8_standard --scene materials1 --width 1920 --height 1080 --bounces 12
profile_workload environments1_1920x1080_b16_spp8_standard --scene environments1 --width 1920 --height 1080 --bounces 16
cat gpurun_out/${TAG}_bench.json | cut -c1-400
