#!/bin/bash
# Round-3 evidence on the GPU box: tools/profile_r03.sh  ->  gpurun_out/r03_*; the summaries are then copied into profiles/.
#   r03_bench.json                  the default bench line (all legs)
#   r03_kernel_stats_serial.csv     rocprofv3 --kernel-trace --stats of the headline workload with one lane (kernels one at a time,
#                                   production batches: the average durations bench.py's hipEvent pass must agree with)
#   r03_pmc_traffic.json            PMC passes of the four workloads folded by tools/pmc_traffic.py (bench.py reads it from profiles/)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
echo "bench done"
BENCH="python3 bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-kernel-timing --no-secondary"
LUPIN_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_stats_serial -- $BENCH > gpurun_out/r03_stats_serial.log 2>&1
cp gpurun_out/r03_stats_serial/*/*kernel_stats.csv gpurun_out/r03_kernel_stats_serial.csv
rm -rf gpurun_out/r03_stats_serial
echo "kernel stats done"
export LUPIN_PMC_OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_pmc_traffic.json
rm -f $LUPIN_PMC_OUT
profile_workload() {  # key, bench args...
  KEY=$1; shift
  tools/pmc_passes.sh r03_pmc_${KEY}_ python3 bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-kernel-timing --no-secondary "$@"
  UNITS=$(grep -h '^{"metric"' gpurun_out/r03_pmc_${KEY}_1.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['path_bounces']*2.0)")
  python3 tools/pmc_traffic.py $KEY $UNITS profiles/r02_fetch_size_calibration.jsonl gpurun_out/r03_pmc_${KEY}_1 gpurun_out/r03_pmc_${KEY}_2 gpurun_out/r03_pmc_${KEY}_3 gpurun_out/r03_pmc_${KEY}_4 gpurun_out/r03_pmc_${KEY}_5 > gpurun_out/r03_pmc_${KEY}_derived.json
  rm -rf gpurun_out/r03_pmc_${KEY}_[1-5]
  echo "pmc $KEY done"
}
profile_workload bistro_class_3840x2160_b16_spp8_standard
profile_workload cornellbox_1024x1024_b8_spp8_standard --scene cornellbox --width 1024 --height 1024 --bounces 8
profile_workload materials1_1920x1080_b12_spp8_standard --scene materials1 --width 1920 --height 1080 --bounces 12
profile_workload environments1_1920x1080_b16_spp8_standard --scene environments1 --width 1920 --height 1080 --bounces 16
cut -c1-600 gpurun_out/r03_bench.json
