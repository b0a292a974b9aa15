#!/bin/bash
# Collects the round's evidence on the GPU box: tools/profile_round.sh <tag>   (e.g. r01_v4)
#   gpurun_out/<tag>_bench.json               default bench line
#   gpurun_out/<tag>_stats_serial/            rocprofv3 --kernel-trace --stats, LUPIN_LANES=1 (kernels one at a time: the
#                                             average durations the bench's hipEvent pass must agree with)
#   gpurun_out/<tag>_stats_overlap/           the same with the default 3 lanes
#   gpurun_out/<tag>_pmc.json                 folded PMC passes (tools/pmc_passes.sh)
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
LUPIN_LANES=1 python3 bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench_serial.json 2>> gpurun_out/${TAG}_bench.err
export LUPIN_LANES=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats_serial -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_stats_serial.log 2>&1
unset LUPIN_LANES
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats_overlap -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_stats_overlap.log 2>&1
tools/pmc_passes.sh ${TAG}_pmc python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing
UNITS=$(grep -h '^{"metric"' gpurun_out/${TAG}_pmc1.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['path_bounces']*1.5)")
python3 tools/pmc_summary.py $UNITS gpurun_out/${TAG}_pmc.json gpurun_out/${TAG}_pmc1 gpurun_out/${TAG}_pmc2 gpurun_out/${TAG}_pmc3 gpurun_out/${TAG}_pmc4 gpurun_out/${TAG}_pmc5 > gpurun_out/${TAG}_pmc_derived.json
cp gpurun_out/${TAG}_stats_serial/*/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats_serial.csv
cp gpurun_out/${TAG}_stats_overlap/*/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats_overlap.csv
cat gpurun_out/${TAG}_bench.json
