#!/bin/bash
# A/B in one call on one box, alternating: tools/r03_ab2.sh "<env A>" "<env B>" [rounds]
cd "$GRAFT_REPO_ROOT"
A=$1; B=$2; R=${3:-3}
run() { env $1 python3 tools/scene_bench.py bistro_class --width 3840 --height 2160 --bounces 16 --steps 16 --warmup 8 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-50s %8.1f Msamples/s  %7.2f ms/step  extend %.1f shade %.1f ms/2 steps' % (sys.argv[1], d['Msamples_per_s'], d['ms_per_step'], d['kernel_ms_2steps']['extend'], d['kernel_ms_2steps']['shade']))" "$1"; }
for r in $(seq $R); do run "$A"; run "$B"; done
