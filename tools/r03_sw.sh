#!/bin/bash
# switch sweep under the current defaults: tools/r03_sw.sh "<env>" ...
cd "$GRAFT_REPO_ROOT"
run() { python3 tools/scene_bench.py "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-14s type %d %4dx%-4d  %8.1f Msamples/s  %7.2f ms/step  extend %.1f shade %.1f ms/2 steps' % (d['scene'], d['type'], d['width'], d['height'], d['Msamples_per_s'], d['ms_per_step'], d['kernel_ms_2steps']['extend'], d['kernel_ms_2steps']['shade']))"; }
for E in "$@"; do
  echo "== $E"
  env $E bash -c "$(declare -f run); run bistro_class --width 3840 --height 2160 --bounces 16 --steps 16 --warmup 8; run materials1 --bounces 12 --steps 32 --warmup 16; run environments1 --bounces 16 --steps 32 --warmup 16"
done
