#!/bin/bash
# MIS / Direct on the 1080p scenes under a list of environments
cd "$GRAFT_REPO_ROOT"
run() { python3 tools/scene_bench.py "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-14s type %d %4dx%-4d  %8.1f Msamples/s  %7.2f ms/step  extend %.1f shade %.1f ms/2 steps' % (d['scene'], d['type'], d['width'], d['height'], d['Msamples_per_s'], d['ms_per_step'], d['kernel_ms_2steps']['extend'], d['kernel_ms_2steps']['shade']))"; }
for E in "$@"; do
  echo "== $E"
  env $E bash -c "$(declare -f run); run materials1 --bounces 12 --steps 16 --warmup 8 --type 1; run environments1 --bounces 16 --steps 16 --warmup 8 --type 1; run materials1 --bounces 12 --steps 16 --warmup 8 --type 3; run cornellbox_builtin --width 1024 --height 1024 --bounces 8 --steps 32 --warmup 32 --type 1"
done
