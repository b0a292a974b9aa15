#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for E in "$@"; do
env $E python3 tools/scene_bench.py cornellbox_builtin --width 1024 --height 1024 --bounces 8 --steps 64 --warmup 64 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-50s %8.1f Msamples/s  %7.2f ms/step  extend %.1f shade %.1f ms/2 steps' % (sys.argv[1], d['Msamples_per_s'], d['ms_per_step'], d['kernel_ms_2steps']['extend'], d['kernel_ms_2steps']['shade']))" "$E"
done
