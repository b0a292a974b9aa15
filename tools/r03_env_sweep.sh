#!/bin/bash
# bistro-class 4K under a list of environments: tools/r03_env_sweep.sh "<env>" ...
cd "$GRAFT_REPO_ROOT"
for E in "$@"; do
env $E python3 tools/scene_bench.py bistro_class --width 3840 --height 2160 --bounces 16 --steps 16 --warmup 8 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-60s %8.1f Msamples/s  %7.2f ms/step  extend %.1f shade %.1f ms/2 steps' % (sys.argv[1], d['Msamples_per_s'], d['ms_per_step'], d['kernel_ms_2steps']['extend'], d['kernel_ms_2steps']['shade']))" "$E"
done
