#!/bin/bash
# ./experiments/scenes_ab.sh "<env assignments>" ...   -- Msamples/s of the three big configs under each environment
for E in "$@"; do
  echo "== $E"
  for S in "materials1 --bounces 12" "environments1 --bounces 16" "bistro_class --bounces 16 --steps 4"; do
    env $E timeout -k 10 300 python tools/scene_bench.py $S 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  %-14s %8.1f Msamples/s'%(d['scene'], d['Msamples_per_s']), {k: round(v,1) for k,v in d['kernel_ms_2steps'].items()})" || exit 1
  done
done
