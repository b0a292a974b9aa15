#!/bin/bash
# ./experiments/ab_env.sh "<env assignments>" ...  -- headline bench under each environment
for E in "$@"; do
  echo "== $E"
  env $E python bench.py --steps 32 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Msamples/s'%d['value'], d['kernel_ms'])" || exit 1
done
