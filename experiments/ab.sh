#!/bin/bash
# A/B bench of kernel variants: ./experiments/ab.sh lib1.so lib2.so ...
for L in "$@"; do
  echo "== $L"
  LUPIN_HIP_LIB=$PWD/$L python bench.py --steps 32 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Msamples/s'%d['value'], d['kernel_ms'])"
done
