#!/bin/bash
# ./experiments/ab_scene.sh <scene> lib1.so lib2.so ...
S=$1; shift
for L in "$@"; do
  LUPIN_HIP_LIB=$PWD/$L python tools/scene_bench.py $S --bounces 12 --steps 6 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['scene'], '%.1f Msamples/s'%d['Msamples_per_s'], d['kernel_ms_2steps'])"
done
