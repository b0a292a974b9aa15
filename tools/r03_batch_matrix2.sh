#!/bin/bash
# frames per wavefront x lanes with the short-stack tracer: Msamples/s of the BASELINE configs
cd "$GRAFT_REPO_ROOT"
run() { python3 tools/scene_bench.py "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-14s %4dx%-4d  %8.1f Msamples/s  %7.2f ms/step' % (d['scene'], d['width'], d['height'], d['Msamples_per_s'], d['ms_per_step']))"; }
for cfg in "$@"; do
  set -- $cfg
  if [ "$1" = "0" ]; then unset LUPIN_BATCH LUPIN_LANES; echo "== the library's defaults"; else export LUPIN_BATCH=$1 LUPIN_LANES=$2; echo "== batch $1 lanes $2"; fi
  run bistro_class --width 3840 --height 2160 --bounces 16 --steps 16 --warmup 8
  run materials1 --bounces 12 --steps 32 --warmup 16
  run environments1 --bounces 16 --steps 32 --warmup 16
  run cornellbox_builtin --width 1024 --height 1024 --bounces 8 --steps 64 --warmup 64
done
