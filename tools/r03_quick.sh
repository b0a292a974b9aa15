#!/bin/bash
# default-configuration Msamples/s of the four BASELINE workloads (and MIS / Direct on the headline frame)
cd "$GRAFT_REPO_ROOT"
run() { python3 tools/scene_bench.py "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-14s type %d %4dx%-4d  %8.1f Msamples/s  %7.2f ms/step' % (d['scene'], d['type'], d['width'], d['height'], d['Msamples_per_s'], d['ms_per_step']))"; }
run bistro_class --width 3840 --height 2160 --bounces 16 --steps 16 --warmup 8
run materials1 --bounces 12 --steps 32 --warmup 16
run environments1 --bounces 16 --steps 32 --warmup 16
run cornellbox_builtin --width 1024 --height 1024 --bounces 8 --steps 64 --warmup 64
if [ "$1" = "all" ]; then
run bistro_class --width 3840 --height 2160 --bounces 16 --steps 8 --warmup 4 --type 1
run bistro_class --width 3840 --height 2160 --bounces 16 --steps 8 --warmup 4 --type 3
run bistro_class --width 3840 --height 2160 --bounces 16 --steps 8 --warmup 4 --type 2
fi
