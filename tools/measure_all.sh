#!/bin/bash
# Msamples/s of every BASELINE.json configuration (and the other integrators) on one GPU -> one JSON line per run.
#   tools/measure_all.sh > gpurun_out/measure_all.jsonl
set -e
run() { timeout -k 10 400 python3 tools/scene_bench.py "$@" 2>/dev/null; }
for t in 0 1 2 3; do run cornellbox_builtin --width 1024 --height 1024 --bounces 8 --steps 16 --type $t; done
run materials1 --bounces 12
run environments1 --bounces 16
run materials4 --cam 1 --bounces 12
run bistro_class --bounces 16 --steps 4
run bistro_class --bounces 16 --steps 4 --type 1
