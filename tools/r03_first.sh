#!/bin/bash
# round 3, first GPU run of the wide tracer: its own tests, the whole -m gpu suite, then wide-vs-binary on the headline scenes
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_wide_traversal.py -m gpu -x -q -s > gpurun_out/r03_first_wide_tests.log 2>&1 || { tail -40 gpurun_out/r03_first_wide_tests.log; exit 1; }
tail -15 gpurun_out/r03_first_wide_tests.log
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_first_gpu_suite.log 2>&1 || { tail -40 gpurun_out/r03_first_gpu_suite.log; exit 1; }
tail -5 gpurun_out/r03_first_gpu_suite.log
for tv in wide binary; do
  for sc in "bistro_class --width 3840 --height 2160 --bounces 16 --steps 4" "materials1 --bounces 12" "environments1 --bounces 16"; do
    echo "== LUPIN_TRAVERSAL=$tv $sc"
    LUPIN_TRAVERSAL=$tv timeout -k 10 400 python3 tools/scene_bench.py $sc 2>&1 | tail -1 | tee -a gpurun_out/r03_first_scene_bench.jsonl
  done
done
