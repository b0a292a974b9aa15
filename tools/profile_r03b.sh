#!/bin/bash
# Round-3 evidence for the serial regime (short-stack tracer, one lane, eight frames per wavefront) -> gpurun_out/r03_*.txt
cd "$GRAFT_REPO_ROOT"
bash tools/r03_batch_matrix2.sh "1 8" "4 4" "4 1" "8 1" "8 2" "16 1" "16 4" "0 0" > gpurun_out/r03_batch_matrix_short.txt 2>&1
echo "matrix done"
{ for S in 0 31 26 20 16 12; do LUPIN_SHORT_STACK=$S python3 tools/scene_bench.py bistro_class --width 3840 --height 2160 --bounces 16 --steps 16 --warmup 8 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('LUPIN_SHORT_STACK=%-3s %8.1f Msamples/s  %7.2f ms/step  tracer %.1f shade %.1f ms per 2 steps (serial)  handed to the full-stack pass: %d of %d queries' % (sys.argv[1], d['Msamples_per_s'], d['ms_per_step'], d['kernel_ms_2steps']['extend'], d['kernel_ms_2steps']['shade'], d['retraced'], d['first_pass_queries']))" $S; done; } > gpurun_out/r03_short_stack_sweep.txt 2>&1
echo "sweep done"
python3 tools/strong_scaling_preview.py --steps 16 > gpurun_out/r03_strong_scaling_preview.txt 2>&1
echo "preview done"
bash tools/r03_quick.sh all > gpurun_out/r03_quick_all.txt 2>&1
cat gpurun_out/r03_quick_all.txt
