#!/bin/bash
# tools/ktrace.sh <tag> [bench args]: rocprofv3 kernel trace of the headline bench (lanes as configured) -> overlap summary
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_ktrace_dir -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary "$@" > gpurun_out/${TAG}_ktrace.log 2>&1
python3 tools/ktrace_overlap.py gpurun_out/${TAG}_ktrace_dir/*/*kernel_trace.csv
rm -rf gpurun_out/${TAG}_ktrace_dir
