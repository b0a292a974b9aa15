#!/bin/bash
# rocprofv3 PMC passes (counters only, one group per run, serial lanes so counters belong to one kernel at a time):
#   tools/pmc_passes.sh <out_prefix under gpurun_out/> <program> [args...]
# then fold with tools/pmc_summary.py.
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export LUPIN_LANES=1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${OUT}1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -- "$@" > gpurun_out/${OUT}1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${OUT}2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS -- "$@" > gpurun_out/${OUT}2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${OUT}3 --pmc FETCH_SIZE TCC_HIT -- "$@" > gpurun_out/${OUT}3.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${OUT}4 --pmc WRITE_SIZE TCC_MISS TCC_REQ -- "$@" > gpurun_out/${OUT}4.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${OUT}5 --pmc TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ_LATENCY GRBM_GUI_ACTIVE -- "$@" > gpurun_out/${OUT}5.log 2>&1 || echo "pass 5 (TCP) unavailable"
echo done
