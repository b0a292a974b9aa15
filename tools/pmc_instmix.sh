#!/bin/bash
# VALU instruction mix of the stage kernels on the headline workload (f64 polynomial steps, f32 adds / muls / fmas,
# transcendental-unit ops = the v_rcp / v_sqrt of the correctly rounded division and sqrt expansions, conversions, integer):
#   tools/pmc_instmix.sh <tag> [bench args]
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export LUPIN_LANES=1
BENCH="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-secondary $@"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_mixa --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 -- $BENCH > gpurun_out/${TAG}_mixa.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_mixb --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT -- $BENCH > gpurun_out/${TAG}_mixb.log 2>&1
python3 - $TAG <<'PY'
import csv,glob,collections,json,sys
tag=sys.argv[1]
k=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f'gpurun_out/{tag}_mix[ab]/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].split('(')[0].replace('void ','').split('<')[0]
        k[n][r['Counter_Name']]+=float(r['Counter_Value'])
out={}
for n,c in k.items():
    if 'shade' in n or 'extend' in n:
        tot=c.get('SQ_INSTS_VALU',1.0)
        out[n]={a: round(b/tot,4) for a,b in sorted(c.items()) if a!='SQ_INSTS_VALU'}
        out[n]['SQ_INSTS_VALU']=tot
print(json.dumps(out, indent=1))
PY
