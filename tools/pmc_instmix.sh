cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export LUPIN_LANES=1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/f64a --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 -- python3 tools/scene_bench.py materials1 --bounces 12 --steps 1 --warmup 0 > gpurun_out/f64a.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/f64b --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT -- python3 tools/scene_bench.py materials1 --bounces 12 --steps 1 --warmup 0 > gpurun_out/f64b.log 2>&1
python3 - <<'PY'
import csv,glob,collections
k=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('gpurun_out/f64[ab]/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].split('(')[0].replace('void ','').split('<')[0]
        k[n][r['Counter_Name']]+=float(r['Counter_Value'])
for n,c in k.items():
    if 'shade' in n or 'extend' in n:
        print(n, {a: '%.3g'%b for a,b in sorted(c.items())})
PY
tail -2 gpurun_out/f64a.log | cut -c1-300
