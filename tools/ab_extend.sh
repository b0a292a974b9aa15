run() { echo "== $*"; env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],1), round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['kernel_ms'].items()}, {k:round(v,2) for k,v in d['roofline']['bytes_per_unit_terms'].items()})"; }
for a in "$@"; do run $a; done
