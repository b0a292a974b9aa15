#!/bin/bash
# instruction-cache counters of the headline bench's kernels (one PMC pass) -> gpurun_out/r03_icache.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export LUPIN_LANES=1
rocprofv3 -L 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH\|SQC_INST[A-Z_]*" | sort -u | tr '\n' ' ' > gpurun_out/r03_icache_counters.txt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03_ic --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -- python3 bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-kernel-timing --no-secondary > gpurun_out/r03_ic.log 2>&1
python3 - gpurun_out/r03_ic/*/*counter_collection.csv > gpurun_out/r03_icache.txt <<'PY'
import csv, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:60]
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:6]:
    req = c.get("SQC_ICACHE_REQ", 0); miss = c.get("SQC_ICACHE_MISSES", 0)
    print("%-62s icache req %.3g  misses %.3g  miss rate %.4f  SQ_IFETCH %.3g  wait_inst_any / wave_cycles %.3f" % (k, req, miss, miss / req if req else 0, c.get("SQ_IFETCH", 0), c.get("SQ_WAIT_INST_ANY", 0) / max(1.0, c.get("SQ_WAVE_CYCLES", 1))))
PY
rm -rf gpurun_out/r03_ic
cat gpurun_out/r03_icache_counters.txt; echo; cat gpurun_out/r03_icache.txt; tail -3 gpurun_out/r03_ic.log | cut -c1-200
