#!/bin/bash
# tools/r03_ab.sh <tag> [bench args]: kernel stats (serial lanes) + two PMC passes of the headline bench under the caller's environment;
# per-kernel totals keyed by the full template name go to gpurun_out/<tag>_pmc.json
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export LUPIN_LANES=1
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary $@"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_kstats_dir -- $BENCH > gpurun_out/${TAG}_kstats.log 2>&1
cp gpurun_out/${TAG}_kstats_dir/*/*kernel_stats.csv gpurun_out/${TAG}_kstats.csv
rm -rf gpurun_out/${TAG}_kstats_dir
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_p1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -- $BENCH > gpurun_out/${TAG}_p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_p5 --pmc TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ_LATENCY GRBM_GUI_ACTIVE -- $BENCH > gpurun_out/${TAG}_p5.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_p2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS -- $BENCH > gpurun_out/${TAG}_p2.log 2>&1
python3 - ${TAG} <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
units = None
for line in open(f"gpurun_out/{tag}_p1.log"):
    if line.startswith('{"metric"'):
        units = json.loads(line)["path_bounces"] * 1.5   # 2 timed steps + 1 warm-up, each a third ... (steps 2 + warmup 1) / 2
k = collections.defaultdict(lambda: collections.defaultdict(float))
for p in ("p1", "p5", "p2"):
    for f in glob.glob(f"gpurun_out/{tag}_{p}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").split("(")[0]
            k[name][r["Counter_Name"]] += float(r["Counter_Value"])
out = {"units_path_bounces": units, "kernels": {}}
for name, c in k.items():
    d = dict(c)
    if c.get("SQ_ACTIVE_INST_VALU"):
        d["lane_util"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64)
        d["wait_any_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if units:
        for key in ("TCP_TCC_READ_REQ", "TCP_TOTAL_CACHE_ACCESSES", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_SALU"):
            if key in c: d[key + "_per_unit"] = c[key] / units
    out["kernels"][name] = d
json.dump(out, open(f"gpurun_out/{tag}_pmc.json", "w"), indent=1)
for r in csv.DictReader(open(f"gpurun_out/{tag}_kstats.csv")):
    if float(r["Percentage"]) > 0.5:
        print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms  {int(r["Calls"]):5d} calls  avg {float(r["AverageNs"])/1e3:8.1f} us  {float(r["Percentage"]):5.1f} %  {r["Name"][:90]}')
for name, d in out["kernels"].items():
    if "k_extend_persistent" in name:
        print(name[:60], {x: round(v, 3) for x, v in d.items() if x.endswith("_per_unit") or x in ("lane_util", "wait_any_frac")})
PY
rm -rf gpurun_out/${TAG}_p1 gpurun_out/${TAG}_p5 gpurun_out/${TAG}_p2
