#!/usr/bin/env python3
"""Folds the PMC passes of one workload (tools/pmc_passes.sh) into profiles/r02_pmc_traffic.json, the file bench.py reads for
`roofline.traffic` / `roofline.l2_hit`.

    python tools/pmc_traffic.py <workload key> <path_bounces of the profiled run> <calibration jsonl> <pass dir> ...

HBM-side bytes per kernel = FETCH_SIZE x 1024 x c + WRITE_SIZE x 1024, with the read correction c taken from THIS build's
calibration (tools/calib/fetch_calib.hip under the same counters, MI355X_MICROARCH.md "calibrate on a known byte count in
your own access pattern"): random 64-byte records (the tracer's node / instance fetches) are counted exactly (c = 1), wide
coalesced streams at half (c = 2).  The closest-hit kernel's reads are gathers (its 36 B of coalesced path state per unit are
under 4 % of its traffic), so c = 1; k_shade mixes streamed path state with gathers, so both bounds are recorded.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    key, units, calib_path = sys.argv[1], float(sys.argv[2]), sys.argv[3]
    kernels = collections.defaultdict(lambda: collections.defaultdict(float))
    for d in sys.argv[4:]:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
                if k.startswith("k_"):
                    kernels[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calib = [json.loads(l) for l in open(calib_path) if l.startswith("{")] if os.path.exists(calib_path) else []
    rec = {"path_bounces_profiled": units, "hbm_bytes_per_unit": {}, "hbm_bytes_per_unit_if_streamed": {}, "l2_hit": {}, "valu_lane_utilisation": {},
           "calibration": {c["kernel"]: {"requested_bytes": c["requested_bytes"], "FETCH_SIZE_bytes": c["FETCH_SIZE_bytes"],
                                         "requested_over_fetch": c["ratio_requested_over_fetch"]} for c in calib},
           "counters_per_unit": {}}
    for k, c in kernels.items():
        name = "k_extend" if k.startswith("k_extend") else k
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rec["hbm_bytes_per_unit"][name] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / units
            rec["hbm_bytes_per_unit_if_streamed"][name] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / units
        if "TCC_HIT" in c and "TCC_MISS" in c and c["TCC_HIT"] + c["TCC_MISS"] > 0:
            rec["l2_hit"][name] = c["TCC_HIT"] / (c["TCC_HIT"] + c["TCC_MISS"])
        if c.get("SQ_ACTIVE_INST_VALU"):
            rec["valu_lane_utilisation"][name] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64)
        rec["counters_per_unit"][name] = {n: v / units for n, v in sorted(c.items())}
    out = os.environ.get("LUPIN_PMC_OUT") or os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")
    allrec = json.load(open(out)) if os.path.exists(out) else {}
    allrec[key] = rec
    json.dump(allrec, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: rec[k] for k in ("hbm_bytes_per_unit", "l2_hit", "valu_lane_utilisation")}, indent=1))


if __name__ == "__main__":
    main()
