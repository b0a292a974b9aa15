#!/usr/bin/env python3
"""Folds rocprofv3 --pmc counter_collection.csv files (one directory per pass) into profiles/<name>.json and
refreshes profiles/pmc_traffic_last.json (HBM bytes per path-bounce per kernel, used by bench.py's roofline.traffic).

    python tools/pmc_summary.py <units> <out.json> <pass_dir> [<pass_dir> ...]
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    units = float(sys.argv[1])
    out_path = sys.argv[2]
    kernels = collections.defaultdict(lambda: collections.defaultdict(float))
    for d in sys.argv[3:]:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
                if k.startswith("k_"):
                    kernels[k][r["Counter_Name"]] += float(r["Counter_Value"])
    derived = {}
    traffic = {}
    for k, c in kernels.items():
        dd = {}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            dd["hbm_read_bytes_per_unit_fetch_x2"] = c["FETCH_SIZE"] * 1024 * 2 / units
            dd["hbm_write_bytes_per_unit"] = c["WRITE_SIZE"] * 1024 / units
            traffic[k] = dd["hbm_read_bytes_per_unit_fetch_x2"] + dd["hbm_write_bytes_per_unit"]
        if "SQ_ACTIVE_INST_VALU" in c:
            dd["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64)
            dd["wait_any_over_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
            dd["valu_wave_insts_per_unit"] = c["SQ_INSTS_VALU"] / units
        if "TCC_HIT" in c and "TCC_MISS" in c:
            dd["tcc_hit_rate"] = c["TCC_HIT"] / (c["TCC_HIT"] + c["TCC_MISS"])
        derived[k] = dd
    json.dump({"path_bounces": units, "kernels": kernels, "derived": derived}, open(out_path, "w"), indent=1)
    if traffic:
        json.dump({"source": os.path.basename(out_path), "hbm_bytes_per_unit": traffic},
                  open(os.path.join(os.path.dirname(out_path), "pmc_traffic_last.json"), "w"), indent=1)
    print(json.dumps(derived, indent=1))


if __name__ == "__main__":
    main()
