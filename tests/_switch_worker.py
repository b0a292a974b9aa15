"""Worker of tests/test_gpu_switches.py: renders the cases of a reference file under THIS process's LUPIN_* environment and
compares every image with the oracle's (computed once by the parent).  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lupinpathtracer_amd import api   # noqa: E402
from tests import util   # noqa: E402


def main():
    ref = np.load(sys.argv[1])
    cases = json.loads(str(ref["cases"]))
    ctx = api.Context(0)
    out = {"differing_words": {}, "stats": {}}
    for c in cases:
        scene, cams = util.load_scene(c["scene"], ctx)
        ctx.stats_reset(0)
        got = util.gpu_accumulate(ctx, scene, cams[0], c["w"], c["h"], c["frames"], c["spp"], max_bounces=c["bounces"], ptype=c["type"])
        st = ctx.stats()
        key = f"{c['scene']}:{c['type']}"
        out["differing_words"][key] = util.f16_words_differ(got, ref[key])
        out["stats"][key] = {k: st[k] for k in ("frames_in_flight", "frames_per_wavefront", "short_stack_entries", "wide_traversal", "wide_queries", "wide_retraced",
                                                    "verify_checked", "verify_mismatches")}
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
