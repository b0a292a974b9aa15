"""Every run-time switch the library still reads (DESIGN.md 5 "Run-time switches"), each value in a fresh process, against the
oracle: a mixed-material scene traversed from global memory (bistro-class stand-in, small) under three integrators and the
matte Cornell box (staged in LDS) under two.  The default pipeline is the first row; a switch may change which kernels run,
never a texel."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import util

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CASES = [dict(scene="bistro_class_small", type=t, w=96, h=64, frames=3, spp=2, bounces=6) for t in (0, 1, 3)] + \
        [dict(scene="cornellbox_builtin", type=t, w=64, h=64, frames=3, spp=2, bounces=6) for t in (0, 1)]

SWITCHES = [
    {},                                                      # the default pipeline
    {"LUPIN_PATH_RECORDS": "0"}, {"LUPIN_PATH_RECORDS": "1"},
    {"LUPIN_SORT_SHADE": "0"}, {"LUPIN_SORT_SHADE": "1"},
    {"LUPIN_EXTEND": "simple"},
    {"LUPIN_SHADOW": "simple"},
    {"LUPIN_LDS_GEOMETRY": "0"},
    {"LUPIN_LIGHT_STAGE": "0"}, {"LUPIN_LIGHT_STAGE": "1"},
    {"LUPIN_SIMPLE_SHADE": "0"},
    {"LUPIN_GRAPH": "1"},
    {"LUPIN_LANES": "1"}, {"LUPIN_LANES": "8"},
    {"LUPIN_BATCH": "1"}, {"LUPIN_BATCH": "8"},
    {"LUPIN_TRAVERSAL": "wide"},
    {"LUPIN_TRAVERSAL": "wide", "LUPIN_VERIFY_WIDE": "1"},
    {"LUPIN_TRAVERSAL": "wide", "LUPIN_LDS_GEOMETRY": "0", "LUPIN_GRAPH": "1", "LUPIN_BATCH": "8"},
    {"LUPIN_DEBUG_SYNC": "1"},
    # the binary tracer's first pass on a stack far too short for the scene: most queries overflow and take the second pass
    {"LUPIN_SHORT_STACK": "5"}, {"LUPIN_SHORT_STACK": "0"},
    {"LUPIN_SHORT_STACK": "5", "LUPIN_BATCH": "1", "LUPIN_LANES": "1"},
    {"LUPIN_SHORT_STACK": "3", "LUPIN_GRAPH": "1"},
]


@pytest.fixture(scope="module")
def reference(tmp_path_factory, built):
    """The oracle's image of every case, once per session (host-side scenes: no GPU involved)."""
    path = str(tmp_path_factory.mktemp("switches") / "reference.npz")
    arrays = {"cases": json.dumps(CASES)}
    for c in CASES:
        scene, cams = util.load_scene(c["scene"], None)
        arrays[f"{c['scene']}:{c['type']}"] = util.oracle_accumulate(scene, cams[0], c["w"], c["h"], c["frames"], c["spp"], max_bounces=c["bounces"], ptype=c["type"])
    np.savez(path, **arrays)
    return path


@pytest.mark.gpu
@pytest.mark.parametrize("switch", SWITCHES, ids=lambda s: ",".join(f"{k[6:]}={v}" for k, v in s.items()) or "default")
def test_switch_value_renders_the_oracles_image(reference, switch):
    env = {k: v for k, v in os.environ.items() if not k.startswith("LUPIN_") or k in ("LUPIN_HIP_LIB",)}
    env.update(switch)
    p = subprocess.run([sys.executable, os.path.join(HERE, "_switch_worker.py"), reference], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert all(v == 0 for v in res["differing_words"].values()), res["differing_words"]
    big = res["stats"]["bistro_class_small:0"]
    if switch.get("LUPIN_TRAVERSAL") == "wide":
        assert big["wide_traversal"] == 1 and big["wide_queries"] > 0
        if switch.get("LUPIN_VERIFY_WIDE"):
            assert big["verify_checked"] > 0 and big["verify_mismatches"] == 0
    else:
        assert big["wide_traversal"] == 0
    if not switch:
        # the default pipeline: the case's three chained frames ran as ONE wavefront on one lane (DESIGN 5 "One wavefront at a time")
        assert big["frames_per_wavefront"] == 3 and big["frames_in_flight"] == 1, big
        small = res["stats"]["cornellbox_builtin:0"]
        assert small["frames_per_wavefront"] == 3 and small["frames_in_flight"] == 4 and small["short_stack_entries"] == 0, small   # LDS-resident: four lanes, one-ray-per-lane tracer
    if switch.get("LUPIN_BATCH") == "1":
        assert big["frames_per_wavefront"] == 1 and big["frames_in_flight"] == int(switch.get("LUPIN_LANES", "8")), big
    if "LUPIN_LANES" in switch:
        assert big["frames_in_flight"] == int(switch["LUPIN_LANES"])
    if int(switch.get("LUPIN_SHORT_STACK", "0")) > 0 and "LUPIN_TRAVERSAL" not in switch:
        # the cases chain three frames: one wavefront on one lane, where the short first pass applies
        assert big["short_stack_entries"] == int(switch["LUPIN_SHORT_STACK"]) and 0 < big["wide_retraced"] < big["wide_queries"], big
    elif switch.get("LUPIN_SHORT_STACK") == "0":
        assert big["short_stack_entries"] == 0 and big["wide_queries"] == 0
