#!/usr/bin/env python3
"""How the persistent tracer's waves spend their scheduling rounds, wide vs binary (device counters of the work-counting build).
    python tools/round_stats.py [scene] [width] [height] [bounces]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lupinpathtracer_amd import api
from tests import util

name = sys.argv[1] if len(sys.argv) > 1 else "bistro_class"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
B = int(sys.argv[4]) if len(sys.argv) > 4 else 16
ctx = api.Context(0)
scene, cams = util.load_scene(name, ctx)
cam = cams[0]
params = api.CameraParams(**{**cam.params.__dict__, "aspect": W / H})
res = api.build_pathtrace_resources(ctx, api.BakedPathtraceParams(max_bounces=B, samples_per_pixel=8))
tex = api.Texture(ctx, W, H)
desc = api.PathtraceDesc(camera_params=params, camera_transform=cam.transform)
for mode in ("wide", "binary"):
    ctx.set_traversal(mode)
    api.pathtrace_scene(ctx, res, scene, tex, 0, desc)
    ctx.stats_reset(2)
    api.pathtrace_scene(ctx, res, scene, tex, 0, desc)
    st = ctx.stats()
    ctx.stats_reset(0)
    u = st["path_bounces"]
    r = st["tracer_rounds"]
    out = {"mode": mode, "scene": name, "path_bounces": u,
           "per_unit": {"binary_nodes": st["node_visits"][0] / u, "wide_nodes": st["wide_node_visits"][0] / u, "tris": st["tri_tests"][0] / u,
                        "instances": st["instance_entries"][0] / u},
           "retraced_frac": st["wide_retraced"] / max(1, st["wide_queries"]),
           "wave_rounds_per_64_units": {"refill": 64 * r[0] / u, "N": 64 * r[1] / u, "N_steps": 64 * r[2] / u, "T": 64 * r[4] / u, "I": 64 * r[6] / u, "F": 64 * r[8] / u},
           "cycles_per_wave_round": {"refill": st["tracer_cycles"][0] / max(1, r[0]), "N": st["tracer_cycles"][1] / max(1, r[1]), "N_per_step": st["tracer_cycles"][1] / max(1, r[2]),
                                     "T": st["tracer_cycles"][2] / max(1, r[4]), "I": st["tracer_cycles"][3] / max(1, r[6]), "F": st["tracer_cycles"][4] / max(1, r[8])},
           "cycle_share": {k: st["tracer_cycles"][i] / max(1, st["tracer_cycles"][5]) for i, k in enumerate(["refill", "N", "T", "I", "F"])},
           "lanes_per_step": {"N": r[3] / max(1, r[2]), "T": r[5] / max(1, r[4]), "I": r[7] / max(1, r[6]), "F": r[9] / max(1, r[8])}}
    print(json.dumps(out))
