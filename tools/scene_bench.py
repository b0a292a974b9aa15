#!/usr/bin/env python3
"""Msamples/s of the HIP path on the other BASELINE.json configs (fixture scenes under tests/golden/scenes).

    python tools/scene_bench.py materials1 --cam 0 --width 1920 --height 1080 --bounces 12 --spp 8 --steps 8
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("--cam", type=int, default=0)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--type", type=int, default=0)
    ap.add_argument("--blas", default="sah", choices=["sah", "lbvh"], help="BLAS builder: reference CPU SAH or the device LBVH")
    args = ap.parse_args()
    from lupinpathtracer_amd import api
    from tests import util
    ctx = api.Context(0)
    t = time.perf_counter()
    if args.blas == "sah":
        scene, cams = util.load_scene(args.scene, ctx)
    else:
        from lupinpathtracer_amd import loader
        if args.scene.startswith("bistro_class"):
            scene, cams = loader.build_scene_bistro_class(ctx, util.SHARED, blas_builder="lbvh")
        else:
            scene, cams = loader.load_scene_yoctogl_v24(os.path.join(util.SCENES, args.scene, args.scene + ".json"), ctx,
                                                        asset_dirs=[util.SHARED], blas_builder="lbvh")
    load_s = time.perf_counter() - t
    cam = cams[args.cam]
    params = api.CameraParams(**{**cam.params.__dict__, "aspect": args.width / args.height})
    res = api.build_pathtrace_resources(ctx, api.BakedPathtraceParams(max_bounces=args.bounces, samples_per_pixel=args.spp))
    out = api.DoubleBufferedTexture(ctx, args.width, args.height)
    ctx.reserve_path_state(args.width * args.height, args.bounces, args.spp)   # a full batch's path state now, not inside the timed steps
    k = 0

    def step():
        nonlocal k
        api.pathtrace_scene(ctx, res, scene, out.front(), args.type,
                            api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), k), camera_params=params,
                                              camera_transform=cam.transform))
        out.flip()
        k += 1
    for _ in range(args.warmup):
        step()
    ctx.sync()
    ctx.stats_reset(False)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    ctx.stats_reset(True)
    for _ in range(min(2, args.steps)):
        step()
    kst = ctx.stats()
    # fraction of the HBM roofline (8 TB/s) from the algorithmic bytes per path-bounce of the oracle's work accounting
    # (tests/golden/work_counters.json: per-unit figures do not depend on the resolution) and the serial kernel times
    roofline = None
    with open(os.path.join(ROOT, "tests", "golden", "work_counters.json")) as f:
        wc = json.load(f)
    for rec in wc.values():
        if rec["scene"] == args.scene and rec["max_bounces"] == args.bounces and rec["pathtrace_type"] == args.type and kst["path_bounces"] > 0:
            roofline = {}
            for kname, ms, per_unit in (("k_extend", kst["extend_ms"], rec["extend_bytes_per_unit"]), ("k_shade", kst["shade_ms"], rec["shade_bytes_per_unit"])):
                achieved = per_unit * kst["path_bounces"] / (ms * 1e-3) / 1e9
                roofline[kname] = {"bound": "hbm", "bytes_per_unit": per_unit, "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0}
    print(json.dumps({"scene": args.scene, "camera": args.cam, "width": args.width, "height": args.height, "bounces": args.bounces,
                      "spp_per_step": args.spp, "steps": args.steps, "type": args.type, "Msamples_per_s": st["path_bounces"] / dt / 1e6,
                      "Mpaths_per_s": st["paths"] / dt / 1e6, "ms_per_step": dt / args.steps * 1e3,
                      "bounces_per_path": st["path_bounces"] / st["paths"], "scene_stats": scene.stats, "blas_builder": args.blas, "load_and_build_s": load_s,
                      "kernel_ms_2steps": {"extend": kst["extend_ms"], "shade": kst["shade_ms"], "total": kst["total_ms"]}, "roofline": roofline,
                      "first_pass_queries": st.get("wide_queries", 0), "retraced": st.get("wide_retraced", 0)}))


if __name__ == "__main__":
    main()
