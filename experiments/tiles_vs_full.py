"""Single-dispatch vs tile-set dispatch (world 1) throughput on the bench workload."""
import sys, time
sys.path.insert(0, ".")
from lupinpathtracer_amd import api, loader
ctx = api.Context(0)
scene, cams = loader.build_scene_cornell_box(ctx)
cam = cams[0]
res = api.build_pathtrace_resources(ctx, api.BakedPathtraceParams(max_bounces=8, samples_per_pixel=8))
for mode in ("full", "tiles32", "tiles8", "full", "tiles32"):
    out = api.DoubleBufferedTexture(ctx, 1024, 1024)
    k = 0
    def step():
        global k
        desc = api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), k), camera_params=cam.params, camera_transform=cam.transform)
        if mode == "full": api.pathtrace_scene(ctx, res, scene, out.front(), 0, desc)
        else: api.pathtrace_scene_tiles(ctx, res, scene, out.front(), 0, desc, int(mode[5:]), 0, 1)
        out.flip(); k += 1
    for _ in range(6): step()
    ctx.sync(); ctx.stats_reset(False)
    t0 = time.perf_counter()
    for _ in range(32): step()
    ctx.sync()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    print(mode, "%.0f Msamples/s" % (st["path_bounces"] / dt / 1e6), "%.3f ms/step" % (dt / 32 * 1e3), flush=True)
