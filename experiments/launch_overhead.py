"""How long does the host need to enqueue one pathtrace_scene call, and how fast do small frames go?"""
import sys, time
sys.path.insert(0, ".")
from lupinpathtracer_amd import api, loader
ctx = api.Context(0)
scene, cams = loader.build_scene_cornell_box(ctx)
cam = cams[0]
for size, spp in ((64, 1), (256, 1), (256, 8), (512, 8), (1024, 8)):
    res = api.build_pathtrace_resources(ctx, api.BakedPathtraceParams(max_bounces=8, samples_per_pixel=spp))
    out = api.DoubleBufferedTexture(ctx, size, size)
    def step(k):
        api.pathtrace_scene(ctx, res, scene, out.front(), 0, api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), k), camera_params=cam.params, camera_transform=cam.transform))
        out.flip()
    for k in range(8): step(k)
    ctx.sync()
    n = 64
    t0 = time.perf_counter()
    for k in range(n): step(8 + k)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print(f"{size}x{size} spp {spp}: enqueue {1e3*(t1-t0)/n:.3f} ms/frame, end-to-end {1e3*(t2-t0)/n:.3f} ms/frame, launches/frame {2*spp*9+3}")
