"""What each rank of an N-GPU bench run would do, rendered one rank at a time on ONE GPU: units, time and
throughput per rank (no communication) -> the ceiling of the weak-scaling efficiency, and the tile balance."""
import sys, time, math
sys.path.insert(0, ".")
from lupinpathtracer_amd import api, loader
ctx = api.Context(0)
scene, cams = loader.build_scene_cornell_box(ctx)
cam = cams[0]
res = api.build_pathtrace_resources(ctx, api.BakedPathtraceParams(max_bounces=8, samples_per_pixel=8))
TILE = 8
base = None
for world in (1, 2, 4, 8):
    tiles = max(1, int(round(1024 * math.sqrt(world) / (TILE * 4))))
    if world > 1 and tiles % world == 0: tiles += 1
    side = tiles * TILE * 4
    out = api.DoubleBufferedTexture(ctx, side, side)
    rates, units = [], []
    for rank in (range(world) if world <= 4 else (0, 3, 7)):
        k = 0
        def step():
            global k
            desc = api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), k), camera_params=cam.params, camera_transform=cam.transform)
            api.pathtrace_scene_tiles(ctx, res, scene, out.front(), 0, desc, TILE, rank, world)
            out.flip(); k += 1
        for _ in range(5): step()
        ctx.sync(); ctx.stats_reset(False)
        t0 = time.perf_counter()
        for _ in range(24): step()
        ctx.sync()
        dt = time.perf_counter() - t0
        st = ctx.stats()
        rates.append(st["path_bounces"] / dt / 1e6); units.append(st["path_bounces"] / 24)
    if base is None: base = rates[0]
    print(f"world {world}: image {side}^2, per-rank Msamples/s {[round(r) for r in rates]}, units/step {[int(u) for u in units]}, "
          f"ceiling efficiency {min(rates) / base:.3f}", flush=True)
