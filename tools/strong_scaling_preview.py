"""What each rank of `bench.py --gpus N` would do on the headline frame, rendered one rank at a time on ONE GPU (no communication):
per-rank step time and share of the work -> the ceiling of the strong-scaling speed-up (T(1) / max_r T(N, r)) and the tile balance.

    python tools/strong_scaling_preview.py [--tile-size 8] [--steps 6]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lupinpathtracer_amd import api
from tests import util

ap = argparse.ArgumentParser()
ap.add_argument("--tile-size", type=int, default=8)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--worlds", default="1,2,4,8")
ap.add_argument("--first-rank-only", action="store_true")
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
args = ap.parse_args()
ctx = api.Context(0)
scene, cams = util.load_scene("bistro_class", ctx)
cam = cams[0]
res = api.build_pathtrace_resources(ctx, api.BakedPathtraceParams(max_bounces=16, samples_per_pixel=8))
out = api.DoubleBufferedTexture(ctx, args.width, args.height)
ctx.reserve_path_state(args.width * args.height, 16, 8)
base = None
for world in [int(w) for w in args.worlds.split(",")]:
    times, units = [], []
    for rank in range(1 if args.first_rank_only else world):
        k = [0]
        def step():
            desc = api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), k[0]), camera_params=cam.params, camera_transform=cam.transform)
            api.pathtrace_scene_tiles(ctx, res, scene, out.front(), 0, desc, args.tile_size, rank, world)
            out.flip(); k[0] += 1
        for _ in range(2): step()
        ctx.sync(); ctx.stats_reset()
        t0 = time.perf_counter()
        for _ in range(args.steps): step()
        ctx.sync()
        times.append((time.perf_counter() - t0) / args.steps * 1e3)
        units.append(ctx.stats()["path_bounces"] / args.steps)
        if rank == 0:   # serial kernel sums of the same work (one frame at a time, hipEvent-bracketed stages)
            ctx.stats_reset(1)
            for _ in range(2): step()
            ctx.sync()
            st = ctx.stats()
            ktime = (st["extend_ms"] / 2, st["shade_ms"] / 2, st["total_ms"] / 2)
            ctx.stats_reset(0)
    if base is None: base = times[0]
    print(f"world {world}: ms/step per rank {[round(t, 1) for t in times]}  work share {[round(u / sum(units), 3) for u in units]}  "
          f"speed-up ceiling {base / max(times):.2f}  rank-0 serial kernels per step: extend {ktime[0]:.1f} shade {ktime[1]:.1f} total {ktime[2]:.1f} ms", flush=True)
