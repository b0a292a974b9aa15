"""Does a live torch.distributed / RCCL process group slow the renderer down?  (world 1, one GPU)"""
import os, sys, time
sys.path.insert(0, ".")
import torch
from lupinpathtracer_amd import api, loader
mode = sys.argv[1]
torch.cuda.set_device(0)
if mode == "pg":
    import torch.distributed as dist
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
if mode == "pg_lazy":
    import torch.distributed as dist
    dist.init_process_group(backend="nccl")
ctx = api.Context(0)
scene, cams = loader.build_scene_cornell_box(ctx)
cam = cams[0]
res = api.build_pathtrace_resources(ctx, api.BakedPathtraceParams(max_bounces=8, samples_per_pixel=8))
out = api.DoubleBufferedTexture(ctx, 1024, 1024)
k = 0
def step():
    global k
    desc = api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), k), camera_params=cam.params, camera_transform=cam.transform)
    api.pathtrace_scene_tiles(ctx, res, scene, out.front(), 0, desc, 32, 0, 1)
    out.flip(); k += 1
for rep in range(3):
    for _ in range(6): step()
    ctx.sync(); ctx.stats_reset(False)
    t0 = time.perf_counter()
    for _ in range(32): step()
    t1 = time.perf_counter()
    ctx.sync()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    print(mode, rep, "%.0f Msamples/s" % (st["path_bounces"] / dt / 1e6), "enqueue %.3f ms/step" % ((t1 - t0) / 32 * 1e3), "total %.3f ms/step" % (dt / 32 * 1e3), flush=True)
    if mode.startswith("pg") and rep == 0:
        t = torch.zeros(1024, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize(); print("collective done", flush=True)
