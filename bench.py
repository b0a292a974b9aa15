#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (path-bounces per second) of the HIP path tracer.

    python bench.py --gpus N --steps K --warmup W

A step = one `pathtrace_scene` accumulation frame (samples_per_pixel = 8, 8 bounces, Standard integrator) of
the Cornell box -- BASELINE.json configs[1] (cornellbox 1024 x 1024, 8 bounces; its 1024 spp are 128 such
frames, Msamples/s does not depend on how many are timed).  With N > 1 (one rank per GPU, torch.distributed
over RCCL) the image grows with N (weak scaling: the same view at N x 1024^2 pixels) and its tiles are dealt round-robin
to the ranks; no collective runs while accumulating, the timed region ends with the one all-gather of tile
payloads that a readback needs.

The JSON line also carries
  roofline      algorithmic bytes (oracle work counters, tests/golden/work_counters.json) of the dominant kernel
                divided by its hipEvent-measured duration, against the 8 TB/s HBM3E peak;
  cpu_baseline  the CPU oracle (a restatement of the reference megakernel -- the reference has no CPU path) on one
                full step of the same workload on this host's cores.
"""
import argparse
import json
import math
import os
import sys
import time

# The frames in flight need their own hardware queues (3 lanes + torch + RCCL streams; measured: 2 queues cost 13 %).
# The HIP runtime reads this when it initialises, i.e. after this line.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s


def image_size_for(n_gpus, base, tile_px):
    """Weak scaling: base^2 pixels per GPU of the SAME view -- a square image of about n * base^2 pixels whose side is a
    whole number of tiles (1 -> 1024, 2 -> 1440, 4 -> 2080, 8 -> 2912 with 32-pixel tiles: within 3.2 % of 1024^2 per
    GPU), so every rank owns whole tiles only (equal pixel counts whatever the round-robin pattern) carrying the same
    mix of paths as the single-GPU frame.  A wider image would instead add empty space around the box: cheaper paths,
    a different workload."""
    tiles = max(1, int(round(base * math.sqrt(n_gpus) / tile_px)))
    if n_gpus > 1 and tiles % n_gpus == 0:
        tiles += 1   # measured: 2080^2 on 4 ranks runs each rank at 98 % of the single-GPU rate, 2048^2 (64 tiles per row) at 96 %
    return tiles * tile_px, tiles * tile_px


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--size", type=int, default=1024, help="pixels per side per GPU")
    ap.add_argument("--spp", type=int, default=8, help="samples per pixel per step (baked)")
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--tile-size", type=int, default=8, help="tile edge in 4-px workgroups for multi-GPU sharding")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-kernel hipEvent pass (roofline = null)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import numpy as np
    from lupinpathtracer_amd import api, loader, distributed

    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    # The renderer's context (its HIP streams) is created BEFORE the process group, and the group is initialised lazily
    # (no device_id): measured on one GPU, an eagerly initialised RCCL communicator that exists before the context costs
    # the renderer 14 % (7.5 -> 6.45 Gsamples/s -- the frames in flight stop overlapping as well), a lazily initialised
    # one costs nothing, before or after its first collective.
    ctx = api.Context(local_rank)
    dist = None
    if world > 1 or os.environ.get("LUPIN_BENCH_FORCE_DIST") == "1":   # the flag exercises the N > 1 code path on one GPU
        import torch.distributed as dist
        dist.init_process_group(backend="nccl")

    scene, cams = loader.build_scene_cornell_box(ctx)
    cam = cams[0]
    W, H = image_size_for(world, args.size, args.tile_size * 4)
    cam_params = api.CameraParams(**{**cam.params.__dict__, "aspect": cam.params.aspect * W / H})
    res = api.build_pathtrace_resources(ctx, api.BakedPathtraceParams(max_bounces=args.bounces, samples_per_pixel=args.spp))
    out = api.DoubleBufferedTexture(ctx, W, H)
    ptype = api.PathtraceType.Standard

    frame = [0]

    def step():
        desc = api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), frame[0]), camera_params=cam_params,
                                 camera_transform=cam.transform)
        if dist is None:
            api.pathtrace_scene(ctx, res, scene, out.front(), ptype, desc)
        else:
            api.pathtrace_scene_tiles(ctx, res, scene, out.front(), ptype, desc, args.tile_size, rank, world)
        out.flip()
        frame[0] += 1

    def full_sync():
        ctx.sync()
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize(device)

    ops = distributed.HipTileOps(torch, ctx, device)

    for _ in range(args.warmup):
        step()
    if dist is not None:   # warm the collective too
        out.flip()
        distributed.gather_framebuffer(dist, ops, out.front(), W, H, args.tile_size, rank, world)
        out.flip()
    full_sync()
    ctx.stats_reset(False)

    # ---- timed region: exactly K steps + the readback gather ----
    full_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if dist is not None:
        out.flip()   # front = last rendered frame
        distributed.gather_framebuffer(dist, ops, out.front(), W, H, args.tile_size, rank, world)
        out.flip()
    full_sync()
    elapsed = time.perf_counter() - t0

    st = ctx.stats()
    units = torch.tensor([float(st["path_bounces"]), float(st["paths"])], dtype=torch.float64, device=device)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    total_units, total_paths, elapsed = float(units[0]), float(units[1]), float(tmax[0])

    # ---- per-kernel pass (rank 0, N = 1): hipEvents around every extend / shade launch, on the kernels' stream ----
    roofline = None
    kernel_ms = None
    if rank == 0 and world == 1 and not args.no_kernel_timing:
        ksteps = min(args.steps, 8)
        ctx.stats_reset(True)
        for _ in range(ksteps):
            step()
        kst = ctx.stats()
        ctx.stats_reset(False)
        with open(os.path.join(ROOT, "tests", "golden", "work_counters.json")) as f:
            wc = json.load(f)
        key = f"cornellbox_1024x1024_b{args.bounces}_spp{args.spp}_standard"
        if key in wc and args.size == 1024 and kst["extend_launches"] > 0:
            launches = kst["extend_launches"]
            kernel_ms = {"extend": kst["extend_ms"], "shade": kst["shade_ms"], "total": kst["total_ms"], "launches": launches,
                         "steps": ksteps}
            # HBM bytes per launch from the PMC passes of the same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
            # runs, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md); summary committed under profiles/
            pmc = {}
            pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
            if os.path.exists(pmc_path):
                with open(pmc_path) as f:
                    pmc = json.load(f).get("hbm_bytes_per_unit", {})

            def roof(name):
                per_unit = wc[key]["shade_bytes_per_unit" if name == "k_shade" else "extend_bytes_per_unit"]
                ms = kst["shade_ms"] if name == "k_shade" else kst["extend_ms"]
                avg_launch_s = ms * 1e-3 / launches
                achieved = per_unit * kst["path_bounces"] / launches / avg_launch_s / 1e9
                traffic = pmc[name] * kst["path_bounces"] / launches if name in pmc else None
                return {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                        "bytes_per_unit": per_unit, "units_per_launch": kst["path_bounces"] / launches,
                        "avg_launch_us": avg_launch_s * 1e6}

            # the two stage kernels take the same time to within a few percent; the line's `roofline` is the slower one
            both = {n: roof(n) for n in ("k_extend", "k_shade")}
            roofline = dict(both["k_shade" if kst["shade_ms"] >= kst["extend_ms"] else "k_extend"])
            roofline["note"] = ("algorithmic bytes in the reference's layout over the kernel's serial launch time (per-kernel pass runs "
                                "one frame at a time); the scene is LDS/cache-resident, `traffic` is the measured HBM bytes per launch")
            roofline["other_kernel"] = both["k_extend" if roofline["kernel"] == "k_shade" else "k_shade"]

    # ---- CPU baseline (rank 0, N = 1): one full step of the same workload on the oracle ----
    cpu_baseline = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        host_scene, host_cams = loader.build_scene_cornell_box(None)
        hc = host_cams[0]
        threads = oracle.num_threads()
        side = args.size
        oracle.pathtrace(host_scene, 64, 64, hc.params, hc.transform, args.bounces, 1)   # page the library in
        c0 = time.perf_counter()
        cpu_img, cnt = oracle.pathtrace(host_scene, side, side, hc.params, hc.transform, args.bounces, args.spp)
        cdt = time.perf_counter() - c0
        # the same frame (accum_counter 0) through the HIP path: BASELINE's RMSE figure, against the restatement
        chk = api.Texture(ctx, side, side)
        api.pathtrace_scene(ctx, res, scene, chk, ptype, api.PathtraceDesc(camera_params=hc.params, camera_transform=hc.transform))
        gpu_img = chk.download()
        diff = gpu_img[..., :3].astype(np.float32) - cpu_img[..., :3].astype(np.float32)
        parity = {"rmse_vs_cpu_restatement": float(np.sqrt((diff ** 2).mean())),
                  "differing_f16_words": int((gpu_img.view(np.uint16) != cpu_img.view(np.uint16)).sum()),
                  "sample": f"frame 0 of the workload, {side}x{side}, {args.spp} spp"}
        cpu_baseline = {"value": cnt["path_bounces"] / cdt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
                        "sample": f"1 step: cornellbox {side}x{side}, {args.bounces} bounces, {args.spp} spp "
                                  f"({cnt['path_bounces']} path-bounces, {cdt:.2f} s, OpenMP over rows)"}

    if rank == 0:
        value = total_units / elapsed / 1e6
        line = {
            "metric": "Msamples/sec (paths x bounces)", "value": value, "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cornellbox {W}x{H} ({W * H / world / 1e6:.3f} Mpx per GPU), {args.bounces} bounces, "
                                   f"{args.spp} spp per step, Standard integrator, software BVH",
                       "scene": "built-in Cornell box (8 instances, 36 triangles, 1 area light)",
                       "samples_per_pixel_per_step": args.spp, "spp_total_timed": args.spp * args.steps,
                       "frames_in_flight": int(os.environ.get("LUPIN_LANES", "3")),
                       "sharding": "single dispatch" if world == 1 else f"tile-sharded, tile {args.tile_size * 4}px, round-robin, RCCL all-gather at readback"},
            "Mpaths_per_s": total_paths / elapsed / 1e6,
            "path_bounces": total_units,
            "roofline": roofline, "kernel_ms": kernel_ms, "cpu_baseline": cpu_baseline, "parity": parity,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
