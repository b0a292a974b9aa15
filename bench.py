#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (path-bounces per second) of the HIP path tracer on the north-star workload.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[4], the frame the north star is quoted on, on however many GPUs there are): the
bistro-class scene (seeded procedural stand-in for bistroexterior, which is not in the container -- SURVEY 8d: 2.88 M
triangles, 501 instances, 100 emissive quads, sky HDRI) at 3840 x 2160, 16 bounces, Standard integrator, software BVH.
A step = one `pathtrace_scene` accumulation frame of 8 samples per pixel (its 4096 spp are 512 such frames; Msamples/s does
not depend on how many are timed).

N = 1 renders the whole frame on one GPU.  N > 1 renders the SAME frame tile-sharded ("scaling": "strong"): one process per
GPU, 32-pixel tiles dealt round-robin (include/lupin_tiles.h), no collective while accumulating; the timed region ends with
the one exchange a readback needs, `lupin_hip_gather_framebuffer` (pack -> ncclAllGather over xGMI -> unpack, in the C ABI).
The ranks rendezvous through a file (RCCL unique id); no torch in the process, so the library runs on the HIP runtime it
was built and tested against.  Launch with `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT from the environment), or plain `python bench.py --gpus N`, which starts the
N rank processes itself.

The JSON line also carries (rank 0, N = 1)
  roofline      dominant kernel: bytes it requests per path-bounce in THIS build's layout after light culling (64 B per
                node visit, 48 B per triangle test, 64 B per instance entry, 56 B of path state; counted on the device by
                the kernel's work-counting instantiation) x units per launch / its hipEvent-measured launch time, against
                the nominal 8 TB/s and a device-copy peak measured in the same run; `traffic` / `l2_hit` from the PMC
                passes committed under profiles/;
  cpu_baseline  the CPU oracle (a restatement of the reference megakernel -- the reference has no CPU path) on the same
                scene and view at reduced size, on this host's cores;
  parity        HIP output vs the oracle: the reduced-size frame and tiles of the full-size frame, bit for bit;
  configs       the same measurements for BASELINE configs[1] (Cornell box 1024^2, 8 bounces) and configs[2]
                (materials1 1920 x 1080, 12 bounces), plus the accuracy line of config 2 (RMSE at 1024 spp in both
                accumulation modes).
"""
import argparse
import json
import os
import subprocess
import sys
import time

# The frames in flight need their own hardware queues (3 lanes + RCCL's streams; measured: 2 queues cost 13 %).
# The HIP runtime reads this when it initialises, i.e. after this line.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s
SHARED = os.path.join(ROOT, "tests", "golden", "scenes", "_shared")
SCENES = os.path.join(ROOT, "tests", "golden", "scenes")
PATH_STATE_BYTES_EXTEND = 56.0   # queue entry 4 + ori|rng 16 + dir|meta 16 read, hit record 16 + triangle 4 written


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N rank processes (before anything touches the GPU) and relay
    rank 0's line."""
    port = str(29500 + os.getpid() % 2000)
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    codes = [p.wait() for p in procs]
    sys.stdout.write(out)
    if any(codes):
        raise SystemExit(f"rank exit codes {codes}")


def load_workload(name, ctx):
    from lupinpathtracer_amd import loader
    if name == "bistro_class":
        return loader.build_scene_bistro_class(ctx, SHARED)
    if name == "cornellbox":
        return loader.build_scene_cornell_box(ctx)
    return loader.load_scene_yoctogl_v24(os.path.join(SCENES, name, name + ".json"), ctx, asset_dirs=[SHARED])


def camera_for(api, cam, width, height, keep_aspect=False):
    """The scene's camera rendered at width x height (aspect overridden to the image's, as tools/scene_bench.py does)."""
    if keep_aspect:
        return cam.params
    return api.CameraParams(**{**cam.params.__dict__, "aspect": width / height})


def own_layout_bytes(kst, mode=0):
    """Bytes the tracing kernel requests per path-bounce in this build's layout, from its device-side work counters."""
    u = max(1, kst["path_bounces"])
    n, t, i = kst["node_visits"][mode] / u, kst["tri_tests"][mode] / u, kst["instance_entries"][mode] / u
    return {"node_visits": n, "tri_tests": t, "instance_entries": i,
            "bytes_per_unit": 64.0 * n + 48.0 * t + 64.0 * i + PATH_STATE_BYTES_EXTEND}


def pmc_record(workload_key):
    """HBM traffic and L2 hit rate per kernel from the committed rocprofv3 --pmc passes of this workload (profiles/)."""
    for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json"):   # the newest committed passes that cover this workload
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                rec = json.load(f).get(workload_key)
            if rec:
                return dict(rec, source=name)
    return {}


def request_ceiling():
    """Random 64 / 128-byte record fetches per second the chip sustains, measured by tools/calib/gather_probe (profiles/): the
    ceiling for a kernel whose loads are dependent gathers, in L1-miss requests per second."""
    path = os.path.join(ROOT, "profiles", "r02_gather_probe.jsonl")
    if not os.path.exists(path):
        return None
    best = 0.0
    with open(path) as f:
        for line in f:
            if line.startswith("{"):
                d = json.loads(line)
                if d.get("stream") == "none" and d.get("record_bytes") in (64, 128):
                    best = max(best, float(d["Grecords_per_s"]))
    return best or None


def measure_single_gpu(api, ctx, scene, cam, width, height, bounces, spp, steps, warmup, ptype, workload_key, peak_measured, keep_aspect=False):
    """Throughput + per-kernel roofline of one workload on one GPU (frames overlap as in production for `value`; the per-kernel
    pass runs them one at a time so that a duration belongs to one kernel)."""
    params = camera_for(api, cam, width, height, keep_aspect)
    res = api.build_pathtrace_resources(ctx, api.BakedPathtraceParams(max_bounces=bounces, samples_per_pixel=spp))
    out = api.DoubleBufferedTexture(ctx, width, height)
    ctx.reserve_path_state(width * height, bounces, spp)
    frame = [0]

    def step():
        api.pathtrace_scene(ctx, res, scene, out.front(), ptype,
                            api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), frame[0]), camera_params=params,
                                              camera_transform=cam.transform))
        out.flip()
        frame[0] += 1

    for _ in range(warmup):
        step()
    ctx.sync()
    ctx.stats_reset(0)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    rec = {"value": st["path_bounces"] / dt / 1e6, "unit": "Msamples/s", "Mpaths_per_s": st["paths"] / dt / 1e6,
           "ms_per_step": dt / steps * 1e3, "steps": steps, "path_bounces": st["path_bounces"]}

    # one full wavefront as the library forms them by default (lupin_hip_set_batch_frames: sixteen calls up to 4 M pixels, eight
    # above): the timed launches are production launches
    ksteps = 16 if width * height <= (4 << 20) else 8
    rec["frames_per_wavefront"] = ksteps
    ctx.stats_reset(1)
    for _ in range(ksteps):
        step()
    kst = ctx.stats()
    assert int(kst["frames_per_wavefront"]) == ksteps, (kst["frames_per_wavefront"], ksteps)
    ctx.stats_reset(2)
    frame[0] -= ksteps          # the same frames again: the counters belong to the launches that were timed
    for _ in range(ksteps):
        step()
    wst = ctx.stats()
    ctx.stats_reset(0)
    rec["kernel_ms"] = {"extend": kst["extend_ms"], "shade": kst["shade_ms"], "total": kst["total_ms"],
                        "launches": kst["extend_launches"], "steps": ksteps}
    if kst["extend_launches"] > 0 and kst["extend_ms"] > 0:
        launches = kst["extend_launches"]
        own = own_layout_bytes(wst)
        units_per_launch = kst["path_bounces"] / launches
        avg_s = kst["extend_ms"] * 1e-3 / launches
        achieved = own["bytes_per_unit"] * units_per_launch / avg_s / 1e9
        pmc = pmc_record(workload_key)
        traffic_per_unit = pmc.get("hbm_bytes_per_unit", {}).get("k_extend")
        rec["roofline"] = {
            # a scene staged in LDS (Cornell box) is served on-chip: its tracer is bound by LDS reads / instruction issue, and a
            # fraction of the HBM peak says nothing about it (round 2 printed 0.98 there, 1.27 of the measured peak)
            "bound": "lds/issue" if scene_is_lds_resident(scene) else "hbm",
            "kernel": "k_extend_persistent" if not scene_is_lds_resident(scene) else "k_extend",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "peak_measured": peak_measured, "frac_of_measured_peak": achieved / peak_measured if peak_measured else None,
            "traffic": traffic_per_unit * units_per_launch if traffic_per_unit else None,
            "traffic_bytes_per_unit": traffic_per_unit, "l2_hit": pmc.get("l2_hit", {}).get("k_extend"), "pmc_source": pmc.get("source"),
            "bytes_per_unit": own["bytes_per_unit"], "bytes_per_unit_terms": {k: own[k] for k in own if k != "bytes_per_unit"},
            "units_per_launch": units_per_launch, "avg_launch_us": avg_s * 1e6, "launches_timed": launches,
            "share_of_frame_time": kst["extend_ms"] / kst["total_ms"] if kst["total_ms"] else None,
            "definition": "bytes REQUESTED by the closest-hit kernel in this build's layout after culling (64 B/node visit + 48 B/triangle "
                          "test + 64 B/instance entry + 56 B path state, device-counted) x units per launch / hipEvent launch time",
        }
        if traffic_per_unit:
            # the same kernel at DRAM level (PMC FETCH_SIZE + WRITE_SIZE, calibrated): what fraction of the HBM peak actually crosses
            # the fabric -- the caches serve the rest of the requested bytes
            dram = traffic_per_unit * units_per_launch / avg_s / 1e9
            rec["roofline"]["dram"] = {"bytes_per_unit": traffic_per_unit, "achieved_GBps": dram, "frac_of_peak": dram / HBM_PEAK_GBS,
                                       "frac_of_measured_peak": dram / peak_measured if peak_measured else None}
        l1_miss = pmc.get("counters_per_unit", {}).get("k_extend", {}).get("TCP_TCC_READ_REQ")
        ceiling = request_ceiling()
        if l1_miss and ceiling and not scene_is_lds_resident(scene):
            greq = l1_miss * units_per_launch / avg_s / 1e9
            shade_miss = sum(c.get("TCP_TCC_READ_REQ", 0.0) for k, c in pmc.get("counters_per_unit", {}).items() if k != "k_extend")   # k_shade, k_sort_queue, ...
            frame_bound_ms = (l1_miss + shade_miss) * (st["path_bounces"] / steps) / (ceiling * 1e9) * 1e3
            rec["roofline"]["request_rate"] = {
                "l1_miss_requests_per_unit": l1_miss, "achieved_Greq_per_s": greq, "ceiling_Greq_per_s": ceiling, "frac": greq / ceiling,
                "whole_frame": {"requests_per_unit_all_kernels": l1_miss + shade_miss, "bound_ms_per_step": frame_bound_ms,
                                "measured_ms_per_step": rec["ms_per_step"], "frac": frame_bound_ms / rec["ms_per_step"]},
                "definition": "TCP_TCC_READ_REQ per path-bounce (PMC pass in profiles/) x units per launch / launch time, against the rate of "
                              "independent random 64- / 128-byte record fetches measured by tools/calib/gather_probe on this chip",
                "note": "NOT the binding limit (round 3): the four-wide traversal needs 39 % fewer of these requests per ray and runs in the same "
                        "time, launch for launch (profiles/r03_tracer_per_iteration_*.txt), and with six blocks per CU the tracer issues "
                        "more of them per second than the probe's independent gathers -- it is bound by latency x waves in flight, "
                        "with the vector ALU at about 60 % (DESIGN.md 5)"}
        if scene_is_lds_resident(scene):
            rec["roofline"]["note"] = ("geometry is staged in LDS (scene < 24 KB): requests are served on-chip; `achieved` / `frac` are the "
                                       "requested bytes against the HBM peak for reference only, not a bound")
    return rec, res, params


def scene_is_lds_resident(scene):
    st = getattr(scene, "stats", None) or {}
    return st.get("total_tri_count", 1 << 30) <= 128


def parity_vs_oracle(api, oracle, ctx, gpu_scene, host_scene, cam_params, cam_transform, width, height, bounces, spp, ptype, tiles=()):
    """HIP frame 0 vs the oracle: whole frame when `tiles` is empty, else those TileParams tiles of the frame (reference tile rule)."""
    import numpy as np
    res = api.build_pathtrace_resources(ctx, api.BakedPathtraceParams(max_bounces=bounces, samples_per_pixel=spp))
    tex = api.Texture(ctx, width, height)
    out = {"size": f"{width}x{height}", "spp": spp, "bounces": bounces}
    if not tiles:
        api.pathtrace_scene(ctx, res, gpu_scene, tex, ptype, api.PathtraceDesc(camera_params=cam_params, camera_transform=cam_transform))
        got = tex.download()
        c0 = time.perf_counter()
        want, cnt = oracle.pathtrace(host_scene, width, height, cam_params, cam_transform, bounces, spp, ptype)
        out["oracle_seconds"] = time.perf_counter() - c0
        out["oracle_path_bounces"] = cnt["path_bounces"]
        regions = [(slice(None), slice(None))]
    else:
        regions = []
        want = np.zeros((height, width, 4), np.float16)
        for ts, ti in tiles:
            tp = api.TileParams(tile_size=ts, tile_idx=ti)
            api.pathtrace_scene(ctx, res, gpu_scene, tex, ptype, api.PathtraceDesc(camera_params=cam_params, camera_transform=cam_transform, tile_params=tp))
            oracle.pathtrace(host_scene, width, height, cam_params, cam_transform, bounces, spp, ptype, tile_params=tp, out=want)
            (ox, oy), gx, gy = oracle.dispatch_extent(width, height, tp)
            regions.append((slice(oy, oy + gy * 4), slice(ox, ox + gx * 4)))
        got = tex.download()
        out["tiles"] = [list(t) for t in tiles]
    bad, npx, se = 0, 0, 0.0
    for ry, rx in regions:
        g, w = got[ry, rx], want[ry, rx]
        bad += int((g.view(np.uint16) != w.view(np.uint16)).sum())
        d = g[..., :3].astype(np.float32) - w[..., :3].astype(np.float32)
        se += float((d.astype(np.float64) ** 2).sum())
        npx += d.size
    out["differing_f16_words"] = bad
    out["rmse_vs_cpu_restatement"] = (se / max(1, npx)) ** 0.5
    out["pixels_compared"] = npx // 3
    return out


def accuracy_1024spp(api, oracle, ctx, gpu_scene, host_scene, cam, size, bounces, spp, frames, band_tile):
    """SURVEY 8d accuracy line on config 2: per-pixel RMSE of the HIP image against the oracle after `frames` accumulation
    frames (frame 0 is discarded by the reference's 1/accum_counter blend, so frames - 1 of them contribute: 128 x 8 = 1024
    spp) in both accumulation modes, on one tile of the frame (reference TileParams rule; the oracle renders only that tile):
      f16 running average -- pathtracer.wgsl:275-289 as written: prev_frame is an Rgba16Float texel, re-quantised every frame;
      f32 accumulate      -- the same recurrence on unquantised values.
    Also the cost of the f16 accumulation itself: oracle f16 mode vs oracle f32 mode (RMSE and relative MSE)."""
    import numpy as np
    ts, ti = band_tile
    tp = api.TileParams(tile_size=ts, tile_idx=ti)
    res = api.build_pathtrace_resources(ctx, api.BakedPathtraceParams(max_bounces=bounces, samples_per_pixel=spp))
    (ox, oy), gx, gy = oracle.dispatch_extent(size, size, tp)
    ry, rx = slice(oy, oy + gy * 4), slice(ox, ox + gx * 4)
    out = {"spp_total": (frames - 1) * spp, "frames": frames, "region": f"tile {ti} of tile_size {ts}: {gx * 4}x{gy * 4} px at ({ox},{oy})"}
    t0 = time.perf_counter()
    ref16 = np.zeros((size, size, 4), np.float16)
    ref32 = np.zeros((size, size, 3), np.float32)
    scratch = np.zeros((size, size, 4), np.float16)
    for k in range(frames):
        oracle.pathtrace(host_scene, size, size, cam.params, cam.transform, bounces, spp, 0, accum_counter=k, prev_frame=ref16.copy(),
                         tile_params=tp, out=ref16)
        oracle.pathtrace(host_scene, size, size, cam.params, cam.transform, bounces, spp, 0, accum_counter=k, prev_frame_f32=ref32.copy(),
                         tile_params=tp, out=scratch, want_f32=ref32)
    out["oracle_seconds"] = time.perf_counter() - t0

    def rmse(a, b):
        d = a.astype(np.float64) - b.astype(np.float64)
        return float(np.sqrt((d ** 2).mean()))

    for mode, name in ((0, "f16_running_average"), (1, "f32_accumulate")):
        ctx.set_accumulation_mode(mode)
        dbuf = api.DoubleBufferedTexture(ctx, size, size)
        for k in range(frames):
            api.pathtrace_scene(ctx, res, gpu_scene, dbuf.front(), 0,
                                api.PathtraceDesc(accum_params=api.AccumulationParams(dbuf.back(), k), camera_params=cam.params,
                                                  camera_transform=cam.transform, tile_params=tp))
            dbuf.flip()
        dbuf.flip()
        got16 = dbuf.front().download()
        if mode == 0:
            out[name] = {"rmse_vs_oracle": rmse(got16[ry, rx, :3], ref16[ry, rx, :3]),
                         "differing_f16_words": int((got16[ry, rx].view(np.uint16) != ref16[ry, rx].view(np.uint16)).sum())}
        else:
            got32 = dbuf.front().download_f32()
            out[name] = {"rmse_vs_oracle": rmse(got32[ry, rx, :3], ref32[ry, rx]),
                         "differing_f32_words": int((got32[ry, rx, :3].view(np.uint32) != ref32[ry, rx].view(np.uint32)).sum()),
                         "f16_view_rmse_vs_oracle_f32": rmse(got16[ry, rx, :3], ref32[ry, rx])}
        del dbuf
    ctx.set_accumulation_mode(0)
    a, b = ref16[ry, rx, :3].astype(np.float64), ref32[ry, rx].astype(np.float64)
    out["f16_mode_vs_f32_mode"] = {"rmse": float(np.sqrt(((a - b) ** 2).mean())),
                                   "relative_mse": float((((a - b) / np.maximum(b, 1e-2)) ** 2).mean()),
                                   "mean_ratio": float(a.mean() / b.mean()) if b.mean() > 0 else None}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)   # two wavefronts of eight 4K frames; one of sixteen for the smaller per-rank dispatches of --gpus N
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="bistro_class", help="bistro_class (default) | cornellbox | a fixture scene name")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=8, help="samples per pixel per step (baked)")
    ap.add_argument("--bounces", type=int, default=16)
    ap.add_argument("--integrator", default="Standard", choices=["Standard", "MIS", "Naive", "Direct"], help="pathtrace type (the headline is Standard)")
    ap.add_argument("--tile-size", type=int, default=8, help="tile edge in 4-px workgroups for multi-GPU sharding")
    ap.add_argument("--gather", default="root", choices=["root", "all"], help="N > 1 readback: tiles to rank 0 (ncclSend / ncclRecv) or to every rank (ncclAllGather)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle legs (cpu_baseline, parity, accuracy)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-kernel passes (roofline = null)")
    ap.add_argument("--no-secondary", action="store_true", help="skip BASELINE configs 2 and 3")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from lupinpathtracer_amd import api, distributed

    ctx = api.Context(local_rank)
    info = api.runtime_info()
    assert info["num_hip_runtimes_mapped"] == 1, info   # one HIP runtime in the process: the one the library was built against
    comm = None
    if world > 1 or os.environ.get("LUPIN_BENCH_FORCE_DIST") == "1":   # the flag exercises the N > 1 code path on one GPU
        comm = distributed.rendezvous(ctx, rank, world)

    t_load = time.perf_counter()
    scene, cams = load_workload(args.scene, ctx)
    load_s = time.perf_counter() - t_load
    cam = cams[0]
    W, H = args.width, args.height
    keep_aspect = args.scene == "cornellbox" and W == H
    cam_params = camera_for(api, cam, W, H, keep_aspect)
    res = api.build_pathtrace_resources(ctx, api.BakedPathtraceParams(max_bounces=args.bounces, samples_per_pixel=args.spp))
    out = api.DoubleBufferedTexture(ctx, W, H)
    # setup, like the two lines above: the path state of all frames in flight (a rank's share of the tiles when sharded)
    tpx = args.tile_size * 4                                  # LUPIN_WORKGROUP_SIZE pixels per tile_size unit (renderer.rs:807-829)
    tiles = ((W - 1) // tpx + 1) * ((H - 1) // tpx + 1)
    ctx.reserve_path_state(W * H if comm is None else -(-tiles // world) * tpx * tpx, args.bounces, args.spp)
    ptype = api.PathtraceType[args.integrator]
    frame = [0]

    def step():
        desc = api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), frame[0]), camera_params=cam_params,
                                 camera_transform=cam.transform)
        if comm is None:
            api.pathtrace_scene(ctx, res, scene, out.front(), ptype, desc)
        else:
            api.pathtrace_scene_tiles(ctx, res, scene, out.front(), ptype, desc, args.tile_size, rank, world)
        out.flip()
        frame[0] += 1

    def gather():
        out.flip()   # front = last rendered frame
        if args.gather == "root":
            comm.gather_framebuffer_to(out.front(), args.tile_size, 0)   # the readback: rank 0 receives the other ranks' tiles
        else:
            comm.gather_framebuffer(out.front(), args.tile_size)         # every rank ends up with the whole frame
        out.flip()

    def full_sync():
        ctx.sync()              # hipStreamSynchronize on every stream of the context
        if comm is not None:
            comm.barrier()      # all ranks (an RCCL all-reduce; returns after the device finished it)
        ctx.sync()

    for _ in range(args.warmup):
        step()
    if comm is not None:   # warm the collective too
        gather()
    full_sync()
    ctx.stats_reset(0)

    # ---- timed region: exactly K steps + the readback gather ----
    full_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    render_s = gather_s = None
    if comm is not None:
        ctx.sync()                         # this rank's frames are done (the gather below waits for them anyway)
        render_s = time.perf_counter() - t0
        gather()
        ctx.sync()
        gather_s = time.perf_counter() - t0 - render_s
    full_sync()
    elapsed = time.perf_counter() - t0

    st = ctx.stats()
    lanes_used = int(st["frames_in_flight"])   # what the library used for this scene, not what an environment variable might say
    total_units, total_paths = float(st["path_bounces"]), float(st["paths"])
    rank_units = total_units
    if comm is not None:
        total_units, total_paths = (float(v) for v in comm.allreduce([total_units, total_paths], "sum"))
        elapsed = float(comm.allreduce([elapsed], "max")[0])
        lo = float(-comm.allreduce([-rank_units], "max")[0])
        hi = float(comm.allreduce([rank_units], "max")[0])
        # per-phase times of the ranks (diagnosis of a scaling curve: is it the slowest rank's rendering or the exchange?)
        rmax, gmax = (float(v) for v in comm.allreduce([render_s, gather_s], "max"))
        rmin, gmin = (float(-v) for v in comm.allreduce([-render_s, -gather_s], "max"))
        phases = {"render_ms_min_max": [rmin * 1e3, rmax * 1e3], "gather_ms_min_max": [gmin * 1e3, gmax * 1e3],
                  "gather": "ncclSend/ncclRecv to rank 0" if args.gather == "root" else "ncclAllGather",
                  "gather_payload_bytes": W * H * 8}
    else:
        lo = hi = rank_units
        phases = None

    extras = {}
    if rank == 0 and world == 1 and comm is None:
        extras = single_gpu_extras(args, api, ctx, scene, cam, cam_params, W, H, ptype)

    if rank == 0:
        value = total_units / elapsed / 1e6
        line = {
            "metric": "Msamples/sec (paths x bounces)", "value": value, "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scene} {W}x{H}, {args.bounces} bounces, {args.spp} spp per step, {args.integrator} integrator, software BVH "
                                   f"(BASELINE configs[4] frame; stand-in scene for bistroexterior, SURVEY 8d)",
                       "scene": getattr(scene, "stats", None), "samples_per_pixel_per_step": args.spp,
                       "spp_total_timed": args.spp * args.steps, "frames_in_flight": lanes_used, "frames_per_wavefront": int(st["frames_per_wavefront"]),
                       "first_pass_stack_entries": int(st["short_stack_entries"]), "traversal": "wide" if st["wide_traversal"] else "binary",
                       "scene_load_and_build_s": load_s,
                       "sharding": "single dispatch" if comm is None else
                                   f"same frame tile-sharded over {world} ranks, tile {args.tile_size * 4}px round-robin, RCCL gather at readback (C ABI, --gather {args.gather})",
                       "rank_path_bounces_min_max": [lo, hi], "rank_phases": phases},
            "Mpaths_per_s": total_paths / elapsed / 1e6,
            "path_bounces": total_units, "timed_seconds": elapsed,
            "hip_runtime": info,
        }
        line.update({"roofline": None, "cpu_baseline": None, "parity": None})
        line.update(extras)
        print(json.dumps(line))
        sys.stdout.flush()
    if comm is not None:
        comm.barrier()
        comm.close()


def single_gpu_extras(args, api, ctx, scene, cam, cam_params, W, H, ptype):
    """roofline / cpu_baseline / parity of the headline workload and the secondary BASELINE configs (rank 0, N = 1)."""
    extras = {}
    peak_measured = None
    if not args.no_kernel_timing:
        peak_measured = ctx.measure_copy_bandwidth(1 << 31, 6)
        key = f"{args.scene}_{W}x{H}_b{args.bounces}_spp{args.spp}_{args.integrator.lower()}"
        rec, _, _ = measure_single_gpu(api, ctx, scene, cam, W, H, args.bounces, args.spp, 2, 0, ptype, key, peak_measured,
                                       keep_aspect=(cam_params is cam.params))
        extras["roofline"] = rec.get("roofline")
        extras["kernel_ms"] = rec.get("kernel_ms")
        extras["hbm_peak"] = {"nominal_GBps": HBM_PEAK_GBS, "measured_copy_GBps": peak_measured,
                              "method": "lupin_hip_measure_copy_bandwidth: 2 GiB device-to-device copy kernel, read + written bytes / time"}

    oracle = None
    if not args.no_cpu_baseline:
        from oracle import oracle
        host_scene, host_cams = load_workload(args.scene, None)
        hc = host_cams[0]
        threads = oracle.num_threads()
        # bounded sample of the same workload: the same scene and view at 1/4 of the width and height
        sw, sh = max(64, W // 4), max(64, H // 4)
        sparams = camera_for(api, hc, sw, sh, keep_aspect=(cam_params is cam.params))
        oracle.pathtrace(host_scene, 64, 64, sparams, hc.transform, args.bounces, 1)   # page the library in
        par = parity_vs_oracle(api, oracle, ctx, scene, host_scene, sparams, hc.transform, sw, sh, args.bounces, args.spp, ptype)
        extras["cpu_baseline"] = {"value": par["oracle_path_bounces"] / par["oracle_seconds"] / 1e6, "unit": "Msamples/s", "cores": threads,
                                  "kind": "port",
                                  "sample": f"1 step of the same scene and view at {sw}x{sh}, {args.bounces} bounces, {args.spp} spp "
                                            f"({par['oracle_path_bounces']} path-bounces, {par['oracle_seconds']:.2f} s, OpenMP over rows)"}
        # ... and tiles of the full-size frame (reference TileParams rule): first, a middle one, the last full one
        ntiles = api.get_num_tiles(args.tile_size, W, H)
        ntx = (W - 1) // (args.tile_size * 4) + 1
        tiles = sorted({0, (ntiles // 2) + ntx // 3, ntiles - ntx - 2}) if ntiles > 2 * ntx + 2 else [0]
        full = parity_vs_oracle(api, oracle, ctx, scene, host_scene, cam_params, hc.transform, W, H, args.bounces, args.spp, ptype,
                                tiles=[(args.tile_size, t) for t in tiles])
        extras["parity"] = {"differing_f16_words": par["differing_f16_words"] + full["differing_f16_words"],
                            "rmse_vs_cpu_restatement": max(par["rmse_vs_cpu_restatement"], full["rmse_vs_cpu_restatement"]),
                            "reduced_size_frame": par, "full_size_tiles": full}

    if not args.no_secondary and args.scene == "bistro_class":
        extras["configs"] = []
        for name, scene_name, w, h, bounces, steps, warm, aspect_keep in (("configs[1] cornellbox 1024x1024 b8", "cornellbox", 1024, 1024, 8, 64, 64, True),
                                                                         ("configs[2] materials1 1920x1080 b12", "materials1", 1920, 1080, 12, 32, 16, False),
                                                                         ("configs[3] environments1 1920x1080 b16", "environments1", 1920, 1080, 16, 32, 16, False)):
            # (warm-up = full wavefronts on every lane the scene will use -- four for the LDS-resident Cornell box -- so that no
            # lane allocates its path state inside the timed steps)
            sc2, cams2 = load_workload(scene_name, ctx)
            key = f"{scene_name}_{w}x{h}_b{bounces}_spp{args.spp}_standard"
            if args.no_kernel_timing:
                continue
            rec, _, params2 = measure_single_gpu(api, ctx, sc2, cams2[0], w, h, bounces, args.spp, steps, warm, ptype, key, peak_measured, keep_aspect=aspect_keep)
            rec["workload"] = f"{name}, {args.spp} spp per step, Standard"
            if oracle is not None:
                hs2, hc2 = load_workload(scene_name, None)
                ts = 8
                ntx = (w - 1) // (ts * 4) + 1
                nt = api.get_num_tiles(ts, w, h)
                band = [(ts, t) for t in sorted({nt // 2 + ntx // 4, nt // 2 + ntx // 2, nt // 3})]
                rec["parity"] = parity_vs_oracle(api, oracle, ctx, sc2, hs2, params2, hc2[0].transform, w, h, bounces, args.spp, ptype, tiles=band)
                if scene_name == "cornellbox":
                    rec["accuracy_1024spp"] = accuracy_1024spp(api, oracle, ctx, sc2, hs2, hc2[0], 1024, bounces, args.spp, 129, (32, 8 * 3 + 4))
            extras["configs"].append(rec)
            del sc2
    return extras


if __name__ == "__main__":
    main()
