"""Worker of test_bench_cli.py::test_gather_through_the_c_abi (its own process: it loads librccl).

1. world 3 on one GPU without a collective: every "rank" renders its tiles, the payloads are laid out as an all-gather
   would leave them, lupin_hip_unpack_gathered_tiles scatters them -- every rank's frame must equal the single dispatch.
2. lupin_hip_gather_framebuffer through a real RCCL communicator of world size 1 (rendezvous file, ncclCommInitRank,
   ncclAllGather on the context's stream), with HIP-graph replay on when LUPIN_GRAPH=1 is set by the caller.
3. exactly one HIP runtime is mapped in the process, the one the library was built against.
4. lupin_hip_comm_init_all / lupin_hip_gather_framebuffer_all (one process, one context per device) with the one device at hand."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lupinpathtracer_amd import api, distributed, loader   # noqa: E402
from tests import util   # noqa: E402

assert "torch" not in sys.modules
ctx = api.Context(0)
info = api.runtime_info()
assert info["num_hip_runtimes_mapped"] == 1 and info["build_hip_version"] // 100000 == info["runtime_hip_version"] // 100000, info
scene, cams = loader.build_scene_cornell_box(ctx)
cam = cams[0]
W, H, ts, world = 200, 136, 4, 3          # 13 x 9 tiles of 16 px, partial tiles on both edges
res = api.build_pathtrace_resources(ctx, api.BakedPathtraceParams(max_bounces=5, samples_per_pixel=2))
desc = api.PathtraceDesc(camera_params=cam.params, camera_transform=cam.transform)
full = api.Texture(ctx, W, H)
api.pathtrace_scene(ctx, res, scene, full, 0, desc)
want = full.download()

capacity = max(distributed.packed_pixels(W, H, ts, r, world) for r in range(world))
gathered = api.Texture(ctx, capacity * world, 1)      # any device buffer of world * capacity * 8 bytes
shards = []
for r in range(world):
    t = api.Texture(ctx, W, H)
    api.pathtrace_scene_tiles(ctx, res, scene, t, 0, desc, ts, r, world)
    shards.append(t)
    assert api.pack_tiles(ctx, t, ts, r, world, gathered.device_ptr() + r * capacity * 8) == distributed.packed_pixels(W, H, ts, r, world)
for r in range(world):
    api.unpack_gathered_tiles(ctx, shards[r], ts, r, world, gathered.device_ptr(), capacity)
    assert util.f16_words_differ(shards[r].download(), want) == 0, r
print("SCATTER OK")

comm = distributed.rendezvous(ctx, 0, 1, path=os.path.join(os.environ.get("TMPDIR", "/tmp"), f"lupin_rdzv_test_{os.getpid()}"))
assert (comm.rank, comm.world) == (0, 1)
out = api.DoubleBufferedTexture(ctx, W, H)
for k in range(4):   # frames, a gather in the middle, more frames: the sequence that faulted under graph replay in round 1
    api.pathtrace_scene_tiles(ctx, res, scene, out.front(), 0,
                              api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), k), camera_params=cam.params, camera_transform=cam.transform),
                              ts, 0, 1)
    if k == 1:
        comm.gather_framebuffer(out.front(), ts)
    if k == 2:
        comm.gather_framebuffer_to(out.front(), ts, 0)   # the readback form (world 1: pack only; the call must still be well-formed)
    out.flip()
out.flip()
comm.gather_framebuffer_to(out.front(), ts, 0)
comm.gather_framebuffer(out.front(), ts)
got = out.front().download()
ref = util.gpu_accumulate(ctx, scene, cam, W, H, frames=4, spp=2, max_bounces=5)
assert util.f16_words_differ(got, ref) == 0
assert float(comm.allreduce([3.0, 4.0], "sum")[1]) == 4.0 and float(comm.allreduce([5.0], "max")[0]) == 5.0
try:
    comm.gather_framebuffer_to(out.front(), ts, 1)   # no such rank
    raise SystemExit("root out of range was accepted")
except api.LupinError:
    pass
comm.barrier()
comm.close()
print("GATHER OK")

# 2b. f32 accumulation mode: tiles that arrive through an unpack must refresh the f32 accumulator too (advisor, round 2)
import numpy as np   # noqa: E402
ctx.set_accumulation_mode(1)
try:
    frames32 = []
    for r in range(world):
        t = api.Texture(ctx, W, H)
        api.pathtrace_scene_tiles(ctx, res, scene, t, 0, desc, ts, r, world)
        frames32.append(t)
        api.pack_tiles(ctx, t, ts, r, world, gathered.device_ptr() + r * capacity * 8)
    for r in range(world):
        api.unpack_gathered_tiles(ctx, frames32[r], ts, r, world, gathered.device_ptr(), capacity)
        f32 = frames32[r].download_f32()
        f16 = frames32[r].download()
        assert util.f16_words_differ(f16, want) == 0
        # every texel of the accumulator is what its f16 view shows (own tiles: the unrounded value, within one f16 ulp; tiles of
        # another rank: the f16 texel widened) -- in particular no zero or stale holes where other ranks rendered
        assert np.allclose(f32[..., :3], f16[..., :3].astype(np.float32), rtol=1.5e-3, atol=1e-7)
        assert np.all(f32[..., 3] == 1.0)
finally:
    ctx.set_accumulation_mode(0)
print("F32 UNPACK OK")

# 4. the one-process-N-contexts layout (lupin_hip_comm_init_all + lupin_hip_gather_framebuffer_all, one RCCL group) with the
#    one device this box has: same frames, same result
(comm1,) = api.Comm.init_all([ctx])
assert (comm1.rank, comm1.world) == (0, 1)
out = api.DoubleBufferedTexture(ctx, W, H)
for k in range(3):
    api.pathtrace_scene_tiles(ctx, res, scene, out.front(), 0,
                              api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), k), camera_params=cam.params, camera_transform=cam.transform),
                              ts, 0, 1)
    api.gather_framebuffer_all([comm1], [out.front()], ts)
    out.flip()
out.flip()
ref = util.gpu_accumulate(ctx, scene, cam, W, H, frames=3, spp=2, max_bounces=5)
assert util.f16_words_differ(out.front().download(), ref) == 0
comm1.close()
print("INIT_ALL OK")
