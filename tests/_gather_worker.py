"""Worker of test_bench_cli.py::test_gather_framebuffer_with_device_payloads: world 3 on one GPU, the other ranks'
payloads come from a stand-in collective, the gathered frame must equal the single-dispatch frame."""
import os
import sys

import torch   # first: torch ships its own HIP runtime

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.cuda.set_device(0)

from lupinpathtracer_amd import api, distributed, loader   # noqa: E402
from tests import util   # noqa: E402

ctx = api.Context(0)
scene, cams = loader.build_scene_cornell_box(ctx)
cam = cams[0]
W, H, ts, world = 200, 136, 4, 3          # 13 x 9 tiles of 16 px, partial tiles on both edges
res = api.build_pathtrace_resources(ctx, api.BakedPathtraceParams(max_bounces=5, samples_per_pixel=2))
desc = api.PathtraceDesc(camera_params=cam.params, camera_transform=cam.transform)
full = api.Texture(ctx, W, H)
api.pathtrace_scene(ctx, res, scene, full, 0, desc)
ops = distributed.HipTileOps(torch, ctx, torch.device("cuda", 0))
capacity = max(distributed.packed_pixels(W, H, ts, r, world) for r in range(world))
shards, payloads = [], []
for r in range(world):
    t = api.Texture(ctx, W, H)
    api.pathtrace_scene_tiles(ctx, res, scene, t, 0, desc, ts, r, world)
    shards.append(t)
    payloads.append(ops.pack(t, ts, r, world, capacity))


class StandInCollective:
    def all_gather_into_tensor(self, out, mine):
        for r in range(world):
            out[r * mine.numel():(r + 1) * mine.numel()].copy_(payloads[r])


want = full.download()
for rank in range(world):
    distributed.gather_framebuffer(StandInCollective(), ops, shards[rank], W, H, ts, rank, world)
    assert util.f16_words_differ(shards[rank].download(), want) == 0, rank
print("GATHER OK")
