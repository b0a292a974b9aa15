"""Worker for tests/test_distributed_cpu.py: one rank of a world_size-N gloo job (CPU).
Renders its own tiles with the oracle, then runs the product's gather logic (lupinpathtracer_amd.distributed)
with the numpy device double."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from lupinpathtracer_amd import api, loader, distributed
    from oracle import oracle

    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    W, H, ts = 48, 40, 2          # 8-px tiles: 6 x 5 = 30 tiles
    scene, cams = loader.build_scene_cornell_box(None)
    cam = cams[0]
    fb = np.zeros((H, W, 4), np.float16)
    for t in distributed.owned_tiles(W, H, ts, rank, world):
        oracle.pathtrace(scene, W, H, cam.params, cam.transform, 4, 2, tile_params=api.TileParams(ts, t), out=fb, num_threads=1)
    ops = distributed.NumpyTileOps(torch)
    own = fb.copy()
    # readback form first (only the last rank receives), on a copy; then the all-gather on the real framebuffer
    root = world - 1
    fb_to = own.copy()
    sent = distributed.gather_framebuffer_to(dist, ops, fb_to, W, H, ts, rank, world, root=root)
    others_untouched = bool(rank == root or np.array_equal(fb_to.view(np.uint16), own.view(np.uint16)))
    flags = [None] * world
    dist.all_gather_object(flags, (others_untouched, int(sent)))
    if rank == root and root != 0:
        dist.send(torch.from_numpy(fb_to.view(np.int64).reshape(-1).copy()), dst=0)
    if rank == 0 and root != 0:
        buf = torch.empty(W * H, dtype=torch.int64)
        dist.recv(buf, src=root)
        fb_to = buf.numpy().view(np.float16).reshape(H, W, 4)
    nbytes = distributed.gather_framebuffer(dist, ops, fb, W, H, ts, rank, world)
    dist.barrier()
    if rank == 0:
        full, _ = oracle.pathtrace(scene, W, H, cam.params, cam.transform, 4, 2, num_threads=1)
        np.savez(out_path, gathered=fb, full=full, nbytes=nbytes, gathered_to=fb_to,
                 others_untouched=np.array([f[0] for f in flags]), sent=np.array([f[1] for f in flags]))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
