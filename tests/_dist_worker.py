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
    nbytes = distributed.gather_framebuffer(dist, ops, fb, W, H, ts, rank, world)
    dist.barrier()
    if rank == 0:
        full, _ = oracle.pathtrace(scene, W, H, cam.params, cam.transform, 4, 2, num_threads=1)
        np.savez(out_path, gathered=fb, full=full, nbytes=nbytes)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
