"""The N > 1 path on the CPU: tile partitioning, the packed payload layout, and the gather over a real
world_size-2 (and 3) torch.distributed job with the gloo backend."""
import os
import subprocess
import sys

import numpy as np
import pytest

from lupinpathtracer_amd import api, distributed

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("w,h,ts,world", [(1024, 1024, 32, 8), (2048, 1024, 32, 2), (100, 37, 3, 3), (64, 64, 16, 5), (7, 5, 1, 2),
                                           (1000, 700, 25, 5), (2048, 2048, 8, 4), (90, 50, 5, 3)])
def test_tile_partition_is_exact(built, w, h, ts, world):
    seen = np.zeros((h, w), np.int32)
    total = 0
    for r in range(world):
        px = 0
        for t in distributed.owned_tiles(w, h, ts, r, world):
            ox, oy, tw, th = distributed.tile_rect(w, h, ts, t)
            seen[oy:oy + th, ox:ox + tw] += 1
            px += tw * th
        assert px == distributed.packed_pixels(w, h, ts, r, world) == api.packed_tile_pixels(w, h, ts, r, world)
        total += px
    assert np.all(seen == 1) and total == w * h
    assert sum(len(distributed.owned_tiles(w, h, ts, r, world)) for r in range(world)) == api.get_num_tiles(ts, w, h)
    # no rank is confined to fixed columns (rows are rotated when a row holds a multiple of `world` tiles)
    ntx, nty, _ = distributed.tile_grid(w, h, ts)
    if nty >= world and ntx >= world:
        for r in range(world):
            cols = {t % ntx for t in distributed.owned_tiles(w, h, ts, r, world)}
            assert len(cols) == ntx, (r, sorted(cols))


def test_pack_unpack_round_trip(built):
    rng = np.random.default_rng(0)
    img = rng.random((37, 100, 4)).astype(np.float16)
    out = np.zeros_like(img)
    for r in range(3):
        p = distributed.pack_tiles_numpy(img, 3, r, 3)
        assert len(p) == distributed.packed_pixels(100, 37, 3, r, 3)
        distributed.unpack_tiles_numpy(out, p, 3, r, 3)
    assert np.array_equal(out.view(np.uint16), img.view(np.uint16))


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_gather_reproduces_single_process_render(built, tmp_path, world):
    out = str(tmp_path / "gathered.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + world + os.getpid() % 200), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], os.path.join(HERE, "_dist_worker.py"), out]
    subprocess.run(cmd, check=True, env=env, timeout=600)
    z = np.load(out)
    assert np.array_equal(z["gathered"].view(np.uint16), z["full"].view(np.uint16))
    assert int(z["nbytes"]) >= 48 * 40 * 8 // world
