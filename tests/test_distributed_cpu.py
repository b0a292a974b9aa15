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
    # the readback form (gather_framebuffer_to): the root holds the whole frame, nobody else's framebuffer changed, and the
    # payloads sent add up to exactly the pixels the root does not own
    assert np.array_equal(z["gathered_to"].view(np.uint16), z["full"].view(np.uint16))
    assert z["others_untouched"].all()
    root = world - 1
    assert int(z["sent"].sum()) == (48 * 40 - distributed.packed_pixels(48, 40, 2, root, world)) * 8 and int(z["sent"][root]) == 0


def test_rendezvous_ignores_a_stale_file(built, tmp_path, monkeypatch):
    """A leftover id file of a dead job (same path) must not be taken for this job's: readers match the job nonce."""
    import threading
    path = str(tmp_path / "rdzv")
    monkeypatch.setenv("MASTER_PORT", "29999")
    nonce = distributed.job_nonce()
    assert len(nonce) == 32 and nonce == distributed.job_nonce()
    monkeypatch.setenv("LUPIN_RDZV_NONCE", "another job")
    stale_nonce = distributed.job_nonce()
    assert stale_nonce != nonce
    monkeypatch.delenv("LUPIN_RDZV_NONCE")
    with open(path, "wb") as f:                      # what a job that died before its clean-up leaves behind
        f.write(bytes(range(128)) + stale_nonce)
    with pytest.raises(TimeoutError):
        distributed.wait_unique_id(path, nonce, timeout=0.2, rank=1)
    with open(path, "wb") as f:                      # a pre-nonce 128-byte file is not accepted either
        f.write(bytes(128))
    with pytest.raises(TimeoutError):
        distributed.wait_unique_id(path, nonce, timeout=0.1, rank=1)
    uid = bytes(reversed(range(128)))
    t = threading.Timer(0.1, distributed.publish_unique_id, (path, uid, nonce))
    t.start()
    assert distributed.wait_unique_id(path, nonce, timeout=5.0, rank=1) == uid
    t.join()


def test_missing_rccl_is_an_error_code_not_a_crash(built, tmp_path):
    """LUPIN_RCCL_LIB names the library to load; one that cannot be loaded must come back as LUPIN_ERR_RCCL with dlopen's
    message (the failure path once dereferenced a second dlerror() == NULL)."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from lupinpathtracer_amd import _abi\n"
            "import ctypes as C\n"
            "buf = (C.c_uint8 * 128)()\n"
            "rc = _abi.lib().lupin_hip_comm_get_unique_id(C.cast(buf, C.c_void_p))\n"
            "msg = _abi.lib().lupin_hip_last_error().decode()\n"
            "print(rc, msg)\n"
            "assert rc == -8 and 'librccl could not be loaded' in msg and 'no_such_rccl' in msg, (rc, msg)\n") % os.path.dirname(HERE)
    env = dict(os.environ, LUPIN_RCCL_LIB=str(tmp_path / "no_such_rccl.so"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
