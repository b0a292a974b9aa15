"""Tile-sharded rendering across the GPUs of one node.

The path shards by pixel tiles: every pixel owns its RNG stream (seeded from the GLOBAL pixel index and
accum_counter, pathtracer.wgsl:224-226) and its output texel, so ranks never exchange anything while
accumulating.  Tiles (tile_size x 4 pixels, numbered row-major like renderer.rs:816-817) are dealt
round-robin, rank r owns tiles r, r + world, ... (rows rotated when that would make column stripes, see
`owned_tiles`) -- interleaved because per-tile cost varies a lot (sky vs geometry).  The one exchange step is the gather of per-tile Rgba16Float payloads at readback: one all-gather of
equally sized packed buffers (RCCL over xGMI on the GPU box; gloo in the CPU tests).

On the GPU the whole exchange is one C-ABI call, `lupin_hip_gather_framebuffer` (`api.Comm.gather_framebuffer`): pack
kernel -> ncclAllGather -> unpack kernel on the context's stream, no torch in the process.  `rendezvous` hands rank 0's
RCCL unique id to the other ranks of a one-process-per-GPU job through a file (the ranks of one node share /tmp).

The functions below restate the tile arithmetic of include/lupin_tiles.h and the payload layout in numpy; `gather_framebuffer`
runs the same pack -> all-gather -> unpack sequence over any `torch.distributed`-like collective (the CPU tests drive it
with gloo and `NumpyTileOps`), which pins the layout the HIP kernels must produce.
"""
import os
import time

import numpy as np

WORKGROUP_SIZE = 4


def tile_grid(width, height, tile_size):
    tpx = tile_size * WORKGROUP_SIZE
    return (max(1, width) - 1) // tpx + 1, (max(1, height) - 1) // tpx + 1, tpx


def owned_tiles(width, height, tile_size, rank, world):
    """include/lupin_tiles.h restated: round-robin over the row-major tile index; when a row holds a multiple of `world`
    tiles (which would give every rank fixed columns) each row is rotated by one more: owner = (tx + ty) % world."""
    ntx, nty, _ = tile_grid(width, height, tile_size)
    if world > 1 and ntx % world == 0:
        return [ty * ntx + tx for ty in range(nty) for tx in range((rank - ty) % world, ntx, world)]
    return list(range(rank, ntx * nty, world))


def tile_rect(width, height, tile_size, t):
    ntx, nty, tpx = tile_grid(width, height, tile_size)
    ox, oy = (t % ntx) * tpx, (t // ntx) * tpx
    return ox, oy, min(tpx, width - ox), min(tpx, height - oy)


def packed_pixels(width, height, tile_size, rank, world):
    return sum(w * h for (_, _, w, h) in (tile_rect(width, height, tile_size, t) for t in owned_tiles(width, height, tile_size, rank, world)))


def pack_tiles_numpy(image, tile_size, rank, world):
    """Reference layout of the packed payload: owned tiles in ascending order, each tile row-major, 8 B per pixel.
    image: (H, W, 4) float16.  (The HIP kernel k_pack_tiles produces exactly this.)"""
    h, w = image.shape[:2]
    parts = []
    for t in owned_tiles(w, h, tile_size, rank, world):
        ox, oy, tw, th = tile_rect(w, h, tile_size, t)
        parts.append(image[oy:oy + th, ox:ox + tw].reshape(-1, 4))
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, 4), image.dtype)


def unpack_tiles_numpy(image, packed, tile_size, rank, world):
    h, w = image.shape[:2]
    pos = 0
    for t in owned_tiles(w, h, tile_size, rank, world):
        ox, oy, tw, th = tile_rect(w, h, tile_size, t)
        image[oy:oy + th, ox:ox + tw] = packed[pos:pos + tw * th].reshape(th, tw, 4)
        pos += tw * th
    return image


class NumpyTileOps:
    """Test double for the device side: framebuffers are (H, W, 4) float16 numpy arrays."""

    def __init__(self, torch):
        self.torch = torch

    def pack(self, image, tile_size, rank, world, capacity_pixels):
        p = pack_tiles_numpy(image, tile_size, rank, world)
        buf = np.zeros((capacity_pixels, 4), np.float16)
        buf[:len(p)] = p
        return self.torch.from_numpy(buf.view(np.int64).reshape(-1).copy())   # one int64 word per Rgba16Float pixel

    def unpack(self, image, payload, tile_size, rank, world):
        packed = payload.contiguous().numpy().view(np.float16).reshape(-1, 4)
        unpack_tiles_numpy(image, packed, tile_size, rank, world)


def rendezvous_path(env=None):
    """Where rank 0 publishes the RCCL unique id.  Ranks started by one launcher share their parent process
    (`torch.distributed.run`'s agent, or bench.py's own spawner), so (parent pid, MASTER_PORT) names the job."""
    env = os.environ if env is None else env
    if env.get("LUPIN_RDZV_FILE"):
        return env["LUPIN_RDZV_FILE"]
    return os.path.join(env.get("TMPDIR", "/tmp"), f"lupin_rdzv_{os.getppid()}_{env.get('MASTER_PORT', '0')}")


def job_nonce(env=None):
    """32 bytes naming THIS job: the launcher's pid, its start time (so a recycled pid is another job) and MASTER_PORT.
    Every rank of one launch computes the same value; a file left behind by a job that died carries another one."""
    import hashlib
    env = os.environ if env is None else env
    ppid = os.getppid()
    start = "?"
    try:
        with open(f"/proc/{ppid}/stat") as f:
            start = f.read().rsplit(")", 1)[1].split()[19]   # field 22: starttime in clock ticks since boot
    except (OSError, IndexError):
        pass
    return hashlib.sha256(f"{ppid}:{start}:{env.get('MASTER_PORT', '0')}:{env.get('LUPIN_RDZV_NONCE', '')}".encode()).digest()


def publish_unique_id(path, uid, nonce):
    """Rank 0: remove whatever is at `path` (a leftover of a dead job), then publish id + nonce atomically."""
    try:
        os.remove(path)
    except OSError:
        pass
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "wb") as f:
        f.write(uid + nonce)
    os.replace(tmp, path)


def wait_unique_id(path, nonce, timeout=300.0, rank=-1):
    """Other ranks: wait for a file that carries THIS job's nonce; anything else at the path is ignored."""
    deadline = time.monotonic() + timeout
    while True:
        try:
            with open(path, "rb") as f:
                blob = f.read()
            if len(blob) == 128 + len(nonce) and blob[128:] == nonce:
                return blob[:128]
        except FileNotFoundError:
            pass
        if time.monotonic() > deadline:
            raise TimeoutError(f"rank {rank}: no RCCL unique id of this job at {path} after {timeout:.0f} s")
        time.sleep(0.01)


def rendezvous(ctx, rank, world, path=None, timeout=300.0):
    """One-process-per-GPU communicator: rank 0 makes the unique id (ncclGetUniqueId) and writes it, tagged with the job's
    nonce, atomically to `path`; the others wait for a file with that nonce (a stale file of another job would make
    ncclCommInitRank hang on a mismatched id); everyone then joins with ncclCommInitRank.  Returns an api.Comm."""
    from . import api
    path = path or rendezvous_path()
    nonce = job_nonce()
    if rank == 0:
        uid = api.Comm.unique_id()
        publish_unique_id(path, uid, nonce)
    else:
        uid = wait_unique_id(path, nonce, timeout, rank)
    comm = api.Comm.init_rank(ctx, uid, rank, world)
    comm.barrier()   # everyone has read the file
    if rank == 0:
        try:
            os.remove(path)
        except OSError:
            pass
    return comm


def gather_framebuffer_to(dist, ops, framebuffer, width, height, tile_size, rank, world, root=0):
    """Host-logic twin of lupin_hip_gather_framebuffer_to: every rank but `root` sends its exact tile payload, the root
    receives them and scatters.  Returns the payload bytes this rank sent (0 on the root)."""
    torch = ops.torch
    if rank == root:
        for r in range(world):
            if r == root:
                continue
            n = packed_pixels(width, height, tile_size, r, world)
            if n == 0:
                continue
            buf = torch.empty(n, dtype=torch.int64)
            dist.recv(buf, src=r)
            ops.unpack(framebuffer, buf, tile_size, r, world)
        if hasattr(ops, "finish"):
            ops.finish()
        return 0
    n = packed_pixels(width, height, tile_size, rank, world)
    if n:
        dist.send(ops.pack(framebuffer, tile_size, rank, world, n), dst=root)
    return n * 8


def gather_framebuffer(dist, ops, framebuffer, width, height, tile_size, rank, world):
    """Host-logic twin of lupin_hip_gather_framebuffer over a torch.distributed-like collective: all-gather the per-rank tile
    payloads and scatter them into `framebuffer` on every rank.  Returns the payload bytes this rank contributed."""
    capacity = max(packed_pixels(width, height, tile_size, r, world) for r in range(world))
    mine = ops.pack(framebuffer, tile_size, rank, world, capacity)
    torch = ops.torch
    gathered = torch.empty(world * mine.numel(), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(gathered, mine)
    for r in range(world):
        if r != rank:
            ops.unpack(framebuffer, gathered[r * mine.numel():(r + 1) * mine.numel()], tile_size, r, world)
    if hasattr(ops, "finish"):
        ops.finish()
    return capacity * 8
