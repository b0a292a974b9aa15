"""numpy restatement of the device BLAS builder (lupinpathtracer_amd/csrc/lbvh.hip) -- the checker for that kernel
chain.  The builder is this repository's own algorithm (the reference builds its BVH on the CPU only), so this file,
not oracle/, is its specification: same f32 operations in the same order, stable sort, complete binary tree."""
import numpy as np

from lupinpathtracer_amd._abi import BVH_NODE_DTYPE


def lbvh_depth(n):
    d = 0
    while d < 31 and (2 << d) < n:
        d += 1
    return d


def _spread10(v):
    v = v.astype(np.uint64)
    v = (v * 0x00010001) & 0xFF0000FF
    v = (v * 0x00000101) & 0x0F00F00F
    v = (v * 0x00000011) & 0xC30C30C3
    v = (v * 0x00000005) & 0x49249249
    return v.astype(np.uint32)


def build(verts_pos4, indices):
    v = np.ascontiguousarray(verts_pos4, np.float32).reshape(-1, 4)[:, :3]
    idx = np.ascontiguousarray(indices, np.uint32).reshape(-1, 3)
    n = len(idx)
    tri = v[idx]                                               # (n, 3 verts, 3)
    centre = ((tri.min(axis=1) + tri.max(axis=1)) * np.float32(0.5)).astype(np.float32)
    lo, hi = centre.min(axis=0), centre.max(axis=0)
    ext = (hi - lo).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        f = ((centre - lo) / ext * np.float32(1024.0)).astype(np.float32)
    f = np.where(ext > 0, f, np.float32(0.0))
    q = np.clip(f, np.float32(0.0), np.float32(1023.0)).astype(np.uint32)
    keys = (_spread10(q[:, 0]) << 2) | (_spread10(q[:, 1]) << 1) | _spread10(q[:, 2])
    order = np.argsort(keys, kind="stable")
    depth = lbvh_depth(n)
    leaves = 1 << depth
    nodes = np.zeros(2 * leaves - 1, BVH_NODE_DTYPE)
    bounds = (np.arange(leaves + 1, dtype=np.uint64) * np.uint64(n)) >> np.uint64(depth)
    sorted_tri = tri[order]
    for k in range(leaves):
        a, b = int(bounds[k]), int(bounds[k + 1])
        pts = sorted_tri[a:b].reshape(-1, 3)
        nd = nodes[leaves - 1 + k]
        nd["aabb_min"] = pts.min(axis=0)
        nd["aabb_max"] = pts.max(axis=0)
        nd["tri_begin_or_first_child"] = a
        nd["tri_count"] = b - a
    for level in range(depth - 1, -1, -1):
        for k in range(1 << level):
            i = (1 << level) - 1 + k
            l, r = nodes[2 * i + 1], nodes[2 * i + 2]
            nodes[i]["aabb_min"] = np.minimum(l["aabb_min"], r["aabb_min"])
            nodes[i]["aabb_max"] = np.maximum(l["aabb_max"], r["aabb_max"])
            nodes[i]["tri_begin_or_first_child"] = 2 * i + 1
            nodes[i]["tri_count"] = 0
    return nodes, idx[order].reshape(-1)
