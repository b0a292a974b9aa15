"""lupin_build_bvh against an independent numpy restatement of the reference's split rule (no reference fixture pins
the node arrays, so the property the reference's CODE fixes is replayed instead).  For every internal node of the built tree:
its triangle set -> choose_split (data_structures.rs:366-466: centroid bounds padded by 0.001, five bins per axis by
floor((c - cmin) * 5 / (cmax - cmin)), half-area x count costs, first strictly cheaper plane wins, axes in x, y, z order)
must give exactly the stored children: their boxes bit for bit, and the partition `centroid[axis] <= pos` (:249-262) must put
exactly the left child's triangles on the left.  Also the leaves: a node is a leaf exactly when no plane beats its own cost
(or the partition leaves one side empty, or the depth cap holds it)."""
import os

import numpy as np
import pytest

from lupinpathtracer_amd import api, loader
from tests import util

F = np.float32
NUM_BINS = 5
F32_MAX, F32_MIN = np.finfo(np.float32).max, np.finfo(np.float32).min   # Rust f32::MAX / f32::MIN (= -MAX)


def half_area_cost(size, count):
    """node_cost (data_structures.rs:468-475): (x * (y + z) + y * z) * count, f32, left to right."""
    with np.errstate(over="ignore", invalid="ignore"):
        return F(F(F(size[0] * F(size[1] + size[2])) + F(size[1] * size[2])) * F(count))


def choose_split(node_lo, node_hi, cent, blo, bhi, count_total):
    """One node: returns None (no plane is cheaper than the node itself) or (axis, pos, left box, right box)."""
    best_cost = half_area_cost((node_hi - node_lo).astype(F), count_total)
    best = None
    for axis in range(3):
        c = cent[:, axis]
        cmin, cmax = c.min(), c.max()
        if cmin == cmax:
            continue
        cmin = F(cmin - F(0.001)); cmax = F(cmax + F(0.001))
        scale = F(F(NUM_BINS) / F(cmax - cmin))
        bins = np.clip(np.floor(((c - cmin).astype(F) * scale).astype(F)).astype(np.int64), 0, NUM_BINS - 1)
        bin_lo = np.full((NUM_BINS, 3), F32_MAX, F); bin_hi = np.full((NUM_BINS, 3), F32_MIN, F)
        bin_n = np.zeros(NUM_BINS, np.int64)
        for b in range(NUM_BINS):
            m = bins == b
            bin_n[b] = m.sum()
            if bin_n[b]:
                bin_lo[b] = blo[m].min(axis=0); bin_hi[b] = bhi[m].max(axis=0)
        step = F(F(cmax - cmin) / F(NUM_BINS))
        for i in range(NUM_BINS - 1):
            l_lo = np.minimum.reduce(bin_lo[:i + 1]); l_hi = np.maximum.reduce(bin_hi[:i + 1])
            r_lo = np.minimum.reduce(bin_lo[i + 1:]); r_hi = np.maximum.reduce(bin_hi[i + 1:])
            ln, rn = int(bin_n[:i + 1].sum()), int(bin_n[i + 1:].sum())
            with np.errstate(over="ignore", invalid="ignore"):
                cost = F(half_area_cost((l_hi - l_lo).astype(F), ln) + half_area_cost((r_hi - r_lo).astype(F), rn))
            if cost < best_cost:      # NaN / inf costs of an empty side never win (f32::MIN - f32::MAX overflows to -inf, times 0 = NaN)
                best_cost = cost
                best = (axis, F(cmin + F(step * F(i + 1))), l_lo.copy(), l_hi.copy(), r_lo.copy(), r_hi.copy())
    return best


def replay(verts, indices, max_nodes=None):
    v = np.ascontiguousarray(verts, F).reshape(-1, 4)[:, :3]
    nodes, idx = api.build_bvh(np.ascontiguousarray(verts, F), indices)
    tri = v[idx.reshape(-1, 3)]
    with np.errstate(over="ignore"):
        cent = ((tri[:, 0] + tri[:, 1]).astype(F) + tri[:, 2]).astype(F) / F(3.0)        # compute_tri_centroid (base.rs:1156-1159)
    blo, bhi = tri.min(axis=1), tri.max(axis=1)
    # triangle range of every node = union of its leaves (children are consecutive: first_child, first_child + 1)
    rng = {}

    def span(n):
        nd = nodes[n]
        if nd["tri_count"] > 0:
            rng[n] = (int(nd["tri_begin_or_first_child"]), int(nd["tri_begin_or_first_child"] + nd["tri_count"]))
        else:
            c = int(nd["tri_begin_or_first_child"])
            a, b = span(c), span(c + 1)
            assert a[1] == b[0], "children's triangle ranges are adjacent, left first"
            rng[n] = (a[0], b[1])
        return rng[n]
    import sys
    sys.setrecursionlimit(100000)
    span(0)
    depth = {0: 1}
    checked = leaves_checked = 0
    order = [0]
    while order:
        n = order.pop()
        nd = nodes[n]
        b, e = rng[n]
        if max_nodes is not None and checked + leaves_checked >= max_nodes:
            break
        got = choose_split(nd["aabb_min"].astype(F), nd["aabb_max"].astype(F), cent[b:e], blo[b:e], bhi[b:e], e - b)
        if nd["tri_count"] > 0:
            # a leaf: no plane won, or the winning plane's partition left one side empty, or the depth cap stopped the recursion
            if got is not None and depth[n] < 25:      # (a node at the cap is still SPLIT once by its parent's loop; its children are not)
                axis, pos = got[0], got[1]
                left = cent[b:e, axis] <= pos
                assert left.all() or (~left).all(), f"leaf {n} had a usable split"
            leaves_checked += 1
            continue
        assert got is not None, f"node {n} was split although no plane is cheaper than the node"
        axis, pos, l_lo, l_hi, r_lo, r_hi = got
        c = int(nd["tri_begin_or_first_child"])
        L, R = nodes[c], nodes[c + 1]
        assert np.array_equal(L["aabb_min"].view(np.uint32), l_lo.view(np.uint32)) and np.array_equal(L["aabb_max"].view(np.uint32), l_hi.view(np.uint32)), n
        assert np.array_equal(R["aabb_min"].view(np.uint32), r_lo.view(np.uint32)) and np.array_equal(R["aabb_max"].view(np.uint32), r_hi.view(np.uint32)), n
        lb, le = rng[c]
        assert (cent[lb:le, axis] <= pos).all() and not (cent[rng[c + 1][0]:rng[c + 1][1], axis] <= pos).any(), n
        depth[c] = depth[c + 1] = depth[n] + 1
        order += [c, c + 1]
        checked += 1
    return checked, leaves_checked, len(nodes)


def test_synthetic_soups_replay(built):
    rng = np.random.default_rng(4)
    for ntris in (2, 7, 64, 900):
        pos = np.zeros((ntris * 3, 4), F)
        centres = rng.random((ntris, 1, 3), dtype=F) * F(8)
        pos[:, :3] = (centres + rng.normal(size=(ntris, 3, 3)).astype(F) * F(0.15)).reshape(-1, 3)
        checked, leaves, total = replay(pos, np.arange(ntris * 3, dtype=np.uint32))
        assert checked + leaves == total


@pytest.mark.parametrize("mesh_rank", [0, 1])
def test_fixture_meshes_replay(built, mesh_rank):
    """The two largest meshes of a reference scene (shapes such as the bunny): every internal node up to 3 000 nodes, top down."""
    scene_cpu, _, _, _ = loader.load_scene_cpu_yoctogl_v24(os.path.join(util.SCENES, "materials1", "materials1.json"), [util.SHARED])
    meshes = sorted(zip(scene_cpu.verts_pos_array, scene_cpu.indices_array), key=lambda p: -len(p[1]))
    v, idx = meshes[mesh_rank]
    checked, leaves, total = replay(v, idx, max_nodes=3000)
    print(f"{len(idx) // 3} triangles, {total} nodes: {checked} internal nodes and {leaves} leaves replayed")
    assert checked > 500
