"""The reference's BLAS builder on the device (csrc/sahbvh.hip): the tree it builds is lp::build_bvh's tree (data_structures.rs:196-475,
CPU restatement lupin_build_bvh) -- same boxes, same split into children, same triangle set in every leaf -- up to node numbering
and the order of triangles inside a leaf; images and closest hits are the same through either builder."""
import os
import time

import numpy as np
import pytest

from lupinpathtracer_amd import api, loader
from tests import util
from tests.test_lbvh import random_mesh


def assert_same_tree(cpu, dev, verts):
    """Walk both trees together, level by level (vectorised): every pair of corresponding nodes has equal boxes, both are leaves or
    both are inner nodes; corresponding leaves cover the same index range with the same multiset of triangles."""
    (na, ia), (nb, ib) = cpu, dev
    assert len(na) == len(nb)
    a_ids, b_ids = np.array([0]), np.array([0])
    leaf_begin, leaf_count = [], []
    visited = 0
    depth = 0
    while len(a_ids):
        a, b = na[a_ids], nb[b_ids]
        visited += len(a_ids)
        assert np.array_equal(a["aabb_min"], b["aabb_min"]) and np.array_equal(a["aabb_max"], b["aabb_max"]), f"boxes differ at depth {depth}"
        assert np.array_equal(a["tri_count"], b["tri_count"]), f"leaf / inner or leaf size differs at depth {depth}"
        leaf = a["tri_count"] > 0
        assert np.array_equal(a["tri_begin_or_first_child"][leaf], b["tri_begin_or_first_child"][leaf])
        leaf_begin.append(a["tri_begin_or_first_child"][leaf]); leaf_count.append(a["tri_count"][leaf])
        fa, fb = a["tri_begin_or_first_child"][~leaf].astype(np.int64), b["tri_begin_or_first_child"][~leaf].astype(np.int64)
        a_ids, b_ids = np.concatenate([fa, fa + 1]), np.concatenate([fb, fb + 1])    # children adjacent (bvh_custom.wgsl:240-241)
        depth += 1
    assert visited == len(na) and depth <= 25
    begin, count = np.concatenate(leaf_begin).astype(np.int64), np.concatenate(leaf_count).astype(np.int64)
    n_tris = len(ia) // 3
    order = np.argsort(begin)
    begin, count = begin[order], count[order]
    assert begin[0] == 0 and np.array_equal(begin[1:], (begin + count)[:-1]) and begin[-1] + count[-1] == n_tris   # leaves tile the index buffer
    leaf_of = np.repeat(np.arange(len(begin)), count)
    ta, tb = ia.reshape(-1, 3).astype(np.int64), ib.reshape(-1, 3).astype(np.int64)
    sa = np.lexsort((ta[:, 2], ta[:, 1], ta[:, 0], leaf_of))
    sb = np.lexsort((tb[:, 2], tb[:, 1], tb[:, 0], leaf_of))
    assert np.array_equal(ta[sa], tb[sb]), "a leaf holds different triangles"
    return depth


def caterpillar_mesh(n):
    """Triangles whose centroids double in distance: every SAH split peels one bin off, so the tree runs into the depth cap."""
    verts = np.zeros((3 * n, 4), np.float32)
    x = (2.0 ** np.arange(n, dtype=np.float64) * 1e-3).astype(np.float32)
    verts[0::3, 0] = x; verts[1::3, 0] = x * 1.01; verts[2::3, 0] = x
    verts[1::3, 1] = x * 0.01; verts[2::3, 2] = x * 0.01
    return verts, np.arange(3 * n, dtype=np.uint32)


@pytest.mark.gpu
@pytest.mark.parametrize("n,clustered", [(1, False), (2, False), (3, False), (5, False), (63, False), (64, False), (65, False), (1000, False), (4099, False),
                                         (100_003, False), (300, True), (5000, True)])
def test_same_tree_as_the_reference_builder(gpu_ctx, n, clustered):
    verts, idx = random_mesh(n, 700 + n, clustered)
    assert_same_tree(api.build_bvh(verts, idx), api.build_bvh_sah_device(gpu_ctx, verts, idx), verts)


@pytest.mark.gpu
def test_degenerate_inputs(gpu_ctx):
    # all centroids equal on every axis -> no split, one leaf; and equal on two axes
    verts, idx = random_mesh(40, 3)
    verts[:, :3] = np.tile(verts[:3, :3], (40, 1))
    assert_same_tree(api.build_bvh(verts, idx), api.build_bvh_sah_device(gpu_ctx, verts, idx), verts)
    verts, idx = random_mesh(500, 4)
    verts[:, 1:3] = np.tile(verts[:3, 1:3], (500, 1))
    assert_same_tree(api.build_bvh(verts, idx), api.build_bvh_sah_device(gpu_ctx, verts, idx), verts)
    # shared vertices, duplicated triangles
    verts, idx = random_mesh(256, 5)
    idx = np.concatenate([idx, idx[:300]])
    assert_same_tree(api.build_bvh(verts, idx), api.build_bvh_sah_device(gpu_ctx, verts, idx), verts)
    # depth cap of the reference's 25-entry stack
    verts, idx = caterpillar_mesh(60)
    d = assert_same_tree(api.build_bvh(verts, idx), api.build_bvh_sah_device(gpu_ctx, verts, idx), verts)
    assert d == 25
    bad = idx.copy(); bad[7] = 10_000
    with pytest.raises(api.LupinError):
        api.build_bvh_sah_device(gpu_ctx, verts, bad)
    nan = verts.copy(); nan[5, 1] = np.nan       # ordered-integer min / max and fminf / fmaxf agree on finite values only: refused
    with pytest.raises(api.LupinError):
        api.build_bvh_sah_device(gpu_ctx, nan, idx)


@pytest.mark.gpu
def test_deterministic(gpu_ctx):
    verts, idx = random_mesh(30_000, 11)
    a, b = api.build_bvh_sah_device(gpu_ctx, verts, idx), api.build_bvh_sah_device(gpu_ctx, verts, idx)
    assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1])


@pytest.mark.gpu
def test_bistro_class_meshes_same_tree_and_build_time(gpu_ctx):
    """Every mesh of the bistro-class stand-in (2.88 M triangles in 20 meshes): same tree; the whole device build stays under 0.5 s."""
    scene_cpu, _, _, _ = loader.build_scene_bistro_class_cpu(util.SHARED)
    meshes = [(np.ascontiguousarray(v, np.float32).reshape(-1, 4), np.ascontiguousarray(i, np.uint32)) for v, i in zip(scene_cpu.verts_pos_array, scene_cpu.indices_array)]
    api.build_bvh_sah_device(gpu_ctx, *meshes[0])   # warm-up: code object load
    t0 = time.perf_counter()
    dev = [api.build_bvh_sah_device(gpu_ctx, v, i) for v, i in meshes]
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    cpu = [api.build_bvh(v, i) for v, i in meshes]
    t_cpu = time.perf_counter() - t0
    print(f"bistro_class BLAS build: device {t_dev * 1e3:.1f} ms, CPU (1 thread) {t_cpu * 1e3:.1f} ms, {sum(len(i) for _, i in meshes) // 3} triangles")
    for (v, _), c, d in zip(meshes, cpu, dev):
        assert_same_tree(c, d, v)
    assert t_dev < 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["materials1", "bistro_class"])
def test_images_and_hits_equal_through_either_builder(gpu_ctx, name):
    if name == "materials1":
        path = os.path.join(util.SCENES, "materials1", "materials1.json")
        dev, cams = loader.load_scene_yoctogl_v24(path, gpu_ctx, asset_dirs=[util.SHARED], blas_builder="sah_device")
        cam = cams[1]
    else:
        dev, cams = loader.build_scene_bistro_class(gpu_ctx, util.SHARED, blas_builder="sah_device")
        cam = cams[0]
    sah, _ = util.load_scene(name, gpu_ctx)
    W, H = 480, 270
    a = util.gpu_accumulate(gpu_ctx, sah, cam, W, H, frames=2, spp=2)
    b = util.gpu_accumulate(gpu_ctx, dev, cam, W, H, frames=2, spp=2)
    assert util.f16_words_differ(a, b) == 0
    rng = np.random.default_rng(9)
    n = 200_000
    ori = np.tile(np.asarray(cam.transform, np.float32).reshape(4, 3)[3], (n, 1)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ha, hb = api.trace_rays(gpu_ctx, sah, ori, d), api.trace_rays(gpu_ctx, dev, ori, d)
    assert np.array_equal(ha[0], hb[0])
    hit = ha[0].astype(bool)
    assert hit.sum() > 1000 and np.array_equal(ha[1][hit].view(np.uint32), hb[1][hit].view(np.uint32))
