"""Device BLAS builder (csrc/lbvh.hip, "next" row 8f-1): structure checks of its numpy restatement on the CPU, the
kernels against that restatement byte for byte on the GPU, and builder-independence of the traced results."""
import os

import numpy as np
import pytest

from lupinpathtracer_amd import api, loader
from tests import lbvh_ref, util


def random_mesh(n_tris, seed, clustered=False):
    rng = np.random.default_rng(seed)
    centres = rng.random((n_tris, 1, 3), dtype=np.float32) * (0.0 if clustered else 10.0) - 3.0
    tri = centres + (rng.random((n_tris, 3, 3), dtype=np.float32) - 0.5) * 0.4
    verts = np.zeros((n_tris * 3, 4), np.float32)
    verts[:, :3] = tri.reshape(-1, 3)
    idx = rng.permutation(n_tris * 3).astype(np.uint32) if n_tris > 1 else np.arange(3, dtype=np.uint32)
    # shuffle which vertex a triangle uses but keep triangles well formed
    idx = np.arange(n_tris * 3, dtype=np.uint32).reshape(-1, 3)[rng.permutation(n_tris)].reshape(-1)
    return verts, idx


def check_tree(nodes, reordered, verts, n_tris):
    depth = lbvh_ref.lbvh_depth(n_tris)
    assert len(nodes) == (2 << depth) - 1 and depth <= 24          # inside the reference's 25-entry stack (renderer.rs:296)
    tri = verts[reordered.reshape(-1, 3)][:, :, :3]
    seen = np.zeros(n_tris, bool)

    def walk(i, d):
        nd = nodes[i]
        if nd["tri_count"] > 0:
            a, c = int(nd["tri_begin_or_first_child"]), int(nd["tri_count"])
            assert d == depth and 1 <= c <= 2 and not seen[a:a + c].any()
            seen[a:a + c] = True
            pts = tri[a:a + c].reshape(-1, 3)
            assert np.array_equal(nd["aabb_min"], pts.min(0)) and np.array_equal(nd["aabb_max"], pts.max(0))
            return
        l = int(nd["tri_begin_or_first_child"])
        for ch in (l, l + 1):                                       # children adjacent (bvh_custom.wgsl:240-241)
            assert np.all(nodes[ch]["aabb_min"] >= nd["aabb_min"]) and np.all(nodes[ch]["aabb_max"] <= nd["aabb_max"])
            walk(ch, d + 1)
    walk(0, 0)
    assert seen.all()


@pytest.mark.parametrize("n", [1, 2, 3, 5, 64, 777])
def test_restatement_builds_a_valid_reference_format_tree(n):
    verts, idx = random_mesh(n, n)
    nodes, reordered = lbvh_ref.build(verts, idx)
    assert sorted(map(tuple, reordered.reshape(-1, 3))) == sorted(map(tuple, idx.reshape(-1, 3)))   # a permutation of the triangles
    check_tree(nodes, reordered, verts, n)


def test_oracle_traces_the_same_hits_through_either_builder():
    """The reference's traversal (oracle) over the LBVH-format nodes finds the hits it finds over the SAH nodes:
    the node format is the reference's own, and results are builder-independent (up to exact ties)."""
    from oracle import oracle
    scene_cpu, textures, envs, cams = loader.load_scene_cpu_yoctogl_v24(os.path.join(util.SCENES, "shapes1", "shapes1.json"), [util.SHARED])
    sah = api.build_accel_structures_and_upload(None, scene_cpu, textures, envs)
    lbvh = api.build_accel_structures_and_upload(None, scene_cpu, textures, envs, blas_builder=lambda v, i: lbvh_ref.build(v, i) if 192 <= len(i) <= 3 * 6144 else api.build_bvh(v, i))   # 3 of the 8 meshes
    rng = np.random.default_rng(11)
    n = 4000
    ori = np.tile(np.asarray(cams[0].transform, np.float32).reshape(4, 3)[3], (n, 1)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a = oracle.trace_rays(sah, ori, d)
    b = oracle.trace_rays(lbvh, ori, d)
    assert np.array_equal(a[0], b[0])                                  # hit flags
    hit = a[0].astype(bool)
    assert hit.sum() > 100
    same_t = a[1][hit].view(np.uint32) == b[1][hit].view(np.uint32)    # identical triangle => identical t, bit for bit
    assert same_t.mean() > 0.999
    assert np.array_equal(a[3][hit][same_t], b[3][hit][same_t])        # instance


@pytest.mark.gpu
@pytest.mark.parametrize("n,clustered", [(1, False), (2, False), (3, False), (5, False), (64, False), (1000, False), (4099, False), (300, True)])
def test_device_builder_matches_restatement(gpu_ctx, n, clustered):
    verts, idx = random_mesh(n, 100 + n, clustered)
    want_nodes, want_idx = lbvh_ref.build(verts, idx)
    nodes, reordered = api.build_bvh_device(gpu_ctx, verts, idx)
    assert np.array_equal(reordered, want_idx)
    assert nodes.tobytes() == want_nodes.tobytes()


@pytest.mark.gpu
def test_device_builder_errors(gpu_ctx):
    verts, idx = random_mesh(8, 1)
    bad = idx.copy(); bad[5] = 10_000
    with pytest.raises(api.LupinError):
        api.build_bvh_device(gpu_ctx, verts, bad)


@pytest.mark.gpu
def test_lbvh_scene_renders_like_the_oracle_and_hits_like_sah(gpu_ctx):
    """materials1 (bunny meshes, 156 k triangles) with device-built BLASes: the HIP path equals the oracle on the same
    scene bit for bit, and closest hits equal those of the SAH-built scene."""
    path = os.path.join(util.SCENES, "materials1", "materials1.json")
    lbvh, cams = loader.load_scene_yoctogl_v24(path, gpu_ctx, asset_dirs=[util.SHARED], blas_builder="lbvh")
    sah, _ = util.load_scene("materials1", gpu_ctx)
    cam = cams[1]
    W, H = 160, 64
    got = util.gpu_accumulate(gpu_ctx, lbvh, cam, W, H, frames=2, spp=2)
    ref = util.oracle_accumulate(lbvh, cam, W, H, frames=2, spp=2)
    assert util.f16_words_differ(got, ref) <= 1e-3 * got.size
    assert float(np.sqrt(((got.astype(np.float32) - ref.astype(np.float32))[..., :3] ** 2).mean())) < 1e-3
    rng = np.random.default_rng(5)
    n = 20000
    ori = np.tile(np.asarray(cam.transform, np.float32).reshape(4, 3)[3], (n, 1)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a, b = api.trace_rays(gpu_ctx, sah, ori, d), api.trace_rays(gpu_ctx, lbvh, ori, d)
    assert np.array_equal(a[0], b[0])
    hit = a[0].astype(bool)
    same_t = a[1][hit].view(np.uint32) == b[1][hit].view(np.uint32)
    assert hit.sum() > 1000 and same_t.mean() > 0.999
