"""Randomised parity: small synthetic scenes that mix what the reference's test scenes do not (every material type incl.
subsurface / gltfpbr, emission / roughness / scattering / normal textures, vertex colours, partial opacity, several
environments with and without textures, scaled and sheared instances, orthographic and thin-lens cameras), rendered by
the HIP path and by the oracle on the same RNG streams.  SURVEY 8d: "oracle-only coverage via synthetic materials"."""
import numpy as np
import pytest

from lupinpathtracer_amd import api, loader
from lupinpathtracer_amd._abi import ENVIRONMENT_DTYPE, INSTANCE_DTYPE, MATERIAL_DTYPE, MESH_INFO_DTYPE
from tests import util

pytestmark = pytest.mark.gpu


def box_mesh():
    v = np.array([(x, y, z) for x in (-0.5, 0.5) for y in (-0.5, 0.5) for z in (-0.5, 0.5)], np.float32)
    f = [(0, 1, 3), (0, 3, 2), (4, 6, 7), (4, 7, 5), (0, 4, 5), (0, 5, 1), (2, 3, 7), (2, 7, 6), (0, 2, 6), (0, 6, 4), (1, 5, 7), (1, 7, 3)]
    return v, np.array(f, np.uint32).reshape(-1)


def grid_mesh(n, rng):
    """(n+1)^2 vertices, bumpy height field with uvs, normals, colours."""
    u, w = np.meshgrid(np.linspace(0, 1, n + 1), np.linspace(0, 1, n + 1), indexing="ij")
    h = 0.08 * np.sin(7 * u + rng.uniform(0, 6)) * np.cos(5 * w + rng.uniform(0, 6))
    v = np.stack([u - 0.5, h, w - 0.5], -1).reshape(-1, 3).astype(np.float32)
    idx = []
    for i in range(n):
        for j in range(n):
            a, b, c, d = i * (n + 1) + j, (i + 1) * (n + 1) + j, (i + 1) * (n + 1) + j + 1, i * (n + 1) + j + 1
            idx += [a, c, b, a, d, c]
    uv = np.stack([u * 3, w * 2], -1).reshape(-1, 2).astype(np.float32)
    nrm = np.zeros_like(v); nrm[:, 1] = 1.0
    nrm[:, 0] = -0.3 * np.cos(7 * v[:, 0]); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    col = rng.uniform(0.4, 1.0, (len(v), 4)).astype(np.float32)
    return v, np.array(idx, np.uint32), uv, nrm.astype(np.float32), col


def pad4(a, w=0.0):
    out = np.full((len(a), 4), w, np.float32)
    out[:, :a.shape[1]] = a
    return out


def random_scene(seed):
    rng = np.random.default_rng(seed)
    s = api.SceneCPU()
    infos = []
    bv, bi = box_mesh()
    s.verts_pos_array.append(pad4(bv)); s.indices_array.append(bi); infos.append(api.default_mesh_info())            # 0: box
    gv, gi, guv, gn, gc = grid_mesh(6, rng)
    s.verts_pos_array.append(pad4(gv)); s.indices_array.append(gi)
    s.verts_texcoord_array.append(guv); s.verts_normal_array.append(pad4(gn)); s.verts_color_array.append(gc)
    info = api.default_mesh_info(); info["texcoords_buf_idx"] = 0; info["normals_buf_idx"] = 0; info["colors_buf_idx"] = 0
    infos.append(info)                                                                                              # 1: grid with every attribute
    gv2, gi2, guv2, _, _ = grid_mesh(3, rng)
    s.verts_pos_array.append(pad4(gv2)); s.indices_array.append(gi2); s.verts_texcoord_array.append(guv2)
    info = api.default_mesh_info(); info["texcoords_buf_idx"] = 1
    infos.append(info)                                                                                              # 2: grid with uvs only
    s.mesh_infos = np.array(infos, MESH_INFO_DTYPE)

    # textures: 0 RGBA8 colour with alpha holes, 1 RGBA8 roughness / metallic, 2 RGBA16F emission, 3 RGBA8 normal map, 4 RGBA16F sky
    t0 = rng.integers(0, 256, (16, 16, 4), dtype=np.uint8); t0[..., 3] = np.where(rng.random((16, 16)) < 0.3, 90, 255)
    t1 = rng.integers(0, 256, (8, 8, 4), dtype=np.uint8)
    t2 = np.ones((8, 4, 4), np.float16); t2[..., :3] = rng.uniform(0, 2, (8, 4, 3)).astype(np.float16)
    t3 = np.zeros((16, 16, 4), np.uint8); t3[..., :2] = rng.integers(96, 160, (16, 16, 2)); t3[..., 2] = 255; t3[..., 3] = 255
    sky32 = np.ones((8, 16, 4), np.float32); sky32[..., :3] = rng.uniform(0.05, 1.5, (8, 16, 3)); sky32[2, 5, :3] = 40.0
    textures = [api.TextureCPU(t0), api.TextureCPU(t1), api.TextureCPU(t2), api.TextureCPU(t3), api.TextureCPU(sky32.astype(np.float16))]

    mats = []
    for t in range(8):
        for variant in range(2):
            m = api.default_material()
            m["mat_type"] = t
            m["color"] = (*rng.uniform(0.2, 0.95, 3), 1.0 if variant == 0 else rng.uniform(0.3, 1.0))
            m["roughness"] = [0.0, 0.25, 0.2, 0.0, 0.0, 0.15, 0.0, 0.35][t] if variant == 0 else rng.uniform(0.0, 0.6)
            m["metallic"] = rng.uniform(0, 1)
            m["ior"] = rng.uniform(1.1, 2.0)
            m["scattering"][:3] = rng.uniform(0.05, 0.9, 3)
            m["sc_anisotropy"] = rng.uniform(-0.6, 0.6)
            m["tr_depth"] = rng.uniform(0.02, 0.5)
            if variant == 1:
                m["color_tex_idx"] = 0 if t % 2 == 0 else api.SENTINEL_IDX
                m["roughness_tex_idx"] = 1 if t in (1, 2, 7) else api.SENTINEL_IDX
                m["scattering_tex_idx"] = 1 if t in (5, 6) else api.SENTINEL_IDX
                m["normal_tex_idx"] = 3 if t in (0, 1, 2, 7) else api.SENTINEL_IDX
            mats.append(m)
    em = api.default_material(); em["emission"][:3] = (14, 11, 6); mats.append(em)                       # plain emitter
    em2 = api.default_material(); em2["emission"][:3] = (5, 6, 9); em2["emission_tex_idx"] = 2; mats.append(em2)   # textured emitter
    s.materials = np.array(mats, MATERIAL_DTYPE)

    def frame(scale3, yaw, pitch, pos, shear=0.0):
        cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
        r = np.array([[cy, 0, -sy], [sy * sp, cp, cy * sp], [sy * cp, -sp, cy * cp]], np.float64)   # rows = columns of the frame
        f = np.zeros((4, 3), np.float32)
        f[0] = r[0] * scale3[0]; f[1] = r[1] * scale3[1] + shear * r[0]; f[2] = r[2] * scale3[2]; f[3] = pos
        return f

    insts = [api.instance_from_transform(frame((6, 6, 6), 0.0, 0.0, (0, -0.2, 2.5)), 2, 0)]              # floor (matte, uvs)
    for i in range(14):
        mesh = int(rng.integers(0, 3))
        mat = int(rng.integers(0, 16))
        sc = rng.uniform(0.3, 0.9, 3) * (1.0 if mesh == 0 else 1.6)
        pos = (rng.uniform(-1.6, 1.6), rng.uniform(0.0, 1.2), rng.uniform(1.2, 3.8))
        insts.append(api.instance_from_transform(frame(sc, rng.uniform(0, 6.28), rng.uniform(-0.5, 0.5), pos, shear=rng.uniform(-0.2, 0.2)), mesh, mat))
    insts.append(api.instance_from_transform(frame((0.5, 0.5, 0.5), 0.3, 0.0, (0.3, 2.2, 2.4)), 0, 16))   # box light
    insts.append(api.instance_from_transform(frame((0.8, 0.8, 0.8), 1.0, 0.4, (-1.0, 1.8, 3.0)), 2, 17))  # textured emitter with uvs
    s.instances = np.array(insts, INSTANCE_DTYPE)

    envs = []
    e0 = api.default_environment(); e0["emission"] = (0.5, 0.55, 0.7); e0["emission_tex_idx"] = 4
    rot = frame((1, 1, 1), 0.7, 0.2, (0, 0, 0)); e0["transform"][:3, :3] = rot[:3]
    envs.append(e0)
    if seed % 2 == 0:
        e1 = api.default_environment(); e1["emission"] = (0.05, 0.04, 0.03); envs.append(e1)   # constant environment, no texture
    s.environments = np.array(envs, ENVIRONMENT_DTYPE)
    envs_info = [api.EnvMapInfo(sky32, 16, 8)] + ([api.EnvMapInfo(np.ones((1, 1, 4), np.float32), 1, 1)] if seed % 2 == 0 else [])
    api.validate_scene(s, len(textures), len(textures))

    cam = loader.SceneCamera(transform=np.array([[1, 0, 0], [0, 0.96, 0.28], [0, -0.28, 0.96], [0, 1.3, -1.2]], np.float32),
                             params=api.CameraParams(lens=0.03, film=0.036, aspect=1.5, focus=3.5, aperture=0.0))
    cam_dof = loader.SceneCamera(transform=cam.transform, params=api.CameraParams(lens=0.05, film=0.036, aspect=1.5, focus=3.0, aperture=0.08))
    cam_ortho = loader.SceneCamera(transform=cam.transform, params=api.CameraParams(is_orthographic=True, lens=0.05, film=3.0, aspect=1.5, focus=3.0, aperture=0.0))
    return s, textures, envs_info, [cam, cam_dof, cam_ortho]


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_scene_parity(gpu_ctx, seed):
    scene_cpu, textures, envs_info, cams = random_scene(seed)
    scene = api.build_accel_structures_and_upload(gpu_ctx, scene_cpu, textures, envs_info)
    W, H = 120, 80
    for ptype in range(4):
        cam = cams[(seed + ptype) % 3]
        got = util.gpu_accumulate(gpu_ctx, scene, cam, W, H, frames=2, spp=3, max_bounces=7, ptype=ptype)
        ref = util.oracle_accumulate(scene, cam, W, H, frames=2, spp=3, max_bounces=7, ptype=ptype)
        g, r = got.astype(np.float32), ref.astype(np.float32)
        assert np.isfinite(g).all()
        rmse = float(np.sqrt(((g[..., :3] - r[..., :3]) ** 2).mean()))
        nbad = util.f16_words_differ(got, ref)
        assert rmse < 1e-3 and nbad <= 1e-3 * got.size, f"seed {seed} type {ptype}: rmse {rmse}, {nbad} words differ"
        assert g[..., :3].max() > 0.05   # the scene is lit
    # G-buffers and heat maps on the same scene
    from oracle import oracle
    res = api.build_pathtrace_resources(gpu_ctx, api.BakedPathtraceParams(max_bounces=7, samples_per_pixel=2))
    desc = api.PathtraceDesc(camera_params=cams[0].params, camera_transform=cams[0].transform)
    tex = api.Texture(gpu_ctx, W, H)
    for ft in (api.FalsecolorType.Albedo, api.FalsecolorType.Normals, api.FalsecolorType.Emission, api.FalsecolorType.Opacity, api.FalsecolorType.IsDelta):
        api.pathtrace_scene_falsecolor(gpu_ctx, res, scene, tex, ft, desc)
        ref, _ = oracle.pathtrace(scene, W, H, cams[0].params, cams[0].transform, 7, 2, falsecolor_type=ft)
        assert util.f16_words_differ(tex.download(), ref) <= 1e-3 * ref.size, ft
    dd = api.DebugVizDesc(api.DebugVizType.BVHTriChecks, 0.0, 300.0, False)
    api.pathtrace_scene_debug(gpu_ctx, res, scene, tex, dd, desc)
    ref, _ = oracle.pathtrace(scene, W, H, cams[0].params, cams[0].transform, 7, 2, debug_desc=dd)
    assert util.f16_words_differ(tex.download(), ref) <= 1e-3 * ref.size


def _random_api_mix(lanes, seed, n_ops):
    import os
    old = os.environ.get("LUPIN_LANES")
    os.environ["LUPIN_LANES"] = str(lanes)
    try:
        ctx = api.Context(0)
    finally:
        if old is None:
            os.environ.pop("LUPIN_LANES", None)
        else:
            os.environ["LUPIN_LANES"] = old
    try:
        rng = np.random.default_rng(seed)
        scene, cams = loader.build_scene_cornell_box(ctx)
        cam = cams[0]
        res = api.build_pathtrace_resources(ctx, api.BakedPathtraceParams(max_bounces=5, samples_per_pixel=2))
        W, H = 160, 96
        out = api.DoubleBufferedTexture(ctx, W, H)
        extra = api.Texture(ctx, W, H)
        k, digest = 0, []
        for _ in range(n_ops):
            op = int(rng.integers(0, 10))
            desc = api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), k), camera_params=cam.params, camera_transform=cam.transform)
            if op <= 4:
                api.pathtrace_scene(ctx, res, scene, out.front(), int(rng.integers(0, 4)), desc)
                out.flip()
                k += 1
            elif op == 5:
                t = int(rng.integers(0, api.get_num_tiles(6, W, H)))
                api.pathtrace_scene(ctx, res, scene, out.front(), 0,
                                    api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), k), tile_params=api.TileParams(6, t),
                                                      camera_params=cam.params, camera_transform=cam.transform))
            elif op == 6:
                out.copy_front_to_back()
            elif op == 7:
                digest.append(int(out.front().download().view(np.uint16).astype(np.uint64).sum()))
            elif op == 8:
                api.pathtrace_scene_falsecolor(ctx, res, scene, extra, int(rng.integers(0, 12)),
                                               api.PathtraceDesc(camera_params=cam.params, camera_transform=cam.transform))
                digest.append(int(extra.download().view(np.uint16).astype(np.uint64).sum()))
            else:
                out.back().upload(out.back().download())
        return digest, out.front().download().copy(), out.back().download().copy()
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", [1, 2])
def test_frames_in_flight_equal_a_serial_context(seed):
    """600 random API calls (all integrators, tiles, copies, uploads, downloads, G-buffers) on a context with three frames
    in flight and on a one-lane context: every checkpoint and both final textures are identical."""
    a = _random_api_mix(3, seed, 600)
    b = _random_api_mix(1, seed, 600)
    assert a[0] == b[0] and len(a[0]) > 50
    assert util.f16_words_differ(a[1], b[1]) == 0 and util.f16_words_differ(a[2], b[2]) == 0


def caterpillar_bvh(verts, indices):
    """A maximally unbalanced BLAS in the reference's node format: every internal node has one single-triangle leaf and
    one internal child (the last one two leaves), children adjacent, boxes exact.  n triangles -> depth n - 1."""
    from lupinpathtracer_amd._abi import BVH_NODE_DTYPE
    v = np.asarray(verts, np.float32).reshape(-1, 4)[:, :3]
    idx = np.asarray(indices, np.uint32)
    tri = v[idx.reshape(-1, 3)]
    n = len(tri)
    lo, hi = tri.min(axis=1), tri.max(axis=1)
    nodes = np.zeros(2 * n - 1, BVH_NODE_DTYPE)
    suffix_lo = np.minimum.accumulate(lo[::-1], axis=0)[::-1]   # bounds of triangles i..n-1
    suffix_hi = np.maximum.accumulate(hi[::-1], axis=0)[::-1]
    cur = 0                                                     # node holding triangles i..n-1
    for i in range(n - 1):
        nodes[cur]["aabb_min"], nodes[cur]["aabb_max"] = suffix_lo[i], suffix_hi[i]
        nodes[cur]["tri_begin_or_first_child"], nodes[cur]["tri_count"] = 2 * i + 1, 0
        leaf, rest = 2 * i + 1, 2 * i + 2
        nodes[leaf]["aabb_min"], nodes[leaf]["aabb_max"] = lo[i], hi[i]
        nodes[leaf]["tri_begin_or_first_child"], nodes[leaf]["tri_count"] = i, 1
        cur = rest
    nodes[cur]["aabb_min"], nodes[cur]["aabb_max"] = lo[n - 1], hi[n - 1]
    nodes[cur]["tri_begin_or_first_child"], nodes[cur]["tri_count"] = n - 1, 1
    return nodes, idx.copy()


def _deep_scene(n_instances):
    """25 triangles under a depth-24 BLAS (the reference's BVH_MAX_DEPTH - 1), instanced along a line (-> a deep
    agglomerative TLAS), a box emitter, a constant environment."""
    rng = np.random.default_rng(9)
    s = api.SceneCPU()
    centres = rng.uniform(-0.8, 0.8, (25, 1, 3))
    tri = (centres + rng.uniform(-0.35, 0.35, (25, 3, 3))).astype(np.float32)
    s.verts_pos_array.append(pad4(tri.reshape(-1, 3))); s.indices_array.append(np.arange(75, dtype=np.uint32))
    bv, bi = box_mesh()
    s.verts_pos_array.append(pad4(bv)); s.indices_array.append(bi)
    s.mesh_infos = np.array([api.default_mesh_info(), api.default_mesh_info()], MESH_INFO_DTYPE)
    m0 = api.default_material(); m0["color"] = (0.8, 0.7, 0.6, 1.0)
    m1 = api.default_material(); m1["mat_type"] = 2; m1["color"] = (0.9, 0.9, 0.9, 1.0); m1["roughness"] = 0.1
    em = api.default_material(); em["emission"][:3] = (9, 9, 8)
    s.materials = np.array([m0, m1, em], MATERIAL_DTYPE)
    insts = []
    for k in range(n_instances):
        f = np.zeros((4, 3), np.float32)
        sc = 0.12 * (1.0 + 0.5 * (k % 3))
        f[0] = (sc, 0, 0); f[1] = (0, sc, 0); f[2] = (0, 0, sc); f[3] = (-1.6 + 3.2 * k / n_instances, -0.4 + 0.9 * ((k * 7) % 11) / 11.0, 2.0 + 0.004 * k)
        insts.append(api.instance_from_transform(f, 0, k % 2))
    f = np.zeros((4, 3), np.float32); f[0] = (0.6, 0, 0); f[1] = (0, 0.6, 0); f[2] = (0, 0, 0.6); f[3] = (0.0, 1.6, 2.5)
    insts.append(api.instance_from_transform(f, 1, 2))
    s.instances = np.array(insts, INSTANCE_DTYPE)
    e = api.default_environment(); e["emission"] = (0.2, 0.25, 0.3)
    s.environments = np.array([e], ENVIRONMENT_DTYPE)
    api.validate_scene(s, 0, 0)
    cam = loader.SceneCamera(transform=np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 0.3, -1.5]], np.float32),
                             params=api.CameraParams(lens=0.03, film=0.036, aspect=1.0, focus=3.0, aperture=0.0))
    return s, [api.EnvMapInfo(np.ones((1, 1, 4), np.float32), 1, 1)], cam


def test_limits_deep_hierarchies_long_paths_odd_sizes(gpu_ctx):
    """Edges of the input space: a BLAS at the reference's depth cap and a 300-instance TLAS (the LDS traversal stack is
    sized from the real depths), images that are not multiples of the 4-pixel workgroup down to 1 x 1, many bounces and
    samples per call (656 iterations in one call), closest hits against the oracle."""
    from oracle import oracle
    scene_cpu, envs_info, cam = _deep_scene(n_instances=300)
    builder = lambda v, i: caterpillar_bvh(v, i) if len(i) == 75 else api.build_bvh(v, i)
    nodes, _ = caterpillar_bvh(scene_cpu.verts_pos_array[0], scene_cpu.indices_array[0])
    depth = {0: 0}
    for i, nd in enumerate(nodes):
        if nd["tri_count"] == 0:
            depth[int(nd["tri_begin_or_first_child"])] = depth[int(nd["tri_begin_or_first_child"]) + 1] = depth[i] + 1
    assert max(depth.values()) == 24        # BVH_MAX_DEPTH - 1 (renderer.rs:296): the deepest tree the reference's 26-entry stack walks
    scene = api.build_accel_structures_and_upload(gpu_ctx, scene_cpu, [], envs_info, blas_builder=builder)
    rng = np.random.default_rng(2)
    n = 6000
    ori = np.tile(np.array([0.0, 0.3, -1.5], np.float32), (n, 1))
    d = rng.normal(size=(n, 3)).astype(np.float32) * np.array([0.6, 0.35, 0.0], np.float32) + np.array([0.0, 0.0, 1.0], np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    g, o = api.trace_rays(gpu_ctx, scene, ori, d), oracle.trace_rays(scene, ori, d)
    assert np.array_equal(g[0], o[0]) and g[0].sum() > 500
    hit = g[0].astype(bool)
    for k in (1, 3, 4):
        assert np.array_equal(np.asarray(g[k])[hit].view(np.uint32), np.asarray(o[k])[hit].view(np.uint32))
    for (W, H) in ((1, 1), (3, 5), (5, 3), (33, 17)):
        got = util.gpu_accumulate(gpu_ctx, scene, cam, W, H, frames=2, spp=2, max_bounces=6)
        ref = util.oracle_accumulate(scene, cam, W, H, frames=2, spp=2, max_bounces=6)
        assert util.f16_words_differ(got, ref) == 0, (W, H)
    got = util.gpu_accumulate(gpu_ctx, scene, cam, 24, 16, frames=1, spp=16, max_bounces=40)   # 16 x 41 iterations in one call
    ref = util.oracle_accumulate(scene, cam, 24, 16, frames=1, spp=16, max_bounces=40)
    assert util.f16_words_differ(got, ref) == 0
    for ptype in (1, 3):
        got = util.gpu_accumulate(gpu_ctx, scene, cam, 20, 12, frames=1, spp=3, max_bounces=12, ptype=ptype)
        ref = util.oracle_accumulate(scene, cam, 20, 12, frames=1, spp=3, max_bounces=12, ptype=ptype)
        assert util.f16_words_differ(got, ref) == 0, ptype
