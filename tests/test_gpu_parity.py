"""Parity of the HIP path (through the C ABI) with the CPU oracle on identical inputs, plus
size-independent properties at the BASELINE sizes and statistical comparison with the reference's goldens.

Tolerance (BASELINE.json north_star): per-pixel RMSE < 1e-3 on identical RNG seeds.  The arithmetic contract
 is stricter -- same IEEE operations in the same order on both sides -- so these tests also require the
number of differing f16 words to stay below 0.1 %, and bit-exactness for integer outputs (hits, counters)."""
import numpy as np
import pytest

from lupinpathtracer_amd import api, loader
from tests import util

pytestmark = pytest.mark.gpu

RMSE_TOL = 1e-3          # north_star tolerance
WORD_FRACTION_TOL = 1e-3  # differing f16 words / all words


def compare(got, ref, what):
    g, r = got.astype(np.float32), ref.astype(np.float32)
    rmse = float(np.sqrt(((g[..., :3] - r[..., :3]) ** 2).mean()))
    nbad = util.f16_words_differ(got, ref)
    assert rmse < RMSE_TOL, f"{what}: rmse {rmse}"
    assert nbad <= WORD_FRACTION_TOL * got.size, f"{what}: {nbad} of {got.size} f16 words differ"
    return rmse, nbad


@pytest.mark.parametrize("ptype", [0, 1, 2, 3])
def test_cornell_all_integrators_bit_exact(gpu_ctx, ptype):
    scene, cams = util.load_scene("cornellbox_builtin", gpu_ctx)
    cam = cams[0]
    W = H = 96
    got = util.gpu_accumulate(gpu_ctx, scene, cam, W, H, frames=2, spp=8, ptype=ptype)
    ref = util.oracle_accumulate(scene, cam, W, H, frames=2, spp=8, ptype=ptype)
    assert util.f16_words_differ(got, ref) == 0


SCENE_CASES = [("cornellbox", 0, 0), ("materials1", 0, 0), ("materials2", 1, 0), ("materials3", 1, 0), ("materials4", 2, 0),
               ("materials5", 1, 0), ("environments1", 1, 0), ("environments2", 2, 0), ("features1", 1, 0), ("shapes1", 1, 0),
               ("instances1", 1, 0), ("arealights1", 2, 0), ("furnace1", 0, 0), ("furnace2", 0, 0),
               ("materials4", 1, 1), ("materials4", 1, 2), ("materials4", 1, 3), ("environments1", 2, 1), ("features1", 2, 3),
               ("materials2", 2, 1), ("bistro_class_small", 0, 0), ("bistro_class_small", 0, 1)]


@pytest.mark.parametrize("name,cam_i,ptype", SCENE_CASES)
def test_scene_parity_with_oracle(gpu_ctx, name, cam_i, ptype):
    """Every test scene of the reference (textures, HDR environments + alias tables, normal maps, opacity,
    refractive / volumetric / glossy / reflective materials, instancing), two accumulation frames."""
    scene, cams = util.load_scene(name, gpu_ctx)
    cam = cams[cam_i]
    W = 192
    H = max(4, int(W / cam.params.aspect)) // 4 * 4
    got = util.gpu_accumulate(gpu_ctx, scene, cam, W, H, frames=2, spp=4, ptype=ptype)
    ref = util.oracle_accumulate(scene, cam, W, H, frames=2, spp=4, ptype=ptype)
    compare(got, ref, f"{name} cam{cam_i} type{ptype}")


def test_orthographic_and_dof_cameras(gpu_ctx):
    scene, cams = util.load_scene("materials1", gpu_ctx)
    picked = [c for c in cams if c.params.is_orthographic or c.params.aperture > 0]
    assert picked, "fixture scene should carry orthographic / depth-of-field cameras"
    for cam in picked[:3]:
        got = util.gpu_accumulate(gpu_ctx, scene, cam, 96, 40, frames=1, spp=4)
        ref = util.oracle_accumulate(scene, cam, 96, 40, frames=1, spp=4)
        compare(got, ref, "camera variants")


@pytest.mark.parametrize("name", ["cornellbox_builtin", "materials1", "instances1"])
def test_closest_hit_kernel_exact(gpu_ctx, name):
    """ray_scene_intersection alone (bvh_custom.wgsl:7-110): hit / instance / triangle identical, dst and uv bit-equal."""
    from oracle import oracle
    scene, cams = util.load_scene(name, gpu_ctx)
    rng = np.random.default_rng(5)
    n = 200000
    inst = scene.desc.num_instances
    assert inst > 0
    ori = (rng.random((n, 3), dtype=np.float32) * 2 - 1) * np.float32(3.0) + np.array([0, 1, 0], np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    d[:100] = np.array([0, -1, 0], np.float32)      # axis-aligned rays: zero direction components -> inf inv_dir
    d[100:200] = np.array([1, 0, 0], np.float32)
    g = api.trace_rays(gpu_ctx, scene, ori, d)
    o = oracle.trace_rays(scene, ori, d)
    assert np.array_equal(g[0], o[0])
    hit = o[0] == 1
    assert hit.sum() > n // 10
    assert np.array_equal(g[3][hit], o[3][hit]) and np.array_equal(g[4][hit], o[4][hit])
    assert np.array_equal(g[1][hit].view(np.uint32), o[1][hit].view(np.uint32))
    assert np.array_equal(g[2][hit].view(np.uint32), o[2][hit].view(np.uint32))


def test_tiled_dispatch_matches_oracle_tiles(gpu_ctx):
    """TileParams sub-dispatch incl. the floor-division edge behaviour (renderer.rs:807-829)."""
    from oracle import oracle
    scene, cams = util.load_scene("cornellbox_builtin", gpu_ctx)
    cam = cams[0]
    W, H, ts = 70, 53, 4
    res = api.build_pathtrace_resources(gpu_ctx, api.BakedPathtraceParams(max_bounces=4, samples_per_pixel=2))
    tex = api.Texture(gpu_ctx, W, H)
    ref = np.zeros((H, W, 4), np.float16)
    nt = api.get_num_tiles(ts, W, H)
    for t in range(nt):
        desc = api.PathtraceDesc(tile_params=api.TileParams(ts, t), camera_params=cam.params, camera_transform=cam.transform)
        api.pathtrace_scene(gpu_ctx, res, scene, tex, 0, desc)
        oracle.pathtrace(scene, W, H, cam.params, cam.transform, 4, 2, tile_params=api.TileParams(ts, t), out=ref)
    got = tex.download()
    assert util.f16_words_differ(got, ref) == 0
    assert np.all(got[52:, :, 3] == 0) and np.all(got[:, 68:, 3] == 0)   # remainder pixels never written
    with pytest.raises(api.LupinError) as e:
        api.pathtrace_scene(gpu_ctx, res, scene, tex, 0, api.PathtraceDesc(tile_params=api.TileParams(ts, nt)))
    assert e.value.code == -5   # "tile_idx out of range!" (renderer.rs:814)


def test_overlapped_frames_keep_accumulation_order(gpu_ctx):
    """Consecutive calls run on alternating lanes (streams) and only meet at the resolve: a long accumulation, a
    mid-sequence texture upload and copy_front_to_back on the primary stream must give the serial result bit for bit."""
    scene, cams = util.load_scene("cornellbox_builtin", gpu_ctx)
    cam = cams[0]
    W, H, frames = 96, 64, 9
    want = util.oracle_accumulate(scene, cam, W, H, frames=frames, spp=2, max_bounces=6)
    got = util.gpu_accumulate(gpu_ctx, scene, cam, W, H, frames=frames, spp=2, max_bounces=6)
    assert util.f16_words_differ(got, want) == 0
    # download in the middle, re-upload into the back buffer, continue: same pixels
    res = api.build_pathtrace_resources(gpu_ctx, api.BakedPathtraceParams(max_bounces=6, samples_per_pixel=2))
    out = api.DoubleBufferedTexture(gpu_ctx, W, H)
    for k in range(frames):
        desc = api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), k), camera_params=cam.params, camera_transform=cam.transform)
        api.pathtrace_scene(gpu_ctx, res, scene, out.front(), 0, desc)
        if k == 3:
            mid = out.front().download()
            out.front().upload(mid)
        if k == 5:
            out.copy_front_to_back()   # back == front: flipping now changes nothing
        out.flip()
    out.flip()
    assert util.f16_words_differ(out.front().download(), want) == 0


@pytest.mark.parametrize("name,ptype", [("cornellbox_builtin", 0), ("materials4", 1), ("bistro_class_small", 0), ("bistro_class_small", 3)])
def test_frames_per_wavefront_equal_one_call_per_wavefront(gpu_ctx, name, ptype):
    """lupin_hip_set_batch_frames: consecutive chained calls run as ONE wavefront.  With a camera that moves every frame, an
    accumulation that restarts in the middle (accum_counter back to 0), a tile-set call and a broken texture chain thrown in,
    eight (three, sixteen, the library's own choice) frames per wavefront must store exactly what one call per wavefront stores --
    and that is the oracle's image."""
    scene, cams = util.load_scene(name, gpu_ctx)
    cam = cams[1 % len(cams)] if name == "materials4" else cams[0]
    W, H, spp, bounces, frames = 88, 56, 2, 5, 11
    res = api.build_pathtrace_resources(gpu_ctx, api.BakedPathtraceParams(max_bounces=bounces, samples_per_pixel=spp))

    def cam_for(k):
        t = np.array(cam.transform, np.float32).copy()
        t[3, 0] += np.float32(0.01 * k)           # the camera drifts: every frame of a batch has its own
        return t, api.CameraParams(**{**cam.params.__dict__, "aspect": W / H, "lens": cam.params.lens * (1.0 + 0.01 * (k % 3))})

    def counter(k):
        return k if k < 6 else k - 6              # a restart at frame 6: the blend chain begins again inside a batch

    def render(batch):
        gpu_ctx.set_batch_frames(batch)
        out = api.DoubleBufferedTexture(gpu_ctx, W, H)
        stray = api.Texture(gpu_ctx, W, H)
        snaps = []
        for k in range(frames):
            t, cp = cam_for(k)
            desc = api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), counter(k)), camera_params=cp, camera_transform=t)
            if k == 4:
                api.pathtrace_scene_tiles(gpu_ctx, res, scene, out.front(), ptype, desc, 2, 0, 1)   # another dispatch shape: its own wavefront
            else:
                api.pathtrace_scene(gpu_ctx, res, scene, out.front(), ptype, desc)
            if k == 8:   # a call outside the chain (reads `stray`, which no call of the batch wrote)
                api.pathtrace_scene(gpu_ctx, res, scene, stray, ptype,
                                    api.PathtraceDesc(accum_params=api.AccumulationParams(out.front(), 3), camera_params=cp, camera_transform=t))
            if k in (2, 9):
                snaps.append(out.front().download())   # a read in the middle of a batch
            out.flip()
        out.flip()
        snaps += [out.front().download(), out.back().download(), stray.download()]
        return snaps

    try:
        one = render(1)
        eight = render(8)
        three = render(3)
        sixteen = render(16)
        auto = render(0)          # the library's choice by dispatch size (sixteen for a frame this small)
    finally:
        gpu_ctx.set_batch_frames(0)
    for a, b, c, d, e in zip(one, eight, three, sixteen, auto):
        assert util.f16_words_differ(a, b) == 0 and util.f16_words_differ(a, c) == 0 and util.f16_words_differ(a, d) == 0 and util.f16_words_differ(a, e) == 0
    # and the serial result is the oracle's (frames 6.. restart the accumulation, so the last image depends on frames 6..10 only)
    from oracle import oracle
    prev = np.zeros((H, W, 4), np.float16)
    for k in range(6, frames):
        t, cp = cam_for(k)
        prev, _ = oracle.pathtrace(scene, W, H, cp, t, bounces, spp, ptype, accum_counter=counter(k), prev_frame=prev)
    assert util.f16_words_differ(one[-3], prev) == 0
    with pytest.raises(api.LupinError):
        gpu_ctx.set_batch_frames(17)


def test_error_behaviour(gpu_ctx):
    scene, cams = util.load_scene("cornellbox_builtin", gpu_ctx)
    res = api.build_pathtrace_resources(gpu_ctx, api.BakedPathtraceParams())
    tex = api.Texture(gpu_ctx, 16, 16)
    with pytest.raises(api.LupinError) as e:   # render_target must differ from prev_frame (renderer.rs:754-755)
        api.pathtrace_scene(gpu_ctx, res, scene, tex, 0, api.PathtraceDesc(accum_params=api.AccumulationParams(tex, 1)))
    assert e.value.code == -6
    with pytest.raises(api.LupinError):
        api.pathtrace_scene(gpu_ctx, res, scene, tex, 7, api.PathtraceDesc())
    with pytest.raises(api.LupinError):
        api.build_pathtrace_resources(gpu_ctx, api.BakedPathtraceParams(samples_per_pixel=0))


def test_malformed_hierarchies_are_rejected(gpu_ctx):
    """A node array with a cycle would hang the reference's traversal; scene creation refuses it."""
    from lupinpathtracer_amd import loader
    scene_cpu = loader.cornell_box_scene_cpu()[0]

    def cyclic(v, i):
        nodes, idx = api.build_bvh(v, i)
        if len(nodes) >= 3:
            nodes = nodes.copy()
            inner = [k for k in range(len(nodes)) if nodes[k]["tri_count"] == 0]
            nodes[inner[-1]]["tri_begin_or_first_child"] = 0        # the deepest internal node points back at the root
        return nodes, idx
    with pytest.raises(api.LupinError) as e:
        api.build_accel_structures_and_upload(gpu_ctx, scene_cpu, [], [], blas_builder=cyclic)
    assert "not a tree" in str(e.value) or e.value.code == -1


def test_empty_scene_and_rne_mode(gpu_ctx):
    from oracle import oracle
    empty = loader.build_scene_empty(gpu_ctx)
    res = api.build_pathtrace_resources(gpu_ctx, api.BakedPathtraceParams(max_bounces=3, samples_per_pixel=2))
    tex = api.Texture(gpu_ctx, 12, 8)
    api.pathtrace_scene(gpu_ctx, res, empty, tex, 0, api.PathtraceDesc())
    img = tex.download()
    assert np.all(img[..., :3] == 0) and np.all(img[..., 3] == 1)
    scene, cams = util.load_scene("cornellbox_builtin", gpu_ctx)
    cam = cams[0]
    gpu_ctx.set_f16_store_rounding(1)
    try:
        tex = api.Texture(gpu_ctx, 32, 32)
        api.pathtrace_scene(gpu_ctx, res, scene, tex, 0, api.PathtraceDesc(camera_params=cam.params, camera_transform=cam.transform))
        ref, _ = oracle.pathtrace(scene, 32, 32, cam.params, cam.transform, 3, 2, store_rounding=1)
        assert util.f16_words_differ(tex.download(), ref) == 0
    finally:
        gpu_ctx.set_f16_store_rounding(0)


def test_work_counters_equal_oracle(gpu_ctx):
    """The unit of the headline metric (path-bounces) is counted identically on both sides."""
    from oracle import oracle
    scene, cams = util.load_scene("materials1", gpu_ctx)
    cam = cams[0]
    W, H = 160, 64
    res = api.build_pathtrace_resources(gpu_ctx, api.BakedPathtraceParams(max_bounces=8, samples_per_pixel=4))
    tex = api.Texture(gpu_ctx, W, H)
    for ptype in (0, 1):
        gpu_ctx.stats_reset(False)
        api.pathtrace_scene(gpu_ctx, res, scene, tex, ptype, api.PathtraceDesc(camera_params=cam.params, camera_transform=cam.transform))
        st = gpu_ctx.stats()
        _, cnt = oracle.pathtrace(scene, W, H, cam.params, cam.transform, 8, 4, ptype)
        assert st["paths"] == cnt["paths"] == W * H * 4
        assert st["path_bounces"] == cnt["path_bounces"]


def test_full_size_properties_cornell_1024(gpu_ctx):
    """BASELINE config 2 size (1024 x 1024, 8 bounces): determinism, tile-set union == full-screen dispatch
    (what the multi-GPU sharding relies on), pack/unpack round trip, and oracle parity on sampled rows."""
    from oracle import oracle
    scene, cams = util.load_scene("cornellbox_builtin", gpu_ctx)
    cam = cams[0]
    W = H = 1024
    res = api.build_pathtrace_resources(gpu_ctx, api.BakedPathtraceParams(max_bounces=8, samples_per_pixel=8))
    desc = api.PathtraceDesc(camera_params=cam.params, camera_transform=cam.transform)
    a, b = api.Texture(gpu_ctx, W, H), api.Texture(gpu_ctx, W, H)
    api.pathtrace_scene(gpu_ctx, res, scene, a, 0, desc)
    api.pathtrace_scene(gpu_ctx, res, scene, b, 0, desc)
    full = a.download()
    assert util.f16_words_differ(full, b.download()) == 0
    # tile-set sharding over 3 "ranks", then pack -> unpack into a fresh target
    world, ts = 3, 16
    shards = api.Texture(gpu_ctx, W, H)
    for r in range(world):
        api.pathtrace_scene_tiles(gpu_ctx, res, scene, shards, 0, desc, ts, r, world)
    assert util.f16_words_differ(full, shards.download()) == 0
    import ctypes as C
    from lupinpathtracer_amd import _abi
    gathered = api.Texture(gpu_ctx, W, H)
    for r in range(world):
        npx = api.packed_tile_pixels(W, H, ts, r, world)
        buf = api.Texture(gpu_ctx, npx, 1)          # any device buffer of npx * 8 bytes
        assert api.pack_tiles(gpu_ctx, shards, ts, r, world, buf.device_ptr()) == npx
        api.unpack_tiles(gpu_ctx, gathered, ts, r, world, buf.device_ptr())
        gpu_ctx.sync()
    assert util.f16_words_differ(full, gathered.download()) == 0
    # the same with 16 tiles per row and 4 ranks: a multiple, where ownership rotates row by row (include/lupin_tiles.h)
    world = 4
    shards4, gathered4 = api.Texture(gpu_ctx, W, H), api.Texture(gpu_ctx, W, H)
    for r in range(world):
        api.pathtrace_scene_tiles(gpu_ctx, res, scene, shards4, 0, desc, ts, r, world)
    assert util.f16_words_differ(full, shards4.download()) == 0
    from lupinpathtracer_amd import distributed
    host = shards4.download()
    for r in range(world):
        npx = api.packed_tile_pixels(W, H, ts, r, world)
        buf = api.Texture(gpu_ctx, npx, 1)
        assert api.pack_tiles(gpu_ctx, shards4, ts, r, world, buf.device_ptr()) == npx
        assert np.array_equal(buf.download().reshape(-1, 4).view(np.uint16), distributed.pack_tiles_numpy(host, ts, r, world).view(np.uint16))
        api.unpack_tiles(gpu_ctx, gathered4, ts, r, world, buf.device_ptr())
        gpu_ctx.sync()
    assert util.f16_words_differ(full, gathered4.download()) == 0
    # oracle parity on a band of rows via the reference's own tiling (tile_size 256 groups wide, 2 groups tall)
    band = np.zeros((H, W, 4), np.float16)
    oracle.pathtrace(scene, W, H, cam.params, cam.transform, 8, 8, 0, tile_params=api.TileParams(4, 7 * 64 + 20), out=band)
    ys, xs = np.nonzero(band[..., 3])
    assert len(ys) == 256
    assert util.f16_words_differ(full[ys, xs], band[ys, xs]) == 0


GOLDEN_GPU = [("furnace1", 0), ("materials1", 1), ("materials1", 2), ("materials2", 1), ("materials3", 1), ("materials4", 1),
              ("materials4", 2), ("materials5", 2), ("environments1", 1), ("environments2", 2), ("features1", 1),
              ("shapes1", 2), ("instances1", 1), ("arealights1", 1), ("arealights1", 2)]


@pytest.mark.parametrize("name,cam_i", GOLDEN_GPU)
def test_gpu_vs_reference_golden_renders(gpu_ctx, name, cam_i):
    """lupin_tests' protocol (10 spp x 101 frames, 8 bounces, Standard, max_radiance 10) at half the golden
    resolution, compared with the reference's golden render: mean within 2 %, 8x8-block averages within 5 %
    relative RMSE (goldens: RGBE-quantised, author's device BVH -- statistical pin)."""
    scene, cams = util.load_scene(name, gpu_ctx)
    cam = cams[cam_i]
    small, full_shape, full_mean = util.golden_render(name, cam_i)      # golden box-filtered 4x4
    H, W = small.shape[0] * 2, small.shape[1] * 2                      # render at 1/2 size
    adv = api.AdvancedParams(max_radiance=10.0)
    img = util.gpu_accumulate(gpu_ctx, scene, cam, W, H, frames=101, spp=10, advanced=adv).astype(np.float32)[..., :3]
    mine = img.reshape(H // 2, 2, W // 2, 2, 3).mean(axis=(1, 3))
    g = small
    assert abs(mine.mean() - g.mean()) / g.mean() < 0.02, (mine.mean(), g.mean())
    bh, bw = (g.shape[0] // 8) * 8, (g.shape[1] // 8) * 8
    a = mine[:bh, :bw].reshape(bh // 8, 8, bw // 8, 8, 3).mean(axis=(1, 3))
    b = g[:bh, :bw].reshape(bh // 8, 8, bw // 8, 8, 3).mean(axis=(1, 3))
    rel_rmse = float(np.sqrt(((a - b) ** 2).mean()) / g.mean())
    print(f"golden {name} cam{cam_i}: mean ratio {mine.mean() / g.mean():.4f}, block rel-rmse {rel_rmse:.4f}")
    assert rel_rmse < 0.05, rel_rmse


@pytest.mark.parametrize("name,cam_i", [("cornellbox_builtin", 0), ("features1", 1), ("materials4", 2)])
def test_falsecolor_entry_point(gpu_ctx, name, cam_i):
    """lp::pathtrace_scene_falsecolor (renderer.rs:872-948, pathtracer.wgsl:296-452): all 12 FalsecolorType views,
    two accumulation frames, against the oracle; plus one tiled call."""
    from oracle import oracle
    scene, cams = util.load_scene(name, gpu_ctx)
    cam = cams[cam_i]
    W = 128
    H = max(4, int(W / cam.params.aspect)) // 4 * 4
    res = api.build_pathtrace_resources(gpu_ctx, api.BakedPathtraceParams(max_bounces=8, samples_per_pixel=3))
    for ftype in api.FalsecolorType:
        out = api.DoubleBufferedTexture(gpu_ctx, W, H)
        prev = np.zeros((H, W, 4), np.float16)
        for k in range(2):
            desc = api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), k), camera_params=cam.params, camera_transform=cam.transform)
            api.pathtrace_scene_falsecolor(gpu_ctx, res, scene, out.front(), ftype, desc)
            ref, _ = oracle.pathtrace(scene, W, H, cam.params, cam.transform, 8, 3, accum_counter=k, prev_frame=prev, falsecolor_type=ftype)
            prev = ref
            out.flip()
        out.flip()
        got = out.front().download()
        compare(got, ref, f"falsecolor {ftype.name} {name}")
        if ftype in (api.FalsecolorType.Normals,):
            nrm = got[..., :3].astype(np.float32)
            hit = np.abs(nrm).sum(axis=2) > 0
            assert hit.any()
    # tiled
    tex = api.Texture(gpu_ctx, W, H)
    ref = np.zeros((H, W, 4), np.float16)
    for t in range(api.get_num_tiles(4, W, H)):
        api.pathtrace_scene_falsecolor(gpu_ctx, res, scene, tex, api.FalsecolorType.Instance,
                                       api.PathtraceDesc(tile_params=api.TileParams(4, t), camera_params=cam.params, camera_transform=cam.transform))
        oracle.pathtrace(scene, W, H, cam.params, cam.transform, 8, 3, tile_params=api.TileParams(4, t), out=ref,
                         falsecolor_type=api.FalsecolorType.Instance)
    assert util.f16_words_differ(tex.download(), ref) == 0
    with pytest.raises(api.LupinError):
        api.pathtrace_scene_falsecolor(gpu_ctx, res, scene, tex, 12, api.PathtraceDesc())


@pytest.mark.parametrize("name,cam_i", [("cornellbox_builtin", 0), ("arealights1", 1), ("materials4", 2)])
def test_debug_heatmap_entry_point(gpu_ctx, name, cam_i):
    """lp::pathtrace_scene_debug (renderer.rs:966-1041, pathtracer.wgsl:457-503, :2806-2872): box-test, triangle-test
    and bounce-count heat maps, first-hit-only and whole-path, two accumulation frames, against the oracle.  The
    colour is a function of an integer count, so equality also pins the per-pixel counts (incl. light-pdf marching,
    where the product path culls lights and the debug view must not)."""
    from oracle import oracle
    scene, cams = util.load_scene(name, gpu_ctx)
    cam = cams[cam_i]
    W = 96
    H = max(4, int(W / cam.params.aspect)) // 4 * 4
    res = api.build_pathtrace_resources(gpu_ctx, api.BakedPathtraceParams(max_bounces=6, samples_per_pixel=3))
    cases = [(api.DebugVizType.BVHAABBChecks, True, 0.0, 60.0), (api.DebugVizType.BVHTriChecks, True, 0.0, 20.0),
             (api.DebugVizType.BVHAABBChecks, False, 0.0, 400.0), (api.DebugVizType.BVHTriChecks, False, 5.0, 150.0),
             (api.DebugVizType.NumBounces, False, 0.0, 7.0), (api.DebugVizType.NumBounces, True, 0.0, 3.0)]
    for viz, first, lo, hi in cases:
        dd = api.DebugVizDesc(viz, lo, hi, first)
        out = api.DoubleBufferedTexture(gpu_ctx, W, H)
        prev = np.zeros((H, W, 4), np.float16)
        for k in range(2):
            desc = api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), k), camera_params=cam.params, camera_transform=cam.transform)
            api.pathtrace_scene_debug(gpu_ctx, res, scene, out.front(), dd, desc)
            ref, _ = oracle.pathtrace(scene, W, H, cam.params, cam.transform, 6, 3, accum_counter=k, prev_frame=prev, debug_desc=dd)
            prev = ref
            out.flip()
        out.flip()
        got = out.front().download()
        assert util.f16_words_differ(got, ref) == 0, f"debug {viz.name} first_hit_only={first} {name}"
        assert len(np.unique(got[..., :3].astype(np.float32))) > 3   # an actual heat map, not one colour
    with pytest.raises(api.LupinError):
        api.pathtrace_scene_debug(gpu_ctx, res, scene, api.Texture(gpu_ctx, W, H), api.DebugVizDesc(7, 0.0, 1.0, False),
                                  api.PathtraceDesc(camera_params=cam.params, camera_transform=cam.transform))


def test_tonemap_and_fit_aspect(gpu_ctx):
    """lp::tonemap_and_fit_aspect (tonemapping.rs:155-224): k_tonemap against the oracle's restatement, byte for byte,
    on a rendered Cornell frame: default desc, letterboxed targets, viewport, exposure + filmic, clear=False."""
    from oracle import oracle
    scene, cams = util.load_scene("cornellbox_builtin", gpu_ctx)
    img = util.gpu_accumulate(gpu_ctx, scene, cams[0], 96, 64, frames=2, spp=4)
    tex = api.Texture(gpu_ctx, 96, 64)
    tex.upload(img)
    prev = np.random.default_rng(3).integers(0, 255, (120, 100, 4), dtype=np.uint8)
    cases = [(96, 64, api.TonemapDesc(), None), (200, 90, api.TonemapDesc(), None), (50, 120, api.TonemapDesc(exposure=1.5, filmic=True), None),
             (100, 120, api.TonemapDesc(viewport=api.Viewport(7, 11, 80, 33), exposure=-0.75, clear=False), prev),
             (100, 120, api.TonemapDesc(viewport=api.Viewport(40.5, 60.25, 300, 300), srgb=False), None)]
    for w, h, desc, dst in cases:
        got = api.tonemap_and_fit_aspect(gpu_ctx, tex, w, h, desc, dst)
        want = oracle.tonemap(img, w, h, desc, dst)
        assert np.array_equal(got, want), (w, h, desc)
    assert got[..., :3].max() > 0
