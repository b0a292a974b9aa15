"""Every BASELINE.json configuration at its own size through the HIP path (`-m gpu`):

  config 2  cornellbox 1024 x 1024, 8 bounces                  (tests/test_gpu_parity.py::test_full_size_properties_cornell_1024)
  config 3  materials1 1920 x 1080, 12 bounces
  config 4  environments1 1920 x 1080, 16 bounces, world 4     (stand-in for classroom + HDRI)
  config 5  bistro_class 3840 x 2160, 16 bounces, world 8      (the full 2.88 M-triangle stand-in for bistroexterior)

Per configuration, one 8-spp frame: determinism, oracle parity on tiles of the full-size frame (the reference's own
TileParams rule, so the oracle only renders those tiles), path-bounce and traversal work counters equal to the oracle's
on those tiles, tile-set union == full dispatch for the configuration's world size.  Plus the accuracy modes (f32
accumulation) and the exact closest-hit probe on the large scene.  Integer outputs are compared for equality, images bit for bit."""
import numpy as np
import pytest

from lupinpathtracer_amd import api
from tests import util

pytestmark = pytest.mark.gpu

CONFIGS = [
    # name, scene, camera, W, H, bounces, world sizes, tile_size of the sharding, tiles compared with the oracle (tile_size 8 = 32 px)
    ("config3", "materials1", 0, 1920, 1080, 12, (2,), 8, (0.30, 0.55, 0.80)),
    ("config4", "environments1", 0, 1920, 1080, 16, (4,), 8, (0.25, 0.50, 0.75)),
    ("config5", "bistro_class", 0, 3840, 2160, 16, (4, 8), 8, (0.35, 0.52, 0.71, 0.93)),
]


def pick_tiles(ts, w, h, fractions):
    ntx = (w - 1) // (ts * 4) + 1
    nty = (h - 1) // (ts * 4) + 1
    tiles = []
    for k, f in enumerate(fractions):
        ty = min(nty - 2, int(f * nty))
        tx = (ntx * (2 * k + 1)) // (2 * len(fractions))
        tiles.append(ty * ntx + tx)
    return tiles


@pytest.mark.parametrize("label,name,cam_i,W,H,bounces,worlds,shard_ts,fractions", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_baseline_config_at_full_size(gpu_ctx, label, name, cam_i, W, H, bounces, worlds, shard_ts, fractions):
    from oracle import oracle
    scene, cams = util.load_scene(name, gpu_ctx)
    cam = cams[cam_i]
    params = api.CameraParams(**{**cam.params.__dict__, "aspect": W / H})
    spp = 8
    res = api.build_pathtrace_resources(gpu_ctx, api.BakedPathtraceParams(max_bounces=bounces, samples_per_pixel=spp))
    desc = api.PathtraceDesc(camera_params=params, camera_transform=cam.transform)
    a, b = api.Texture(gpu_ctx, W, H), api.Texture(gpu_ctx, W, H)
    gpu_ctx.stats_reset(0)
    api.pathtrace_scene(gpu_ctx, res, scene, a, 0, desc)
    full_stats = gpu_ctx.stats()
    api.pathtrace_scene(gpu_ctx, res, scene, b, 0, desc)
    full = a.download()
    assert util.f16_words_differ(full, b.download()) == 0                      # determinism
    assert full_stats["paths"] == W * H * spp and full_stats["path_bounces"] >= full_stats["paths"]
    assert np.isfinite(full.astype(np.float32)).all() and float(full[..., :3].astype(np.float32).mean()) > 0.01

    # tile-set union == full dispatch for the configuration's world sizes, and the ranks' path-bounces add up
    for world in worlds:
        shards = api.Texture(gpu_ctx, W, H)
        gpu_ctx.stats_reset(0)
        per_rank = []
        for r in range(world):
            before = gpu_ctx.stats()["path_bounces"]
            api.pathtrace_scene_tiles(gpu_ctx, res, scene, shards, 0, desc, shard_ts, r, world)
            per_rank.append(gpu_ctx.stats()["path_bounces"] - before)
        assert util.f16_words_differ(full, shards.download()) == 0, f"{label}: world {world}"
        assert sum(per_rank) == full_stats["path_bounces"]
        assert max(per_rank) < 1.25 * (sum(per_rank) / world), f"{label}: round-robin tiles unbalanced: {per_rank}"

    # oracle parity on tiles of the full-size frame, with equal work counters
    ts = 8
    tex = api.Texture(gpu_ctx, W, H)
    ref = np.zeros((H, W, 4), np.float16)
    for t in pick_tiles(ts, W, H, fractions):
        tp = api.TileParams(tile_size=ts, tile_idx=t)
        # the wide tracer + re-trace first: counters of its own layout, image compared below
        gpu_ctx.set_traversal("wide")
        gpu_ctx.stats_reset(2)
        api.pathtrace_scene(gpu_ctx, res, scene, tex, 0, api.PathtraceDesc(camera_params=params, camera_transform=cam.transform, tile_params=tp))
        wst = gpu_ctx.stats()
        wide_img = tex.download()
        assert wst["wide_traversal"] == 1 and wst["wide_node_visits"][0] > 0 and wst["wide_queries"] >= wst["path_bounces"]
        # then the reference's order alone, whose work must equal the oracle's count for count
        gpu_ctx.set_traversal("binary")
        gpu_ctx.stats_reset(2)
        api.pathtrace_scene(gpu_ctx, res, scene, tex, 0, api.PathtraceDesc(camera_params=params, camera_transform=cam.transform, tile_params=tp))
        st = gpu_ctx.stats()
        gpu_ctx.stats_reset(0)
        assert st["wide_traversal"] == 0 and util.f16_words_differ(wide_img, tex.download()) == 0
        # a wide-node visit replaces up to three of the binary tree's (deep BLASes: about two; a sky tile that only touches the
        # top of the TLAS: hardly any), and the re-traced share is small
        print(f"{label} tile {t}: node fetches per path-bounce {st['node_visits'][0] / st['path_bounces']:.1f} binary -> "
              f"{(wst['wide_node_visits'][0] + wst['node_visits'][0]) / st['path_bounces']:.1f} wide + re-trace")
        assert wst["wide_node_visits"][0] + wst["node_visits"][0] < st["node_visits"][0], f"{label}: tile {t}"
        assert wst["wide_retraced"] < 0.06 * wst["wide_queries"]
        _, cnt = oracle.pathtrace(scene, W, H, params, cam.transform, bounces, spp, 0, tile_params=tp, out=ref)
        (ox, oy), gx, gy = oracle.dispatch_extent(W, H, tp)
        assert gx == gy == ts
        got = tex.download()[oy:oy + 32, ox:ox + 32]
        assert util.f16_words_differ(got, ref[oy:oy + 32, ox:ox + 32]) == 0, f"{label}: tile {t}"
        assert util.f16_words_differ(got, full[oy:oy + 32, ox:ox + 32]) == 0     # tiled == full-screen dispatch
        assert st["paths"] == cnt["paths"] == 32 * 32 * spp
        assert st["path_bounces"] == cnt["path_bounces"]
        # traversal work of the closest-hit queries: one wide-node visit = the reference's two child-box tests
        assert 2 * st["node_visits"][0] == cnt["tlas_aabb"][0] + cnt["blas_aabb"][0], f"{label}: tile {t}"
        assert st["tri_tests"][0] == cnt["tri_tests"][0]
        assert st["instance_entries"][0] == cnt["instances_entered"][0]


def test_closest_hit_kernel_exact_on_the_large_scene(gpu_ctx):
    """ray_scene_intersection alone on the 2.88 M-triangle scene (deep BLASes, 501 instances): exact hit tables."""
    from oracle import oracle
    scene, cams = util.load_scene("bistro_class", gpu_ctx)
    rng = np.random.default_rng(11)
    n = 300000
    ori = (rng.random((n, 3), dtype=np.float32) * 2 - 1) * np.array([30, 4, 30], np.float32) + np.array([0, 5, 0], np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    d[:1000, 1] = -np.abs(d[:1000, 1])
    g = api.trace_rays(gpu_ctx, scene, ori, d)
    o = oracle.trace_rays(scene, ori, d)
    assert np.array_equal(g[0], o[0])
    hit = o[0] == 1
    assert hit.sum() > n // 10
    assert np.array_equal(g[3][hit], o[3][hit]) and np.array_equal(g[4][hit], o[4][hit])
    assert np.array_equal(g[1][hit].view(np.uint32), o[1][hit].view(np.uint32))
    assert np.array_equal(g[2][hit].view(np.uint32), o[2][hit].view(np.uint32))


def test_f32_accumulation_mode(gpu_ctx):
    """lupin_hip_set_accumulation_mode: the f32 mode runs pathtracer.wgsl:275-289's recurrence on unquantised values (bit-equal
    to the oracle's f32-prev replay), its f16 view is the rounded value, the f16 mode is untouched by the switch, and the
    shadow follows copy_front_to_back / is dropped by an upload."""
    from oracle import oracle
    scene, cams = util.load_scene("cornellbox_builtin", gpu_ctx)
    cam = cams[0]
    W, H, frames, spp = 80, 64, 7, 4
    ref16 = util.oracle_accumulate(scene, cam, W, H, frames=frames, spp=spp)
    ref32 = np.zeros((H, W, 3), np.float32)
    scratch = np.zeros((H, W, 4), np.float16)
    for k in range(frames):
        oracle.pathtrace(scene, W, H, cam.params, cam.transform, 8, spp, 0, accum_counter=k, prev_frame_f32=ref32.copy(), out=scratch, want_f32=ref32)
    res = api.build_pathtrace_resources(gpu_ctx, api.BakedPathtraceParams(max_bounces=8, samples_per_pixel=spp))
    try:
        gpu_ctx.set_accumulation_mode(1)
        out = api.DoubleBufferedTexture(gpu_ctx, W, H)
        for k in range(frames):
            api.pathtrace_scene(gpu_ctx, res, scene, out.front(), 0,
                                api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), k), camera_params=cam.params, camera_transform=cam.transform))
            out.flip()
        out.flip()
        got32 = out.front().download_f32()
        assert np.array_equal(got32[..., :3].view(np.uint32), ref32.view(np.uint32))
        assert np.all(got32[..., 3] == 1.0)
        assert util.f16_words_differ(out.front().download(), scratch) == 0          # the f16 view = the oracle's rounded f32 result
        assert util.f16_words_differ(out.front().download(), ref16) > 0              # and it is not the f16 running average
        out.copy_front_to_back()
        assert np.array_equal(out.back().download_f32(), got32)
        out.front().upload(np.zeros((H, W, 4), np.float16))                           # the f16 texels are the truth now
        with pytest.raises(api.LupinError):
            out.front().download_f32()
    finally:
        gpu_ctx.set_accumulation_mode(0)
    assert util.f16_words_differ(util.gpu_accumulate(gpu_ctx, scene, cam, W, H, frames=frames, spp=spp), ref16) == 0
    with pytest.raises(api.LupinError):
        gpu_ctx.set_accumulation_mode(7)


def test_work_counting_does_not_change_the_image(gpu_ctx):
    scene, cams = util.load_scene("materials4", gpu_ctx)
    cam = cams[1]
    W, H = 160, 64
    res = api.build_pathtrace_resources(gpu_ctx, api.BakedPathtraceParams(max_bounces=8, samples_per_pixel=4))
    desc = api.PathtraceDesc(camera_params=cam.params, camera_transform=cam.transform)
    a, b = api.Texture(gpu_ctx, W, H), api.Texture(gpu_ctx, W, H)
    for ptype in (0, 1, 3):
        gpu_ctx.stats_reset(0)
        api.pathtrace_scene(gpu_ctx, res, scene, a, ptype, desc)
        plain = gpu_ctx.stats()
        gpu_ctx.stats_reset(2)
        api.pathtrace_scene(gpu_ctx, res, scene, b, ptype, desc)
        counted = gpu_ctx.stats()
        gpu_ctx.stats_reset(0)
        assert util.f16_words_differ(a.download(), b.download()) == 0
        assert plain["path_bounces"] == counted["path_bounces"] and plain["node_visits"] == [0, 0, 0]
        assert counted["node_visits"][0] + counted["wide_node_visits"][0] > 0 and counted["tri_tests"][0] > 0 and counted["instance_entries"][0] > 0


def test_measured_copy_bandwidth_and_runtime(gpu_ctx):
    gbps = gpu_ctx.measure_copy_bandwidth(1 << 30, 4)
    assert 1000.0 < gbps < 8000.0 * 1.05, gbps          # between a sanity floor and the nominal HBM3E peak
    info = api.runtime_info()
    assert info["num_hip_runtimes_mapped"] == 1
    assert info["build_hip_version"] // 100000 == info["runtime_hip_version"] // 100000


@pytest.mark.gpu
def test_objects_that_outlive_their_context_fail_cleanly(built):
    """Hosts that tear down in the wrong order (or a garbage collector) hand stale handles to the C side: it answers with an error,
    frees what it can and never dereferences the dead context."""
    import ctypes as C
    from lupinpathtracer_amd import _abi
    lib = _abi.lib()
    ctx = C.c_void_p()
    assert lib.lupin_hip_create_context(0, C.byref(ctx)) == 0
    tex, dbuf = C.c_void_p(), C.c_void_p()
    assert lib.lupin_hip_texture_create(ctx, 64, 32, C.byref(tex)) == 0
    assert lib.lupin_hip_dbuf_create(ctx, 64, 32, C.byref(dbuf)) == 0
    lib.lupin_hip_destroy_context(ctx)
    lib.lupin_hip_destroy_context(ctx)                                   # twice: the second call is a no-op
    buf = np.zeros((32, 64, 4), np.uint16)
    assert lib.lupin_hip_texture_download_rgba16f(tex, buf.ctypes.data_as(C.c_void_p)) != 0
    assert b"destroyed" in lib.lupin_hip_last_error()
    assert lib.lupin_hip_texture_upload_rgba16f(tex, buf.ctypes.data_as(C.c_void_p)) != 0
    assert lib.lupin_hip_dbuf_copy_front_to_back(dbuf) != 0
    assert lib.lupin_hip_sync(ctx) != 0
    other = C.c_void_p()
    assert lib.lupin_hip_texture_create(ctx, 8, 8, C.byref(other)) != 0
    lib.lupin_hip_texture_destroy(tex)                                   # frees the device memory, no sync on the dead context
    lib.lupin_hip_dbuf_destroy(dbuf)


@pytest.mark.gpu
def test_reserving_path_state_changes_nothing_but_the_allocation_time(built):
    """lupin_hip_reserve_path_state allocates every lane's path state up front; frames rendered afterwards equal those of a context
    that allocates lazily."""
    from lupinpathtracer_amd import api as lp
    a, b = lp.Context(0), lp.Context(0)
    b.reserve_path_state(96 * 64, 8, 2)
    with pytest.raises(lp.LupinError):
        b.reserve_path_state(0, 8, 2)
    outs = []
    for ctx in (a, b):
        scene, cams = util.load_scene("arealights1", ctx)
        outs.append(util.gpu_accumulate(ctx, scene, cams[0], 96, 64, frames=10, spp=2, max_bounces=8))   # more frames than lanes
    assert util.f16_words_differ(outs[0], outs[1]) == 0
    a.close(); b.close()
