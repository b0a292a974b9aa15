"""Light-pdf stage (LUPIN_LIGHT_STAGE=1): sample_lights_pdf (pathtracer.wgsl:2516-2549 -> bvh_custom.wgsl:112-152) runs in its own
stage -- k_light_pdf between k_shade and the next extend for the Standard integrator (weight update, roulette, epilogue),
k_light_pdf_mis between k_shade and the shadow rays for MIS (the two power-heuristic weights).  Same draws, same arithmetic:
the image equals the inline build's word for word and the oracle's."""
import os

import numpy as np
import pytest

from lupinpathtracer_amd import api
from tests import util


@pytest.fixture(scope="module")
def staged_ctx(built):
    old = os.environ.get("LUPIN_LIGHT_STAGE")
    os.environ["LUPIN_LIGHT_STAGE"] = "1"     # read at context creation
    try:
        ctx = api.Context(0)
    finally:
        if old is None:
            os.environ.pop("LUPIN_LIGHT_STAGE", None)
        else:
            os.environ["LUPIN_LIGHT_STAGE"] = old
    yield ctx
    ctx.close()


# area lights; textured + environment; volumes (the medium branch parks phase_eval / phase_pdf); many emissive instances
@pytest.mark.gpu
@pytest.mark.parametrize("name,cam,bounces", [("arealights1", 0, 8), ("environments1", 0, 8), ("materials4", 0, 12), ("materials2", 0, 8),
                                              ("features1", 0, 8), ("bistro_class_small", 0, 16), ("cornellbox_builtin", 0, 8)])
@pytest.mark.parametrize("ptype", [0, 1], ids=["standard", "mis"])
def test_stage_equals_inline_and_oracle(gpu_ctx, staged_ctx, name, cam, bounces, ptype):
    inline_scene, cams = util.load_scene(name, gpu_ctx)
    staged_scene, _ = util.load_scene(name, staged_ctx)
    W, H = 200, 120
    a = util.gpu_accumulate(gpu_ctx, inline_scene, cams[cam], W, H, frames=2, spp=3, max_bounces=bounces, ptype=ptype)
    b = util.gpu_accumulate(staged_ctx, staged_scene, cams[cam], W, H, frames=2, spp=3, max_bounces=bounces, ptype=ptype)
    assert util.f16_words_differ(a, b) == 0
    ref = util.oracle_accumulate(staged_scene, cams[cam], W, H, frames=2, spp=3, max_bounces=bounces, ptype=ptype)
    assert util.f16_words_differ(b, ref) == 0


@pytest.mark.gpu
def test_stage_tiles_and_counters(gpu_ctx, staged_ctx):
    """Tile dispatches (edge tiles, offsets) and the path-bounce counters go through the stage unchanged."""
    a_scene, cams = util.load_scene("arealights1", gpu_ctx)
    b_scene, _ = util.load_scene("arealights1", staged_ctx)
    W, H = 150, 90
    outs = []
    for ctx, scene in ((gpu_ctx, a_scene), (staged_ctx, b_scene)):
        res = api.build_pathtrace_resources(ctx, api.BakedPathtraceParams(max_bounces=8, samples_per_pixel=4))
        tex = api.Texture(ctx, W, H)
        ctx.stats_reset()
        ts = 4                                           # 16 x 16-pixel tiles: 10 x 6 of them, the last column / row partial
        for t in range(((W - 1) // (4 * ts) + 1) * ((H - 1) // (4 * ts) + 1)):
            api.pathtrace_scene(ctx, res, scene, tex, api.PathtraceType.Standard,
                                api.PathtraceDesc(tile_params=api.TileParams(ts, t), camera_params=cams[0].params, camera_transform=cams[0].transform))
        ctx.sync()
        st = ctx.stats()
        outs.append((tex.download(), st["path_bounces"], st["paths"]))
    assert util.f16_words_differ(outs[0][0], outs[1][0]) == 0
    assert outs[0][1] == outs[1][1] and outs[0][2] == outs[1][2]
