"""Pins for what the reference's golden renders do not cover ("parity unpinned" rows a11 / a12 / a13).

CPU (oracle) side, no GPU needed:
  * closed-form white-furnace answers (the reference's furnace1 scene with the sphere's material swapped) for every
    material family incl. subsurface and gltfpbr, under all four integrators;
  * the MIS integrator's environment double count, a reference quirk that these furnaces expose.

GPU side (`-m gpu`): the same furnaces bit-exact against the oracle, and MIS / Naive / Direct renders under the reference
test app's protocol landing on the reference's Standard-integrator goldens where the estimators agree by construction.
"""
import numpy as np
import pytest

from lupinpathtracer_amd import api
from tests import util

MT = api.MaterialType
E = 0.5   # furnace1: constant environment emission (test_scenes/furnace1/furnace1.json)
SC = np.array([0.8, 0.8, 0.8, 0.0], np.float32)
# (label, material type, fields, lower bound of the sphere's radiance / E, is the answer exactly E?)
FURNACES = [
    ("matte white", MT.Matte, dict(roughness=0.0), 1.0, True),                       # Lambert, albedo 1: radiance = E
    ("refractive smooth", MT.Refractive, dict(roughness=0.0), 1.0, True),            # delta lobes, no absorption (color 1 => density 0)
    ("transparent smooth", MT.Transparent, dict(roughness=0.0), 1.0, True),
    ("volumetric", MT.Volumetric, dict(roughness=0.0, scattering=SC), 1.0, True),    # density 0: the medium does nothing
    ("glossy", MT.Glossy, dict(roughness=0.3), 0.95, False),
    ("reflective smooth", MT.Reflective, dict(roughness=0.0), 0.95, False),          # conductor Fresnel of reflectivity clamped to 0.99
    ("refractive rough", MT.Refractive, dict(roughness=0.3), 0.90, False),           # single-scattering GGX loses energy, never gains
    ("subsurface rough", MT.Subsurface, dict(roughness=0.3, scattering=SC), 0.90, False),
    ("gltfpbr metal", MT.GltfPbr, dict(roughness=0.4, metallic=1.0), 0.90, False),
    ("gltfpbr dielectric", MT.GltfPbr, dict(roughness=0.4, metallic=0.0), 0.95, False),
    ("gltfpbr half metal", MT.GltfPbr, dict(roughness=0.2, metallic=0.5), 0.60, False),
]
# The Direct integrator's light ray is a plain closest-hit query (pathtracer.wgsl:1126): sampled THROUGH a rough transmissive
# surface it stops at the far side of the same object, so that energy is lost (reference behaviour, reproduced): for these
# materials Direct is only bounded from above.
DIRECT_LOSES_ENERGY = {"refractive rough", "subsurface rough"}
W, H = 240, 100
SPHERE = (slice(H // 2 - 15, H // 2 + 15), slice(W // 2 - 15, W // 2 + 15))   # pixels well inside the sphere's silhouette


def oracle_furnace(mat_type, fields, ptype, spp=32):
    from oracle import oracle
    scene, cams = util.furnace_variant(None, mat_type, **fields)
    cam = cams[0]
    _, _, rgb = oracle.pathtrace(scene, W, H, cam.params, cam.transform, 8, spp, ptype, want_f32=True)
    return rgb


@pytest.mark.parametrize("label,mat_type,fields,lower,exact", FURNACES, ids=[f[0].replace(" ", "_") for f in FURNACES])
def test_white_furnace_energy(built, label, mat_type, fields, lower, exact):
    """Closed-form pin: inside a constant environment of radiance E a surface that absorbs nothing returns exactly E
    (matte albedo 1; smooth dielectrics) and no surface returns more than E (energy conservation); the Standard, Naive and
    Direct integrators are estimators of that same value.  Pixels that miss the sphere read E exactly."""
    for ptype in (0, 2, 3):
        rgb = oracle_furnace(mat_type, fields, ptype)
        sphere = float(rgb[SPHERE].mean())
        assert np.all(rgb[:8, :8] == np.float32(E)), label                    # background: the environment itself
        assert sphere <= E * 1.01, (label, ptype, sphere)                    # never brighter than the furnace (1 % Monte-Carlo slack)
        if not (ptype == 3 and label in DIRECT_LOSES_ENERGY):
            assert sphere >= E * lower - 0.006, (label, ptype, sphere)
        if exact:
            assert abs(sphere - E) < 0.006, (label, ptype, sphere)


def test_mis_integrator_double_counts_environments(built):
    """Reference quirk, reproduced on purpose: pathtrace_mis adds the environment on EVERY miss (pathtracer.wgsl:757-761, not
    gated by next_emission like the surface emission at :794-796) although both MIS shadow rays already added their weighted
    environment terms (:831-849).  In the white furnace a non-delta surface therefore returns 2 E; delta surfaces (no shadow
    rays, next_emission stays true) return E.  Scenes without environments are unaffected (see the golden pin below)."""
    rgb = oracle_furnace(MT.Matte, dict(roughness=0.0), 1)
    assert abs(float(rgb[SPHERE].mean()) - 2 * E) < 0.01
    rgb = oracle_furnace(MT.Refractive, dict(roughness=0.0), 1)
    assert abs(float(rgb[SPHERE].mean()) - E) < 0.006
    assert np.all(rgb[:8, :8] == np.float32(E))


@pytest.mark.gpu
@pytest.mark.parametrize("label,mat_type,fields,lower,exact", FURNACES, ids=[f[0].replace(" ", "_") for f in FURNACES])
def test_furnaces_gpu_bit_exact(gpu_ctx, label, mat_type, fields, lower, exact):
    scene, cams = util.furnace_variant(gpu_ctx, mat_type, **fields)
    cam = cams[0]
    for ptype in (0, 1, 2, 3):
        got = util.gpu_accumulate(gpu_ctx, scene, cam, W, H, frames=2, spp=8, ptype=ptype)
        ref = util.oracle_accumulate(scene, cam, W, H, frames=2, spp=8, ptype=ptype)
        assert util.f16_words_differ(got, ref) == 0, (label, ptype)


# MIS / Naive / Direct against the reference's goldens (made with Standard).  The four integrators of the reference are NOT
# exact estimators of one image: every sample is clamped at max_radiance = 10 (the clamp's bias depends on the estimator's
# sample distribution), pathtrace_mis double counts environments (above), Direct's light ray stops at the first surface, and
# sample_lights_pdf has no visibility term.  What the reference data can pin is therefore agreement at the few-percent level
# where those effects are small; measured on MI355X (gpurun_out/r2_pins.log, 1010 spp at half the golden resolution):
#   materials1 cam1   Naive 1.006 / 3.7 %    Direct 0.969 / 14.9 %       (mean ratio / 8x8-block relative RMSE)
#   environments1 cam1                       Direct 0.980 /  8.6 %       Naive 0.689: BSDF-only sampling of a sun of radiance
#                                                                        15 552 is almost always clamped -- not comparable
#   arealights1 cam1  MIS 1.017   Naive 0.962   Direct 0.986;   cam2  MIS 1.013     (no environment: MIS is comparable here)
OTHER_INTEGRATOR_PINS = [
    # scene, camera, integrator, mean tolerance, block rel-RMSE tolerance
    ("materials1", 1, 2, 0.015, 0.06), ("materials1", 1, 3, 0.045, 0.20),
    ("environments1", 1, 3, 0.03, 0.12),
    ("arealights1", 1, 1, 0.025, None), ("arealights1", 1, 2, 0.05, None), ("arealights1", 1, 3, 0.025, None),
    ("arealights1", 2, 1, 0.025, None),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,cam_i,ptype,mean_tol,rmse_tol", OTHER_INTEGRATOR_PINS)
def test_other_integrators_land_on_the_standard_goldens(gpu_ctx, name, cam_i, ptype, mean_tol, rmse_tol):
    """lupin_tests' protocol (10 spp x 101 frames, 8 bounces, max_radiance 10, lupin_tests/src/main.rs:29-35,125-138) with
    pathtrace_type = MIS / Naive / Direct against the reference's golden render of the same view (made with Standard)."""
    scene, cams = util.load_scene(name, gpu_ctx)
    cam = cams[cam_i]
    small, _, _ = util.golden_render(name, cam_i)
    Hh, Wh = small.shape[0] * 2, small.shape[1] * 2
    adv = api.AdvancedParams(max_radiance=10.0)
    img = util.gpu_accumulate(gpu_ctx, scene, cam, Wh, Hh, frames=101, spp=10, ptype=ptype, advanced=adv).astype(np.float32)[..., :3]
    mine = img.reshape(Hh // 2, 2, Wh // 2, 2, 3).mean(axis=(1, 3))
    ratio = float(mine.mean() / small.mean())
    bh, bw = (small.shape[0] // 8) * 8, (small.shape[1] // 8) * 8
    a = mine[:bh, :bw].reshape(bh // 8, 8, bw // 8, 8, 3).mean(axis=(1, 3))
    b = small[:bh, :bw].reshape(bh // 8, 8, bw // 8, 8, 3).mean(axis=(1, 3))
    rel_rmse = float(np.sqrt(((a - b) ** 2).mean()) / small.mean())
    print(f"PIN {name} cam{cam_i} type{ptype}: mean ratio {ratio:.4f}, block rel-rmse {rel_rmse:.4f}")
    assert abs(ratio - 1.0) < mean_tol, ratio
    if rmse_tol is not None:
        assert rel_rmse < rmse_tol, rel_rmse


# The same four integrators WITHOUT the per-sample clamp (max_radiance = 1e30) and with f32 accumulation: what is left between
# them is Monte-Carlo noise and the reference's documented quirks, not clamp bias.  Measured on MI355X, 1024 spp at 480 x 270
# (tools/integrator_agreement.py, profiles/r03_integrator_agreement.jsonl), mean ratio to the Standard integrator:
#   arealights1 (no environment)  cam1  MIS 0.9998  Naive 1.0003  Direct 0.9989     cam2  MIS 1.0001  Naive 1.0003  Direct 1.0033
#   materials1 cam1  Naive 1.0001  Direct 0.9987   materials4 cam1  Naive 1.0001  Direct 1.0020   features1  1.0006 / 1.0004
#   MIS on the scenes WITH an environment: 1.30 - 1.46 -- the reference's double-counted environment (pathtracer.wgsl:757-761)
# So the 2 - 5 % offsets of the clamped pins above are the clamp's estimator-dependent bias; the implementations of the four
# integrators agree to 0.4 % where the reference's quirks do not apply.
AGREEMENT = [
    # scene, camera, integrators compared with Standard
    ("arealights1", 1, (1, 2, 3)), ("arealights1", 2, (1, 2, 3)),
    ("materials1", 1, (2, 3)), ("materials4", 1, (2, 3)), ("features1", 1, (2, 3)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,cam_i,types", AGREEMENT)
def test_integrators_agree_once_the_clamp_is_out_of_the_way(gpu_ctx, name, cam_i, types):
    scene, cams = util.load_scene(name, gpu_ctx)
    cam = cams[cam_i]
    Wd, Hd, spp, frames = 480, 270, 16, 65
    params = api.CameraParams(**{**cam.params.__dict__, "aspect": Wd / Hd})
    adv = api.AdvancedParams(max_radiance=1e30)
    gpu_ctx.set_accumulation_mode(1)
    try:
        means = {}
        for ptype in (0,) + tuple(types):
            res = api.build_pathtrace_resources(gpu_ctx, api.BakedPathtraceParams(max_bounces=8, samples_per_pixel=spp))
            out = api.DoubleBufferedTexture(gpu_ctx, Wd, Hd)
            for k in range(frames):
                api.pathtrace_scene(gpu_ctx, res, scene, out.front(), ptype,
                                    api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), k), camera_params=params,
                                                      camera_transform=cam.transform, advanced=adv))
                out.flip()
            out.flip()
            means[ptype] = float(out.front().download_f32()[..., :3].astype(np.float64).mean())
    finally:
        gpu_ctx.set_accumulation_mode(0)
    for t in types:
        ratio = means[t] / means[0]
        print(f"AGREE {name} cam{cam_i} type{t}: mean ratio to Standard {ratio:.4f}")
        assert abs(ratio - 1.0) < 0.01, (name, cam_i, t, ratio)
