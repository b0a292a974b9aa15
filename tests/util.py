"""Shared helpers for the test-suite (fixtures live under tests/golden/, see make_fixtures.py)."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
SCENES = os.path.join(GOLDEN, "scenes")
SHARED = os.path.join(SCENES, "_shared")

_scene_cache = {}


def load_scene(name, ctx=None):
    """(Scene, cameras) for a fixture scene; host-side preprocessing is cached per (name, ctx is None)."""
    from lupinpathtracer_amd import loader
    key = (name, id(ctx) if ctx is not None else None)
    if key not in _scene_cache:
        if name == "cornellbox_builtin":
            _scene_cache[key] = loader.build_scene_cornell_box(ctx)
        elif name.startswith("bistro_class"):
            # "bistro_class" = full stand-in (20 meshes / 400 instances / 100 lights); "bistro_class_small" for tests
            kw = dict(n_meshes=3, n_instances=40, n_lights=12, n_materials=24) if name.endswith("small") else {}
            _scene_cache[key] = loader.build_scene_bistro_class(ctx, SHARED, **kw)
        else:
            path = os.path.join(SCENES, name, name + ".json")
            _scene_cache[key] = loader.load_scene_yoctogl_v24(path, ctx, asset_dirs=[SHARED])
    return _scene_cache[key]


def golden_render(name, cam):
    """Downsampled (4x4 box) reference golden render and its full-size mean."""
    z = np.load(os.path.join(GOLDEN, "renders", f"{name}_cam{cam}.npz"))
    return z["small"].astype(np.float32), z["full_shape"], z["full_mean"]


def f16_words_differ(a, b):
    return int((np.ascontiguousarray(a).view(np.uint16) != np.ascontiguousarray(b).view(np.uint16)).sum())


def oracle_accumulate(scene, cam, width, height, frames, spp, max_bounces=8, ptype=0, advanced=None, start=0):
    """example1.rs:37-52 on the CPU oracle: frames accum_counter = start..start+frames-1, f16 running average."""
    from oracle import oracle
    prev = np.zeros((height, width, 4), np.float16)
    out = prev
    for k in range(start, start + frames):
        out, _ = oracle.pathtrace(scene, width, height, cam.params, cam.transform, max_bounces, spp, ptype,
                                  accum_counter=k, prev_frame=prev, advanced=advanced)
        prev = out
    return out


def gpu_accumulate(ctx, scene, cam, width, height, frames, spp, max_bounces=8, ptype=0, advanced=None, start=0):
    """The same loop through the C ABI (DoubleBufferedTexture + pathtrace_scene + flip)."""
    from lupinpathtracer_amd import api
    res = api.build_pathtrace_resources(ctx, api.BakedPathtraceParams(max_bounces=max_bounces, samples_per_pixel=spp))
    out = api.DoubleBufferedTexture(ctx, width, height)
    for k in range(start, start + frames):
        desc = api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), k), camera_params=cam.params,
                                 camera_transform=cam.transform, advanced=advanced or api.AdvancedParams())
        api.pathtrace_scene(ctx, res, scene, out.front(), ptype, desc)
        out.flip()
    out.flip()
    return out.front().download()


def furnace_variant(ctx, mat_type, **fields):
    """The reference's white-furnace scene (test_scenes/furnace1: constant environment 0.5, one sphere) with the sphere's
    material replaced: a closed-form energy test for materials no reference scene uses (subsurface, gltfpbr)."""
    from lupinpathtracer_amd import api, loader
    path = os.path.join(SCENES, "furnace1", "furnace1.json")
    scene_cpu, textures, envs_info, cams = loader.load_scene_cpu_yoctogl_v24(path, [SHARED])
    m = scene_cpu.materials[0]
    m["mat_type"] = int(mat_type)
    for k, v in fields.items():
        m[k] = v
    return api.build_accel_structures_and_upload(ctx, scene_cpu, textures, envs_info, True), cams
