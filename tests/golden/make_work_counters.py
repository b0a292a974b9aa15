"""Writes tests/golden/work_counters.json: the oracle's deterministic work accounting per bench workload
(SURVEY 8d).  bench.py reads this file (data only) to turn measured units/s into algorithmic bytes/s.

Algorithmic bytes per path-bounce, in the REFERENCE's data layout (what its shader has to touch):
    traversal(ctx) = 48 * tlas_aabb[ctx] + 64 * instances_entered[ctx] + 32 * blas_aabb[ctx] + 60 * tri_tests[ctx]
    extend  = traversal(0)                                  closest hit of the integrator loop
    shade   = traversal(1) + traversal(2)                   light-pdf marching + shadow / MIS rays
            + 184 * surface_hits + 48 * normal_fetches + 24 * uv_fetches + 48 * color_fetches
            + 16 * tex_ldr + 32 * tex_hdr                   (instance 64 + material 96 + mesh info 12 + indices 12; bilinear taps)
            + 80 * light_mesh + 92 * light_env              (light 8 + alias bin 12 + 3 positions 48 + indices 12 | alias 12 + env 80)
    resolve = 16 bytes per pixel per frame (prev read + store)
all divided by the number of path-bounces of the same run.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def per_unit(cnt, pixels, frames):
    u = cnt["path_bounces"]
    trav = lambda k: 48 * cnt["tlas_aabb"][k] + 64 * cnt["instances_entered"][k] + 32 * cnt["blas_aabb"][k] + 60 * cnt["tri_tests"][k]
    shade = (trav(1) + trav(2) + 184 * cnt["surface_hits"] + 48 * cnt["normal_fetches"] + 24 * cnt["uv_fetches"] +
             48 * cnt["color_fetches"] + 16 * cnt["tex_ldr"] + 32 * cnt["tex_hdr"] + 80 * cnt["light_mesh"] + 92 * cnt["light_env"])
    return {"extend_bytes_per_unit": trav(0) / u, "shade_bytes_per_unit": shade / u,
            "resolve_bytes_per_unit": 16.0 * pixels * frames / u, "bounces_per_path": u / cnt["paths"]}


def main():
    from lupinpathtracer_amd import api
    from oracle import oracle
    from tests import util
    out = {}
    workloads = [
        # key, scene, camera, width, height, max_bounces, spp, type
        ("cornellbox_1024x1024_b8_spp8_standard", "cornellbox_builtin", 0, 1024, 1024, 8, 8, 0),
        ("materials1_cam0_960x540_b12_spp4_standard", "materials1", 0, 960, 540, 12, 4, 0),
        ("environments1_cam0_960x540_b16_spp4_standard", "environments1", 0, 960, 540, 16, 4, 0),
        ("bistro_class_cam0_480x270_b16_spp2_standard", "bistro_class", 0, 480, 270, 16, 2, 0),
    ]
    for key, name, cam_i, w, h, mb, spp, ptype in workloads:
        scene, cams = util.load_scene(name)
        cam = cams[cam_i]
        params = cam.params
        if name.startswith("bistro_class"):   # rendered at 16:9 like tools/scene_bench.py does
            params = api.CameraParams(**{**cam.params.__dict__, "aspect": w / h})
        _, cnt = oracle.pathtrace(scene, w, h, params, cam.transform, mb, spp, ptype)
        rec = {"scene": name, "camera": cam_i, "width": w, "height": h, "max_bounces": mb, "samples_per_pixel": spp,
               "pathtrace_type": ptype, "frames": 1, "accum_counter": 0, "counters": cnt}
        rec.update(per_unit(cnt, w * h, 1))
        out[key] = rec
        print(key, {k: round(v, 2) for k, v in rec.items() if k.endswith("per_unit") or k == "bounces_per_path"})
    with open(os.path.join(HERE, "work_counters.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
