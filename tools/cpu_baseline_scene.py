#!/usr/bin/env python3
"""CPU restatement (oracle) throughput on a fixture scene, for the GPU / CPU ratio of configs other than bench.py's.
    python tools/cpu_baseline_scene.py bistro_class --width 480 --height 270 --bounces 16 --spp 2
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene"); ap.add_argument("--cam", type=int, default=0)
    ap.add_argument("--width", type=int, default=480); ap.add_argument("--height", type=int, default=270)
    ap.add_argument("--bounces", type=int, default=16); ap.add_argument("--spp", type=int, default=2)
    a = ap.parse_args()
    from lupinpathtracer_amd import api
    from oracle import oracle
    from tests import util
    scene, cams = util.load_scene(a.scene, None)
    cam = cams[a.cam]
    params = api.CameraParams(**{**cam.params.__dict__, "aspect": a.width / a.height})
    oracle.pathtrace(scene, 64, 36, params, cam.transform, a.bounces, 1)
    t0 = time.perf_counter()
    _, cnt = oracle.pathtrace(scene, a.width, a.height, params, cam.transform, a.bounces, a.spp)
    dt = time.perf_counter() - t0
    print(json.dumps({"scene": a.scene, "size": [a.width, a.height], "spp": a.spp, "bounces": a.bounces, "threads": oracle.num_threads(),
                      "seconds": dt, "path_bounces": cnt["path_bounces"], "Msamples_per_s": cnt["path_bounces"] / dt / 1e6}))

if __name__ == "__main__":
    main()
