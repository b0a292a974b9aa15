"""Regenerates tests/golden/ from the reference checkout (run in the build container only):

  scenes/<name>/<name>.json      the reference's test_scenes JSON, verbatim (data)
  scenes/_shared/{shapes,textures}/   the PLY / PNG / HDR assets those scenes name, de-duplicated
                                  (all 15 scenes share byte-identical copies)
  renders/<name>_cam<N>.npz      the reference's golden renders `render_cam<N>.hdr` (RGBE, 1920 wide,
                                  ~1000 spp), decoded and box-filtered 4x4 -> float16, plus the full-size mean
  renders/furnace1_cam0.hdr      the white-furnace known-answer golden, verbatim

These are fixtures (inputs and expected outputs), not reference source code.
"""
import glob
import hashlib
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/test_scenes"


def main():
    from lupinpathtracer_amd import loader
    shared = os.path.join(HERE, "scenes", "_shared")
    seen = {}
    for scene_dir in sorted(glob.glob(os.path.join(REF, "*"))):
        name = os.path.basename(scene_dir)
        js = os.path.join(scene_dir, name + ".json")
        if not os.path.exists(js):
            continue
        out = os.path.join(HERE, "scenes", name)
        os.makedirs(out, exist_ok=True)
        shutil.copyfile(js, os.path.join(out, name + ".json"))
        for sub in ("shapes", "textures"):
            for f in sorted(glob.glob(os.path.join(scene_dir, sub, "*"))):
                rel = os.path.join(sub, os.path.basename(f))
                h = hashlib.md5(open(f, "rb").read()).hexdigest()
                if rel in seen:
                    if seen[rel] != h:   # same name, different bytes: keep it scene-local
                        os.makedirs(os.path.join(out, sub), exist_ok=True)
                        shutil.copyfile(f, os.path.join(out, rel))
                    continue
                seen[rel] = h
                os.makedirs(os.path.join(shared, sub), exist_ok=True)
                shutil.copyfile(f, os.path.join(shared, rel))
        for g in sorted(glob.glob(os.path.join(scene_dir, "render_cam*.hdr"))):
            cam = os.path.basename(g)[len("render_cam"):-len(".hdr")]
            img = loader.read_hdr(g)
            h, w, _ = img.shape
            hh, ww = (h // 4) * 4, (w // 4) * 4
            small = img[:hh, :ww].reshape(hh // 4, 4, ww // 4, 4, 3).mean(axis=(1, 3))
            os.makedirs(os.path.join(HERE, "renders"), exist_ok=True)
            np.savez_compressed(os.path.join(HERE, "renders", f"{name}_cam{cam}.npz"), small=small.astype(np.float16),
                                full_shape=np.array([h, w]), full_mean=img.mean(axis=(0, 1)).astype(np.float32))
            if name == "furnace1":
                shutil.copyfile(g, os.path.join(HERE, "renders", f"{name}_cam{cam}.hdr"))
    print("fixtures written under", HERE)


if __name__ == "__main__":
    main()
