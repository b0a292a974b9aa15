#!/usr/bin/env python3
"""HIP path vs every golden render the reference ships (tests/golden/renders), with lupin_tests' protocol:
full golden resolution, 10 spp x 101 frames (accum_counter 0..100), 8 bounces, Standard, max_radiance 10.
Both images are box-filtered 4x4 (the fixtures keep the goldens that way); prints one JSON line per golden:
mean ratio, RMSE and mean-relative RMSE of the filtered images, and the reference's own per-pixel L2 acceptance
statistic (lupin_tests/src/main.rs:35: L2 over rgb <= 5.0) on the filtered pixels.

    python tools/golden_compare.py [name ...] > gpurun_out/golden_compare.jsonl
"""
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from lupinpathtracer_amd import api
    from tests import util
    ctx = api.Context(0)
    want = set(sys.argv[1:])
    for path in sorted(glob.glob(os.path.join(util.GOLDEN, "renders", "*.npz"))):
        stem = os.path.basename(path)[:-4]
        name, cam_i = stem.rsplit("_cam", 1)
        if want and name not in want:
            continue
        cam_i = int(cam_i)
        small, full_shape, full_mean = util.golden_render(name, cam_i)
        H, W = small.shape[0] * 4, small.shape[1] * 4
        scene, cams = util.load_scene(name, ctx)
        t0 = time.perf_counter()
        img = util.gpu_accumulate(ctx, scene, cams[cam_i], W, H, frames=101, spp=10, advanced=api.AdvancedParams(max_radiance=10.0))
        dt = time.perf_counter() - t0
        mine = img[..., :3].astype(np.float32).reshape(H // 4, 4, W // 4, 4, 3).mean(axis=(1, 3))
        d = mine - small
        l2 = np.sqrt((d ** 2).sum(axis=2))
        print(json.dumps({"golden": stem, "size": [W, H], "seconds": round(dt, 2), "mean_golden": float(small.mean()),
                          "mean_ratio": float(mine.mean() / small.mean()), "rmse_4x4": float(np.sqrt((d ** 2).mean())),
                          "rel_rmse_4x4": float(np.sqrt((d ** 2).mean()) / small.mean()),
                          "max_pixel_l2_4x4": float(l2.max()), "pixels_over_l2_5": int((l2 > 5.0).sum())}), flush=True)


if __name__ == "__main__":
    main()
