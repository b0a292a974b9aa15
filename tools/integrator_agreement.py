#!/usr/bin/env python3
"""The four integrators against each other WITHOUT the per-sample clamp (max_radiance = 1e30), f32 accumulation: what is left
between them is estimator noise and the reference's documented quirks, not clamp bias.
    python tools/integrator_agreement.py arealights1 1 [width height spp frames]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lupinpathtracer_amd import api
from tests import util

name = sys.argv[1]; cam_i = int(sys.argv[2])
W, H, spp, frames = (int(x) for x in sys.argv[3:7]) if len(sys.argv) > 6 else (480, 270, 16, 65)
ctx = api.Context(0)
scene, cams = util.load_scene(name, ctx)
cam = cams[cam_i]
params = api.CameraParams(**{**cam.params.__dict__, "aspect": W / H})
ctx.set_accumulation_mode(1)
imgs = {}
for ptype in (0, 1, 2, 3):
    res = api.build_pathtrace_resources(ctx, api.BakedPathtraceParams(max_bounces=8, samples_per_pixel=spp))
    out = api.DoubleBufferedTexture(ctx, W, H)
    for k in range(frames):
        api.pathtrace_scene(ctx, res, scene, out.front(), ptype,
                            api.PathtraceDesc(accum_params=api.AccumulationParams(out.back(), k), camera_params=params, camera_transform=cam.transform,
                                              advanced=api.AdvancedParams(max_radiance=1e30)))
        out.flip()
    out.flip()
    imgs[ptype] = out.front().download_f32()[..., :3].astype(np.float64)
ref = imgs[0]
bh, bw = (H // 16) * 16, (W // 16) * 16
blk = lambda a: a[:bh, :bw].reshape(bh // 16, 16, bw // 16, 16, 3).mean(axis=(1, 3))
for t in (1, 2, 3):
    ratio = imgs[t].mean() / ref.mean()
    rel = np.sqrt(((blk(imgs[t]) - blk(ref)) ** 2).mean()) / ref.mean()
    print(json.dumps({"scene": name, "cam": cam_i, "type": t, "mean_ratio_vs_standard": ratio, "block16_rel_rmse": rel, "spp": spp * (frames - 1)}))
