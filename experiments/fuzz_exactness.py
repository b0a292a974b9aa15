import sys; sys.path.insert(0, ".")
import numpy as np
from tests.test_gpu_fuzz import random_scene
from tests import util
from lupinpathtracer_amd import api
ctx = api.Context(0)
tot = 0
for seed in range(1, 9):
    s, t, e, c = random_scene(seed)
    sc = api.build_accel_structures_and_upload(ctx, s, t, e)
    for p in range(4):
        cam = c[(seed + p) % 3]
        got = util.gpu_accumulate(ctx, sc, cam, 120, 80, frames=2, spp=3, max_bounces=7, ptype=p)
        ref = util.oracle_accumulate(sc, cam, 120, 80, frames=2, spp=3, max_bounces=7, ptype=p)
        nb = util.f16_words_differ(got, ref); tot += nb
        print(seed, p, nb, flush=True)
print("total differing f16 words:", tot)
