"""Bit-exactness campaign: seeded synthetic scenes (tests/test_gpu_fuzz.random_scene: every material type, every texture slot, vertex
colours, opacity, several environments, shear, ortho / DOF cameras) x all four integrators, HIP path vs the oracle, with the
light-pdf stage off and on.  Prints one line per (seed, integrator) with differing f16 words and a total.

    python tools/fuzz_campaign.py <first seed> <last seed> [--light-stage]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--light-stage" in sys.argv:
    os.environ["LUPIN_LIGHT_STAGE"] = "1"
from tests.test_gpu_fuzz import random_scene
from tests import util
from lupinpathtracer_amd import api

first, last = int(sys.argv[1]), int(sys.argv[2])
ctx = api.Context(0)
tot = 0
for seed in range(first, last + 1):
    s, t, e, c = random_scene(seed)
    sc = api.build_accel_structures_and_upload(ctx, s, t, e)
    for p in range(4):
        cam = c[(seed + p) % 3]
        got = util.gpu_accumulate(ctx, sc, cam, 120, 80, frames=2, spp=3, max_bounces=7, ptype=p)
        ref = util.oracle_accumulate(sc, cam, 120, 80, frames=2, spp=3, max_bounces=7, ptype=p)
        nb = util.f16_words_differ(got, ref); tot += nb
        if nb: print("MISMATCH seed", seed, "integrator", p, "words", nb, flush=True)
    if seed % 10 == 0: print("seed", seed, "done, total so far", tot, flush=True)
print("seeds %d..%d: total differing f16 words: %d" % (first, last, tot))
