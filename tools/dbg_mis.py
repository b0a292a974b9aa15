import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from lupinpathtracer_amd import api
from tests import util
name = sys.argv[1] if len(sys.argv) > 1 else "bistro_class_small"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (160, 96)
check = len(sys.argv) <= 4
ctx = api.Context(0)
scene, cams = util.load_scene(name, ctx)
for ptype in (0, 1, 3):
    ctx.stats_reset(0)
    got = util.gpu_accumulate(ctx, scene, cams[0], W, H, 2, 3, max_bounces=6, ptype=ptype)
    st = ctx.stats()
    bad = []
    if check:
        ref = util.oracle_accumulate(scene, cams[0], W, H, 2, 3, max_bounces=6, ptype=ptype)
        bad = np.argwhere(got.view(np.uint16) != ref.view(np.uint16))
    print("type", ptype, "differing words", len(bad), {k: st[k] for k in st if k.startswith("verify") or k.startswith("wide_q") or k.startswith("wide_r")})
