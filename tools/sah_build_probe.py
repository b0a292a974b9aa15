"""Times the device SAH builder on the bistro-class meshes (per mesh wall time); run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lupinpathtracer_amd import api, loader
from tests import util

ctx = api.Context()
scene_cpu, _, _, _ = loader.build_scene_bistro_class_cpu(util.SHARED)
meshes = [(np.ascontiguousarray(v, np.float32).reshape(-1, 4), np.ascontiguousarray(i, np.uint32)) for v, i in zip(scene_cpu.verts_pos_array, scene_cpu.indices_array)]
api.build_bvh_sah_device(ctx, *meshes[0])
tot = 0.0
for v, i in meshes:
    t0 = time.perf_counter(); nodes, _ = api.build_bvh_sah_device(ctx, v, i); dt = time.perf_counter() - t0
    tot += dt
    print(f"{len(i) // 3:8d} tris {len(nodes):8d} nodes {dt * 1e3:7.2f} ms")
print(f"total {tot * 1e3:.1f} ms")
# one mesh holding everything (2.88 M triangles): the frontier grows to ~1 M nodes
offs = np.cumsum([0] + [len(v) for v, _ in meshes[:-1]])
big_v = np.concatenate([v + np.array([[3.0 * k, 0, 0, 0]], np.float32) for k, (v, _) in enumerate(meshes)])
big_i = np.concatenate([i + np.uint32(o) for (_, i), o in zip(meshes, offs)])
for _ in range(2):
    t0 = time.perf_counter(); nodes, _ = api.build_bvh_sah_device(ctx, big_v, big_i); dt = time.perf_counter() - t0
    print(f"one mesh: {len(big_i) // 3} tris {len(nodes)} nodes {dt * 1e3:.1f} ms")
