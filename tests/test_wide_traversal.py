"""The four-wide traversal of the persistent tracer (DESIGN.md 5 "Wide traversal"): the collapsed hierarchy (CPU), the
certificate (unflagged rays equal the reference's order bit for bit, GPU), and the pipeline with its re-trace (GPU)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from lupinpathtracer_amd import _abi, api
from tests import util

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_LEAF = 0x80000000
REF_NONE = 0xFFFFFFFF
REF_LEAKY = 0x40000000
REF_INDEX = 0x3FFFFFFF


def collapse(nodes, verts=None, idx=None, ntris=0):
    """lupin_hip_collapse_bvh4 on a BVH_NODE_DTYPE array -> (wide nodes as (n, 32) u32 words, root reference, triangle flags)."""
    nodes = np.ascontiguousarray(nodes, _abi.BVH_NODE_DTYPE)
    root = C.c_uint32()
    if verts is not None:
        verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 4)
        idx = np.ascontiguousarray(idx, np.uint32)
        ntris = len(idx) // 3
    nverts = 0 if verts is None else len(verts)
    args = (_abi.ptr(nodes), len(nodes), _abi.ptr(verts), nverts, _abi.ptr(idx), ntris * 3)
    n = _abi.lib().lupin_hip_collapse_bvh4(*args, None, 0, C.byref(root), None)
    assert n >= 0, _abi.lib().lupin_hip_last_error()
    out = np.zeros((max(n, 1), 32), np.uint32)
    flags = np.zeros(max(ntris, 1), np.uint8)
    n2 = _abi.lib().lupin_hip_collapse_bvh4(*args, _abi.ptr(out), n, C.byref(root), _abi.ptr(flags) if verts is not None else None)
    assert n2 == n
    return out[:n], root.value, flags[:ntris]


def leaky_by_numpy(nodes, verts, idx):
    """Independent restatement: (triangles not inside every stored box above them, nodes whose box misses a triangle below)."""
    v = np.ascontiguousarray(verts, np.float32).reshape(-1, 4)[:, :3]
    tri = v[np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)]            # (t, 3, 3)
    tlo, thi = tri.min(axis=1), tri.max(axis=1)
    leaky_tri = np.zeros(len(tri), bool)
    leaky_node = np.zeros(len(nodes), bool)

    def walk(n, clip_lo, clip_hi):
        nd = nodes[n]
        clip_lo = np.maximum(clip_lo, nd["aabb_min"]); clip_hi = np.minimum(clip_hi, nd["aabb_max"])
        if nd["tri_count"] > 0:
            b, c = int(nd["tri_begin_or_first_child"]), int(nd["tri_count"])
            leaky_tri[b:b + c] = ~(np.all(tlo[b:b + c] >= clip_lo, axis=1) & np.all(thi[b:b + c] <= clip_hi, axis=1))
            lo, hi = tlo[b:b + c].min(axis=0), thi[b:b + c].max(axis=0)
        else:
            c0 = int(nd["tri_begin_or_first_child"])
            l0, h0 = walk(c0, clip_lo, clip_hi)
            l1, h1 = walk(c0 + 1, clip_lo, clip_hi)
            lo, hi = np.minimum(l0, l1), np.maximum(h0, h1)
        leaky_node[n] = not (np.all(lo >= nd["aabb_min"]) and np.all(hi <= nd["aabb_max"]))
        return lo, hi
    import sys
    sys.setrecursionlimit(10000)
    walk(0, np.full(3, -np.inf, np.float32), np.full(3, np.inf, np.float32))
    return leaky_tri, leaky_node


def check_collapse(nodes, verts, idx):
    """Every child box of a wide node is a box the reference's tree stores (bit for bit), every leaf of the binary tree hangs
    under exactly one wide slot with its own box, unused slots are NaN / REF_NONE, a node holds two to four children, and
    the 'box does not bound its triangles' marks (bit 30 of a child reference, bit 1 of a triangle's flags) equal an
    independent numpy restatement."""
    ntris = len(idx) // 3
    wide, root, flags = collapse(nodes, verts, idx)
    leaky_tri, leaky_node = leaky_by_numpy(nodes, verts, idx)
    assert np.array_equal((flags & 2) != 0, leaky_tri)
    ends = np.zeros(ntris, bool)
    for nd in nodes:
        if nd["tri_count"] > 0:
            ends[int(nd["tri_begin_or_first_child"]) + int(nd["tri_count"]) - 1] = True
    assert np.array_equal((flags & 1) != 0, ends)
    box = lambda nd: (tuple(nd["aabb_min"].tolist()), tuple(nd["aabb_max"].tolist()))
    if nodes[0]["tri_count"] > 0:
        assert len(wide) == 0 and root == (REF_LEAF | int(nodes[0]["tri_begin_or_first_child"]))
        return 0, int(leaky_tri.sum())
    leaf_of = {int(nd["tri_begin_or_first_child"]): i for i, nd in enumerate(nodes) if nd["tri_count"] > 0}
    internal = {}
    for i, nd in enumerate(nodes):
        if nd["tri_count"] == 0:
            internal.setdefault(box(nd), []).append(i)
    f32 = wide.view(np.float32)
    seen_leaves, visited, stack = [], set(), [root]
    while stack:
        w = stack.pop()
        assert w not in visited and w < len(wide)
        visited.add(w)
        refs = wide[w, 24:28]
        used = refs != REF_NONE
        assert 2 <= used.sum() <= 4 and np.all(wide[w, 28:32] == 0)
        for k in range(4):
            lo = tuple(f32[w, [0 + k, 4 + k, 8 + k]].tolist())
            hi = tuple(f32[w, [12 + k, 16 + k, 20 + k]].tolist())
            if not used[k]:
                assert all(np.isnan(v) for v in lo + hi)
                continue
            r = int(refs[k])
            leaky = bool(r & REF_LEAKY)
            if r & REF_LEAF:
                n = leaf_of[r & REF_INDEX]
                assert box(nodes[n]) == (lo, hi)
                assert leaky == bool(leaky_node[n])
                seen_leaves.append(r & REF_INDEX)
            else:
                cands = internal[(lo, hi)]                       # a box of the reference's tree, bit for bit
                assert leaky in {bool(leaky_node[n]) for n in cands}
                stack.append(r & REF_INDEX)
    assert len(visited) == len(wide)
    assert sorted(seen_leaves) == sorted(leaf_of)
    return len(wide), int(leaky_tri.sum())


def soup(rng, ntris, spread=10.0, size=0.2):
    pos = np.zeros((ntris * 3, 4), np.float32)
    centres = rng.random((ntris, 1, 3), dtype=np.float32) * np.float32(spread)
    pos[:, :3] = (centres + rng.normal(size=(ntris, 3, 3)).astype(np.float32) * np.float32(size)).reshape(-1, 3)
    return pos, np.arange(ntris * 3, dtype=np.uint32)


def test_collapse_on_built_trees(built):
    """lupin_build_bvh's trees of synthetic meshes (1 triangle ... a 2 000-triangle soup) collapse losslessly."""
    rng = np.random.default_rng(11)
    for ntris in (1, 2, 3, 5, 17, 256, 2000):
        pos, idx = soup(rng, ntris)
        nodes, idx2 = api.build_bvh(pos, idx)
        n_wide, _ = check_collapse(nodes, pos, idx2)
        n_internal = int((nodes["tri_count"] == 0).sum())
        if n_internal:
            assert (n_internal + 2) // 3 <= n_wide <= n_internal


def test_collapse_on_the_fixture_meshes(built):
    """... and so do the BLASes of a reference scene's meshes -- on which the reference's builder does leave triangles outside
    the boxes above them (child boxes from centroid bins, partition by comparison: data_structures.rs:366-466 vs :249-262)."""
    from lupinpathtracer_amd import loader
    scene_cpu, _, _, _ = loader.load_scene_cpu_yoctogl_v24(os.path.join(util.SCENES, "materials1", "materials1.json"), [util.SHARED])
    done, leaky = 0, 0
    for v, idx in sorted(zip(scene_cpu.verts_pos_array, scene_cpu.indices_array), key=lambda p: -len(p[1])):
        if len(idx) // 3 > 80000 or done >= 3:
            continue
        nodes, idx2 = api.build_bvh(v, idx)
        leaky += check_collapse(nodes, v, idx2)[1]
        done += 1
    assert done >= 2
    print("triangles outside a box above them:", leaky)


def test_reference_builder_leaves_triangles_outside_their_boxes(built):
    """The bistro-class stand-in's big mesh: lupin_build_bvh (= the reference's binned builder) stores child boxes that do
    not contain every triangle of the child; the marks must find them (here: compared with the numpy restatement), because
    a ray can hit such a triangle where the reference's own traversal never looks (DESIGN.md 5 "Wide traversal")."""
    from lupinpathtracer_amd import loader
    scene_cpu = loader.build_scene_bistro_class_cpu(util.SHARED, n_meshes=3, n_instances=40, n_lights=12, n_materials=24)[0]
    leaky = 0
    for v, idx in zip(scene_cpu.verts_pos_array, scene_cpu.indices_array):
        if len(idx) < 3000:
            continue
        nodes, idx2 = api.build_bvh(v, idx)
        n_wide, lk = check_collapse(nodes, v, idx2)
        print(f"{len(idx) // 3} triangles, {n_wide} wide nodes, {lk} triangles outside a box above them")
        leaky += lk
    assert leaky > 0


def test_collapse_rejects_malformed_input(built):
    nodes = np.zeros(3, _abi.BVH_NODE_DTYPE)
    nodes[0]["tri_begin_or_first_child"] = 5      # children out of range
    root = C.c_uint32()
    assert _abi.lib().lupin_hip_collapse_bvh4(_abi.ptr(nodes), 3, None, 0, None, 12, None, 0, C.byref(root), None) < 0


# ---------------------------------------------------------------- GPU ----------------------------------------------------------------

def _rays(rng, n, scale=3.0, centre=(0, 1, 0)):
    ori = (rng.random((n, 3), dtype=np.float32) * 2 - 1) * np.float32(scale) + np.array(centre, np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    d[:100] = np.array([0, -1, 0], np.float32)      # axis-aligned rays: zero direction components -> inf inv_dir
    d[100:200] = np.array([1, 0, 0], np.float32)
    return ori, d


@pytest.mark.gpu
@pytest.mark.parametrize("name,scale,centre", [("materials1", 3.0, (0, 1, 0)), ("instances1", 3.0, (0, 1, 0)), ("features1", 3.0, (0, 1, 0)),
                                               ("bistro_class_small", 20.0, (0, 5, 0))])
def test_wide_probe_equals_reference_order_unless_flagged(gpu_ctx, name, scale, centre):
    """The certificate on 300 k random rays per scene: a ray the wide traversal does not flag has the oracle's hit, instance,
    triangle, and bit-equal t, u, v; flagged rays are few."""
    from oracle import oracle
    scene, cams = util.load_scene(name, gpu_ctx)
    ori, d = _rays(np.random.default_rng(7), 300000, scale, centre)
    w = api.trace_rays_wide(gpu_ctx, scene, ori, d)
    o = oracle.trace_rays(scene, ori, d)
    b = api.trace_rays(gpu_ctx, scene, ori, d)
    assert np.array_equal(b[0], o[0])
    ok = w[5] == 0
    frac = 1.0 - ok.mean()
    print(f"{name}: {100 * frac:.3f} % of the rays flagged for re-trace, hits {o[0].mean():.2f}")
    assert frac < 0.03
    assert np.array_equal(w[0][ok], o[0][ok])
    hit = ok & (o[0] == 1)
    assert hit.sum() > 1000
    assert np.array_equal(w[3][hit], o[3][hit]) and np.array_equal(w[4][hit], o[4][hit])
    assert np.array_equal(w[1][hit].view(np.uint32), o[1][hit].view(np.uint32))
    assert np.array_equal(w[2][hit].view(np.uint32), o[2][hit].view(np.uint32))


def _coincident_scene(ctx):
    """Exact ties on purpose: the Cornell box with every instance present TWICE at the same place (each hit is an exact tie
    between two instances, the case in which visiting order decides), plus a 3 000-triangle soup above it that makes the
    scene too large for LDS staging, so the persistent tracer and its wide hierarchy serve it."""
    from lupinpathtracer_amd import loader
    scene_cpu, cams = loader.cornell_box_scene_cpu()
    rng = np.random.default_rng(3)
    nt = 3000
    pos = np.zeros((nt * 3, 4), np.float32)
    centres = rng.random((nt, 1, 3), dtype=np.float32) * np.array([1.6, 0.3, 1.6], np.float32) + np.array([-0.8, 1.55, -0.8], np.float32)
    pos[:, :3] = (centres + rng.normal(size=(nt, 3, 3)).astype(np.float32) * 0.03).reshape(-1, 3)
    scene_cpu.verts_pos_array.append(pos)
    scene_cpu.indices_array.append(np.arange(nt * 3, dtype=np.uint32))
    scene_cpu.mesh_infos = np.concatenate([scene_cpu.mesh_infos, np.array([api.default_mesh_info()], _abi.MESH_INFO_DTYPE)])
    soup = api.default_instance()
    soup["mesh_idx"] = len(scene_cpu.verts_pos_array) - 1
    soup["mat_idx"] = 0
    insts = list(scene_cpu.instances) + [soup]
    scene_cpu.instances = np.array(insts + insts, _abi.INSTANCE_DTYPE)
    api.validate_scene(scene_cpu, 0, 0)
    return api.build_accel_structures_and_upload(ctx, scene_cpu, [], [], True), cams


@pytest.mark.gpu
def test_exact_ties_are_flagged_and_the_pipeline_stays_exact(gpu_ctx):
    """Duplicated geometry: the wide probe must flag (nearly) every hit, what it does not flag must still equal the oracle,
    and the pipeline (wide tracer + re-trace in the reference's order) must render the oracle's image."""
    from oracle import oracle
    scene, cams = _coincident_scene(gpu_ctx)
    ori, d = _rays(np.random.default_rng(9), 100000, 0.9, (0, 1, 0))
    w = api.trace_rays_wide(gpu_ctx, scene, ori, d)
    o = oracle.trace_rays(scene, ori, d)
    hit = o[0] == 1
    assert hit.sum() > 50000
    flagged_hits = (w[5] == 1) & hit
    print(f"exact-tie scene: {100 * flagged_hits.sum() / hit.sum():.1f} % of the hits flagged")
    assert flagged_hits.sum() > 0.95 * hit.sum()
    ok = w[5] == 0
    assert np.array_equal(w[0][ok], o[0][ok])
    okh = ok & hit
    assert np.array_equal(w[3][okh], o[3][okh]) and np.array_equal(w[4][okh], o[4][okh])
    assert np.array_equal(w[1][okh].view(np.uint32), o[1][okh].view(np.uint32))
    cam = cams[0]
    gpu_ctx.set_traversal("wide")
    try:
        for ptype in (0, 1):
            gpu_ctx.stats_reset(0)
            got = util.gpu_accumulate(gpu_ctx, scene, cam, 96, 96, 2, 2, max_bounces=5, ptype=ptype)
            st = gpu_ctx.stats()
            ref = util.oracle_accumulate(scene, cam, 96, 96, 2, 2, max_bounces=5, ptype=ptype)
            assert util.f16_words_differ(got, ref) == 0
            assert st["wide_traversal"] == 1 and st["wide_retraced"] > 0.5 * st["wide_queries"]
    finally:
        gpu_ctx.set_traversal("binary")


@pytest.mark.gpu
@pytest.mark.parametrize("ptype", [0, 1, 3])
def test_pipeline_counts_and_reports_the_fallback(gpu_ctx, ptype):
    """With the wide tracer selected, the pipeline on a scene traversed from global memory runs it (stats say so), re-traces a
    small share of the queries, and renders the oracle's image bit for bit; the default (binary) renders the same image."""
    scene, cams = util.load_scene("bistro_class_small", gpu_ctx)
    cam = cams[0]
    W, H = 160, 96
    gpu_ctx.stats_reset(0)
    default = util.gpu_accumulate(gpu_ctx, scene, cam, W, H, 2, 3, max_bounces=6, ptype=ptype)
    st0 = gpu_ctx.stats()   # the default: binary; a first-pass count can only come from its short-stack pass (deep scenes)
    assert st0["wide_traversal"] == 0 and (st0["wide_queries"] == 0 or (st0["short_stack_entries"] > 0 and st0["wide_retraced"] * 1000 < st0["wide_queries"]))
    gpu_ctx.set_traversal("wide")
    try:
        gpu_ctx.stats_reset(0)
        got = util.gpu_accumulate(gpu_ctx, scene, cam, W, H, 2, 3, max_bounces=6, ptype=ptype)
        st = gpu_ctx.stats()
    finally:
        gpu_ctx.set_traversal("binary")
    ref = util.oracle_accumulate(scene, cam, W, H, 2, 3, max_bounces=6, ptype=ptype)
    assert util.f16_words_differ(got, ref) == 0 and util.f16_words_differ(default, ref) == 0
    assert st["wide_traversal"] == 1 and st["wide_queries"] > 0
    rate = st["wide_retraced"] / st["wide_queries"]
    print(f"type {ptype}: {st['wide_queries']} wide queries, {st['wide_retraced']} re-traced ({100 * rate:.3f} %)")
    assert 0 < st["wide_retraced"] and rate < 0.05


def _child(code, env):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


VERIFY = r"""
import json, sys
sys.path.insert(0, %r)
from lupinpathtracer_amd import api
from tests import util
ctx = api.Context(0)
out = {}
for name, W, H in (("bistro_class_small", 320, 180), ("materials1", 320, 180), ("features1", 240, 135)):
    scene, cams = util.load_scene(name, ctx)
    ctx.stats_reset(0)
    img = util.gpu_accumulate(ctx, scene, cams[0], W, H, 2, 4, max_bounces=8, ptype=0)
    st = ctx.stats()
    out[name] = {k: st[k] for k in ("verify_checked", "verify_flagged", "verify_mismatches", "verify_raw_mismatches", "wide_queries", "wide_retraced")}
print("RESULT " + json.dumps(out))
"""


@pytest.mark.gpu
def test_device_side_verification_finds_no_uncertified_difference(built):
    """LUPIN_VERIFY_WIDE=1: every closest-hit query of real path-traced frames, binary vs wide on the device.  No unflagged
    ray may differ; the raw count (flags ignored) shows what the certificate is there for."""
    import json
    out = _child(VERIFY % ROOT, {"LUPIN_VERIFY_WIDE": "1", "LUPIN_TRAVERSAL": "wide"})
    res = json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][0][7:])
    for name, r in res.items():
        print(name, r)
        assert r["verify_checked"] > 100000
        assert r["verify_mismatches"] == 0
        assert r["verify_flagged"] < 0.05 * r["verify_checked"]


@pytest.mark.gpu
def test_traversal_switch_renders_the_same_image(built):
    """LUPIN_TRAVERSAL=wide and the default (the reference's order only) give identical images and the switch is honoured."""
    code = r"""
import sys, hashlib
sys.path.insert(0, %r)
from lupinpathtracer_amd import api
from tests import util
ctx = api.Context(0)
scene, cams = util.load_scene("bistro_class_small", ctx)
ctx.stats_reset(0)
img = util.gpu_accumulate(ctx, scene, cams[0], 200, 120, 2, 3, max_bounces=6, ptype=0)
st = ctx.stats()
print("RESULT", hashlib.sha256(img.tobytes()).hexdigest(), st["wide_traversal"], st["wide_queries"])
""" % ROOT
    a = [l for l in _child(code, {"LUPIN_TRAVERSAL": "wide"}).splitlines() if l.startswith("RESULT")][0].split()
    b = [l for l in _child(code, {}).splitlines() if l.startswith("RESULT")][0].split()
    assert a[1] == b[1]
    assert a[2] == "1" and int(a[3]) > 0 and b[2] == "0"   # (b[3] may count the default's short-stack first pass, see above)
    c = [l for l in _child(code, {"LUPIN_SHORT_STACK": "0"}).splitlines() if l.startswith("RESULT")][0].split()
    assert c[1] == a[1] and c[2] == "0" and int(c[3]) == 0
