"""ctypes wrapper of oracle/liblupin_oracle.so (the CPU restatement of the reference megakernel).

TEST INFRASTRUCTURE: see oracle/lupin_oracle.cpp for the parity status of this oracle.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from lupinpathtracer_amd import _abi
from lupinpathtracer_amd.api import CameraParams, AdvancedParams, scene_flags

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblupin_oracle.so")


class OracleCounters(C.Structure):
    _fields_ = ([("path_bounces", C.c_uint64), ("paths", C.c_uint64)] +
                [(k, C.c_uint64 * 3) for k in ("tlas_aabb", "instances_entered", "blas_aabb", "tri_tests")] +
                [(k, C.c_uint64) for k in ("material_points", "tex_ldr", "tex_hdr", "light_mesh", "light_env",
                                           "closest_hit_queries", "light_pdf_queries", "surface_hits",
                                           "normal_fetches", "uv_fetches", "color_fetches")])


_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        h = C.CDLL(LIB_PATH)
        h.oracle_pathtrace.restype = C.c_int
        h.oracle_pathtrace.argtypes = [C.POINTER(_abi.SceneDesc), C.POINTER(_abi.PushConstants), C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.POINTER(OracleCounters), C.c_int, C.c_int, C.c_int]
        h.oracle_pathtrace_f32prev.restype = C.c_int
        h.oracle_pathtrace_f32prev.argtypes = list(h.oracle_pathtrace.argtypes) + [C.c_void_p]
        h.oracle_trace_rays.restype = C.c_int
        h.oracle_trace_rays.argtypes = [C.POINTER(_abi.SceneDesc), C.c_uint32, C.c_void_p, C.c_void_p, C.c_float, C.c_uint32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        h.oracle_rng_stream.restype = None
        h.oracle_rng_stream.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        h.oracle_bsdf_probe.restype = None
        h.oracle_bsdf_probe.argtypes = [C.c_uint32, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                        C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        h.oracle_float_to_half.restype = C.c_uint16
        h.oracle_float_to_half.argtypes = [C.c_float]
        h.oracle_float_to_half_rtz.restype = C.c_uint16
        h.oracle_float_to_half_rtz.argtypes = [C.c_float]
        h.oracle_half_to_float.restype = C.c_float
        h.oracle_half_to_float.argtypes = [C.c_uint16]
        h.oracle_num_threads.restype = C.c_int
        h.oracle_tonemap.restype = C.c_int
        h.oracle_tonemap.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                     C.c_float, C.c_int, C.c_int, C.c_int]
        _lib = h
    return _lib


def push_constants(scene, camera_params: CameraParams, camera_transform, pathtrace_type=0, accum_counter=0,
                   advanced: AdvancedParams = None, id_offset=(0, 0)):
    """get_push_constants{,_tiled} (renderer.rs:1426-1506)."""
    advanced = advanced or AdvancedParams()
    pc = _abi.PushConstants()
    m = np.asarray(camera_transform, np.float32).reshape(4, 3)
    for c in range(4):
        for r in range(3):
            pc.camera_transform.m[c][r] = float(m[c][r])
        pc.camera_transform.m[c][3] = 1.0 if c == 3 else 0.0
    pc.camera_lens = camera_params.lens
    pc.camera_film = camera_params.film
    pc.camera_aspect = camera_params.aspect
    pc.camera_focus = camera_params.focus
    pc.camera_aperture = camera_params.aperture
    pc.flags = scene_flags(scene, camera_params)
    pc.id_offset[0], pc.id_offset[1] = id_offset
    pc.accum_counter = accum_counter
    pc.pathtrace_type = int(pathtrace_type)
    pc.max_radiance = advanced.max_radiance
    pc.rng_seed = advanced.rng_seed
    pc.ray_epsilon = advanced.ray_epsilon
    return pc


def dispatch_extent(width, height, tile_params=None):
    """(id_offset, groups_x, groups_y) of renderer.rs:807-838."""
    ws = _abi.WORKGROUP_SIZE
    if tile_params is None:
        return (0, 0), (width + ws - 1) // ws, (height + ws - 1) // ws
    ts, ti = tile_params.tile_size, tile_params.tile_idx
    ntx = (max(1, width) - 1) // (ts * ws) + 1
    nty = (max(1, height) - 1) // (ts * ws) + 1
    assert ti < ntx * nty, "tile_idx out of range!"
    ox, oy = (ti % ntx) * ts * ws, (ti // ntx) * ts * ws
    return (ox, oy), min(ts, (width - ox) // ws), min(ts, (height - oy) // ws)


def pathtrace(scene, width, height, camera_params, camera_transform, max_bounces=8, samples_per_pixel=5, pathtrace_type=0,
              accum_counter=0, prev_frame=None, advanced=None, tile_params=None, out=None, num_threads=0, want_f32=False,
              store_rounding=0, falsecolor_type=None, debug_desc=None, prev_frame_f32=None):
    """One pathtrace_scene call on the CPU.  Returns (rgba16f (H,W,4) float16, counters dict[, rgb f32]).
    prev_frame_f32 ((H,W,3) float32): f32-accumulate mode, the blend reads this instead of the f16 prev_frame."""
    (ox, oy), gx, gy = dispatch_extent(width, height, tile_params)
    pc = push_constants(scene, camera_params, camera_transform, pathtrace_type, accum_counter, advanced, (ox, oy))
    if falsecolor_type is not None:
        pc.falsecolor_type = int(falsecolor_type)
        pc.pathtrace_type = 0
    if debug_desc is not None:   # get_push_constants (renderer.rs:1430-1454)
        pc.pathtrace_type = 0
        pc.flags |= {0: 1 << 4, 1: 1 << 3, 2: 1 << 5}[int(debug_desc.viz_type)]
        if debug_desc.first_hit_only:
            pc.flags |= 1 << 6
        pc.heatmap_min = float(debug_desc.heatmap_min)
        pc.heatmap_max = float(debug_desc.heatmap_max)
    if out is None:
        out = np.zeros((height, width, 4), np.float16)
    f32 = want_f32 if isinstance(want_f32, np.ndarray) else (np.zeros((height, width, 3), np.float32) if want_f32 else None)
    prev = None
    if prev_frame is not None:
        prev = np.ascontiguousarray(prev_frame, np.float16)
        assert prev.shape == (height, width, 4)
    cnt = OracleCounters()
    if num_threads <= 0:
        num_threads = globals()["num_threads"]()   # never oversubscribe the cgroup's CPU share: 256 spinning threads on 16 CPUs crawl
    p32 = None
    if prev_frame_f32 is not None:
        p32 = np.ascontiguousarray(prev_frame_f32, np.float32)
        assert p32.shape == (height, width, 3)
    rc = lib().oracle_pathtrace_f32prev(C.byref(scene.desc), C.byref(pc), max_bounces, samples_per_pixel, width, height, gx, gy,
                                        _abi.ptr(prev), _abi.ptr(out), _abi.ptr(f32), C.byref(cnt), num_threads, store_rounding,
                                        2 if debug_desc is not None else (0 if falsecolor_type is None else 1), _abi.ptr(p32))
    if rc != 0:
        raise RuntimeError("oracle_pathtrace failed")
    counters = {}
    for k, t in OracleCounters._fields_:
        v = getattr(cnt, k)
        counters[k] = int(v) if t is C.c_uint64 else [int(x) for x in v]   # 3-vectors: [extend, light-pdf, shadow]
    return (out, counters, f32) if f32 is not None else (out, counters)


def trace_rays(scene, ori, dir_, ray_epsilon=0.001, flags=0):
    ori = np.ascontiguousarray(ori, np.float32).reshape(-1, 3)
    dir_ = np.ascontiguousarray(dir_, np.float32).reshape(-1, 3)
    n = len(ori)
    hit = np.zeros(n, np.uint32)
    dst = np.zeros(n, np.float32)
    uv = np.zeros((n, 2), np.float32)
    inst = np.zeros(n, np.uint32)
    tri = np.zeros(n, np.uint32)
    rc = lib().oracle_trace_rays(C.byref(scene.desc), n, _abi.ptr(ori), _abi.ptr(dir_), ray_epsilon, flags,
                                 _abi.ptr(hit), _abi.ptr(dst), _abi.ptr(uv), _abi.ptr(inst), _abi.ptr(tri))
    assert rc == 0
    return hit, dst, uv, inst, tri


def rng_stream(global_id, accum_counter, count):
    out = np.zeros(count, np.float32)
    lib().oracle_rng_stream(global_id, accum_counter, count, _abi.ptr(out))
    return out


def bsdf_probe(mat_type, color, roughness, metallic, ior, normal, outgoing, rnl, rn):
    color = np.ascontiguousarray(color, np.float32)
    normal = np.ascontiguousarray(normal, np.float32)
    outgoing = np.ascontiguousarray(outgoing, np.float32)
    rn = np.ascontiguousarray(rn, np.float32)
    inc = np.zeros(3, np.float32)
    ev = np.zeros(3, np.float32)
    pdf = C.c_float()
    lib().oracle_bsdf_probe(int(mat_type), _abi.ptr(color), roughness, metallic, ior, _abi.ptr(normal), _abi.ptr(outgoing),
                            rnl, _abi.ptr(rn), _abi.ptr(inc), _abi.ptr(ev), C.byref(pdf))
    return inc, ev, float(pdf.value)


def effective_cpus():
    """CPUs this process may actually use: the cgroup's quota (a GPU box shows all 256 hardware threads but grants 16) and
    the affinity mask, whichever is smaller."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def num_threads():
    """OpenMP threads the oracle runs with when `num_threads=0` is passed to pathtrace."""
    return min(int(lib().oracle_num_threads()), effective_cpus())


def tonemap(src_rgba16f, dst_width, dst_height, desc=None, dst=None):
    """tonemap_and_fit_aspect (tonemapping.rs:155-224) on the CPU: (H,W,4) float16 -> (dst_height,dst_width,4) uint8.
    `desc` is an api.TonemapDesc; `dst` the previous target contents (used when desc.clear is False)."""
    from lupinpathtracer_amd import api
    desc = desc or api.TonemapDesc()
    src = np.ascontiguousarray(src_rgba16f, np.float16)
    out = np.zeros((dst_height, dst_width, 4), np.uint8) if dst is None else np.ascontiguousarray(dst, np.uint8).copy()
    vp = None
    if desc.viewport is not None:
        vp = np.array([desc.viewport.x, desc.viewport.y, desc.viewport.w, desc.viewport.h], np.float32)
    rc = lib().oracle_tonemap(_abi.ptr(src), src.shape[1], src.shape[0], _abi.ptr(out), dst_width, dst_height, _abi.ptr(vp),
                              float(desc.exposure), int(bool(desc.filmic)), int(bool(desc.srgb)), int(bool(desc.clear)))
    if rc != 0:
        raise RuntimeError("oracle_tonemap failed")
    return out
