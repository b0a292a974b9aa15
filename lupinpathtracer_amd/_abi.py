"""ctypes view of include/lupin_hip.h and the loader of liblupin_hip.so.

This is the only place the package touches the shared library.  There is no fallback: if the
library is missing or cannot be loaded, importing any compute entry point raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LUPIN_HIP_LIB overrides the path (A/B runs of kernel variants); it is still the same C ABI, never a fallback
LIB_PATH = os.environ.get("LUPIN_HIP_LIB") or os.path.join(_HERE, "liblupin_hip.so")

SENTINEL_IDX = 0xFFFFFFFF

# ---- record layouts (byte-identical to the reference's #[repr(C)] structs, renderer.rs:94-280) ----

MESH_INFO_DTYPE = np.dtype([("normals_buf_idx", "<u4"), ("texcoords_buf_idx", "<u4"), ("colors_buf_idx", "<u4")])
INSTANCE_DTYPE = np.dtype([("transpose_inverse_transform", "<f4", (3, 4)), ("mesh_idx", "<u4"), ("mat_idx", "<u4"),
                           ("_padding0", "<f4"), ("_padding1", "<f4")])
MATERIAL_DTYPE = np.dtype([("color", "<f4", 4), ("emission", "<f4", 4), ("scattering", "<f4", 4), ("mat_type", "<u4"),
                           ("roughness", "<f4"), ("metallic", "<f4"), ("ior", "<f4"), ("sc_anisotropy", "<f4"),
                           ("tr_depth", "<f4"), ("color_tex_idx", "<u4"), ("emission_tex_idx", "<u4"),
                           ("roughness_tex_idx", "<u4"), ("scattering_tex_idx", "<u4"), ("normal_tex_idx", "<u4"),
                           ("padding0", "<u4")])
ENVIRONMENT_DTYPE = np.dtype([("emission", "<f4", 3), ("emission_tex_idx", "<u4"), ("transform", "<f4", (4, 4))])
LIGHT_DTYPE = np.dtype([("instance_idx", "<u4"), ("area", "<f4")])
ALIAS_BIN_DTYPE = np.dtype([("prob", "<f4"), ("alias_threshold", "<f4"), ("alias", "<u4")])
BVH_NODE_DTYPE = np.dtype([("aabb_min", "<f4", 3), ("tri_begin_or_first_child", "<u4"), ("aabb_max", "<f4", 3),
                           ("tri_count", "<u4")])
TLAS_NODE_DTYPE = np.dtype([("aabb_min", "<f4", 3), ("left", "<u4"), ("aabb_max", "<f4", 3), ("instance_idx", "<u4"),
                            ("right", "<u4"), ("_padding0", "<f4", 3)])
assert MESH_INFO_DTYPE.itemsize == 12 and INSTANCE_DTYPE.itemsize == 64 and MATERIAL_DTYPE.itemsize == 96
assert ENVIRONMENT_DTYPE.itemsize == 80 and LIGHT_DTYPE.itemsize == 8 and ALIAS_BIN_DTYPE.itemsize == 12
assert BVH_NODE_DTYPE.itemsize == 32 and TLAS_NODE_DTYPE.itemsize == 48

TEX_RGBA8_UNORM = 0
TEX_RGBA16_FLOAT = 1

FLAG_CAMERA_ORTHO = 1 << 0
FLAG_ENVS_EMPTY = 1 << 1
FLAG_LIGHTS_EMPTY = 1 << 2
FLAG_INSTANCES_EMPTY = 1 << 7
WORKGROUP_SIZE = 4


class Mat3x4(C.Structure):
    _fields_ = [("m", (C.c_float * 3) * 4)]


class Mat4(C.Structure):
    _fields_ = [("m", (C.c_float * 4) * 4)]


class PushConstants(C.Structure):
    _fields_ = [("camera_transform", Mat4), ("camera_lens", C.c_float), ("camera_film", C.c_float),
                ("camera_aspect", C.c_float), ("camera_focus", C.c_float), ("camera_aperture", C.c_float),
                ("flags", C.c_uint32), ("id_offset", C.c_uint32 * 2), ("accum_counter", C.c_uint32),
                ("heatmap_min", C.c_float), ("heatmap_max", C.c_float), ("falsecolor_type", C.c_uint32),
                ("pathtrace_type", C.c_uint32), ("max_radiance", C.c_float), ("rng_seed", C.c_uint32),
                ("ray_epsilon", C.c_float)]


assert C.sizeof(PushConstants) == 128


class TextureDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32), ("pixels", C.c_void_p)]


class MeshDesc(C.Structure):
    _fields_ = [("verts_pos", C.c_void_p), ("num_verts", C.c_uint32), ("indices", C.c_void_p),
                ("num_indices", C.c_uint32), ("bvh_nodes", C.c_void_p), ("num_bvh_nodes", C.c_uint32)]


class VertexBufferDesc(C.Structure):
    _fields_ = [("data", C.c_void_p), ("num_verts", C.c_uint32)]


class AliasTableDesc(C.Structure):
    _fields_ = [("bins", C.c_void_p), ("num_bins", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("mesh_infos", C.c_void_p), ("meshes", C.POINTER(MeshDesc)), ("num_meshes", C.c_uint32),
                ("verts_normal_array", C.POINTER(VertexBufferDesc)), ("num_normal_buffers", C.c_uint32),
                ("verts_texcoord_array", C.POINTER(VertexBufferDesc)), ("num_texcoord_buffers", C.c_uint32),
                ("verts_color_array", C.POINTER(VertexBufferDesc)), ("num_color_buffers", C.c_uint32),
                ("instances", C.c_void_p), ("num_instances", C.c_uint32),
                ("materials", C.c_void_p), ("num_materials", C.c_uint32),
                ("textures", C.POINTER(TextureDesc)), ("num_textures", C.c_uint32),
                ("environments", C.c_void_p), ("num_environments", C.c_uint32),
                ("tlas_nodes", C.c_void_p), ("num_tlas_nodes", C.c_uint32),
                ("lights", C.c_void_p), ("num_lights", C.c_uint32),
                ("alias_tables", C.POINTER(AliasTableDesc)), ("env_alias_tables", C.POINTER(AliasTableDesc))]


class BakedPathtraceParamsC(C.Structure):
    _fields_ = [("with_runtime_checks", C.c_uint32), ("max_bounces", C.c_uint32), ("samples_per_pixel", C.c_uint32)]


class CameraParamsC(C.Structure):
    _fields_ = [("is_orthographic", C.c_uint32), ("lens", C.c_float), ("film", C.c_float), ("aspect", C.c_float),
                ("focus", C.c_float), ("aperture", C.c_float)]


class AdvancedParamsC(C.Structure):
    _fields_ = [("max_radiance", C.c_float), ("rng_seed", C.c_uint32), ("ray_epsilon", C.c_float)]


class TileParamsC(C.Structure):
    _fields_ = [("tile_size", C.c_uint32), ("tile_idx", C.c_uint32)]


class AccumulationParamsC(C.Structure):
    _fields_ = [("prev_frame", C.c_void_p), ("accum_counter", C.c_uint32)]


class PathtraceDescC(C.Structure):
    _fields_ = [("accum_params", C.POINTER(AccumulationParamsC)), ("tile_params", C.POINTER(TileParamsC)),
                ("camera_params", CameraParamsC), ("camera_transform", Mat3x4), ("force_software_bvh", C.c_uint32),
                ("advanced", AdvancedParamsC)]


class DebugVizDescC(C.Structure):
    _fields_ = [("viz_type", C.c_uint32), ("heatmap_min", C.c_float), ("heatmap_max", C.c_float), ("first_hit_only", C.c_uint32)]


class TonemapDescC(C.Structure):
    _fields_ = [("has_viewport", C.c_uint32), ("viewport_x", C.c_float), ("viewport_y", C.c_float), ("viewport_w", C.c_float),
                ("viewport_h", C.c_float), ("exposure", C.c_float), ("filmic", C.c_uint32), ("srgb", C.c_uint32), ("clear", C.c_uint32)]


class StatsC(C.Structure):
    _fields_ = [("path_bounces", C.c_uint64), ("paths", C.c_uint64), ("extend_launches", C.c_uint64),
                ("extend_ms", C.c_double), ("shade_ms", C.c_double), ("total_ms", C.c_double),
                ("node_visits", C.c_uint64 * 3), ("tri_tests", C.c_uint64 * 3), ("instance_entries", C.c_uint64 * 3),
                ("wide_node_visits", C.c_uint64 * 3), ("tracer_rounds", C.c_uint64 * 10), ("tracer_cycles", C.c_uint64 * 6), ("wide_queries", C.c_uint64), ("wide_retraced", C.c_uint64),
                ("verify_checked", C.c_uint64), ("verify_flagged", C.c_uint64), ("verify_mismatches", C.c_uint64),
                ("verify_raw_mismatches", C.c_uint64), ("verify_reasons", C.c_uint64 * 4), ("frames_in_flight", C.c_uint32),
                ("wide_traversal", C.c_uint32), ("frames_per_wavefront", C.c_uint32), ("short_stack_entries", C.c_uint32)]


class RuntimeInfoC(C.Structure):
    _fields_ = [("build_hip_version", C.c_int32), ("runtime_hip_version", C.c_int32), ("num_hip_runtimes_mapped", C.c_uint32),
                ("hip_runtime_paths", C.c_char * 1012)]


# every symbol include/lupin_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_U32 = C.c_uint32
SYMBOLS = [
    ("lupin_hip_last_error", C.c_char_p, []),
    ("lupin_hip_device_count", C.c_int, []),
    ("lupin_hip_create_context", C.c_int, [C.c_int, _PP]),
    ("lupin_hip_destroy_context", None, [_P]),
    ("lupin_hip_sync", C.c_int, [_P]),
    ("lupin_hip_set_f16_store_rounding", C.c_int, [_P, C.c_int]),
    ("lupin_hip_set_accumulation_mode", C.c_int, [_P, C.c_int]),
    ("lupin_hip_set_traversal", C.c_int, [_P, C.c_int]),
    ("lupin_hip_set_batch_frames", C.c_int, [_P, _U32]),
    ("lupin_hip_reserve_path_state", C.c_int, [_P, C.c_uint64, _U32, _U32]),
    ("lupin_hip_texture_download_rgba32f", C.c_int, [_P, _P]),
    ("lupin_hip_measure_copy_bandwidth", C.c_int, [_P, C.c_uint64, _U32, C.POINTER(C.c_double)]),
    ("lupin_hip_runtime_info", C.c_int, [C.POINTER(RuntimeInfoC)]),
    ("lupin_hip_build_pathtrace_resources", C.c_int, [_P, C.POINTER(BakedPathtraceParamsC), _PP]),
    ("lupin_hip_destroy_pathtrace_resources", None, [_P]),
    ("lupin_hip_scene_create", C.c_int, [_P, C.POINTER(SceneDesc), _PP]),
    ("lupin_hip_scene_destroy", None, [_P]),
    ("lupin_hip_texture_create", C.c_int, [_P, _U32, _U32, _PP]),
    ("lupin_hip_texture_destroy", None, [_P]),
    ("lupin_hip_texture_width", _U32, [_P]),
    ("lupin_hip_texture_height", _U32, [_P]),
    ("lupin_hip_texture_device_ptr", _P, [_P]),
    ("lupin_hip_texture_upload_rgba16f", C.c_int, [_P, _P]),
    ("lupin_hip_texture_download_rgba16f", C.c_int, [_P, _P]),
    ("lupin_hip_dbuf_create", C.c_int, [_P, _U32, _U32, _PP]),
    ("lupin_hip_dbuf_destroy", None, [_P]),
    ("lupin_hip_dbuf_front", _P, [_P]),
    ("lupin_hip_dbuf_back", _P, [_P]),
    ("lupin_hip_dbuf_copy_front_to_back", C.c_int, [_P]),
    ("lupin_hip_dbuf_flip", None, [_P]),
    ("lupin_hip_dbuf_resize", C.c_int, [_P, _U32, _U32]),
    ("lupin_hip_get_num_tiles", _U32, [_U32, _U32, _U32]),
    ("lupin_hip_pathtrace_scene", C.c_int, [_P, _P, _P, _P, _U32, C.POINTER(PathtraceDescC)]),
    ("lupin_hip_pathtrace_scene_falsecolor", C.c_int, [_P, _P, _P, _P, _U32, C.POINTER(PathtraceDescC)]),
    ("lupin_hip_pathtrace_scene_debug", C.c_int, [_P, _P, _P, _P, C.POINTER(DebugVizDescC), C.POINTER(PathtraceDescC)]),
    ("lupin_hip_pathtrace_scene_tiles", C.c_int, [_P, _P, _P, _P, _U32, C.POINTER(PathtraceDescC), _U32, _U32, _U32]),
    ("lupin_hip_stats_reset", C.c_int, [_P, C.c_int]),
    ("lupin_hip_stats_get", C.c_int, [_P, C.POINTER(StatsC)]),
    ("lupin_hip_trace_rays", C.c_int, [_P, _P, _U32, _P, _P, C.c_float, _P, _P, _P, _P, _P]),
    ("lupin_hip_collapse_bvh4", C.c_int64, [_P, _U32, _P, _U32, _P, _U32, _P, C.c_uint64, C.POINTER(C.c_uint32), _P]),
    ("lupin_hip_trace_rays_wide", C.c_int, [_P, _P, _U32, _P, _P, C.c_float, _P, _P, _P, _P, _P, _P]),
    ("lupin_hip_detmath_probe", C.c_int, [_P, C.c_int, _U32, _P, _P, _P]),
    ("lupin_hip_tonemap_and_fit_aspect", C.c_int, [_P, _P, _P, _U32, _U32, C.POINTER(TonemapDescC)]),
    ("lupin_hip_lbvh_depth", _U32, [_U32]),
    ("lupin_hip_lbvh_node_count", C.c_uint64, [_U32]),
    ("lupin_hip_build_bvh_device", C.c_int64, [_P, _P, _U32, _P, _U32, _P, C.c_uint64]),
    ("lupin_hip_build_bvh_sah_device", C.c_int64, [_P, _P, _U32, _P, _U32, _P, C.c_uint64]),
    ("lupin_hip_pack_tiles", C.c_int, [_P, _P, _U32, _U32, _U32, _P, C.POINTER(C.c_uint64)]),
    ("lupin_hip_unpack_tiles", C.c_int, [_P, _P, _U32, _U32, _U32, _P]),
    ("lupin_hip_unpack_gathered_tiles", C.c_int, [_P, _P, _U32, _U32, _U32, _P, C.c_uint64]),
    ("lupin_hip_packed_tile_pixels", C.c_uint64, [_U32, _U32, _U32, _U32, _U32]),
    ("lupin_hip_comm_get_unique_id", C.c_int, [_P]),
    ("lupin_hip_comm_init_rank", C.c_int, [_P, _P, _U32, _U32, _PP]),
    ("lupin_hip_comm_init_all", C.c_int, [_PP, _U32, _PP]),
    ("lupin_hip_comm_from_nccl", C.c_int, [_P, _P, _U32, _U32, _PP]),
    ("lupin_hip_comm_destroy", None, [_P]),
    ("lupin_hip_comm_rank", _U32, [_P]),
    ("lupin_hip_comm_world", _U32, [_P]),
    ("lupin_hip_gather_framebuffer", C.c_int, [_P, _P, _U32]),
    ("lupin_hip_gather_framebuffer_all", C.c_int, [_PP, _PP, _U32, _U32]),
    ("lupin_hip_gather_framebuffer_to", C.c_int, [_P, _P, _U32, _U32]),
    ("lupin_hip_comm_allreduce_f64", C.c_int, [_P, _P, _U32, _U32]),
    ("lupin_hip_comm_barrier", C.c_int, [_P]),
    ("lupin_build_bvh", C.c_int64, [_P, _U32, _P, _U32, _P, C.c_uint64]),
    ("lupin_build_tlas", C.c_int64, [_P, _U32, _P, _U32, _P]),
    ("lupin_build_alias_table", C.c_int64, [_P, C.c_uint64, _P]),
    ("lupin_mesh_light_weights", C.c_float, [_P, _P, _U32, _P]),
    ("lupin_env_light_weights", None, [_P, _U32, _U32, _P, _P]),
    ("lupin_mat3x4_inverse", None, [C.POINTER(Mat3x4), C.POINTER(Mat3x4)]),
]

_lib = None


class LupinError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"lupin_hip error {code}: {message}")
        self.code = code


def lib():
    """Load liblupin_hip.so once.  Raises if it has not been built (`python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build the HIP extension first (make -C lupinpathtracer_amd/csrc); "
                              "there is no CPU fallback")
        handle = C.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(handle, name)   # AttributeError if the library does not export it
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(code):
    if code != 0:
        raise LupinError(code, lib().lupin_hip_last_error().decode("utf-8", "replace"))


def ptr(a):
    """void* of a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)
