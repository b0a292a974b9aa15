"""Host-side mirror of the reference's call surface for the hot path (crate `lupin_pt`, `lp::*`).

Same names, argument meaning and error behaviour as lupin/src/renderer.rs, wgpu_utils.rs and
data_structures.rs, on top of the C ABI in include/lupin_hip.h:

    lp::build_pathtrace_resources        renderer.rs:470      -> build_pathtrace_resources
    lp::pathtrace_scene                  renderer.rs:768      -> pathtrace_scene
    lp::PathtraceDesc / AccumulationParams / TileParams / CameraParams / AdvancedParams / PathtraceType
                                         renderer.rs:644-766  -> dataclasses below
    lp::get_num_tiles                    renderer.rs:675      -> get_num_tiles
    lp::DoubleBufferedTexture            wgpu_utils.rs:279    -> DoubleBufferedTexture
    lp::SceneCPU / validate_scene / build_accel_structures_and_upload
                                         renderer.rs:62-76, data_structures.rs:696-928

Where the reference panics (assert!/panic!), these raise (`LupinError` for ABI errors,
`AssertionError`/`ValueError` for host-side validation).
"""
import ctypes as C
import os
import enum
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _abi
from ._abi import (ALIAS_BIN_DTYPE, BVH_NODE_DTYPE, ENVIRONMENT_DTYPE, INSTANCE_DTYPE, LIGHT_DTYPE, MATERIAL_DTYPE,
                   MESH_INFO_DTYPE, SENTINEL_IDX, TLAS_NODE_DTYPE, LupinError, check, lib, ptr)

WORKGROUP_SIZE = _abi.WORKGROUP_SIZE


class PathtraceType(enum.IntEnum):  # renderer.rs:711-729
    Standard = 0
    MIS = 1
    Naive = 2
    Direct = 3


class FalsecolorType(enum.IntEnum):  # renderer.rs:843-870
    Albedo = 0
    Normals = 1
    NormalsUnsigned = 2
    FrontFacing = 3
    Emission = 4
    Roughness = 5
    Metallic = 6
    Opacity = 7
    MatType = 8
    IsDelta = 9
    Instance = 10
    Tri = 11


class DebugVizType(enum.IntEnum):  # renderer.rs:950-956
    BVHAABBChecks = 0
    BVHTriChecks = 1
    NumBounces = 2


@dataclass
class DebugVizDesc:  # renderer.rs:958-964
    viz_type: int = DebugVizType.BVHAABBChecks
    heatmap_min: float = 0.0
    heatmap_max: float = 100.0
    first_hit_only: bool = False


@dataclass
class Viewport:  # tonemapping.rs:144-151
    x: float = 0.0
    y: float = 0.0
    w: float = 0.0
    h: float = 0.0


@dataclass
class TonemapDesc:  # tonemapping.rs:106-132
    viewport: Optional[Viewport] = None
    exposure: float = 0.0
    filmic: bool = False
    srgb: bool = True
    clear: bool = True


class MaterialType(enum.IntEnum):  # renderer.rs:126-139
    Matte = 0
    Glossy = 1
    Reflective = 2
    Transparent = 3
    Refractive = 4
    Subsurface = 5
    Volumetric = 6
    GltfPbr = 7


@dataclass
class BakedPathtraceParams:  # renderer.rs:451-468
    with_runtime_checks: bool = False
    max_bounces: int = 8
    samples_per_pixel: int = 5


@dataclass
class CameraParams:  # renderer.rs:683-708
    is_orthographic: bool = False
    lens: float = 0.050
    film: float = 0.036
    aspect: float = 1.500
    focus: float = 10000.0
    aperture: float = 0.0


@dataclass
class AdvancedParams:  # renderer.rs:731-749
    max_radiance: float = 100.0
    rng_seed: int = 0
    ray_epsilon: float = 0.001


@dataclass
class TileParams:  # renderer.rs:651-670
    tile_size: int = 100
    tile_idx: int = 0


@dataclass
class AccumulationParams:  # renderer.rs:644-649
    prev_frame: "Texture"
    accum_counter: int


def identity_mat3x4():
    """Mat3x4::IDENTITY (base.rs:651-660): 4 columns x 3 rows."""
    return np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 0, 0]], dtype=np.float32)


@dataclass
class PathtraceDesc:  # renderer.rs:751-766
    accum_params: Optional[AccumulationParams] = None
    tile_params: Optional[TileParams] = None
    camera_params: CameraParams = field(default_factory=CameraParams)
    camera_transform: np.ndarray = field(default_factory=identity_mat3x4)
    force_software_bvh: bool = False
    advanced: AdvancedParams = field(default_factory=AdvancedParams)


def get_num_tiles(tile_size, width, height):
    """renderer.rs:675-681"""
    return int(lib().lupin_hip_get_num_tiles(tile_size, width, height))


# ------------------------------------------------------------------------------------------------
# Device objects
# ------------------------------------------------------------------------------------------------

class Context:
    """One HIP device + stream (the reference's wgpu Device/Queue pair)."""

    def __init__(self, device_ordinal=0):
        h = C.c_void_p()
        check(lib().lupin_hip_create_context(device_ordinal, C.byref(h)))
        self.handle = h
        self.device_ordinal = device_ordinal

    def sync(self):
        check(lib().lupin_hip_sync(self.handle))

    def set_f16_store_rounding(self, mode):
        """0 = toward zero (what the reference's goldens show; default), 1 = nearest even."""
        check(lib().lupin_hip_set_f16_store_rounding(self.handle, int(mode)))

    def stats_reset(self, mode=0):
        """mode: 0 / False plain counters, 1 / True per-kernel hipEvent timing, 2 work counters of the tracing kernels."""
        check(lib().lupin_hip_stats_reset(self.handle, int(mode)))

    def stats(self):
        s = _abi.StatsC()
        check(lib().lupin_hip_stats_get(self.handle, C.byref(s)))
        def conv(v):
            return [conv(x) for x in v] if hasattr(v, "__len__") else v
        return {k: conv(getattr(s, k)) for k, _ in _abi.StatsC._fields_}

    def reserve_path_state(self, pixels, max_bounces, samples_per_pixel):
        """Allocate the path state of every frame in flight now instead of at each lane's first pathtrace call."""
        check(lib().lupin_hip_reserve_path_state(self.handle, int(pixels), int(max_bounces), int(samples_per_pixel)))

    def set_batch_frames(self, frames):
        """Frames per wavefront: how many consecutive, chained pathtrace_scene calls run as one wavefront (1..16;
        0 = by dispatch size, the default: 16 up to 4 M pixels, 8 above)."""
        check(lib().lupin_hip_set_batch_frames(self.handle, int(frames)))

    def set_traversal(self, mode):
        """"binary" (default: the reference's visiting order) or "wide" (four-wide hierarchy + exactness certificate + re-trace)."""
        check(lib().lupin_hip_set_traversal(self.handle, {"binary": 0, "wide": 1}[mode]))

    def set_accumulation_mode(self, mode):
        """0 = f16 running average (reference-faithful, default), 1 = f32 accumulator per texture (pathtracer.wgsl:275-289)."""
        check(lib().lupin_hip_set_accumulation_mode(self.handle, int(mode)))

    def measure_copy_bandwidth(self, nbytes=1 << 30, reps=10):
        """GB/s (read + written) of a device-to-device copy: the measured HBM peak."""
        out = C.c_double()
        check(lib().lupin_hip_measure_copy_bandwidth(self.handle, nbytes, reps, C.byref(out)))
        return float(out.value)

    def close(self):
        if self.handle:
            lib().lupin_hip_destroy_context(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_count():
    return int(lib().lupin_hip_device_count())


def runtime_info():
    """HIP version the library was built against / runs on, and every libamdhip64 mapped into the process."""
    r = _abi.RuntimeInfoC()
    check(lib().lupin_hip_runtime_info(C.byref(r)))
    paths = r.hip_runtime_paths.decode("utf-8", "replace")
    return {"build_hip_version": int(r.build_hip_version), "runtime_hip_version": int(r.runtime_hip_version),
            "num_hip_runtimes_mapped": int(r.num_hip_runtimes_mapped), "hip_runtime_paths": [p for p in paths.split(";") if p]}


class PathtraceResources:
    def __init__(self, ctx, params):
        self.ctx = ctx
        self.params = params
        h = C.c_void_p()
        c = _abi.BakedPathtraceParamsC(1 if params.with_runtime_checks else 0, params.max_bounces, params.samples_per_pixel)
        check(lib().lupin_hip_build_pathtrace_resources(ctx.handle, C.byref(c), C.byref(h)))
        self.handle = h

    def __del__(self):
        try:
            if self.handle and self.ctx.handle:   # a closed context has already released the device
                lib().lupin_hip_destroy_pathtrace_resources(self.handle)
            self.handle = None
        except Exception:
            pass


def build_pathtrace_resources(ctx, baked_pathtrace_params):
    """lp::build_pathtrace_resources (renderer.rs:470): max_bounces / samples_per_pixel are baked."""
    return PathtraceResources(ctx, baked_pathtrace_params)


class Texture:
    """An Rgba16Float render target (row 0 = top)."""

    def __init__(self, ctx, width, height, _handle=None):
        self.ctx = ctx
        self._owned = _handle is None
        if _handle is None:
            h = C.c_void_p()
            check(lib().lupin_hip_texture_create(ctx.handle, width, height, C.byref(h)))
            _handle = h
        self.handle = _handle
        self.width = int(lib().lupin_hip_texture_width(self.handle))
        self.height = int(lib().lupin_hip_texture_height(self.handle))

    def format(self):
        return "Rgba16Float"

    def device_ptr(self):
        return int(lib().lupin_hip_texture_device_ptr(self.handle) or 0)

    def upload(self, rgba16f):
        a = np.ascontiguousarray(rgba16f, dtype=np.float16).reshape(self.height, self.width, 4)
        check(lib().lupin_hip_texture_upload_rgba16f(self.handle, ptr(a)))

    def download(self):
        """(H, W, 4) float16; synchronises (loader.rs download_texture)."""
        out = np.empty((self.height, self.width, 4), dtype=np.float16)
        check(lib().lupin_hip_texture_download_rgba16f(self.handle, ptr(out)))
        return out

    def download_f32(self):
        """(H, W, 4) float32 of the f32 accumulator (frames rendered with Context.set_accumulation_mode(1))."""
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        check(lib().lupin_hip_texture_download_rgba32f(self.handle, ptr(out)))
        return out

    def __del__(self):
        try:
            if self._owned and self.handle and self.ctx.handle:
                lib().lupin_hip_texture_destroy(self.handle)
            self.handle = None
        except Exception:
            pass


class DoubleBufferedTexture:
    """lp::DoubleBufferedTexture (wgpu_utils.rs:279-348)."""

    def __init__(self, ctx, width, height):
        self.ctx = ctx
        h = C.c_void_p()
        check(lib().lupin_hip_dbuf_create(ctx.handle, width, height, C.byref(h)))
        self.handle = h

    @classmethod
    def create(cls, ctx, width, height):
        return cls(ctx, width, height)

    def front(self):
        return Texture(self.ctx, 0, 0, _handle=C.c_void_p(lib().lupin_hip_dbuf_front(self.handle)))

    def back(self):
        return Texture(self.ctx, 0, 0, _handle=C.c_void_p(lib().lupin_hip_dbuf_back(self.handle)))

    def copy_front_to_back(self):
        check(lib().lupin_hip_dbuf_copy_front_to_back(self.handle))

    def flip(self):
        lib().lupin_hip_dbuf_flip(self.handle)

    def resize(self, width, height):
        check(lib().lupin_hip_dbuf_resize(self.handle, width, height))

    def __del__(self):
        try:
            if self.handle and self.ctx.handle:
                lib().lupin_hip_dbuf_destroy(self.handle)
            self.handle = None
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# Scene: CPU description, preprocessing, upload
# ------------------------------------------------------------------------------------------------

@dataclass
class TextureCPU:
    """One texture as the loader hands it over: Rgba8Unorm (H,W,4 uint8) or Rgba16Float (H,W,4 float16)."""
    pixels: np.ndarray

    @property
    def format(self):
        return _abi.TEX_RGBA8_UNORM if self.pixels.dtype == np.uint8 else _abi.TEX_RGBA16_FLOAT


@dataclass
class EnvMapInfo:  # data_structures.rs:13-19: f32 texels used for the env alias table
    data: np.ndarray   # (H, W, 4) float32
    width: int
    height: int


@dataclass
class SceneCPU:  # renderer.rs:62-76
    mesh_infos: np.ndarray = field(default_factory=lambda: np.zeros(0, MESH_INFO_DTYPE))
    verts_pos_array: List[np.ndarray] = field(default_factory=list)        # (n,4) f32
    verts_normal_array: List[np.ndarray] = field(default_factory=list)     # (n,4) f32
    verts_texcoord_array: List[np.ndarray] = field(default_factory=list)   # (n,2) f32
    verts_color_array: List[np.ndarray] = field(default_factory=list)      # (n,4) f32
    indices_array: List[np.ndarray] = field(default_factory=list)          # (3t,) u32
    instances: np.ndarray = field(default_factory=lambda: np.zeros(0, INSTANCE_DTYPE))
    materials: np.ndarray = field(default_factory=lambda: np.zeros(0, MATERIAL_DTYPE))
    environments: np.ndarray = field(default_factory=lambda: np.zeros(0, ENVIRONMENT_DTYPE))


def default_material():
    """Material::default() (renderer.rs:163-185)."""
    m = np.zeros((), MATERIAL_DTYPE)
    m["color"] = (0.0, 0.0, 0.0, 1.0)
    m["ior"] = 1.5
    m["tr_depth"] = 0.01
    for k in ("color_tex_idx", "emission_tex_idx", "roughness_tex_idx", "scattering_tex_idx", "normal_tex_idx"):
        m[k] = SENTINEL_IDX
    return m


def default_mesh_info():
    return np.array((SENTINEL_IDX, SENTINEL_IDX, SENTINEL_IDX), MESH_INFO_DTYPE)


def default_instance():
    """Instance::default(): identity world->local, mesh 0, material 0."""
    i = np.zeros((), INSTANCE_DTYPE)
    i["transpose_inverse_transform"] = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], np.float32)
    return i


def default_environment():
    e = np.zeros((), ENVIRONMENT_DTYPE)
    e["emission_tex_idx"] = SENTINEL_IDX
    e["transform"] = np.eye(4, dtype=np.float32)
    return e


def mat3x4_inverse(m):
    """Mat3x4::inverse (base.rs:708-722); m is (4,3) column-major."""
    a = _abi.Mat3x4()
    b = _abi.Mat3x4()
    m = np.asarray(m, np.float32)
    for c in range(4):
        for r in range(3):
            a.m[c][r] = float(m[c][r])
    lib().lupin_mat3x4_inverse(C.byref(a), C.byref(b))
    return np.array([[b.m[c][r] for r in range(3)] for c in range(4)], np.float32)


def instance_from_transform(local_to_world, mesh_idx, mat_idx):
    """Instance whose transpose_inverse_transform is transpose(inverse(local_to_world)) (loader.rs:653-654)."""
    inv = mat3x4_inverse(local_to_world)      # (4 cols, 3 rows)
    inst = default_instance()
    inst["transpose_inverse_transform"] = inv.T.copy()   # Mat3x4::transpose -> 3 x 4
    inst["mesh_idx"] = mesh_idx
    inst["mat_idx"] = mat_idx
    return inst


def validate_scene(scene: SceneCPU, num_textures: int, num_samplers: int):
    """lp::validate_scene (data_structures.rs:876-928)."""
    assert len(scene.verts_pos_array) == len(scene.mesh_infos)
    assert num_textures == num_samplers
    for i, info in enumerate(scene.mesh_infos):
        for key, arr in (("normals_buf_idx", scene.verts_normal_array), ("texcoords_buf_idx", scene.verts_texcoord_array),
                         ("colors_buf_idx", scene.verts_color_array)):
            idx = int(info[key])
            if idx != SENTINEL_IDX:
                assert idx < len(arr)
                assert len(arr[idx]) == len(scene.verts_pos_array[i])
    for i, indices in enumerate(scene.indices_array):
        if len(indices):
            assert int(indices.max()) < len(scene.verts_pos_array[i])
    for inst in scene.instances:
        assert int(inst["mesh_idx"]) < len(scene.mesh_infos)
        assert int(inst["mat_idx"]) < len(scene.materials)
    for mat in scene.materials:
        for k in ("color_tex_idx", "emission_tex_idx", "roughness_tex_idx", "scattering_tex_idx", "normal_tex_idx"):
            assert int(mat[k]) < num_textures or int(mat[k]) == SENTINEL_IDX
    for env in scene.environments:
        assert (env["emission"] >= 0).all()
        assert int(env["emission_tex_idx"]) < num_textures or int(env["emission_tex_idx"]) == SENTINEL_IDX


def build_bvh(verts_pos, indices):
    """lp::build_bvh (data_structures.rs:196-235): returns (nodes, reordered indices)."""
    verts = np.ascontiguousarray(verts_pos, np.float32).reshape(-1, 4)
    idx = np.ascontiguousarray(indices, np.uint32).copy()
    count = lib().lupin_build_bvh(ptr(verts), len(verts), ptr(idx), len(idx), None, 0)
    if count < 0:
        raise ValueError("build_bvh: invalid input")
    nodes = np.zeros(count, BVH_NODE_DTYPE)
    n2 = lib().lupin_build_bvh(ptr(verts), len(verts), ptr(idx), len(idx), ptr(nodes), count)
    assert n2 == count
    return nodes, idx


def build_bvh_device(ctx, verts_pos, indices):
    """Device BLAS builder (csrc/lbvh.hip): Morton-sorted complete binary tree in the reference's BvhNode format.
    Returns (nodes, reordered indices) like build_bvh; needs a GPU context."""
    verts = np.ascontiguousarray(verts_pos, np.float32).reshape(-1, 4)
    idx = np.ascontiguousarray(indices, np.uint32).copy()
    count = int(lib().lupin_hip_lbvh_node_count(len(idx) // 3))
    nodes = np.zeros(count, BVH_NODE_DTYPE)
    n2 = lib().lupin_hip_build_bvh_device(ctx.handle, ptr(verts), len(verts), ptr(idx), len(idx), ptr(nodes), count)
    if n2 < 0:
        check(int(n2))
    assert n2 == count
    return nodes, idx


def build_bvh_sah_device(ctx, verts_pos, indices):
    """lp::build_bvh (data_structures.rs:196-235) run on the device (csrc/sahbvh.hip): the same tree as build_bvh -- same
    boxes, split planes and triangle set per node -- numbered level by level.  Returns (nodes, reordered indices)."""
    verts = np.ascontiguousarray(verts_pos, np.float32).reshape(-1, 4)
    idx = np.ascontiguousarray(indices, np.uint32).copy()
    cap = max(2 * (len(idx) // 3) - 1, 1)
    nodes = np.zeros(cap, BVH_NODE_DTYPE)
    n = lib().lupin_hip_build_bvh_sah_device(ctx.handle, ptr(verts), len(verts), ptr(idx), len(idx), ptr(nodes), cap)
    if n < 0:
        check(int(n))
    return nodes[:n].copy(), idx


def build_tlas(instances, model_aabbs):
    """lp::build_tlas (data_structures.rs:545-641). model_aabbs: (num_meshes, 6)."""
    inst = np.ascontiguousarray(instances)
    ab = np.ascontiguousarray(model_aabbs, np.float32).reshape(-1, 6)
    if len(inst) == 0 or len(ab) == 0:
        return np.zeros(0, TLAS_NODE_DTYPE)
    out = np.zeros(2 * len(inst), TLAS_NODE_DTYPE)
    n = lib().lupin_build_tlas(ptr(inst), len(inst), ptr(ab), len(ab), ptr(out))
    if n < 0:
        raise ValueError("build_tlas: invalid input")
    return out[:n]


def build_alias_table(weights):
    """lp::build_alias_table (data_structures.rs:116-193)."""
    w = np.ascontiguousarray(weights, np.float32)
    out = np.zeros(len(w), ALIAS_BIN_DTYPE)
    n = lib().lupin_build_alias_table(ptr(w), len(w), ptr(out))
    if n < 0:
        raise ValueError("build_alias_table: invalid input")
    return out[:n]


def build_lights(scene: SceneCPU, envs_info: List[EnvMapInfo]):
    """lp::build_lights (data_structures.rs:20-113): alias tables use the ORIGINAL triangle order."""
    assert len(scene.environments) == len(envs_info), "Mismatching sizes for environment data!"
    lights, alias_tables, env_alias_tables = [], [], []
    for i, inst in enumerate(scene.instances):
        mat = scene.materials[int(inst["mat_idx"])]
        verts = scene.verts_pos_array[int(inst["mesh_idx"])]
        idx = scene.indices_array[int(inst["mesh_idx"])]
        if not np.any(mat["emission"] != 0.0):
            continue
        if len(idx) == 0:
            continue
        verts = np.ascontiguousarray(verts, np.float32)
        idx = np.ascontiguousarray(idx, np.uint32)
        weights = np.zeros(len(idx) // 3, np.float32)
        total = float(lib().lupin_mesh_light_weights(ptr(verts), ptr(idx), len(idx), ptr(weights)))
        if total <= 0.0:
            continue
        table = build_alias_table(weights)
        assert len(table) > 0
        lights.append((i, total))
        alias_tables.append(table)
    for i, env in enumerate(scene.environments):
        info = envs_info[i]
        tex = np.ascontiguousarray(info.data, np.float32).reshape(info.height, info.width, 4)
        if os.environ.get("LUPIN_EXPERIMENT_ENV_F16_WEIGHTS") == "1":   # tools/env_residual.py: weights from the f16 texels the shader samples
            tex = np.ascontiguousarray(tex.astype(np.float16).astype(np.float32))
        scale = np.ascontiguousarray(env["emission"], np.float32)
        weights = np.zeros(info.width * info.height, np.float32)
        lib().lupin_env_light_weights(ptr(tex), info.width, info.height, ptr(scale), ptr(weights))
        table = build_alias_table(weights)
        assert len(table) > 0
        env_alias_tables.append(table)
    return np.array(lights, LIGHT_DTYPE), alias_tables, env_alias_tables


class Scene:
    """lp::Scene in its software-BVH configuration (renderer.rs:17-60): the prepared host arrays, the
    C descriptor that views them, and (when a context is given) the uploaded device scene."""

    def __init__(self):
        self.desc = None
        self.handle = None
        self.ctx = None
        self._keep = []
        self.envs_empty = True
        self.lights_empty = True
        self.instances_empty = True
        self.stats = {}

    def __del__(self):
        try:
            if self.handle and self.ctx is not None and self.ctx.handle:
                lib().lupin_hip_scene_destroy(self.handle)
            self.handle = None
        except Exception:
            pass


def _array_of(struct, items):
    arr = (struct * max(1, len(items)))()
    for i, it in enumerate(items):
        arr[i] = it
    return arr


def build_accel_structures_and_upload(ctx, scene: SceneCPU, textures: List[TextureCPU], envs_info: List[EnvMapInfo],
                                      build_sw_and_hw: bool = True, blas_builder: str = "sah") -> Scene:
    """lp::build_accel_structures_and_upload (data_structures.rs:696-872), software-BVH pipeline.

    ctx may be None: the host-side preprocessing still runs and `Scene.desc` is usable (CPU-only
    tests feed it to the oracle); nothing is uploaded then.
    blas_builder: "sah" = the reference's CPU builder (lupin_build_bvh), "sah_device" = the same tree built on the GPU
    (build_bvh_sah_device; meshes with at least 64 triangles), "lbvh" = the Morton-order device builder
    (build_bvh_device; meshes with at least 64 triangles, smaller ones keep the SAH builder); a callable
    (verts (N,4), indices) -> (nodes, reordered indices) plugs in any other builder that emits the reference's node format.
    """
    if not callable(blas_builder) and blas_builder not in ("sah", "sah_device", "lbvh"):
        raise ValueError("blas_builder must be 'sah', 'sah_device', 'lbvh' or a callable (verts, indices) -> (nodes, reordered indices)")
    if blas_builder in ("lbvh", "sah_device") and ctx is None:
        raise LupinError(_abi_code("LUPIN_ERR_NO_DEVICE"), "the device BLAS builder needs a GPU context")
    out = Scene()
    keep = out._keep

    lights, alias_tables, env_alias_tables = build_lights(scene, envs_info)

    # per mesh: BLAS over a CLONE of the indices; the reordered clone is what the path reads (:724-730)
    mesh_descs, model_aabbs = [], []
    total_tris = 0
    for verts, indices in zip(scene.verts_pos_array, scene.indices_array):
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 4)
        if callable(blas_builder):
            nodes, reordered = blas_builder(v, indices)
            nodes, reordered = np.ascontiguousarray(nodes, BVH_NODE_DTYPE), np.ascontiguousarray(reordered, np.uint32)
        elif blas_builder == "lbvh" and len(indices) >= 3 * 64:
            nodes, reordered = build_bvh_device(ctx, v, indices)
        elif blas_builder == "sah_device" and len(indices) >= 3 * 64:
            nodes, reordered = build_bvh_sah_device(ctx, v, indices)
        else:
            nodes, reordered = build_bvh(v, indices)
        keep += [v, nodes, reordered]
        mesh_descs.append(_abi.MeshDesc(ptr(v), len(v), ptr(reordered), len(reordered), ptr(nodes), len(nodes)))
        total_tris += len(reordered) // 3
        if len(v):
            model_aabbs.append(np.concatenate([v[:, :3].min(axis=0), v[:, :3].max(axis=0)]))
        else:   # Aabb::neutral()
            fm = np.finfo(np.float32).max
            model_aabbs.append(np.array([fm, fm, fm, -fm, -fm, -fm], np.float32))
    model_aabbs = np.array(model_aabbs, np.float32).reshape(-1, 6)
    instances = np.ascontiguousarray(scene.instances)
    tlas = build_tlas(instances, model_aabbs)

    def vbufs(arrs, comps):
        descs = []
        for a in arrs:
            a = np.ascontiguousarray(a, np.float32).reshape(-1, comps)
            keep.append(a)
            descs.append(_abi.VertexBufferDesc(ptr(a), len(a)))
        return descs

    normal_descs = vbufs(scene.verts_normal_array, 4)
    uv_descs = vbufs(scene.verts_texcoord_array, 2)
    color_descs = vbufs(scene.verts_color_array, 4)

    tex_descs = []
    for t in textures:
        px = np.ascontiguousarray(t.pixels)
        assert px.ndim == 3 and px.shape[2] == 4 and px.dtype in (np.uint8, np.float16)
        keep.append(px)
        tex_descs.append(_abi.TextureDesc(px.shape[1], px.shape[0], t.format, ptr(px)))

    alias_descs = [_abi.AliasTableDesc(ptr(t), len(t)) for t in alias_tables]
    env_alias_descs = [_abi.AliasTableDesc(ptr(t), len(t)) for t in env_alias_tables]
    keep += alias_tables + env_alias_tables

    mesh_infos = np.ascontiguousarray(scene.mesh_infos)
    materials = np.ascontiguousarray(scene.materials)
    environments = np.ascontiguousarray(scene.environments)
    keep += [mesh_infos, materials, environments, instances, tlas, lights]

    c_meshes = _array_of(_abi.MeshDesc, mesh_descs)
    c_normals = _array_of(_abi.VertexBufferDesc, normal_descs)
    c_uvs = _array_of(_abi.VertexBufferDesc, uv_descs)
    c_colors = _array_of(_abi.VertexBufferDesc, color_descs)
    c_tex = _array_of(_abi.TextureDesc, tex_descs)
    c_alias = _array_of(_abi.AliasTableDesc, alias_descs)
    c_env_alias = _array_of(_abi.AliasTableDesc, env_alias_descs)
    keep += [c_meshes, c_normals, c_uvs, c_colors, c_tex, c_alias, c_env_alias]

    d = _abi.SceneDesc()
    d.mesh_infos = ptr(mesh_infos) if len(mesh_infos) else None
    d.meshes = c_meshes
    d.num_meshes = len(mesh_descs)
    d.verts_normal_array = c_normals
    d.num_normal_buffers = len(normal_descs)
    d.verts_texcoord_array = c_uvs
    d.num_texcoord_buffers = len(uv_descs)
    d.verts_color_array = c_colors
    d.num_color_buffers = len(color_descs)
    d.instances = ptr(instances) if len(instances) else None
    d.num_instances = len(instances)
    d.materials = ptr(materials) if len(materials) else None
    d.num_materials = len(materials)
    d.textures = c_tex
    d.num_textures = len(tex_descs)
    d.environments = ptr(environments) if len(environments) else None
    d.num_environments = len(environments)
    d.tlas_nodes = ptr(tlas) if len(tlas) else None
    d.num_tlas_nodes = len(tlas)
    d.lights = ptr(lights) if len(lights) else None
    d.num_lights = len(lights)
    d.alias_tables = c_alias
    d.env_alias_tables = c_env_alias
    out.desc = d
    out.envs_empty = len(envs_info) == 0
    out.lights_empty = len(lights) == 0
    out.instances_empty = len(instances) == 0
    out.tlas = tlas
    out.lights = lights
    out.alias_tables = alias_tables
    out.env_alias_tables = env_alias_tables
    out.stats = {"total_tri_count": total_tris, "instances": len(instances), "materials": len(materials),
                 "lights": len(lights), "textures": len(tex_descs)}   # get_scene_stats (data_structures.rs:940-953)

    if ctx is not None:
        h = C.c_void_p()
        check(lib().lupin_hip_scene_create(ctx.handle, C.byref(d), C.byref(h)))
        out.handle = h
        out.ctx = ctx
    return out


def scene_flags(scene: Scene, camera_params: CameraParams):
    """The flag word get_push_constants sets (renderer.rs:1457-1471)."""
    f = 0
    if camera_params.is_orthographic:
        f |= _abi.FLAG_CAMERA_ORTHO
    if scene.envs_empty:
        f |= _abi.FLAG_ENVS_EMPTY
    if scene.lights_empty:
        f |= _abi.FLAG_LIGHTS_EMPTY
    if scene.instances_empty:
        f |= _abi.FLAG_INSTANCES_EMPTY
    return f


def _desc_to_c(desc: PathtraceDesc, keep):
    c = _abi.PathtraceDescC()
    if desc.accum_params is not None:
        ap = _abi.AccumulationParamsC(desc.accum_params.prev_frame.handle, desc.accum_params.accum_counter)
        keep.append(ap)
        c.accum_params = C.pointer(ap)
    if desc.tile_params is not None:
        tp = _abi.TileParamsC(desc.tile_params.tile_size, desc.tile_params.tile_idx)
        keep.append(tp)
        c.tile_params = C.pointer(tp)
    cp = desc.camera_params
    c.camera_params = _abi.CameraParamsC(1 if cp.is_orthographic else 0, cp.lens, cp.film, cp.aspect, cp.focus, cp.aperture)
    m = np.asarray(desc.camera_transform, np.float32).reshape(4, 3)
    for col in range(4):
        for row in range(3):
            c.camera_transform.m[col][row] = float(m[col][row])
    c.force_software_bvh = 1 if desc.force_software_bvh else 0
    c.advanced = _abi.AdvancedParamsC(desc.advanced.max_radiance, desc.advanced.rng_seed, desc.advanced.ray_epsilon)
    return c


def pathtrace_scene(ctx, resources, scene, render_target, pathtrace_type, desc):
    """lp::pathtrace_scene (renderer.rs:768-842): enqueue one accumulation frame (or one tile of it)."""
    assert render_target.format() == "Rgba16Float"
    if scene.handle is None:
        raise LupinError(_abi_code("LUPIN_ERR_NO_DEVICE"), "scene was built without a device context; there is no CPU fallback")
    keep = []
    c = _desc_to_c(desc, keep)
    check(lib().lupin_hip_pathtrace_scene(ctx.handle, resources.handle, scene.handle, render_target.handle,
                                          int(pathtrace_type), C.byref(c)))


def pathtrace_scene_falsecolor(ctx, resources, scene, render_target, falsecolor_type, desc):
    """lp::pathtrace_scene_falsecolor (renderer.rs:872-948)."""
    assert render_target.format() == "Rgba16Float"
    if scene.handle is None:
        raise LupinError(_abi_code("LUPIN_ERR_NO_DEVICE"), "scene was built without a device context; there is no CPU fallback")
    keep = []
    c = _desc_to_c(desc, keep)
    check(lib().lupin_hip_pathtrace_scene_falsecolor(ctx.handle, resources.handle, scene.handle, render_target.handle,
                                                     int(falsecolor_type), C.byref(c)))


def pathtrace_scene_debug(ctx, resources, scene, render_target, debug_desc, desc):
    """lp::pathtrace_scene_debug (renderer.rs:966-1041): BVH-cost / bounce-count heat maps."""
    assert render_target.format() == "Rgba16Float"
    if scene.handle is None:
        raise LupinError(_abi_code("LUPIN_ERR_NO_DEVICE"), "scene was built without a device context; there is no CPU fallback")
    keep = []
    c = _desc_to_c(desc, keep)
    dd = _abi.DebugVizDescC(int(debug_desc.viz_type), float(debug_desc.heatmap_min), float(debug_desc.heatmap_max),
                            1 if debug_desc.first_hit_only else 0)
    check(lib().lupin_hip_pathtrace_scene_debug(ctx.handle, resources.handle, scene.handle, render_target.handle,
                                                C.byref(dd), C.byref(c)))


def tonemap_and_fit_aspect(ctx, src, dst_width, dst_height, desc=None, dst=None):
    """lp::tonemap_and_fit_aspect (tonemapping.rs:155-224): `src` Texture -> (dst_height, dst_width, 4) uint8 (Rgba8Unorm).
    `dst` = previous target contents, used when desc.clear is False."""
    desc = desc or TonemapDesc()
    out = np.zeros((dst_height, dst_width, 4), np.uint8) if dst is None else np.ascontiguousarray(dst, np.uint8).copy()
    assert out.shape == (dst_height, dst_width, 4)
    vp = desc.viewport
    c = _abi.TonemapDescC(0 if vp is None else 1, *( (0.0, 0.0, 0.0, 0.0) if vp is None else (vp.x, vp.y, vp.w, vp.h)),
                          float(desc.exposure), 1 if desc.filmic else 0, 1 if desc.srgb else 0, 1 if desc.clear else 0)
    check(lib().lupin_hip_tonemap_and_fit_aspect(ctx.handle, src.handle, ptr(out), dst_width, dst_height, C.byref(c)))
    return out


def pathtrace_scene_tiles(ctx, resources, scene, render_target, pathtrace_type, desc, tile_size, rank, world):
    """Multi-GPU extension: all tiles owned by `rank` (include/lupin_tiles.h: round-robin, rotated rows) of one accumulation frame, in one launch."""
    assert render_target.format() == "Rgba16Float"
    if scene.handle is None:
        raise LupinError(_abi_code("LUPIN_ERR_NO_DEVICE"), "scene was built without a device context; there is no CPU fallback")
    keep = []
    c = _desc_to_c(desc, keep)
    check(lib().lupin_hip_pathtrace_scene_tiles(ctx.handle, resources.handle, scene.handle, render_target.handle,
                                                int(pathtrace_type), C.byref(c), tile_size, rank, world))


def pack_tiles(ctx, texture, tile_size, rank, world, device_dst_ptr):
    n = C.c_uint64()
    check(lib().lupin_hip_pack_tiles(ctx.handle, texture.handle, tile_size, rank, world, C.c_void_p(device_dst_ptr), C.byref(n)))
    return int(n.value)


def unpack_tiles(ctx, texture, tile_size, rank, world, device_src_ptr):
    check(lib().lupin_hip_unpack_tiles(ctx.handle, texture.handle, tile_size, rank, world, C.c_void_p(device_src_ptr)))


def unpack_gathered_tiles(ctx, texture, tile_size, rank, world, device_gathered_ptr, capacity_pixels):
    """One launch: every tile not owned by `rank` from `world` payloads of `capacity_pixels` pixels each."""
    check(lib().lupin_hip_unpack_gathered_tiles(ctx.handle, texture.handle, tile_size, rank, world, C.c_void_p(device_gathered_ptr), capacity_pixels))


def packed_tile_pixels(width, height, tile_size, rank, world):
    return int(lib().lupin_hip_packed_tile_pixels(width, height, tile_size, rank, world))


class Comm:
    """RCCL communicator of one context (include/lupin_hip.h "the one exchange step"): one per rank / per GPU."""

    def __init__(self, ctx, handle, rank, world):
        self.ctx, self.handle, self.rank, self.world = ctx, handle, rank, world

    @staticmethod
    def unique_id():
        """128 opaque bytes made by rank 0 (ncclGetUniqueId) that every rank passes to `init_rank`."""
        buf = (C.c_uint8 * 128)()
        check(lib().lupin_hip_comm_get_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    @classmethod
    def init_rank(cls, ctx, unique_id, rank, world):
        assert len(unique_id) == 128
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check(lib().lupin_hip_comm_init_rank(ctx.handle, C.cast(buf, C.c_void_p), rank, world, C.byref(h)))
        return cls(ctx, h, rank, world)

    @classmethod
    def init_all(cls, ctxs):
        """One process driving len(ctxs) GPUs (one context per device)."""
        n = len(ctxs)
        cin = (C.c_void_p * n)(*[c.handle for c in ctxs])
        cout = (C.c_void_p * n)()
        check(lib().lupin_hip_comm_init_all(cin, n, cout))
        return [cls(ctxs[i], C.c_void_p(cout[i]), i, n) for i in range(n)]

    def gather_framebuffer(self, texture, tile_size):
        """pack own tiles -> all-gather -> scatter the others' tiles; enqueued, `texture.download()` / `ctx.sync()` waits."""
        check(lib().lupin_hip_gather_framebuffer(self.handle, texture.handle, tile_size))

    def gather_framebuffer_to(self, texture, tile_size, root=0):
        """pack own tiles -> grouped ncclSend / ncclRecv -> the root scatters: only `root` ends up with the whole frame."""
        check(lib().lupin_hip_gather_framebuffer_to(self.handle, texture.handle, tile_size, root))

    def allreduce(self, values, op="sum"):
        a = np.ascontiguousarray(values, dtype=np.float64).copy()
        check(lib().lupin_hip_comm_allreduce_f64(self.handle, ptr(a), a.size, {"sum": 0, "max": 1}[op]))
        return a

    def barrier(self):
        """Every rank's enqueued frames have completed when this returns."""
        check(lib().lupin_hip_comm_barrier(self.handle))

    def close(self):
        if self.handle and self.ctx.handle:
            lib().lupin_hip_comm_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gather_framebuffer_all(comms, textures, tile_size):
    n = len(comms)
    check(lib().lupin_hip_gather_framebuffer_all((C.c_void_p * n)(*[c.handle for c in comms]),
                                                 (C.c_void_p * n)(*[t.handle for t in textures]), n, tile_size))


def _abi_code(name):
    return {"LUPIN_ERR_NO_DEVICE": -2}[name]


def trace_rays(ctx, scene, ori, dir_, ray_epsilon=0.001):
    """Closest-hit probe (bvh_custom.wgsl:7-110) on the device: returns hit, dst, uv, instance, tri arrays."""
    ori = np.ascontiguousarray(ori, np.float32).reshape(-1, 3)
    dir_ = np.ascontiguousarray(dir_, np.float32).reshape(-1, 3)
    n = len(ori)
    hit = np.zeros(n, np.uint32)
    dst = np.zeros(n, np.float32)
    uv = np.zeros((n, 2), np.float32)
    inst = np.zeros(n, np.uint32)
    tri = np.zeros(n, np.uint32)
    check(lib().lupin_hip_trace_rays(ctx.handle, scene.handle, n, ptr(ori), ptr(dir_), ray_epsilon,
                                     ptr(hit), ptr(dst), ptr(uv), ptr(inst), ptr(tri)))
    return hit, dst, uv, inst, tri


def trace_rays_wide(ctx, scene, ori, dir_, ray_epsilon=0.001):
    """The same probe through the four-wide traversal: returns hit, dst, uv, instance, tri and `needs_retrace` (1 = the
    traversal could not certify that ray; the pipeline re-traces such queries with the binary kernel)."""
    ori = np.ascontiguousarray(ori, np.float32).reshape(-1, 3)
    dir_ = np.ascontiguousarray(dir_, np.float32).reshape(-1, 3)
    n = len(ori)
    hit = np.zeros(n, np.uint32)
    dst = np.zeros(n, np.float32)
    uv = np.zeros((n, 2), np.float32)
    inst = np.zeros(n, np.uint32)
    tri = np.zeros(n, np.uint32)
    flag = np.zeros(n, np.uint32)
    check(lib().lupin_hip_trace_rays_wide(ctx.handle, scene.handle, n, ptr(ori), ptr(dir_), ray_epsilon,
                                          ptr(hit), ptr(dst), ptr(uv), ptr(inst), ptr(tri), ptr(flag)))
    return hit, dst, uv, inst, tri, flag
