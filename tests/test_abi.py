"""The C-ABI library loads and exports every symbol include/lupin_hip.h declares; record layouts
match the reference's #[repr(C)] sizes.  No device needed."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from lupinpathtracer_amd import _abi, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "lupin_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lupin_(?:hip_)?[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(built):
    handle = C.CDLL(_abi.LIB_PATH)
    names = header_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(handle, n), f"{n} declared in lupin_hip.h but not exported"
    bound = {n for n, _, _ in _abi.SYMBOLS}
    assert set(names) == bound, (set(names) ^ bound)


def test_record_sizes_match_reference_layouts(built):
    # renderer.rs:94-280 sizes
    assert _abi.MESH_INFO_DTYPE.itemsize == 12
    assert _abi.INSTANCE_DTYPE.itemsize == 64
    assert _abi.MATERIAL_DTYPE.itemsize == 96
    assert _abi.ENVIRONMENT_DTYPE.itemsize == 80
    assert _abi.LIGHT_DTYPE.itemsize == 8
    assert _abi.ALIAS_BIN_DTYPE.itemsize == 12
    assert _abi.BVH_NODE_DTYPE.itemsize == 32
    assert _abi.TLAS_NODE_DTYPE.itemsize == 48
    assert C.sizeof(_abi.PushConstants) == 128
    pc = _abi.PushConstants
    assert pc.camera_lens.offset == 64 and pc.flags.offset == 84 and pc.id_offset.offset == 88
    assert pc.accum_counter.offset == 96 and pc.pathtrace_type.offset == 112 and pc.ray_epsilon.offset == 124
    assert _abi.MATERIAL_DTYPE.fields["mat_type"][1] == 48 and _abi.MATERIAL_DTYPE.fields["normal_tex_idx"][1] == 88
    assert _abi.TLAS_NODE_DTYPE.fields["right"][1] == 32 and _abi.INSTANCE_DTYPE.fields["mesh_idx"][1] == 48


def test_defaults_match_reference(built):
    p = api.BakedPathtraceParams()
    assert (p.with_runtime_checks, p.max_bounces, p.samples_per_pixel) == (False, 8, 5)      # renderer.rs:458-468
    c = api.CameraParams()
    assert (c.lens, c.film, c.aspect, c.focus, c.aperture) == (0.050, 0.036, 1.5, 10000.0, 0.0)  # :695-707
    a = api.AdvancedParams()
    assert (a.max_radiance, a.rng_seed, a.ray_epsilon) == (100.0, 0, 0.001)                   # :739-748
    t = api.TileParams()
    assert (t.tile_size, t.tile_idx) == (100, 0)                                              # :661-669
    m = api.default_material()
    assert tuple(m["color"]) == (0.0, 0.0, 0.0, 1.0) and m["ior"] == np.float32(1.5) and m["tr_depth"] == np.float32(0.01)
    assert int(m["color_tex_idx"]) == 0xFFFFFFFF


@pytest.mark.parametrize("tile_size,w,h", [(100, 1920, 1080), (16, 1024, 1024), (1, 7, 9), (3, 1, 1), (25, 1000, 1000)])
def test_get_num_tiles(built, tile_size, w, h):
    # renderer.rs:675-681
    ntx = (max(1, w) - 1) // (tile_size * 4) + 1
    nty = (max(1, h) - 1) // (tile_size * 4) + 1
    assert api.get_num_tiles(tile_size, w, h) == ntx * nty


def test_no_device_means_error_not_fallback(built):
    """On a machine without a GPU the product refuses to run; it never renders on the CPU."""
    if api.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(api.LupinError) as e:
        api.Context(0)
    assert e.value.code == -2   # LUPIN_ERR_NO_DEVICE
    from lupinpathtracer_amd import loader
    scene, cams = loader.build_scene_cornell_box(None)
    with pytest.raises(api.LupinError):
        api.pathtrace_scene(None, None, scene, type("T", (), {"format": lambda self: "Rgba16Float"})(), 0, api.PathtraceDesc())


def test_packed_tile_pixels_partition(built):
    for (w, h, ts, world) in [(1024, 1024, 16, 8), (1920, 1080, 25, 4), (100, 37, 3, 3), (64, 64, 16, 5)]:
        total = sum(api.packed_tile_pixels(w, h, ts, r, world) for r in range(world))
        assert total == w * h


def test_rust_shim_declares_only_header_symbols_with_matching_struct_sizes():
    """integration/rust/lupin_hip/src/ffi.rs (source only: no Rust toolchain here) must bind existing entry points, and
    every #[repr(C)] struct it declares must have the C struct's field count and byte size."""
    import re
    import ctypes as C
    from lupinpathtracer_amd import _abi
    src = open(os.path.join(ROOT, "integration", "rust", "lupin_hip", "src", "ffi.rs")).read()
    header = open(os.path.join(ROOT, "include", "lupin_hip.h")).read()
    fns = re.findall(r"pub fn (lupin_\w+)\(", src)
    assert len(fns) >= 30
    for f in fns:
        assert re.search(r"\b%s\(" % f, header), f
    for must in ("lupin_hip_pathtrace_scene", "lupin_hip_build_pathtrace_resources", "lupin_hip_scene_create", "lupin_hip_dbuf_flip"):
        assert must in fns
    size = {"u32": 4, "f32": 4, "LupinMat3x4": 48, "LupinMat4x3": 48, "LupinMat4": 64}
    want = {"LupinMeshInfo": 12, "LupinInstance": 64, "LupinMaterial": 96, "LupinEnvironment": 80, "LupinLight": 8, "LupinAliasBin": 12,
            "LupinBvhNode": 32, "LupinTlasNode": 48, "LupinBakedPathtraceParams": 12, "LupinCameraParams": 24, "LupinAdvancedParams": 12,
            "LupinTileParams": 8, "LupinDebugVizDesc": 16, "LupinTonemapDesc": 36}
    for name, nbytes in want.items():
        body = re.search(r"pub struct %s \{(.*?)\}" % name, src, re.S).group(1)
        total = 0
        for ty in re.findall(r"pub \w+: ([^,]+?)\s*(?:,|$)", body.strip()):
            m = re.fullmatch(r"\[(\w+); (\d+)\]", ty.strip())
            total += size[m.group(1)] * int(m.group(2)) if m else size[ty.strip()]
        assert total == nbytes, (name, total, nbytes)
    assert C.sizeof(_abi.TonemapDescC) == 36 and C.sizeof(_abi.DebugVizDescC) == 16


def test_library_asks_for_eight_hardware_queues_unless_the_host_chose(built):
    """Frames in flight run on one HIP stream each; the runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).
    Loading the library sets the variable to 8 when it is unset and leaves a host's choice alone (checked in fresh processes)."""
    import subprocess
    import sys
    code = ("import ctypes, os, sys; sys.path.insert(0, %r); from lupinpathtracer_amd import _abi; _abi.lib(); "
            "g = ctypes.CDLL(None).getenv; g.restype = ctypes.c_char_p; print(g(b'GPU_MAX_HW_QUEUES').decode())") % ROOT
    for preset, want in ((None, "8"), ("2", "2")):
        env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
        if preset is not None:
            env["GPU_MAX_HW_QUEUES"] = preset
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
        assert out.returncode == 0 and out.stdout.strip() == want, out.stdout + out.stderr
