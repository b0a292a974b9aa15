"""include/lupin_loader.hpp (C++ counterpart of the reference's lupin_loader crate: Yocto/GL JSON, PLY, PNG, HDR)
against the Python loader: every array of every fixture scene of the reference, byte for byte.  CPU only."""
import os
import subprocess

import numpy as np
import pytest

from lupinpathtracer_amd import loader
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENE_NAMES = sorted(d for d in os.listdir(util.SCENES) if not d.startswith("_"))


@pytest.fixture(scope="module")
def loader_dump(built, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cpp_loader") / "loader_dump")
    lib_dir = os.path.join(ROOT, "lupinpathtracer_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "loader_dump.cpp"),
                           "-L" + lib_dir, "-llupin_hip", "-L/opt/rocm/lib", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-lz", "-o", exe])
    return exe


def raw(path):
    with open(path, "rb") as f:
        return f.read()


@pytest.mark.parametrize("name", SCENE_NAMES)
def test_cpp_loader_equals_python_loader(loader_dump, tmp_path, name):
    out = str(tmp_path)
    subprocess.check_call([loader_dump, os.path.join(util.SCENES, name, name + ".json"), util.SHARED, out])
    scene, textures, envs_info, cams = loader.load_scene_cpu_yoctogl_v24(os.path.join(util.SCENES, name, name + ".json"), [util.SHARED])
    assert raw(out + "/materials.bin") == np.ascontiguousarray(scene.materials).tobytes()
    assert raw(out + "/instances.bin") == np.ascontiguousarray(scene.instances).tobytes()
    assert raw(out + "/environments.bin") == np.ascontiguousarray(scene.environments).tobytes()
    assert raw(out + "/mesh_infos.bin") == np.ascontiguousarray(scene.mesh_infos).tobytes()
    for i, (pos, idx) in enumerate(zip(scene.verts_pos_array, scene.indices_array)):
        assert raw(f"{out}/pos_{i}.bin") == np.ascontiguousarray(pos, np.float32).tobytes()
        assert raw(f"{out}/idx_{i}.bin") == np.ascontiguousarray(idx, np.uint32).tobytes()
    for tag, arrs in (("nrm", scene.verts_normal_array), ("uv", scene.verts_texcoord_array), ("col", scene.verts_color_array)):
        for i, a in enumerate(arrs):
            assert raw(f"{out}/{tag}_{i}.bin") == np.ascontiguousarray(a, np.float32).tobytes(), (tag, i)
        assert not os.path.exists(f"{out}/{tag}_{len(arrs)}.bin")
    for i, t in enumerate(textures):
        w, h, fmt = np.frombuffer(raw(f"{out}/texmeta_{i}.bin"), np.uint32)
        assert (h, w) == t.pixels.shape[:2] and fmt == (1 if t.pixels.dtype == np.float16 else 0)
        assert raw(f"{out}/tex_{i}.bin") == np.ascontiguousarray(t.pixels).tobytes(), f"texture {i}"
    for i, e in enumerate(envs_info):
        assert raw(f"{out}/env_{i}.bin") == np.ascontiguousarray(e.data, np.float32).tobytes()
    got = np.frombuffer(raw(out + "/cameras.bin"), np.float32).reshape(len(cams), 18)
    for g, c in zip(got, cams):
        assert np.array_equal(g[:12], np.asarray(c.transform, np.float32).reshape(-1))
        p = c.params
        assert np.array_equal(g[12:], np.array([1.0 if p.is_orthographic else 0.0, p.lens, p.film, p.aspect, p.focus, p.aperture], np.float32))


def test_cpp_loader_reports_errors(loader_dump, tmp_path):
    bad = tmp_path / "bad.json"
    bad.write_text('{"shapes": [{"uri": "shapes/missing.ply"}]}')
    p = subprocess.run([loader_dump, str(bad), util.SHARED, str(tmp_path)], capture_output=True, text=True)
    assert p.returncode == 1 and "not found" in p.stderr
    bad.write_text('{"cameras": [')
    p = subprocess.run([loader_dump, str(bad), util.SHARED, str(tmp_path)], capture_output=True, text=True)
    assert p.returncode == 1 and "JSON" in p.stderr
