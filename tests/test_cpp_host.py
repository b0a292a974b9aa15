"""The C++ host mirror (include/lupin.hpp, namespace lp:: / lpl::) and the example1.rs counterpart built on it."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "example1")


@pytest.fixture(scope="module")
def example1(built):
    lib_dir = os.path.join(ROOT, "lupinpathtracer_amd")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "example1.cpp"), "-L" + lib_dir, "-llupin_hip", "-L/opt/rocm/lib",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", EXE]
    subprocess.check_call(cmd)
    return EXE


def test_example_compiles_and_refuses_to_run_without_a_device(example1):
    from lupinpathtracer_amd import api
    if api.device_count() > 0:
        pytest.skip("a HIP device is present")
    p = subprocess.run([example1, "8", "1", "/tmp/_lupin_example.hdr"], capture_output=True, text=True)
    assert p.returncode == 1
    assert "no HIP device" in p.stderr and "no CPU fallback" in p.stderr


@pytest.mark.gpu
def test_example1_matches_python_host(gpu_ctx, example1, tmp_path):
    """example1.rs's loop (5 spp x N frames, flip, extra flip, save) gives the same image through both host mirrors."""
    from lupinpathtracer_amd import loader
    from tests import util
    out = str(tmp_path / "output.hdr")
    preview = str(tmp_path / "preview.ppm")
    subprocess.check_call([example1, "64", "6", out, preview])
    cpp = loader.read_hdr(out)
    scene, cams = util.load_scene("cornellbox_builtin", gpu_ctx)
    img = util.gpu_accumulate(gpu_ctx, scene, cams[0], 64, 64, frames=6, spp=5)
    ref_path = str(tmp_path / "ref.hdr")
    loader.save_texture(ref_path, img)
    ref = loader.read_hdr(ref_path)
    assert cpp.shape == ref.shape == (64, 64, 3)
    assert np.array_equal(cpp, ref)
    # the tonemapped preview (lp::tonemap_and_fit_aspect through the C++ mirror) equals the Python host's
    from lupinpathtracer_amd import api
    raw = open(preview, "rb").read()
    assert raw.startswith(b"P6\n64 64\n255\n")
    ppm = np.frombuffer(raw[len(b"P6\n64 64\n255\n"):], np.uint8).reshape(64, 64, 3)
    tex = api.Texture(gpu_ctx, 64, 64)
    tex.upload(img)
    assert np.array_equal(ppm, api.tonemap_and_fit_aspect(gpu_ctx, tex, 64, 64)[..., :3])


@pytest.fixture(scope="module")
def render_scene(built):
    exe = os.path.join(ROOT, "examples", "render_scene")
    lib_dir = os.path.join(ROOT, "lupinpathtracer_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "render_scene.cpp"),
                           "-L" + lib_dir, "-llupin_hip", "-L/opt/rocm/lib", "-lz", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


@pytest.mark.gpu
def test_render_scene_cpp_loader_matches_python_host(gpu_ctx, render_scene, tmp_path):
    """The whole C++ host chain (lupin_loader.hpp -> lupin.hpp -> C ABI) on a textured fixture scene with an HDR
    environment gives the image the Python host gives."""
    from lupinpathtracer_amd import api, loader
    from tests import util
    name, cam_i, W = "features1", 1, 192
    out = str(tmp_path / "render.hdr")
    subprocess.check_call([render_scene, os.path.join(util.SCENES, name, name + ".json"), str(cam_i), str(W), "3", "4", out, "", util.SHARED])
    cpp = loader.read_hdr(out)
    scene, cams = util.load_scene(name, gpu_ctx)
    H = int(np.float32(W) / np.float32(cams[cam_i].params.aspect))
    img = util.gpu_accumulate(gpu_ctx, scene, cams[cam_i], W, H, frames=3, spp=4, advanced=api.AdvancedParams(max_radiance=10.0))
    ref_path = str(tmp_path / "ref.hdr")
    loader.save_texture(ref_path, img)
    assert np.array_equal(cpp, loader.read_hdr(ref_path))
