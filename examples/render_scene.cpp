// render_scene.cpp -- what the reference's test app does per golden (lupin_tests/src/main.rs:120-170): load a Yocto/GL
// scene through lupin_loader, accumulate N frames of `spp` samples with pathtrace_scene, save the Rgba16Float result as
// .hdr (and, optionally, a tonemapped preview).  C++ host: include/lupin.hpp + include/lupin_loader.hpp.
//
//   g++ -std=c++17 -O2 -Iinclude examples/render_scene.cpp -Llupinpathtracer_amd -llupin_hip -lz -Wl,-rpath,$PWD/lupinpathtracer_amd -o examples/render_scene
//   ./examples/render_scene scene.json [camera=0] [width=1920] [frames=101] [spp=10] [out=render.hdr] [preview.ppm] [asset_dir]
#include <cstdlib>
#include <iostream>

#include "lupin_loader.hpp"

int main(int argc, char **argv)
{
    if (argc < 2) { std::cerr << "usage: render_scene scene.json [camera] [width] [frames] [spp] [out.hdr] [preview.ppm] [asset_dir]\n"; return 2; }
    const uint32_t cam_idx = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 0;
    const uint32_t width = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 1920;
    const uint32_t frames = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 101;
    const uint32_t spp = argc > 5 ? (uint32_t)std::atoi(argv[5]) : 10;
    const std::string out_path = argc > 6 ? argv[6] : "render.hdr";
    try
    {
        lp::Device device(0);
        std::vector<std::string> asset_dirs;
        if (argc > 8) asset_dirs.push_back(argv[8]);
        auto [scene, cameras] = lpl::load_scene_yoctogl_v24(argv[1], device, true, asset_dirs);
        if (cam_idx >= cameras.size()) throw lp::Error(LUPIN_ERR_INVALID_ARGUMENT, "camera index out of range");
        const lpl::SceneCamera &cam = cameras[cam_idx];
        const uint32_t height = (uint32_t)((float)width / cam.params.aspect);   // compute_dimensions_for_1080p's truncation (main.rs:477-484)
        lp::PathtraceResources res = lp::build_pathtrace_resources(device, lp::BakedPathtraceParams{false, 8, spp});
        lp::DoubleBufferedTexture output = lp::DoubleBufferedTexture::create(device, width, height);
        for (uint32_t k = 0; k < frames; k++)
        {
            lp::PathtraceDesc desc;
            desc.accum_params = lp::AccumulationParams{output.back(), k};
            desc.camera_params = cam.params;
            desc.camera_transform = cam.transform;
            desc.advanced.max_radiance = 10.0f;   // the test app's setting
            lp::pathtrace_scene(device, res, scene, output.front(), lp::PathtraceType::Standard, desc);
            output.flip();
        }
        output.flip();
        lpl::save_texture(out_path, output.front());
        if (argc > 7 && argv[7][0])
        {
            std::vector<uint8_t> ldr;
            lp::tonemap_and_fit_aspect(device, output.front(), ldr, width, height);
            lpl::save_rgba8_ppm(argv[7], ldr, width, height);
        }
        std::cout << "wrote " << out_path << " (" << width << "x" << height << ", " << frames << " x " << spp << " spp)\n";
    }
    catch (const std::exception &e)
    {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
