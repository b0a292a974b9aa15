// example1.cpp -- the reference's canonical headless example (lupin_examples/src/example1.rs:6-57) against the
// HIP backend, through the C++ host mirror include/lupin.hpp: Cornell box, 5 spp x N accumulation frames into a
// DoubleBufferedTexture, saved as output.hdr.
//
//   g++ -std=c++17 -O2 -Iinclude examples/example1.cpp -Llupinpathtracer_amd -llupin_hip -Wl,-rpath,$PWD/lupinpathtracer_amd -o examples/example1
//   ./examples/example1 [size=1000] [num_accums=200] [out=output.hdr] [preview.ppm]   (preview: tonemapped, sRGB)
#include <cstdlib>
#include <iostream>

#include "lupin.hpp"

int main(int argc, char **argv)
{
    const uint32_t size = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 1000;
    const uint32_t num_accums = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 200;
    const std::string out_path = argc > 3 ? argv[3] : "output.hdr";
    try
    {
        lp::Device device(0);
        lp::PathtraceResources pathtrace_res = lp::build_pathtrace_resources(device, lp::BakedPathtraceParams{false, 8, 5});
        auto [scene, cameras] = lpl::build_scene_cornell_box(device, false);
        lp::DoubleBufferedTexture output = lp::DoubleBufferedTexture::create(device, size, size);

        for (uint32_t accum_idx = 0; accum_idx < num_accums; accum_idx++)
        {
            lp::PathtraceDesc desc;
            desc.accum_params = lp::AccumulationParams{output.back(), accum_idx};
            desc.camera_params = cameras[0].params;
            desc.camera_transform = cameras[0].transform;
            lp::pathtrace_scene(device, pathtrace_res, scene, output.front(), lp::PathtraceType::Standard, desc);
            output.flip();
        }
        output.flip();
        lpl::save_texture(out_path, output.front());
        if (argc > 4)   // lp::tonemap_and_fit_aspect with the default TonemapDesc, as the reference's viewer / test app present frames
        {
            std::vector<uint8_t> ldr;
            lp::tonemap_and_fit_aspect(device, output.front(), ldr, size, size);
            lpl::save_rgba8_ppm(argv[4], ldr, size, size);
        }
        std::cout << "wrote " << out_path << " (" << size << "x" << size << ", " << num_accums << " x 5 spp)\n";
    }
    catch (const lp::Error &e)
    {
        std::cerr << "lupin error " << e.code << ": " << e.what() << "\n";
        return 1;
    }
    return 0;
}
