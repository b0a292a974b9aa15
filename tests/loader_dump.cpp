// Test helper: loads a Yocto/GL scene through include/lupin_loader.hpp (no device needed) and dumps every array as raw
// bytes into <out_dir>/, for tests/test_cpp_loader.py to compare with the Python loader's arrays.
//   loader_dump <scene.json> <asset_dir> <out_dir>
#include <cstdio>
#include <iostream>

#include "lupin_loader.hpp"

template <typename T>
static void dump(const std::string &path, const T *data, size_t count)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    if (count) std::fwrite(data, sizeof(T), count, f);
    std::fclose(f);
}

int main(int argc, char **argv)
{
    if (argc < 4) { std::cerr << "usage: loader_dump scene.json asset_dir out_dir\n"; return 2; }
    try
    {
        const std::string out = argv[3];
        lpl::LoadedSceneCPU s = lpl::load_scene_cpu_yoctogl_v24(argv[1], {argv[2]});
        dump(out + "/materials.bin", s.scene.materials.data(), s.scene.materials.size());
        dump(out + "/instances.bin", s.scene.instances.data(), s.scene.instances.size());
        dump(out + "/environments.bin", s.scene.environments.data(), s.scene.environments.size());
        dump(out + "/mesh_infos.bin", s.scene.mesh_infos.data(), s.scene.mesh_infos.size());
        for (size_t i = 0; i < s.scene.verts_pos_array.size(); i++)
        {
            dump(out + "/pos_" + std::to_string(i) + ".bin", s.scene.verts_pos_array[i].data(), s.scene.verts_pos_array[i].size());
            dump(out + "/idx_" + std::to_string(i) + ".bin", s.scene.indices_array[i].data(), s.scene.indices_array[i].size());
        }
        for (size_t i = 0; i < s.scene.verts_normal_array.size(); i++) dump(out + "/nrm_" + std::to_string(i) + ".bin", s.scene.verts_normal_array[i].data(), s.scene.verts_normal_array[i].size());
        for (size_t i = 0; i < s.scene.verts_texcoord_array.size(); i++) dump(out + "/uv_" + std::to_string(i) + ".bin", s.scene.verts_texcoord_array[i].data(), s.scene.verts_texcoord_array[i].size());
        for (size_t i = 0; i < s.scene.verts_color_array.size(); i++) dump(out + "/col_" + std::to_string(i) + ".bin", s.scene.verts_color_array[i].data(), s.scene.verts_color_array[i].size());
        for (size_t i = 0; i < s.textures.size(); i++)
        {
            dump(out + "/tex_" + std::to_string(i) + ".bin", s.textures[i].pixels.data(), s.textures[i].pixels.size());
            const uint32_t meta[3] = {s.textures[i].width, s.textures[i].height, s.textures[i].format};
            dump(out + "/texmeta_" + std::to_string(i) + ".bin", meta, 3);
        }
        for (size_t i = 0; i < s.envs_info.size(); i++) dump(out + "/env_" + std::to_string(i) + ".bin", s.envs_info[i].data.data(), s.envs_info[i].data.size());
        std::vector<float> cams;
        for (const lpl::SceneCamera &c : s.cameras)
        {
            for (int col = 0; col < 4; col++) for (int r = 0; r < 3; r++) cams.push_back(c.transform.m[col][r]);
            cams.push_back(c.params.is_orthographic ? 1.0f : 0.0f);
            cams.push_back(c.params.lens); cams.push_back(c.params.film); cams.push_back(c.params.aspect); cams.push_back(c.params.focus); cams.push_back(c.params.aperture);
        }
        dump(out + "/cameras.bin", cams.data(), cams.size());
        std::cout << s.scene.mesh_infos.size() << " meshes, " << s.scene.instances.size() << " instances, " << s.textures.size() << " textures, "
                  << s.cameras.size() << " cameras\n";
    }
    catch (const std::exception &e) { std::cerr << "error: " << e.what() << "\n"; return 1; }
    return 0;
}
