// lupin.hpp -- C++ host-side mirror of the reference's Rust call surface (crate `lupin_pt`, namespace `lp::`,
// and the parts of `lupin_loader`, `lpl::`, that the canonical example needs) on top of the C ABI in lupin_hip.h.
//
// Same names, defaults and error behaviour as the reference:
//   lp::BakedPathtraceParams / build_pathtrace_resources      lupin/src/renderer.rs:451-642
//   lp::PathtraceDesc, AccumulationParams, TileParams, CameraParams, AdvancedParams, PathtraceType
//                                                             renderer.rs:644-766
//   lp::pathtrace_scene                                       renderer.rs:768-842
//   lp::get_num_tiles                                         renderer.rs:675-681
//   lp::DoubleBufferedTexture                                 lupin/src/wgpu_utils.rs:279-348
//   lp::SceneCPU, validate_scene, build_accel_structures_and_upload   renderer.rs:62-76, data_structures.rs:696-928
//   lpl::build_scene_cornell_box, lpl::save_texture (.hdr)    lupin_loader/src/loader.rs:14-207, :1775-1879
//
// Where the reference panics (assert! / panic!), these throw lp::Error.  Header-only; link with -llupin_hip.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <array>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "lupin_hip.h"

namespace lp {

struct Error : std::runtime_error
{
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc)
{
    if (rc != LUPIN_OK) throw Error(rc, lupin_hip_last_error());
}

constexpr uint32_t SENTINEL_IDX = LUPIN_SENTINEL_IDX;
using Mat3x4 = LupinMat3x4;
using MeshInfo = LupinMeshInfo;
using Instance = LupinInstance;
using Material = LupinMaterial;
using Environment = LupinEnvironment;
using Light = LupinLight;
using AliasBin = LupinAliasBin;
using BvhNode = LupinBvhNode;
using TlasNode = LupinTlasNode;
struct Vec4 { float x = 0, y = 0, z = 0, w = 0; };

inline Mat3x4 mat3x4_identity() { return Mat3x4{{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}}}; }
inline MeshInfo default_mesh_info() { return MeshInfo{SENTINEL_IDX, SENTINEL_IDX, SENTINEL_IDX}; }   // renderer.rs:102-113
inline Instance default_instance()
{
    Instance i{};
    i.transpose_inverse_transform = LupinMat4x3{{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}}};
    return i;
}
inline Material default_material()   // renderer.rs:163-185
{
    Material m{};
    m.color[3] = 1.0f;
    m.ior = 1.5f;
    m.tr_depth = 0.01f;
    m.color_tex_idx = m.emission_tex_idx = m.roughness_tex_idx = m.scattering_tex_idx = m.normal_tex_idx = SENTINEL_IDX;
    return m;
}

enum class PathtraceType : uint32_t { Standard = 0, MIS = 1, Naive = 2, Direct = 3 };   // renderer.rs:711-729
enum class FalsecolorType : uint32_t { Albedo = 0, Normals, NormalsUnsigned, FrontFacing, Emission, Roughness, Metallic, Opacity, MatType, IsDelta, Instance, Tri };   // renderer.rs:843-870

struct BakedPathtraceParams   // renderer.rs:451-468
{
    bool with_runtime_checks = false;
    uint32_t max_bounces = 8;
    uint32_t samples_per_pixel = 5;
};
struct CameraParams   // renderer.rs:683-708
{
    bool is_orthographic = false;
    float lens = 0.050f, film = 0.036f, aspect = 1.5f, focus = 10000.0f, aperture = 0.0f;
};
struct AdvancedParams { float max_radiance = 100.0f; uint32_t rng_seed = 0; float ray_epsilon = 0.001f; };   // :731-749
struct TileParams { uint32_t tile_size = 100; uint32_t tile_idx = 0; };                                      // :651-670

inline uint32_t get_num_tiles(uint32_t tile_size, uint32_t width, uint32_t height) { return lupin_hip_get_num_tiles(tile_size, width, height); }

// ---- device objects (RAII over the opaque handles) ----

class Device   // the reference's (wgpu::Device, wgpu::Queue) pair
{
  public:
    explicit Device(int ordinal = 0) { check(lupin_hip_create_context(ordinal, &ctx_)); }
    ~Device() { lupin_hip_destroy_context(ctx_); }
    Device(const Device &) = delete;
    Device &operator=(const Device &) = delete;
    LupinContext *raw() const { return ctx_; }
    void poll_wait() const { check(lupin_hip_sync(ctx_)); }   // device.poll(wait_indefinitely)
    // pathtracer.wgsl:275-289 re-quantises the running mean to f16 every frame (default, LUPIN_ACCUM_F16_RUNNING_AVERAGE);
    // LUPIN_ACCUM_F32 runs the same recurrence on an f32 shadow of each texture and stores the rounded f16 view
    void set_accumulation_mode(int mode) const { check(lupin_hip_set_accumulation_mode(ctx_, mode)); }
    // path state of every frame in flight allocated now instead of at each lane's first pathtrace call
    void reserve_path_state(uint64_t pixels, uint32_t max_bounces, uint32_t samples_per_pixel) const { check(lupin_hip_reserve_path_state(ctx_, pixels, max_bounces, samples_per_pixel)); }
    // calls per wavefront (1..16; 0 = by dispatch size, the default) and the hierarchy the tracer walks (LUPIN_TRAVERSAL_BINARY / _WIDE)
    void set_batch_frames(uint32_t frames) const { check(lupin_hip_set_batch_frames(ctx_, frames)); }
    void set_traversal(int mode) const { check(lupin_hip_set_traversal(ctx_, mode)); }
  private:
    LupinContext *ctx_ = nullptr;
};

class PathtraceResources
{
  public:
    PathtraceResources(const Device &d, const BakedPathtraceParams &p)
    {
        LupinBakedPathtraceParams c{p.with_runtime_checks ? 1u : 0u, p.max_bounces, p.samples_per_pixel};
        check(lupin_hip_build_pathtrace_resources(d.raw(), &c, &res_));
    }
    ~PathtraceResources() { lupin_hip_destroy_pathtrace_resources(res_); }
    PathtraceResources(const PathtraceResources &) = delete;
    LupinPathtraceResources *raw() const { return res_; }
  private:
    LupinPathtraceResources *res_ = nullptr;
};
inline PathtraceResources build_pathtrace_resources(const Device &d, const BakedPathtraceParams &p) { return PathtraceResources(d, p); }

class TextureRef   // a borrowed Rgba16Float render target (wgpu::Texture)
{
  public:
    explicit TextureRef(LupinTexture *t = nullptr) : t_(t) {}
    LupinTexture *raw() const { return t_; }
    uint32_t width() const { return lupin_hip_texture_width(t_); }
    uint32_t height() const { return lupin_hip_texture_height(t_); }
    std::vector<uint16_t> download() const   // synchronises
    {
        std::vector<uint16_t> px((size_t)width() * height() * 4);
        check(lupin_hip_texture_download_rgba16f(t_, px.data()));
        return px;
    }
    std::vector<float> download_f32() const   // the f32 accumulator (LUPIN_ACCUM_F32 frames only)
    {
        std::vector<float> px((size_t)width() * height() * 4);
        check(lupin_hip_texture_download_rgba32f(t_, px.data()));
        return px;
    }
  private:
    LupinTexture *t_;
};

class DoubleBufferedTexture   // wgpu_utils.rs:279-348
{
  public:
    static DoubleBufferedTexture create(const Device &d, uint32_t width, uint32_t height) { return DoubleBufferedTexture(d, width, height); }
    DoubleBufferedTexture(const Device &d, uint32_t w, uint32_t h) { check(lupin_hip_dbuf_create(d.raw(), w, h, &t_)); }
    DoubleBufferedTexture(DoubleBufferedTexture &&o) noexcept : t_(o.t_) { o.t_ = nullptr; }
    DoubleBufferedTexture(const DoubleBufferedTexture &) = delete;
    ~DoubleBufferedTexture() { if (t_) lupin_hip_dbuf_destroy(t_); }
    TextureRef front() const { return TextureRef(lupin_hip_dbuf_front(t_)); }
    TextureRef back() const { return TextureRef(lupin_hip_dbuf_back(t_)); }
    void copy_front_to_back() { check(lupin_hip_dbuf_copy_front_to_back(t_)); }
    void flip() { lupin_hip_dbuf_flip(t_); }
    void resize(uint32_t w, uint32_t h) { check(lupin_hip_dbuf_resize(t_, w, h)); }
  private:
    LupinDoubleBufferedTexture *t_ = nullptr;
};

struct AccumulationParams { TextureRef prev_frame; uint32_t accum_counter = 0; };   // renderer.rs:644-649

struct PathtraceDesc   // renderer.rs:751-766
{
    std::optional<AccumulationParams> accum_params;
    std::optional<TileParams> tile_params;
    CameraParams camera_params;
    Mat3x4 camera_transform = mat3x4_identity();
    bool force_software_bvh = false;
    AdvancedParams advanced;
};

// ---- scene ----

struct TextureCPU { uint32_t width = 0, height = 0; uint32_t format = LUPIN_TEX_RGBA8_UNORM; std::vector<uint8_t> pixels; };
struct EnvMapInfo { std::vector<Vec4> data; uint32_t width = 0, height = 0; };   // data_structures.rs:13-19

struct SceneCPU   // renderer.rs:62-76
{
    std::vector<MeshInfo> mesh_infos;
    std::vector<std::vector<Vec4>> verts_pos_array, verts_normal_array, verts_color_array;
    std::vector<std::vector<float>> verts_texcoord_array;   // 2 floats per vertex
    std::vector<std::vector<uint32_t>> indices_array;
    std::vector<Instance> instances;
    std::vector<Material> materials;
    std::vector<Environment> environments;
};

inline void validate_scene(const SceneCPU &s, uint32_t num_textures, uint32_t num_samplers)   // data_structures.rs:876-928
{
    auto require = [](bool ok, const char *what) { if (!ok) throw Error(LUPIN_ERR_INVALID_ARGUMENT, std::string("validate_scene: ") + what); };
    require(s.verts_pos_array.size() == s.mesh_infos.size(), "verts_pos_array / mesh_infos size mismatch");
    require(num_textures == num_samplers, "textures / samplers mismatch");
    for (size_t i = 0; i < s.mesh_infos.size(); i++)
    {
        const MeshInfo &m = s.mesh_infos[i];
        if (m.normals_buf_idx != SENTINEL_IDX) require(m.normals_buf_idx < s.verts_normal_array.size() && s.verts_normal_array[m.normals_buf_idx].size() == s.verts_pos_array[i].size(), "normals");
        if (m.texcoords_buf_idx != SENTINEL_IDX) require(m.texcoords_buf_idx < s.verts_texcoord_array.size() && s.verts_texcoord_array[m.texcoords_buf_idx].size() == 2 * s.verts_pos_array[i].size(), "texcoords");
        if (m.colors_buf_idx != SENTINEL_IDX) require(m.colors_buf_idx < s.verts_color_array.size() && s.verts_color_array[m.colors_buf_idx].size() == s.verts_pos_array[i].size(), "colors");
    }
    for (size_t i = 0; i < s.indices_array.size(); i++)
        for (uint32_t idx : s.indices_array[i]) require(idx < s.verts_pos_array[i].size(), "vertex index out of range");
    for (const Instance &in : s.instances) require(in.mesh_idx < s.mesh_infos.size() && in.mat_idx < s.materials.size(), "instance indices");
    auto tex_ok = [&](uint32_t t) { return t == SENTINEL_IDX || t < num_textures; };
    for (const Material &m : s.materials) require(tex_ok(m.color_tex_idx) && tex_ok(m.emission_tex_idx) && tex_ok(m.roughness_tex_idx) && tex_ok(m.scattering_tex_idx) && tex_ok(m.normal_tex_idx), "material texture index");
    for (const Environment &e : s.environments) require(e.emission[0] >= 0 && e.emission[1] >= 0 && e.emission[2] >= 0 && tex_ok(e.emission_tex_idx), "environment");
}

class Scene   // lp::Scene, software-BVH configuration (renderer.rs:17-60)
{
  public:
    Scene() = default;
    Scene(Scene &&o) noexcept : scene_(o.scene_) { o.scene_ = nullptr; }
    Scene(const Scene &) = delete;
    ~Scene() { if (scene_) lupin_hip_scene_destroy(scene_); }
    LupinScene *raw() const { return scene_; }
    LupinScene *scene_ = nullptr;
};

// Which builder produces the BLASes: the reference's CPU builder restated (lupin_build_bvh) or the same tree built on the GPU.
enum class BlasBuilder { Cpu, Device };

// lp::build_accel_structures_and_upload (data_structures.rs:696-872): BLAS over a clone of each index buffer,
// TLAS over the instances, lights + alias tables from the ORIGINAL triangle order, then upload.
inline Scene build_accel_structures_and_upload(const Device &d, const SceneCPU &s, const std::vector<TextureCPU> &textures,
                                               const std::vector<EnvMapInfo> &envs_info, bool /*build_sw_and_hw*/ = true,
                                               BlasBuilder blas_builder = BlasBuilder::Cpu)
{
    if (s.environments.size() != envs_info.size()) throw Error(LUPIN_ERR_INVALID_ARGUMENT, "Mismatching sizes for environment data!");
    const size_t nm = s.verts_pos_array.size();
    std::vector<std::vector<uint32_t>> reordered(nm);
    std::vector<std::vector<BvhNode>> bvhs(nm);
    std::vector<float> aabbs(nm * 6);
    std::vector<LupinMeshDesc> meshes(nm);
    for (size_t m = 0; m < nm; m++)
    {
        const auto &v = s.verts_pos_array[m];
        reordered[m] = s.indices_array[m];
        const float *vp = v.empty() ? nullptr : &v[0].x;
        static const float dummy[4] = {0, 0, 0, 0};
        if (!vp) vp = dummy;
        // 2 * triangles - 1 nodes always suffice; BlasBuilder::Device builds the same tree on the GPU (csrc/sahbvh.hip)
        const size_t num_tris = reordered[m].size() / 3;
        bvhs[m].resize(num_tris ? 2 * num_tris - 1 : 1);
        const bool on_device = blas_builder == BlasBuilder::Device && num_tris >= 64;
        const int64_t n = on_device
            ? lupin_hip_build_bvh_sah_device(d.raw(), vp, (uint32_t)v.size(), reordered[m].data(), (uint32_t)reordered[m].size(), bvhs[m].data(), (uint64_t)bvhs[m].size())
            : lupin_build_bvh(vp, (uint32_t)v.size(), reordered[m].data(), (uint32_t)reordered[m].size(), bvhs[m].data(), (uint64_t)bvhs[m].size());
        if (n < 0) throw Error((int)n, "build_bvh failed");
        bvhs[m].resize((size_t)n);
        float lo[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f}, hi[3] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
        for (const Vec4 &p : v) { lo[0] = std::fmin(lo[0], p.x); lo[1] = std::fmin(lo[1], p.y); lo[2] = std::fmin(lo[2], p.z); hi[0] = std::fmax(hi[0], p.x); hi[1] = std::fmax(hi[1], p.y); hi[2] = std::fmax(hi[2], p.z); }
        for (int k = 0; k < 3; k++) { aabbs[m * 6 + k] = lo[k]; aabbs[m * 6 + 3 + k] = hi[k]; }
        meshes[m] = LupinMeshDesc{vp, (uint32_t)v.size(), reordered[m].data(), (uint32_t)reordered[m].size(), bvhs[m].data(), (uint32_t)bvhs[m].size()};
    }
    std::vector<TlasNode> tlas(2 * s.instances.size() + 1);
    int64_t nt = lupin_build_tlas(s.instances.data(), (uint32_t)s.instances.size(), aabbs.data(), (uint32_t)nm, tlas.data());
    if (nt < 0) throw Error((int)nt, "build_tlas failed");
    tlas.resize((size_t)nt);

    // build_lights (data_structures.rs:20-113)
    std::vector<Light> lights;
    std::vector<std::vector<AliasBin>> alias_tables, env_alias_tables;
    for (size_t i = 0; i < s.instances.size(); i++)
    {
        const Instance &in = s.instances[i];
        const Material &mat = s.materials[in.mat_idx];
        const auto &idx = s.indices_array[in.mesh_idx];
        if (mat.emission[0] == 0 && mat.emission[1] == 0 && mat.emission[2] == 0 && mat.emission[3] == 0) continue;
        if (idx.empty()) continue;
        std::vector<float> w(idx.size() / 3);
        float total = lupin_mesh_light_weights(&s.verts_pos_array[in.mesh_idx][0].x, idx.data(), (uint32_t)idx.size(), w.data());
        if (total <= 0.0f) continue;
        std::vector<AliasBin> bins(w.size());
        int64_t nb = lupin_build_alias_table(w.data(), w.size(), bins.data());
        bins.resize((size_t)std::max<int64_t>(nb, 0));
        if (bins.empty()) continue;
        lights.push_back(Light{(uint32_t)i, total});
        alias_tables.push_back(std::move(bins));
    }
    for (size_t i = 0; i < s.environments.size(); i++)
    {
        const EnvMapInfo &e = envs_info[i];
        std::vector<float> w((size_t)e.width * e.height);
        lupin_env_light_weights(&e.data[0].x, e.width, e.height, s.environments[i].emission, w.data());
        std::vector<AliasBin> bins(w.size());
        int64_t nb = lupin_build_alias_table(w.data(), w.size(), bins.data());
        bins.resize((size_t)std::max<int64_t>(nb, 0));
        env_alias_tables.push_back(std::move(bins));
    }

    auto vbufs = [](const std::vector<std::vector<Vec4>> &a) {
        std::vector<LupinVertexBufferDesc> out;
        for (const auto &v : a) out.push_back(LupinVertexBufferDesc{v.empty() ? nullptr : &v[0].x, (uint32_t)v.size()});
        return out;
    };
    std::vector<LupinVertexBufferDesc> nrm = vbufs(s.verts_normal_array), col = vbufs(s.verts_color_array), uvs;
    for (const auto &v : s.verts_texcoord_array) uvs.push_back(LupinVertexBufferDesc{v.data(), (uint32_t)(v.size() / 2)});
    std::vector<LupinTextureDesc> tex;
    for (const TextureCPU &t : textures) tex.push_back(LupinTextureDesc{t.width, t.height, t.format, t.pixels.data()});
    std::vector<LupinAliasTableDesc> at, eat;
    for (const auto &t : alias_tables) at.push_back(LupinAliasTableDesc{t.data(), (uint32_t)t.size()});
    for (const auto &t : env_alias_tables) eat.push_back(LupinAliasTableDesc{t.data(), (uint32_t)t.size()});

    LupinSceneDesc desc{};
    desc.mesh_infos = s.mesh_infos.data(); desc.meshes = meshes.data(); desc.num_meshes = (uint32_t)nm;
    desc.verts_normal_array = nrm.data(); desc.num_normal_buffers = (uint32_t)nrm.size();
    desc.verts_texcoord_array = uvs.data(); desc.num_texcoord_buffers = (uint32_t)uvs.size();
    desc.verts_color_array = col.data(); desc.num_color_buffers = (uint32_t)col.size();
    desc.instances = s.instances.data(); desc.num_instances = (uint32_t)s.instances.size();
    desc.materials = s.materials.data(); desc.num_materials = (uint32_t)s.materials.size();
    desc.textures = tex.data(); desc.num_textures = (uint32_t)tex.size();
    desc.environments = s.environments.data(); desc.num_environments = (uint32_t)s.environments.size();
    desc.tlas_nodes = tlas.data(); desc.num_tlas_nodes = (uint32_t)tlas.size();
    desc.lights = lights.data(); desc.num_lights = (uint32_t)lights.size();
    desc.alias_tables = at.data(); desc.env_alias_tables = eat.data();
    Scene out;
    check(lupin_hip_scene_create(d.raw(), &desc, &out.scene_));
    return out;
}

namespace detail {
inline void fill_desc(const PathtraceDesc &desc, LupinAccumulationParams &ap, LupinTileParams &tp, LupinPathtraceDesc &c)
{
    if (desc.accum_params) { ap.prev_frame = desc.accum_params->prev_frame.raw(); ap.accum_counter = desc.accum_params->accum_counter; c.accum_params = &ap; }
    if (desc.tile_params) { tp.tile_size = desc.tile_params->tile_size; tp.tile_idx = desc.tile_params->tile_idx; c.tile_params = &tp; }
    c.camera_params = LupinCameraParams{desc.camera_params.is_orthographic ? 1u : 0u, desc.camera_params.lens, desc.camera_params.film,
                                        desc.camera_params.aspect, desc.camera_params.focus, desc.camera_params.aperture};
    c.camera_transform = desc.camera_transform;
    c.force_software_bvh = desc.force_software_bvh ? 1u : 0u;
    c.advanced = LupinAdvancedParams{desc.advanced.max_radiance, desc.advanced.rng_seed, desc.advanced.ray_epsilon};
}
}  // namespace detail

// lp::pathtrace_scene (renderer.rs:768-842): enqueues one accumulation frame (or one tile) and returns.
inline void pathtrace_scene(const Device &d, const PathtraceResources &res, const Scene &scene, TextureRef render_target,
                            PathtraceType type, const PathtraceDesc &desc)
{
    LupinAccumulationParams ap{};
    LupinTileParams tp{};
    LupinPathtraceDesc c{};
    detail::fill_desc(desc, ap, tp, c);
    check(lupin_hip_pathtrace_scene(d.raw(), res.raw(), scene.raw(), render_target.raw(), (uint32_t)type, &c));
}

// ---- multi-GPU extension (no counterpart in the reference, which is single-device; tile mathematics renderer.rs:807-829) ----

// All tiles of the frame owned by `rank` (include/lupin_tiles.h) in one launch; desc.tile_params is ignored.
inline void pathtrace_scene_tiles(const Device &d, const PathtraceResources &res, const Scene &scene, TextureRef render_target,
                                  PathtraceType type, const PathtraceDesc &desc, uint32_t tile_size, uint32_t rank, uint32_t world)
{
    LupinAccumulationParams ap{};
    LupinTileParams tp{};
    LupinPathtraceDesc c{};
    detail::fill_desc(desc, ap, tp, c);
    check(lupin_hip_pathtrace_scene_tiles(d.raw(), res.raw(), scene.raw(), render_target.raw(), (uint32_t)type, &c, tile_size, rank, world));
}

// RCCL communicator of one Device; gather_framebuffer = pack own tiles -> ncclAllGather -> scatter the others' tiles.
class Comm
{
  public:
    static std::array<uint8_t, LUPIN_COMM_ID_BYTES> unique_id() { std::array<uint8_t, LUPIN_COMM_ID_BYTES> id{}; check(lupin_hip_comm_get_unique_id(id.data())); return id; }
    Comm(const Device &d, const std::array<uint8_t, LUPIN_COMM_ID_BYTES> &id, uint32_t rank, uint32_t world) { check(lupin_hip_comm_init_rank(d.raw(), id.data(), rank, world, &c_)); }
    explicit Comm(LupinComm *adopted) : c_(adopted) {}
    Comm(Comm &&o) noexcept : c_(o.c_) { o.c_ = nullptr; }
    Comm(const Comm &) = delete;
    ~Comm() { if (c_) lupin_hip_comm_destroy(c_); }
    // one process driving several devices (one context per GPU)
    static std::vector<Comm> init_all(const std::vector<const Device *> &devices)
    {
        std::vector<LupinContext *> ctxs;
        for (const Device *d : devices) ctxs.push_back(d->raw());
        std::vector<LupinComm *> raw(devices.size(), nullptr);
        check(lupin_hip_comm_init_all(ctxs.data(), (uint32_t)ctxs.size(), raw.data()));
        std::vector<Comm> out;
        for (LupinComm *c : raw) out.emplace_back(c);
        return out;
    }
    void gather_framebuffer(TextureRef tex, uint32_t tile_size) const { check(lupin_hip_gather_framebuffer(c_, tex.raw(), tile_size)); }
    // readback form: only `root` receives the other ranks' tiles
    void gather_framebuffer_to(TextureRef tex, uint32_t tile_size, uint32_t root) const { check(lupin_hip_gather_framebuffer_to(c_, tex.raw(), tile_size, root)); }
    void barrier() const { check(lupin_hip_comm_barrier(c_)); }
    uint32_t rank() const { return lupin_hip_comm_rank(c_); }
    uint32_t world() const { return lupin_hip_comm_world(c_); }
    LupinComm *raw() const { return c_; }
  private:
    LupinComm *c_ = nullptr;
};
inline void gather_framebuffer_all(const std::vector<Comm> &comms, const std::vector<TextureRef> &texs, uint32_t tile_size)
{
    std::vector<LupinComm *> c;
    std::vector<LupinTexture *> t;
    for (const Comm &x : comms) c.push_back(x.raw());
    for (const TextureRef &x : texs) t.push_back(x.raw());
    check(lupin_hip_gather_framebuffer_all(c.data(), t.data(), (uint32_t)c.size(), tile_size));
}

// lp::pathtrace_scene_falsecolor (renderer.rs:872-948)
inline void pathtrace_scene_falsecolor(const Device &d, const PathtraceResources &res, const Scene &scene, TextureRef render_target,
                                       FalsecolorType type, const PathtraceDesc &desc)
{
    LupinAccumulationParams ap{};
    LupinTileParams tp{};
    LupinPathtraceDesc c{};
    detail::fill_desc(desc, ap, tp, c);
    check(lupin_hip_pathtrace_scene_falsecolor(d.raw(), res.raw(), scene.raw(), render_target.raw(), (uint32_t)type, &c));
}

// lp::DebugVizType / DebugVizDesc / pathtrace_scene_debug (renderer.rs:950-1041)
enum class DebugVizType : uint32_t { BVHAABBChecks = 0, BVHTriChecks = 1, NumBounces = 2 };
struct DebugVizDesc { DebugVizType viz_type = DebugVizType::BVHAABBChecks; float heatmap_min = 0.0f, heatmap_max = 100.0f; bool first_hit_only = false; };
inline void pathtrace_scene_debug(const Device &d, const PathtraceResources &res, const Scene &scene, TextureRef render_target,
                                  const DebugVizDesc &debug_desc, const PathtraceDesc &desc)
{
    LupinAccumulationParams ap{};
    LupinTileParams tp{};
    LupinPathtraceDesc c{};
    detail::fill_desc(desc, ap, tp, c);
    const LupinDebugVizDesc dd{(uint32_t)debug_desc.viz_type, debug_desc.heatmap_min, debug_desc.heatmap_max, debug_desc.first_hit_only ? 1u : 0u};
    check(lupin_hip_pathtrace_scene_debug(d.raw(), res.raw(), scene.raw(), render_target.raw(), &dd, &c));
}

// lp::Viewport / TonemapDesc / tonemap_and_fit_aspect (tonemapping.rs:106-224); the Rgba8Unorm target is a host image
struct Viewport { float x = 0, y = 0, w = 0, h = 0; };
struct TonemapDesc { std::optional<Viewport> viewport; float exposure = 0.0f; bool filmic = false; bool srgb = true; bool clear = true; };
inline void tonemap_and_fit_aspect(const Device &d, TextureRef src, std::vector<uint8_t> &dst_rgba8, uint32_t dst_width, uint32_t dst_height,
                                   const TonemapDesc &desc = TonemapDesc())
{
    dst_rgba8.resize((size_t)dst_width * dst_height * 4);
    LupinTonemapDesc c{};
    if (desc.viewport) { c.has_viewport = 1; c.viewport_x = desc.viewport->x; c.viewport_y = desc.viewport->y; c.viewport_w = desc.viewport->w; c.viewport_h = desc.viewport->h; }
    c.exposure = desc.exposure; c.filmic = desc.filmic ? 1u : 0u; c.srgb = desc.srgb ? 1u : 0u; c.clear = desc.clear ? 1u : 0u;
    check(lupin_hip_tonemap_and_fit_aspect(d.raw(), src.raw(), dst_rgba8.data(), dst_width, dst_height, &c));
}

}  // namespace lp

namespace lpl {

struct SceneCamera { lp::Mat3x4 transform = lp::mat3x4_identity(); lp::CameraParams params; };   // loader.rs:303-308

// lpl::build_scene_cornell_box (loader.rs:14-207): 8 meshes / instances, 4 materials, values from Yocto/GL.
inline std::pair<lp::Scene, std::vector<SceneCamera>> build_scene_cornell_box(const lp::Device &d, bool build_sw_and_hw = true)
{
    using lp::Vec4;
    lp::SceneCPU s;
    auto mat = [&](float r, float g, float b, float er, float eg, float eb) {
        lp::Material m = lp::default_material();
        if (r >= 0) { m.color[0] = r; m.color[1] = g; m.color[2] = b; m.color[3] = 1.0f; }
        m.emission[0] = er; m.emission[1] = eg; m.emission[2] = eb;
        s.materials.push_back(m);
    };
    mat(0.725f, 0.71f, 0.68f, 0, 0, 0);      // white
    mat(0.63f, 0.065f, 0.05f, 0, 0, 0);      // red
    mat(0.14f, 0.45f, 0.091f, 0, 0, 0);      // green
    mat(-1, 0, 0, 17.0f, 12.0f, 4.0f);       // emissive
    auto mesh = [&](std::vector<Vec4> v, std::vector<uint32_t> idx, uint32_t m) {
        s.mesh_infos.push_back(lp::default_mesh_info());
        s.verts_pos_array.push_back(std::move(v));
        s.indices_array.push_back(std::move(idx));
        lp::Instance in = lp::default_instance();
        in.mesh_idx = (uint32_t)s.mesh_infos.size() - 1;
        in.mat_idx = m;
        s.instances.push_back(in);
    };
    const std::vector<uint32_t> quad = {0, 1, 2, 2, 3, 0}, quad2 = {0, 2, 1, 2, 0, 3};
    const std::vector<uint32_t> box = {0, 2, 1, 2, 0, 3, 4, 6, 5, 6, 4, 7, 8, 10, 9, 10, 8, 11, 12, 14, 13, 14, 12, 15, 16, 18, 17, 18, 16, 19, 20, 22, 21, 22, 20, 23};
    mesh({{-1, 0, 1}, {1, 0, 1}, {1, 0, -1}, {-1, 0, -1}}, quad, 0);                       // floor
    mesh({{-1, 2, 1}, {-1, 2, -1}, {1, 2, -1}, {1, 2, 1}}, quad, 0);                       // ceiling
    mesh({{-1, 0, 1}, {1, 0, 1}, {1, 2, 1}, {-1, 2, 1}}, quad2, 0);                        // back wall
    mesh({{1, 0, -1}, {1, 0, 1}, {1, 2, 1}, {1, 2, -1}}, quad, 2);                         // right wall
    mesh({{-1, 0, 1}, {-1, 0, -1}, {-1, 2, -1}, {-1, 2, 1}}, quad, 1);                     // left wall
    mesh({{0.53f, 0.6f, -0.75f}, {0.7f, 0.6f, -0.17f}, {0.13f, 0.6f, -0.0f}, {-0.05f, 0.6f, -0.57f}, {-0.05f, 0.0f, -0.57f}, {-0.05f, 0.6f, -0.57f},
          {0.13f, 0.6f, -0.0f}, {0.13f, 0.0f, -0.0f}, {0.53f, 0.0f, -0.75f}, {0.53f, 0.6f, -0.75f}, {-0.05f, 0.6f, -0.57f}, {-0.05f, 0.0f, -0.57f},
          {0.7f, 0.0f, -0.17f}, {0.7f, 0.6f, -0.17f}, {0.53f, 0.6f, -0.75f}, {0.53f, 0.0f, -0.75f}, {0.13f, 0.0f, -0.0f}, {0.13f, 0.6f, -0.0f},
          {0.7f, 0.6f, -0.17f}, {0.7f, 0.0f, -0.17f}, {0.53f, 0.0f, -0.75f}, {0.7f, 0.0f, -0.17f}, {0.13f, 0.0f, -0.0f}, {-0.05f, 0.0f, -0.57f}}, box, 0);   // short box
    mesh({{-0.53f, 1.2f, -0.09f}, {0.04f, 1.2f, 0.09f}, {-0.14f, 1.2f, 0.67f}, {-0.71f, 1.2f, 0.49f}, {-0.53f, 0.0f, -0.09f}, {-0.53f, 1.2f, -0.09f},
          {-0.71f, 1.2f, 0.49f}, {-0.71f, 0.0f, 0.49f}, {-0.71f, 0.0f, 0.49f}, {-0.71f, 1.2f, 0.49f}, {-0.14f, 1.2f, 0.67f}, {-0.14f, 0.0f, 0.67f},
          {-0.14f, 0.0f, 0.67f}, {-0.14f, 1.2f, 0.67f}, {0.04f, 1.2f, 0.09f}, {0.04f, 0.0f, 0.09f}, {0.04f, 0.0f, 0.09f}, {0.04f, 1.2f, 0.09f},
          {-0.53f, 1.2f, -0.09f}, {-0.53f, 0.0f, -0.09f}, {-0.53f, 0.0f, -0.09f}, {0.04f, 0.0f, 0.09f}, {-0.14f, 0.0f, 0.67f}, {-0.71f, 0.0f, 0.49f}}, box, 0);   // tall box
    mesh({{-0.25f, 1.99f, -0.25f}, {-0.25f, 1.99f, 0.25f}, {0.25f, 1.99f, 0.25f}, {0.25f, 1.99f, -0.25f}}, quad2, 3);   // light
    lp::validate_scene(s, 0, 0);
    SceneCamera cam;
    cam.transform = lp::Mat3x4{{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 1, -3.9f}}};
    cam.params.is_orthographic = false; cam.params.lens = 0.035f; cam.params.aperture = 0.0f;
    cam.params.focus = 3.9f; cam.params.film = 0.024f; cam.params.aspect = 1.0f;
    return {lp::build_accel_structures_and_upload(d, s, {}, {}, build_sw_and_hw), {cam}};
}

// f16 -> f32 for readback
inline float half_to_float(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1F, man = h & 0x3FFu, u;
    if (exp == 0) { if (!man) u = sign; else { int e = -1; do { man <<= 1; e++; } while (!(man & 0x400u)); man &= 0x3FFu; u = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13); } }
    else if (exp == 31) u = sign | 0x7F800000u | (man << 13);
    else u = sign | ((exp + 112) << 23) | (man << 13);
    float f; std::memcpy(&f, &u, 4); return f;
}

// 8-bit preview of a tonemapped target (lp::tonemap_and_fit_aspect output) as binary PPM; the reference saves Rgba8Unorm
// textures as RGB8 through the `image` crate (loader.rs:1823-1851), PPM keeps this header dependency-free
inline void save_rgba8_ppm(const std::string &path, const std::vector<uint8_t> &rgba8, uint32_t w, uint32_t h)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw lp::Error(LUPIN_ERR_INVALID_ARGUMENT, "cannot open " + path);
    std::fprintf(f, "P6\n%u %u\n255\n", w, h);
    for (size_t i = 0; i < (size_t)w * h; i++) std::fwrite(&rgba8[i * 4], 1, 3, f);
    std::fclose(f);
}

// lpl::save_texture for `.hdr` (loader.rs:1775-1879; alpha dropped): flat RGBE, pixel rule of the `image` crate's encoder
inline void save_texture(const std::string &path, lp::TextureRef tex)
{
    const uint32_t w = tex.width(), h = tex.height();
    const std::vector<uint16_t> px = tex.download();
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw lp::Error(LUPIN_ERR_INVALID_ARGUMENT, "cannot open " + path);
    std::fprintf(f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %u +X %u\n", h, w);
    std::vector<uint8_t> row((size_t)w * 4);
    for (uint32_t y = 0; y < h; y++)
    {
        for (uint32_t x = 0; x < w; x++)
        {
            const uint16_t *p = &px[((size_t)y * w + x) * 4];
            float c[3] = {half_to_float(p[0]), half_to_float(p[1]), half_to_float(p[2])};
            float mx = std::fmax(c[0], std::fmax(c[1], c[2]));
            uint8_t *o = &row[(size_t)x * 4];
            if (!(mx > 0.0f)) { o[0] = o[1] = o[2] = o[3] = 0; continue; }
            int e = (int)std::floor(std::log2(mx)) + 1;
            float mul = std::ldexp(1.0f, e);
            for (int k = 0; k < 3; k++) { float v = std::trunc(c[k] / mul * 256.0f); o[k] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
            o[3] = (uint8_t)(e + 128);
        }
        std::fwrite(row.data(), 1, row.size(), f);
    }
    std::fclose(f);
}

}  // namespace lpl
