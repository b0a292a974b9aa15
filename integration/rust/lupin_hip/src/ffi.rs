//! Raw declarations of include/lupin_hip.h. Every `#[repr(C)]` struct has the byte layout of the C struct of the same
//! name, which in turn equals the reference's own `#[repr(C)]` upload types (lupin/src/renderer.rs:94-280).
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

macro_rules! opaque { ($($n:ident),*) => { $(#[repr(C)] pub struct $n { _p: [u8; 0] })* } }
opaque!(LupinContext, LupinPathtraceResources, LupinScene, LupinTexture, LupinDoubleBufferedTexture, LupinComm);

pub const LUPIN_OK: c_int = 0;
pub const LUPIN_SENTINEL_IDX: u32 = 0xFFFF_FFFF;

#[repr(C)] #[derive(Copy, Clone, Default)] pub struct LupinMat3x4 { pub m: [[f32; 3]; 4] }          // base.rs:634-657: 4 columns x 3 rows
#[repr(C)] #[derive(Copy, Clone, Default)] pub struct LupinMat4x3 { pub m: [[f32; 4]; 3] }          // base.rs:763-768
#[repr(C)] #[derive(Copy, Clone, Default)] pub struct LupinMat4 { pub m: [[f32; 4]; 4] }
#[repr(C)] #[derive(Copy, Clone)] pub struct LupinMeshInfo { pub normals_buf_idx: u32, pub texcoords_buf_idx: u32, pub colors_buf_idx: u32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct LupinInstance { pub transpose_inverse_transform: LupinMat4x3, pub mesh_idx: u32, pub mat_idx: u32, pub padding0: f32, pub padding1: f32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct LupinMaterial {
    pub color: [f32; 4], pub emission: [f32; 4], pub scattering: [f32; 4],
    pub mat_type: u32, pub roughness: f32, pub metallic: f32, pub ior: f32, pub sc_anisotropy: f32, pub tr_depth: f32,
    pub color_tex_idx: u32, pub emission_tex_idx: u32, pub roughness_tex_idx: u32, pub scattering_tex_idx: u32, pub normal_tex_idx: u32, pub padding0: u32,
}
#[repr(C)] #[derive(Copy, Clone)] pub struct LupinEnvironment { pub emission: [f32; 3], pub emission_tex_idx: u32, pub transform: LupinMat4 }
#[repr(C)] #[derive(Copy, Clone)] pub struct LupinLight { pub instance_idx: u32, pub area: f32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct LupinAliasBin { pub prob: f32, pub alias_threshold: f32, pub alias: u32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct LupinBvhNode { pub aabb_min: [f32; 3], pub tri_begin_or_first_child: u32, pub aabb_max: [f32; 3], pub tri_count: u32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct LupinTlasNode { pub aabb_min: [f32; 3], pub left: u32, pub aabb_max: [f32; 3], pub instance_idx: u32, pub right: u32, pub padding: [f32; 3] }

#[repr(C)] pub struct LupinTextureDesc { pub width: u32, pub height: u32, pub format: u32, pub pixels: *const c_void }   // format: 0 Rgba8Unorm, 1 Rgba16Float
#[repr(C)] pub struct LupinMeshDesc { pub verts_pos: *const f32, pub num_verts: u32, pub indices: *const u32, pub num_indices: u32, pub bvh_nodes: *const LupinBvhNode, pub num_bvh_nodes: u32 }
#[repr(C)] pub struct LupinVertexBufferDesc { pub data: *const f32, pub num_verts: u32 }
#[repr(C)] pub struct LupinAliasTableDesc { pub bins: *const LupinAliasBin, pub num_bins: u32 }
#[repr(C)] pub struct LupinSceneDesc {
    pub mesh_infos: *const LupinMeshInfo, pub meshes: *const LupinMeshDesc, pub num_meshes: u32,
    pub verts_normal_array: *const LupinVertexBufferDesc, pub num_normal_buffers: u32,
    pub verts_texcoord_array: *const LupinVertexBufferDesc, pub num_texcoord_buffers: u32,
    pub verts_color_array: *const LupinVertexBufferDesc, pub num_color_buffers: u32,
    pub instances: *const LupinInstance, pub num_instances: u32,
    pub materials: *const LupinMaterial, pub num_materials: u32,
    pub textures: *const LupinTextureDesc, pub num_textures: u32,
    pub environments: *const LupinEnvironment, pub num_environments: u32,
    pub tlas_nodes: *const LupinTlasNode, pub num_tlas_nodes: u32,
    pub lights: *const LupinLight, pub num_lights: u32,
    pub alias_tables: *const LupinAliasTableDesc, pub env_alias_tables: *const LupinAliasTableDesc,
}

#[repr(C)] #[derive(Copy, Clone)] pub struct LupinBakedPathtraceParams { pub with_runtime_checks: u32, pub max_bounces: u32, pub samples_per_pixel: u32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct LupinCameraParams { pub is_orthographic: u32, pub lens: f32, pub film: f32, pub aspect: f32, pub focus: f32, pub aperture: f32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct LupinAdvancedParams { pub max_radiance: f32, pub rng_seed: u32, pub ray_epsilon: f32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct LupinTileParams { pub tile_size: u32, pub tile_idx: u32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct LupinAccumulationParams { pub prev_frame: *const LupinTexture, pub accum_counter: u32 }
#[repr(C)] pub struct LupinPathtraceDesc {
    pub accum_params: *const LupinAccumulationParams, pub tile_params: *const LupinTileParams,
    pub camera_params: LupinCameraParams, pub camera_transform: LupinMat3x4, pub force_software_bvh: u32, pub advanced: LupinAdvancedParams,
}
#[repr(C)] #[derive(Copy, Clone)] pub struct LupinDebugVizDesc { pub viz_type: u32, pub heatmap_min: f32, pub heatmap_max: f32, pub first_hit_only: u32 }
#[repr(C)] #[derive(Copy, Clone)] pub struct LupinTonemapDesc {
    pub has_viewport: u32, pub viewport_x: f32, pub viewport_y: f32, pub viewport_w: f32, pub viewport_h: f32,
    pub exposure: f32, pub filmic: u32, pub srgb: u32, pub clear: u32,
}

extern "C" {
    pub fn lupin_hip_last_error() -> *const c_char;
    pub fn lupin_hip_device_count() -> c_int;
    pub fn lupin_hip_create_context(device_ordinal: c_int, out: *mut *mut LupinContext) -> c_int;
    pub fn lupin_hip_destroy_context(ctx: *mut LupinContext);
    pub fn lupin_hip_sync(ctx: *mut LupinContext) -> c_int;
    pub fn lupin_hip_build_pathtrace_resources(ctx: *mut LupinContext, p: *const LupinBakedPathtraceParams, out: *mut *mut LupinPathtraceResources) -> c_int;
    pub fn lupin_hip_destroy_pathtrace_resources(res: *mut LupinPathtraceResources);
    pub fn lupin_hip_scene_create(ctx: *mut LupinContext, desc: *const LupinSceneDesc, out: *mut *mut LupinScene) -> c_int;
    pub fn lupin_hip_scene_destroy(scene: *mut LupinScene);
    pub fn lupin_hip_texture_create(ctx: *mut LupinContext, w: u32, h: u32, out: *mut *mut LupinTexture) -> c_int;
    pub fn lupin_hip_texture_destroy(tex: *mut LupinTexture);
    pub fn lupin_hip_texture_width(tex: *const LupinTexture) -> u32;
    pub fn lupin_hip_texture_height(tex: *const LupinTexture) -> u32;
    pub fn lupin_hip_texture_upload_rgba16f(tex: *mut LupinTexture, pixels: *const u16) -> c_int;
    pub fn lupin_hip_texture_download_rgba16f(tex: *const LupinTexture, out_pixels: *mut u16) -> c_int;
    pub fn lupin_hip_dbuf_create(ctx: *mut LupinContext, w: u32, h: u32, out: *mut *mut LupinDoubleBufferedTexture) -> c_int;
    pub fn lupin_hip_dbuf_destroy(t: *mut LupinDoubleBufferedTexture);
    pub fn lupin_hip_dbuf_front(t: *mut LupinDoubleBufferedTexture) -> *mut LupinTexture;
    pub fn lupin_hip_dbuf_back(t: *mut LupinDoubleBufferedTexture) -> *mut LupinTexture;
    pub fn lupin_hip_dbuf_flip(t: *mut LupinDoubleBufferedTexture);
    pub fn lupin_hip_dbuf_copy_front_to_back(t: *mut LupinDoubleBufferedTexture) -> c_int;
    pub fn lupin_hip_dbuf_resize(t: *mut LupinDoubleBufferedTexture, w: u32, h: u32) -> c_int;
    pub fn lupin_hip_get_num_tiles(tile_size: u32, w: u32, h: u32) -> u32;
    pub fn lupin_hip_pathtrace_scene(ctx: *mut LupinContext, res: *const LupinPathtraceResources, scene: *const LupinScene,
                                     target: *mut LupinTexture, pathtrace_type: u32, desc: *const LupinPathtraceDesc) -> c_int;
    pub fn lupin_hip_pathtrace_scene_falsecolor(ctx: *mut LupinContext, res: *const LupinPathtraceResources, scene: *const LupinScene,
                                                target: *mut LupinTexture, falsecolor_type: u32, desc: *const LupinPathtraceDesc) -> c_int;
    pub fn lupin_hip_pathtrace_scene_debug(ctx: *mut LupinContext, res: *const LupinPathtraceResources, scene: *const LupinScene,
                                           target: *mut LupinTexture, debug_desc: *const LupinDebugVizDesc, desc: *const LupinPathtraceDesc) -> c_int;
    pub fn lupin_hip_pathtrace_scene_tiles(ctx: *mut LupinContext, res: *const LupinPathtraceResources, scene: *const LupinScene,
                                           target: *mut LupinTexture, pathtrace_type: u32, desc: *const LupinPathtraceDesc,
                                           tile_size: u32, rank: u32, world: u32) -> c_int;
    // multi-GPU: the one exchange step (RCCL all-gather of tile payloads over xGMI) and its communicators
    pub fn lupin_hip_comm_get_unique_id(out_id: *mut u8) -> c_int;
    pub fn lupin_hip_comm_init_rank(ctx: *mut LupinContext, id: *const u8, rank: u32, world: u32, out: *mut *mut LupinComm) -> c_int;
    pub fn lupin_hip_comm_init_all(ctxs: *const *mut LupinContext, n: u32, out_comms: *mut *mut LupinComm) -> c_int;
    pub fn lupin_hip_comm_from_nccl(ctx: *mut LupinContext, nccl_comm: *mut c_void, rank: u32, world: u32, out: *mut *mut LupinComm) -> c_int;
    pub fn lupin_hip_comm_destroy(comm: *mut LupinComm);
    pub fn lupin_hip_comm_rank(comm: *const LupinComm) -> u32;
    pub fn lupin_hip_comm_world(comm: *const LupinComm) -> u32;
    pub fn lupin_hip_gather_framebuffer(comm: *mut LupinComm, tex: *mut LupinTexture, tile_size: u32) -> c_int;
    pub fn lupin_hip_gather_framebuffer_all(comms: *const *mut LupinComm, texs: *const *mut LupinTexture, n: u32, tile_size: u32) -> c_int;
    pub fn lupin_hip_gather_framebuffer_to(comm: *mut LupinComm, tex: *mut LupinTexture, tile_size: u32, root: u32) -> c_int;
    pub fn lupin_hip_comm_barrier(comm: *mut LupinComm) -> c_int;
    pub fn lupin_hip_comm_allreduce_f64(comm: *mut LupinComm, inout: *mut f64, n: u32, op: u32) -> c_int;
    // accumulation mode (f16 running average = reference, or f32 accumulator) and its readback
    pub fn lupin_hip_set_accumulation_mode(ctx: *mut LupinContext, mode: c_int) -> c_int;
    pub fn lupin_hip_reserve_path_state(ctx: *mut LupinContext, pixels: u64, max_bounces: u32, samples_per_pixel: u32) -> c_int;
    pub fn lupin_hip_set_batch_frames(ctx: *mut LupinContext, frames: u32) -> c_int;
    pub fn lupin_hip_set_traversal(ctx: *mut LupinContext, mode: c_int) -> c_int;
    pub fn lupin_hip_texture_download_rgba32f(tex: *const LupinTexture, out_pixels: *mut f32) -> c_int;
    pub fn lupin_hip_tonemap_and_fit_aspect(ctx: *mut LupinContext, src: *const LupinTexture, dst_rgba8: *mut u8, w: u32, h: u32,
                                            desc: *const LupinTonemapDesc) -> c_int;
    // host-side builders with the results of lupin/src/data_structures.rs
    pub fn lupin_build_bvh(verts_pos4: *const f32, num_verts: u32, indices: *mut u32, num_indices: u32, out_nodes: *mut LupinBvhNode, cap: u64) -> i64;
    // the same tree built on the GPU (csrc/sahbvh.hip); cap = 2 * triangles - 1 always suffices
    pub fn lupin_hip_build_bvh_sah_device(ctx: *mut LupinContext, verts_pos4: *const f32, num_verts: u32, indices: *mut u32, num_indices: u32,
                                          out_nodes: *mut LupinBvhNode, cap: u64) -> i64;
    pub fn lupin_build_tlas(instances: *const LupinInstance, n: u32, model_aabbs: *const f32, num_meshes: u32, out: *mut LupinTlasNode) -> i64;
    pub fn lupin_build_alias_table(weights: *const f32, n: u64, out_bins: *mut LupinAliasBin) -> i64;
}
