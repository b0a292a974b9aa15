//! `lupin_hip` -- the reference's public pathtracing surface over the MI355X backend (liblupin_hip.so).
//!
//! SOURCE ONLY: never compiled in this repository (no Rust toolchain in its build environment). It documents,
//! in the reference's own language, how `lp::` maps onto the C ABI of `include/lupin_hip.h`:
//!
//! | reference (crate `lupin_pt`)                         | here                                   |
//! |------------------------------------------------------|----------------------------------------|
//! | `wgpu::Device` + `wgpu::Queue`                       | [`Device`] (one per GPU)               |
//! | `build_pathtrace_resources`, `BakedPathtraceParams`  | [`build_pathtrace_resources`]          |
//! | `Scene`, `build_accel_structures_and_upload`         | [`Scene::upload`] (takes the CPU-built arrays of `SceneCPU` + BVH / TLAS / lights) |
//! | `DoubleBufferedTexture`                              | [`DoubleBufferedTexture`]              |
//! | `pathtrace_scene`, `PathtraceDesc`, `AccumulationParams`, `TileParams` | [`pathtrace_scene`]  |
//! | `pathtrace_scene_falsecolor`, `pathtrace_scene_debug`| [`pathtrace_scene_falsecolor`], [`pathtrace_scene_debug`] |
//! | `tonemap_and_fit_aspect`, `TonemapDesc`              | [`tonemap_and_fit_aspect`]             |
//!
//! Like the reference, failures panic (the reference asserts / panics; the C ABI returns a status + message).
pub mod ffi;
use ffi::*;
use std::ffi::CStr;
use std::ptr;

fn check(rc: i32) {
    if rc != LUPIN_OK {
        let msg = unsafe { CStr::from_ptr(lupin_hip_last_error()) }.to_string_lossy().into_owned();
        panic!("lupin_hip error {}: {}", rc, msg);
    }
}

/// One GPU: HIP context + streams (replaces `wgpu::Device` + `wgpu::Queue`).
pub struct Device { raw: *mut LupinContext }
impl Device {
    pub fn new(ordinal: i32) -> Device { let mut raw = ptr::null_mut(); check(unsafe { lupin_hip_create_context(ordinal, &mut raw) }); Device { raw } }
    /// `device.poll(wait_indefinitely)`
    pub fn sync(&self) { check(unsafe { lupin_hip_sync(self.raw) }); }
}
impl Drop for Device { fn drop(&mut self) { unsafe { lupin_hip_destroy_context(self.raw) } } }

#[derive(Copy, Clone, Debug)] pub struct BakedPathtraceParams { pub with_runtime_checks: bool, pub max_bounces: u32, pub samples_per_pixel: u32 }
impl Default for BakedPathtraceParams { fn default() -> Self { Self { with_runtime_checks: false, max_bounces: 8, samples_per_pixel: 5 } } }   // renderer.rs:458-468
pub struct PathtraceResources { raw: *mut LupinPathtraceResources }
impl Drop for PathtraceResources { fn drop(&mut self) { unsafe { lupin_hip_destroy_pathtrace_resources(self.raw) } } }
pub fn build_pathtrace_resources(device: &Device, p: &BakedPathtraceParams) -> PathtraceResources {
    let c = LupinBakedPathtraceParams { with_runtime_checks: p.with_runtime_checks as u32, max_bounces: p.max_bounces, samples_per_pixel: p.samples_per_pixel };
    let mut raw = ptr::null_mut();
    check(unsafe { lupin_hip_build_pathtrace_resources(device.raw, &c, &mut raw) });
    PathtraceResources { raw }
}

/// Borrowed view of a render target (a `&wgpu::Texture` in the reference).
#[derive(Copy, Clone)] pub struct Texture { raw: *mut LupinTexture }
impl Texture {
    pub fn width(&self) -> u32 { unsafe { lupin_hip_texture_width(self.raw) } }
    pub fn height(&self) -> u32 { unsafe { lupin_hip_texture_height(self.raw) } }
    /// Rgba16Float payload as raw half bits, row 0 = top (`download_texture`, loader.rs:1640-1700); synchronises.
    pub fn download(&self) -> Vec<u16> {
        let mut v = vec![0u16; (self.width() * self.height() * 4) as usize];
        check(unsafe { lupin_hip_texture_download_rgba16f(self.raw, v.as_mut_ptr()) });
        v
    }
}

pub struct DoubleBufferedTexture { raw: *mut LupinDoubleBufferedTexture }   // wgpu_utils.rs:279-348
impl DoubleBufferedTexture {
    pub fn create(device: &Device, width: u32, height: u32) -> Self { let mut raw = ptr::null_mut(); check(unsafe { lupin_hip_dbuf_create(device.raw, width, height, &mut raw) }); Self { raw } }
    pub fn front(&self) -> Texture { Texture { raw: unsafe { lupin_hip_dbuf_front(self.raw) } } }
    pub fn back(&self) -> Texture { Texture { raw: unsafe { lupin_hip_dbuf_back(self.raw) } } }
    pub fn flip(&mut self) { unsafe { lupin_hip_dbuf_flip(self.raw) } }
    pub fn copy_front_to_back(&self) { check(unsafe { lupin_hip_dbuf_copy_front_to_back(self.raw) }); }
    pub fn resize(&mut self, width: u32, height: u32) { check(unsafe { lupin_hip_dbuf_resize(self.raw, width, height) }); }
}
impl Drop for DoubleBufferedTexture { fn drop(&mut self) { unsafe { lupin_hip_dbuf_destroy(self.raw) } } }

/// The arrays `lp::build_accel_structures_and_upload` produces on the CPU before uploading
/// (data_structures.rs:696-872): `SceneCPU`'s vectors plus per-mesh BVH nodes (with the BVH-reordered index
/// buffers), the TLAS and `LightsCPU`. Build them with Lupin's own builders or with `ffi::lupin_build_*`.
pub struct SceneArrays<'a> {
    pub mesh_infos: &'a [LupinMeshInfo],
    pub verts_pos: &'a [Vec<[f32; 4]>], pub indices: &'a [Vec<u32>], pub bvh_nodes: &'a [Vec<LupinBvhNode>],
    pub verts_normal: &'a [Vec<[f32; 4]>], pub verts_texcoord: &'a [Vec<[f32; 2]>], pub verts_color: &'a [Vec<[f32; 4]>],
    pub instances: &'a [LupinInstance], pub materials: &'a [LupinMaterial], pub environments: &'a [LupinEnvironment],
    pub textures: &'a [LupinTextureDesc],
    pub tlas_nodes: &'a [LupinTlasNode], pub lights: &'a [LupinLight],
    pub alias_tables: &'a [Vec<LupinAliasBin>], pub env_alias_tables: &'a [Vec<LupinAliasBin>],
}
pub struct Scene { raw: *mut LupinScene }
impl Scene {
    pub fn upload(device: &Device, s: &SceneArrays) -> Scene {
        let meshes: Vec<LupinMeshDesc> = (0..s.verts_pos.len()).map(|i| LupinMeshDesc {
            verts_pos: s.verts_pos[i].as_ptr() as *const f32, num_verts: s.verts_pos[i].len() as u32,
            indices: s.indices[i].as_ptr(), num_indices: s.indices[i].len() as u32,
            bvh_nodes: s.bvh_nodes[i].as_ptr(), num_bvh_nodes: s.bvh_nodes[i].len() as u32 }).collect();
        let vb4 = |v: &[Vec<[f32; 4]>]| -> Vec<LupinVertexBufferDesc> { v.iter().map(|b| LupinVertexBufferDesc { data: b.as_ptr() as *const f32, num_verts: b.len() as u32 }).collect() };
        let normals = vb4(s.verts_normal);
        let colors = vb4(s.verts_color);
        let uvs: Vec<LupinVertexBufferDesc> = s.verts_texcoord.iter().map(|b| LupinVertexBufferDesc { data: b.as_ptr() as *const f32, num_verts: b.len() as u32 }).collect();
        let tables = |t: &[Vec<LupinAliasBin>]| -> Vec<LupinAliasTableDesc> { t.iter().map(|b| LupinAliasTableDesc { bins: b.as_ptr(), num_bins: b.len() as u32 }).collect() };
        let (alias, env_alias) = (tables(s.alias_tables), tables(s.env_alias_tables));
        let desc = LupinSceneDesc {
            mesh_infos: s.mesh_infos.as_ptr(), meshes: meshes.as_ptr(), num_meshes: meshes.len() as u32,
            verts_normal_array: normals.as_ptr(), num_normal_buffers: normals.len() as u32,
            verts_texcoord_array: uvs.as_ptr(), num_texcoord_buffers: uvs.len() as u32,
            verts_color_array: colors.as_ptr(), num_color_buffers: colors.len() as u32,
            instances: s.instances.as_ptr(), num_instances: s.instances.len() as u32,
            materials: s.materials.as_ptr(), num_materials: s.materials.len() as u32,
            textures: s.textures.as_ptr(), num_textures: s.textures.len() as u32,
            environments: s.environments.as_ptr(), num_environments: s.environments.len() as u32,
            tlas_nodes: s.tlas_nodes.as_ptr(), num_tlas_nodes: s.tlas_nodes.len() as u32,
            lights: s.lights.as_ptr(), num_lights: s.lights.len() as u32,
            alias_tables: alias.as_ptr(), env_alias_tables: env_alias.as_ptr(),
        };
        let mut raw = ptr::null_mut();
        check(unsafe { lupin_hip_scene_create(device.raw, &desc, &mut raw) });
        Scene { raw }
    }
}
impl Drop for Scene { fn drop(&mut self) { unsafe { lupin_hip_scene_destroy(self.raw) } } }

#[repr(u32)] #[derive(Copy, Clone, Debug, PartialEq)] pub enum PathtraceType { Standard = 0, MIS = 1, Naive = 2, Direct = 3 }   // renderer.rs:711-729
#[repr(u32)] #[derive(Copy, Clone, Debug, PartialEq)]
pub enum FalsecolorType { Albedo = 0, Normals, NormalsUnsigned, FrontFacing, Emission, Roughness, Metallic, Opacity, MatType, IsDelta, Instance, Tri }
#[repr(u32)] #[derive(Copy, Clone, Debug, PartialEq)] pub enum DebugVizType { BVHAABBChecks = 0, BVHTriChecks = 1, NumBounces = 2 }
#[derive(Copy, Clone)] pub struct DebugVizDesc { pub viz_type: DebugVizType, pub heatmap_min: f32, pub heatmap_max: f32, pub first_hit_only: bool }

#[derive(Copy, Clone)] pub struct CameraParams { pub is_orthographic: bool, pub lens: f32, pub film: f32, pub aspect: f32, pub focus: f32, pub aperture: f32 }
impl Default for CameraParams { fn default() -> Self { Self { is_orthographic: false, lens: 0.050, film: 0.036, aspect: 1.5, focus: 10000.0, aperture: 0.0 } } }   // renderer.rs:695-707
#[derive(Copy, Clone)] pub struct AdvancedParams { pub max_radiance: f32, pub rng_seed: u32, pub ray_epsilon: f32 }
impl Default for AdvancedParams { fn default() -> Self { Self { max_radiance: 100.0, rng_seed: 0, ray_epsilon: 0.001 } } }   // renderer.rs:739-749
#[derive(Copy, Clone)] pub struct TileParams { pub tile_size: u32, pub tile_idx: u32 }
#[derive(Copy, Clone)] pub struct AccumulationParams { pub prev_frame: Texture, pub accum_counter: u32 }
pub struct PathtraceDesc<'a> {
    pub accum_params: Option<AccumulationParams>, pub tile_params: Option<&'a TileParams>,
    pub camera_params: CameraParams, pub camera_transform: LupinMat3x4, pub force_software_bvh: bool, pub advanced: AdvancedParams,
}

fn with_desc<R>(desc: &PathtraceDesc, f: impl FnOnce(*const LupinPathtraceDesc) -> R) -> R {
    let accum = desc.accum_params.map(|a| LupinAccumulationParams { prev_frame: a.prev_frame.raw, accum_counter: a.accum_counter });
    let tile = desc.tile_params.map(|t| LupinTileParams { tile_size: t.tile_size, tile_idx: t.tile_idx });
    let cp = &desc.camera_params;
    let c = LupinPathtraceDesc {
        accum_params: accum.as_ref().map_or(ptr::null(), |a| a as *const _),
        tile_params: tile.as_ref().map_or(ptr::null(), |t| t as *const _),
        camera_params: LupinCameraParams { is_orthographic: cp.is_orthographic as u32, lens: cp.lens, film: cp.film, aspect: cp.aspect, focus: cp.focus, aperture: cp.aperture },
        camera_transform: desc.camera_transform,
        force_software_bvh: 1,   // the HIP backend is the software-BVH path
        advanced: LupinAdvancedParams { max_radiance: desc.advanced.max_radiance, rng_seed: desc.advanced.rng_seed, ray_epsilon: desc.advanced.ray_epsilon },
    };
    f(&c)
}

pub fn get_num_tiles(tile_size: u32, width: u32, height: u32) -> u32 { unsafe { lupin_hip_get_num_tiles(tile_size, width, height) } }

/// `lp::pathtrace_scene` (renderer.rs:768): enqueues one accumulation frame (or one tile) and returns.
pub fn pathtrace_scene(device: &Device, resources: &PathtraceResources, scene: &Scene, render_target: Texture, pathtrace_type: PathtraceType, desc: &PathtraceDesc) {
    with_desc(desc, |c| check(unsafe { lupin_hip_pathtrace_scene(device.raw, resources.raw, scene.raw, render_target.raw, pathtrace_type as u32, c) }));
}
/// `lp::pathtrace_scene_falsecolor` (renderer.rs:872)
pub fn pathtrace_scene_falsecolor(device: &Device, resources: &PathtraceResources, scene: &Scene, render_target: Texture, falsecolor_type: FalsecolorType, desc: &PathtraceDesc) {
    with_desc(desc, |c| check(unsafe { lupin_hip_pathtrace_scene_falsecolor(device.raw, resources.raw, scene.raw, render_target.raw, falsecolor_type as u32, c) }));
}
/// `lp::pathtrace_scene_debug` (renderer.rs:966)
pub fn pathtrace_scene_debug(device: &Device, resources: &PathtraceResources, scene: &Scene, render_target: Texture, debug_desc: &DebugVizDesc, desc: &PathtraceDesc) {
    let d = LupinDebugVizDesc { viz_type: debug_desc.viz_type as u32, heatmap_min: debug_desc.heatmap_min, heatmap_max: debug_desc.heatmap_max, first_hit_only: debug_desc.first_hit_only as u32 };
    with_desc(desc, |c| check(unsafe { lupin_hip_pathtrace_scene_debug(device.raw, resources.raw, scene.raw, render_target.raw, &d, c) }));
}
/// Multi-GPU extension: every tile `t` with `t % world == rank` of one frame in one launch (one `Device` per GPU).
pub fn pathtrace_scene_tiles(device: &Device, resources: &PathtraceResources, scene: &Scene, render_target: Texture, pathtrace_type: PathtraceType, desc: &PathtraceDesc, tile_size: u32, rank: u32, world: u32) {
    with_desc(desc, |c| check(unsafe { lupin_hip_pathtrace_scene_tiles(device.raw, resources.raw, scene.raw, render_target.raw, pathtrace_type as u32, c, tile_size, rank, world) }));
}

#[derive(Copy, Clone, Default)] pub struct Viewport { pub x: f32, pub y: f32, pub w: f32, pub h: f32 }
#[derive(Copy, Clone)] pub struct TonemapDesc { pub viewport: Option<Viewport>, pub exposure: f32, pub filmic: bool, pub srgb: bool, pub clear: bool }
impl Default for TonemapDesc { fn default() -> Self { Self { viewport: None, exposure: 0.0, filmic: false, srgb: true, clear: true } } }   // tonemapping.rs:120-132
/// `lp::tonemap_and_fit_aspect` (tonemapping.rs:155) into a host Rgba8Unorm image (`dst.len() == w * h * 4`).
pub fn tonemap_and_fit_aspect(device: &Device, src: Texture, dst: &mut [u8], dst_width: u32, dst_height: u32, desc: &TonemapDesc) {
    assert_eq!(dst.len(), (dst_width * dst_height * 4) as usize);
    let v = desc.viewport.unwrap_or_default();
    let c = LupinTonemapDesc { has_viewport: desc.viewport.is_some() as u32, viewport_x: v.x, viewport_y: v.y, viewport_w: v.w, viewport_h: v.h,
                               exposure: desc.exposure, filmic: desc.filmic as u32, srgb: desc.srgb as u32, clear: desc.clear as u32 };
    check(unsafe { lupin_hip_tonemap_and_fit_aspect(device.raw, src.raw, dst.as_mut_ptr(), dst_width, dst_height, &c) });
}
