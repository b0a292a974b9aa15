/*
 * lupin_hip.h -- C ABI of the MI355X (gfx950) software-BVH path tracer that stands in for
 * LupinPathTracer's `lp::pathtrace_scene()` hot path.
 *
 * Every struct below is byte-identical to the `#[repr(C)]` type the reference uploads to its
 * WGSL megakernel, and every entry point names the reference interface it replaces
 * (paths are relative to the reference checkout, `lupin/src/...`).
 *
 * Conventions (reference: renderer.rs:754-766, :768-842):
 *   - handles are opaque, not thread-safe, one context per GPU;
 *   - `lupin_hip_pathtrace_scene` ENQUEUES on the context's HIP stream and returns (the
 *     reference does `queue.submit` and returns, renderer.rs:841); `lupin_hip_sync` or any
 *     download is the sync point;
 *   - no panics across the ABI: every call returns LUPIN_OK or a negative error code and
 *     `lupin_hip_last_error()` holds the message (the reference asserts / panics instead,
 *     renderer.rs:770,776,814);
 *   - matrices are column-major f32, little-endian, exactly as base.rs:500-800 lays them out.
 */
#ifndef LUPIN_HIP_H
#define LUPIN_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * Scene record layouts (renderer.rs:94-250 == pathtracer.wgsl:88-178)
 * ---------------------------------------------------------------------------------------- */

#define LUPIN_SENTINEL_IDX 0xFFFFFFFFu /* renderer.rs SENTINEL_IDX / pathtracer.wgsl:66 */

/* base.rs:634  Mat3x4 { m: [[f32;3];4] }  -- 4 columns x 3 rows, last column = translation */
typedef struct LupinMat3x4 { float m[4][3]; } LupinMat3x4;
/* base.rs:763  Mat4x3 { m: [[f32;4];3] }  -- 3 columns x 4 rows */
typedef struct LupinMat4x3 { float m[3][4]; } LupinMat4x3;
/* base.rs:505  Mat4 { m: [[f32;4];4] } column-major */
typedef struct LupinMat4 { float m[4][4]; } LupinMat4;

/* renderer.rs:94-100 */
typedef struct LupinMeshInfo {
    uint32_t normals_buf_idx;
    uint32_t texcoords_buf_idx;
    uint32_t colors_buf_idx;
} LupinMeshInfo;

/* renderer.rs:115-124 ; 64 bytes. transpose_inverse_transform = rows of the world->local affine. */
typedef struct LupinInstance {
    LupinMat4x3 transpose_inverse_transform;
    uint32_t mesh_idx;
    uint32_t mat_idx;
    float _padding0;
    float _padding1;
} LupinInstance;

/* renderer.rs:126-139 */
enum LupinMaterialType {
    LUPIN_MAT_MATTE = 0,
    LUPIN_MAT_GLOSSY = 1,
    LUPIN_MAT_REFLECTIVE = 2,
    LUPIN_MAT_TRANSPARENT = 3,
    LUPIN_MAT_REFRACTIVE = 4,
    LUPIN_MAT_SUBSURFACE = 5,
    LUPIN_MAT_VOLUMETRIC = 6,
    LUPIN_MAT_GLTFPBR = 7
};

/* renderer.rs:141-161 ; 96 bytes */
typedef struct LupinMaterial {
    float color[4];      /* w = opacity */
    float emission[4];
    float scattering[4];
    uint32_t mat_type;
    float roughness;
    float metallic;
    float ior;
    float sc_anisotropy;
    float tr_depth;
    uint32_t color_tex_idx;
    uint32_t emission_tex_idx;
    uint32_t roughness_tex_idx;
    uint32_t scattering_tex_idx;
    uint32_t normal_tex_idx;
    uint32_t padding0;
} LupinMaterial;

/* renderer.rs:187-194 ; 80 bytes */
typedef struct LupinEnvironment {
    float emission[3];
    uint32_t emission_tex_idx;
    LupinMat4 transform;
} LupinEnvironment;

/* renderer.rs:208-214 */
typedef struct LupinLight {
    uint32_t instance_idx;
    float area;
} LupinLight;

/* renderer.rs:216-223 */
typedef struct LupinAliasBin {
    float prob;
    float alias_threshold;
    uint32_t alias;
} LupinAliasBin;

/* renderer.rs:228-238 ; 32 bytes. tri_count == 0 => internal node, children at first_child, +1 */
typedef struct LupinBvhNode {
    float aabb_min[3];
    uint32_t tri_begin_or_first_child;
    float aabb_max[3];
    uint32_t tri_count;
} LupinBvhNode;

/* renderer.rs:240-250 ; 48 bytes. left == 0 => leaf */
typedef struct LupinTlasNode {
    float aabb_min[3];
    uint32_t left;
    float aabb_max[3];
    uint32_t instance_idx;
    uint32_t right;
    float _padding0[3];
} LupinTlasNode;

/* renderer.rs:252-280 ; 128 bytes. Exposed because the oracle and the kernels consume exactly
 * this record; hosts normally never build it (lupin_hip_pathtrace_scene does, like
 * get_push_constants{,_tiled}, renderer.rs:1426-1506). */
typedef struct LupinPushConstants {
    LupinMat4 camera_transform;
    float camera_lens;
    float camera_film;
    float camera_aspect;
    float camera_focus;
    float camera_aperture;
    uint32_t flags;
    uint32_t id_offset[2];
    uint32_t accum_counter;
    float heatmap_min;
    float heatmap_max;
    uint32_t falsecolor_type;
    uint32_t pathtrace_type;
    float max_radiance;
    uint32_t rng_seed;   /* never read by the shader (pathtracer.wgsl:1565) -- kept for layout */
    float ray_epsilon;
} LupinPushConstants;

/* renderer.rs:284-291 */
#define LUPIN_FLAG_CAMERA_ORTHO         (1u << 0)
#define LUPIN_FLAG_ENVS_EMPTY           (1u << 1)
#define LUPIN_FLAG_LIGHTS_EMPTY         (1u << 2)
#define LUPIN_FLAG_DEBUG_TRI_CHECKS     (1u << 3)
#define LUPIN_FLAG_DEBUG_AABB_CHECKS    (1u << 4)
#define LUPIN_FLAG_DEBUG_NUM_BOUNCES    (1u << 5)
#define LUPIN_FLAG_DEBUG_FIRST_HIT_ONLY (1u << 6)
#define LUPIN_FLAG_INSTANCES_EMPTY      (1u << 7)

/* renderer.rs:294-305 */
#define LUPIN_BVH_MAX_DEPTH   25
#define LUPIN_TLAS_MAX_DEPTH  50
#define LUPIN_WORKGROUP_SIZE  4   /* tile_size is counted in 4x4-pixel workgroups */
#define LUPIN_MAX_ENVS        10

/* Texture formats the loader produces (lupin_loader/src/loader.rs:227-231): LDR = Rgba8Unorm
 * (never ...Srgb, decode happens in the shader), HDR = Rgba16Float. One mip, bilinear, Repeat
 * in u and v (wgpu_utils.rs:244-256). */
enum LupinTextureFormat {
    LUPIN_TEX_RGBA8_UNORM = 0,
    LUPIN_TEX_RGBA16_FLOAT = 1
};

typedef struct LupinTextureDesc {
    uint32_t width;
    uint32_t height;
    uint32_t format;       /* LupinTextureFormat */
    const void *pixels;    /* host pointer, row-major, 4 B or 8 B per texel */
} LupinTextureDesc;

/* One mesh = the per-mesh storage buffers of lp::Scene in its software-BVH configuration
 * (renderer.rs:17-60): positions (Vec4, stride 16), BVH-reordered indices, BLAS nodes. */
typedef struct LupinMeshDesc {
    const float *verts_pos;          /* num_verts * 4 floats (w ignored) */
    uint32_t num_verts;
    const uint32_t *indices;         /* num_indices u32, triangles in BLAS leaf order */
    uint32_t num_indices;
    const LupinBvhNode *bvh_nodes;
    uint32_t num_bvh_nodes;
} LupinMeshDesc;

typedef struct LupinVertexBufferDesc {
    const float *data;               /* normals/colours: 4 floats per vertex; texcoords: 2 */
    uint32_t num_verts;
} LupinVertexBufferDesc;

typedef struct LupinAliasTableDesc {
    const LupinAliasBin *bins;
    uint32_t num_bins;
} LupinAliasTableDesc;

/* Flat restatement of lp::Scene (renderer.rs:17-60) for the software-BVH pipeline. All pointers
 * are host pointers, copied during lupin_hip_scene_create. */
typedef struct LupinSceneDesc {
    const LupinMeshInfo *mesh_infos;         /* num_meshes entries */
    const LupinMeshDesc *meshes;
    uint32_t num_meshes;

    const LupinVertexBufferDesc *verts_normal_array;
    uint32_t num_normal_buffers;
    const LupinVertexBufferDesc *verts_texcoord_array;
    uint32_t num_texcoord_buffers;
    const LupinVertexBufferDesc *verts_color_array;
    uint32_t num_color_buffers;

    const LupinInstance *instances;
    uint32_t num_instances;
    const LupinMaterial *materials;
    uint32_t num_materials;
    const LupinTextureDesc *textures;
    uint32_t num_textures;
    const LupinEnvironment *environments;
    uint32_t num_environments;

    const LupinTlasNode *tlas_nodes;
    uint32_t num_tlas_nodes;

    const LupinLight *lights;
    uint32_t num_lights;
    const LupinAliasTableDesc *alias_tables;      /* num_lights tables */
    const LupinAliasTableDesc *env_alias_tables;  /* num_environments tables */
} LupinSceneDesc;

/* ------------------------------------------------------------------------------------------
 * Call-surface structs (renderer.rs:451-468, :644-766)
 * ---------------------------------------------------------------------------------------- */

/* renderer.rs:451-468 (defaults false, 8, 5) */
typedef struct LupinBakedPathtraceParams {
    uint32_t with_runtime_checks;   /* accepted, no effect: HIP kernels have no naga bounds checks */
    uint32_t max_bounces;
    uint32_t samples_per_pixel;
} LupinBakedPathtraceParams;

/* renderer.rs:683-708 (defaults 0, .050, .036, 1.5, 10000, 0) */
typedef struct LupinCameraParams {
    uint32_t is_orthographic;
    float lens;
    float film;
    float aspect;
    float focus;
    float aperture;
} LupinCameraParams;

/* renderer.rs:711-729 */
enum LupinPathtraceType {
    LUPIN_PATHTRACE_STANDARD = 0,
    LUPIN_PATHTRACE_MIS = 1,
    LUPIN_PATHTRACE_NAIVE = 2,
    LUPIN_PATHTRACE_DIRECT = 3
};

/* renderer.rs:731-749 (defaults 100, 0, 0.001) */
typedef struct LupinAdvancedParams {
    float max_radiance;
    uint32_t rng_seed;
    float ray_epsilon;
} LupinAdvancedParams;

/* renderer.rs:651-670 (defaults 100, 0) */
typedef struct LupinTileParams {
    uint32_t tile_size;   /* in 4-pixel workgroups */
    uint32_t tile_idx;
} LupinTileParams;

typedef struct LupinContext LupinContext;
typedef struct LupinPathtraceResources LupinPathtraceResources;
typedef struct LupinScene LupinScene;
typedef struct LupinTexture LupinTexture;                     /* one Rgba16Float render target */
typedef struct LupinDoubleBufferedTexture LupinDoubleBufferedTexture;
typedef struct LupinComm LupinComm;                           /* RCCL communicator of one context (multi-GPU gather) */

/* renderer.rs:644-649 */
typedef struct LupinAccumulationParams {
    const LupinTexture *prev_frame;
    uint32_t accum_counter;
} LupinAccumulationParams;

/* renderer.rs:751-766. Option<> fields become nullable pointers. */
typedef struct LupinPathtraceDesc {
    const LupinAccumulationParams *accum_params;   /* NULL = None */
    const LupinTileParams *tile_params;            /* NULL = None (full-screen dispatch) */
    LupinCameraParams camera_params;
    LupinMat3x4 camera_transform;
    uint32_t force_software_bvh;                   /* both values select the software BVH here */
    LupinAdvancedParams advanced;
} LupinPathtraceDesc;

enum LupinStatus {
    LUPIN_OK = 0,
    LUPIN_ERR_INVALID_ARGUMENT = -1,
    LUPIN_ERR_NO_DEVICE = -2,        /* no HIP device / extension unusable: there is NO CPU fallback */
    LUPIN_ERR_HIP = -3,
    LUPIN_ERR_NO_SW_BVH = -4,        /* renderer.rs:774-777 */
    LUPIN_ERR_TILE_OUT_OF_RANGE = -5,/* renderer.rs:814 */
    LUPIN_ERR_SAME_TARGET = -6,      /* render_target == prev_frame, renderer.rs:754-755 */
    LUPIN_ERR_OUT_OF_MEMORY = -7,
    LUPIN_ERR_RCCL = -8              /* librccl missing or a collective failed (multi-GPU gather only) */
};

/* ------------------------------------------------------------------------------------------
 * Entry points
 * ---------------------------------------------------------------------------------------- */

const char *lupin_hip_last_error(void);
/* number of visible HIP devices; 0 when none (never initialises a context) */
int lupin_hip_device_count(void);

/* wgpu device/queue acquisition (wgpu_utils.rs:20-120, renderer.rs:307-330) -> one HIP device + its streams.
 * Like a wgpu queue, the context orders work in call order: consecutive pathtrace_scene calls may overlap on the
 * device (they run on alternating internal streams), but each call sees the textures exactly as the calls before it
 * left them; uploads, downloads, copies and lupin_hip_sync wait for everything submitted earlier. */
int lupin_hip_create_context(int device_ordinal, LupinContext **out_ctx);
/* Destroying a context twice is a no-op.  Textures, scenes and communicators may be destroyed after their context (their
 * device memory is freed then); every other use of an object whose context is gone returns LUPIN_ERR_INVALID_ARGUMENT
 * instead of touching freed memory. */
void lupin_hip_destroy_context(LupinContext *ctx);
/* device.poll(wait_indefinitely) (loader.rs:1692,1825) */
int lupin_hip_sync(LupinContext *ctx);

/* f32 -> f16 rounding of the Rgba16Float store (pathtracer.wgsl:288).  WGSL leaves it to the device;
 * the reference's golden renders are reproduced by round-toward-zero, which is the
 * default.  mode: 0 = toward zero, 1 = nearest even. */
#define LUPIN_STORE_ROUND_TOWARD_ZERO 0
#define LUPIN_STORE_ROUND_NEAREST_EVEN 1
int lupin_hip_set_f16_store_rounding(LupinContext *ctx, int mode);

/* Frames per wavefront.  pathtrace_scene is enqueue-and-return like the reference's queue.submit
 * (renderer.rs:841); consecutive calls that differ only in camera and accum_counter and chain their textures (each call's
 * prev_frame is the previous call's render_target -- the reference's front / back / flip loop) are recorded and executed
 * as ONE wavefront of up to `frames` calls: the stage kernels see `frames` times as many paths per launch, the resolve
 * applies the calls' blends per pixel in call order.  Same texels as one wavefront per call, bit for bit.  A batch runs
 * when it is full or as soon as anything needs its result: lupin_hip_sync, texture download / upload / copy, tonemap, pack /
 * gather, falsecolor / debug calls, statistics, a call that cannot join (other scene / integrator / size / parameters /
 * texture chain), a mode setter, teardown of any object.  An error of a recorded call is reported by the call that runs the
 * batch.  frames in [1, 16] (LUPIN_BATCH), or 0 = chosen by dispatch size, the default: sixteen for dispatches of up to 4 M
 * pixels, eight above; 1 = every call is its own wavefront.  f32 accumulation and work
 * counting run unbatched (kernel timing keeps the batches: its launches are the production launches, one lane). */
int lupin_hip_set_batch_frames(LupinContext *ctx, uint32_t frames);

/* Which hierarchy the persistent tracer walks on scenes traversed from global memory (DESIGN.md 5 "Wide traversal").
 * BINARY (default): the reference's own visiting order for every query.  WIDE: the four-wide collapse of the reference's
 * trees with an exactness certificate; queries it cannot certify are re-traced in the reference's order
 * (LupinStats.wide_queries / wide_retraced).  Same images either way; the wide form needs 40 % fewer memory requests
 * per ray and is measured NOT to be faster on MI355X (the tracer is not request-bound), so it is the opt-in.
 * LUPIN_TRAVERSAL=wide sets it at context creation.  Takes effect with the next pathtrace call. */
enum LupinTraversalMode { LUPIN_TRAVERSAL_BINARY = 0, LUPIN_TRAVERSAL_WIDE = 1 };
int lupin_hip_set_traversal(LupinContext *ctx, int mode);

/* lp::build_pathtrace_resources (renderer.rs:470-642): bakes max_bounces / samples_per_pixel */
/* pathtracer.wgsl:275-289 accumulates into an Rgba16Float texture: the running mean is re-quantised to f16 every frame
 * (and stalls once 1/k drops under half an ulp).  LUPIN_ACCUM_F16_RUNNING_AVERAGE reproduces that bit for bit
 * (default); LUPIN_ACCUM_F32 runs the same recurrence on an f32 shadow of each texture (16 B per pixel, allocated on first
 * use) and stores the rounded f16 view, so lupin_hip_texture_download_rgba16f keeps working and
 * lupin_hip_texture_download_rgba32f returns the unquantised mean.  The multi-GPU gather moves the f16 view only. */
enum LupinAccumulationMode { LUPIN_ACCUM_F16_RUNNING_AVERAGE = 0, LUPIN_ACCUM_F32 = 1 };
int lupin_hip_set_accumulation_mode(LupinContext *ctx, int mode);
/* Optional: allocate the path state of every frame in flight up front (dispatches of up to `pixels` pixels, baked
 * max_bounces / samples_per_pixel as in lupin_hip_build_pathtrace_resources).  Without it each of the context's lanes
 * allocates at its first pathtrace call, i.e. inside the first frames of the host's loop.  No counterpart in the reference
 * (wgpu allocates its storage buffers in build_pathtrace_resources, renderer.rs:451-640). */
int lupin_hip_reserve_path_state(LupinContext *ctx, uint64_t pixels, uint32_t max_bounces, uint32_t samples_per_pixel);


int lupin_hip_build_pathtrace_resources(LupinContext *ctx, const LupinBakedPathtraceParams *params,
                                        LupinPathtraceResources **out_res);
void lupin_hip_destroy_pathtrace_resources(LupinPathtraceResources *res);

/* upload half of lp::build_accel_structures_and_upload (data_structures.rs:696-872) */
int lupin_hip_scene_create(LupinContext *ctx, const LupinSceneDesc *desc, LupinScene **out_scene);
void lupin_hip_scene_destroy(LupinScene *scene);

/* Rgba16Float render targets + lp::DoubleBufferedTexture (wgpu_utils.rs:279-348) */
int lupin_hip_texture_create(LupinContext *ctx, uint32_t width, uint32_t height, LupinTexture **out_tex);
void lupin_hip_texture_destroy(LupinTexture *tex);
uint32_t lupin_hip_texture_width(const LupinTexture *tex);
uint32_t lupin_hip_texture_height(const LupinTexture *tex);
/* raw device pointer of the W*H*4 half-float payload (row-major, row 0 = top) for zero-copy
 * wrapping by the host (e.g. RCCL gather of tile payloads) */
void *lupin_hip_texture_device_ptr(const LupinTexture *tex);
int lupin_hip_texture_upload_rgba16f(LupinTexture *tex, const uint16_t *pixels);
/* readback (loader.rs:1775-1879 download path); synchronises the stream */
/* (H, W, 4) f32 of the texture's f32 accumulator; fails unless the last frame rendered into it used LUPIN_ACCUM_F32 */
int lupin_hip_texture_download_rgba32f(const LupinTexture *tex, float *out_pixels);
int lupin_hip_texture_download_rgba16f(const LupinTexture *tex, uint16_t *out_pixels);

int lupin_hip_dbuf_create(LupinContext *ctx, uint32_t width, uint32_t height,
                          LupinDoubleBufferedTexture **out);          /* wgpu_utils.rs:289 */
void lupin_hip_dbuf_destroy(LupinDoubleBufferedTexture *t);
LupinTexture *lupin_hip_dbuf_front(LupinDoubleBufferedTexture *t);    /* :301 */
LupinTexture *lupin_hip_dbuf_back(LupinDoubleBufferedTexture *t);     /* :306 */
int lupin_hip_dbuf_copy_front_to_back(LupinDoubleBufferedTexture *t); /* :321 */
void lupin_hip_dbuf_flip(LupinDoubleBufferedTexture *t);              /* :334 */
int lupin_hip_dbuf_resize(LupinDoubleBufferedTexture *t, uint32_t width, uint32_t height); /* :341 */

/* lp::get_num_tiles (renderer.rs:675-681) */
uint32_t lupin_hip_get_num_tiles(uint32_t tile_size, uint32_t width, uint32_t height);

/* lp::pathtrace_scene (renderer.rs:768-842): enqueues one accumulation frame (or one tile of it) and returns, like
 * queue.submit (:841).  render_target must differ from accum_params->prev_frame (:754-755). */
int lupin_hip_pathtrace_scene(LupinContext *ctx, const LupinPathtraceResources *res,
                              const LupinScene *scene, LupinTexture *render_target,
                              uint32_t pathtrace_type, const LupinPathtraceDesc *desc);

/* renderer.rs:843-870 FalsecolorType */
enum LupinFalsecolorType {
    LUPIN_FALSECOLOR_ALBEDO = 0, LUPIN_FALSECOLOR_NORMALS = 1, LUPIN_FALSECOLOR_NORMALS_UNSIGNED = 2,
    LUPIN_FALSECOLOR_FRONT_FACING = 3, LUPIN_FALSECOLOR_EMISSION = 4, LUPIN_FALSECOLOR_ROUGHNESS = 5,
    LUPIN_FALSECOLOR_METALLIC = 6, LUPIN_FALSECOLOR_OPACITY = 7, LUPIN_FALSECOLOR_MAT_TYPE = 8,
    LUPIN_FALSECOLOR_IS_DELTA = 9, LUPIN_FALSECOLOR_INSTANCE = 10, LUPIN_FALSECOLOR_TRI = 11
};
/* lp::pathtrace_scene_falsecolor (renderer.rs:872-948; shader entry pathtrace_falsecolor_main,
 * pathtracer.wgsl:296-452): G-buffers for denoisers and visual debugging.  Same dispatch / tiling / accumulation
 * rules as lupin_hip_pathtrace_scene. */
int lupin_hip_pathtrace_scene_falsecolor(LupinContext *ctx, const LupinPathtraceResources *res,
                                         const LupinScene *scene, LupinTexture *render_target,
                                         uint32_t falsecolor_type, const LupinPathtraceDesc *desc);

/* renderer.rs:951-964  DebugVizType / DebugVizDesc */
enum LupinDebugVizType { LUPIN_DEBUG_VIZ_BVH_AABB_CHECKS = 0, LUPIN_DEBUG_VIZ_BVH_TRI_CHECKS = 1, LUPIN_DEBUG_VIZ_NUM_BOUNCES = 2 };
typedef struct LupinDebugVizDesc
{
    uint32_t viz_type;        /* LupinDebugVizType */
    float    heatmap_min;
    float    heatmap_max;
    uint32_t first_hit_only;  /* bool */
} LupinDebugVizDesc;

/* renderer.rs:966-1041  lp::pathtrace_scene_debug(device, queue, resources, scene, render_target, debug_desc, desc):
 * pathtrace_debug_main (pathtracer.wgsl:457-503) -- per pixel ONE sample of the first closest-hit query (first_hit_only,
 * except for NumBounces) or of the whole Standard path; the number of box tests / triangle tests / surface hits becomes
 * a heat-map colour (get_heatmap_color, :2806-2872).  Same tiling, accumulation and error rules as pathtrace_scene;
 * baked samples_per_pixel is ignored as in the reference. */
int lupin_hip_pathtrace_scene_debug(LupinContext *ctx, const LupinPathtraceResources *res, const LupinScene *scene,
                                    LupinTexture *render_target, const LupinDebugVizDesc *debug_desc,
                                    const LupinPathtraceDesc *desc);

/* Tile-sharded variant for multi-GPU rendering (extension; the reference renders tiles one
 * sub-dispatch at a time on one device, renderer.rs:807-829): renders, in ONE wavefront launch,
 * every tile t of the frame that lupin_tile_owner(t, tiles_x, world) of include/lupin_tiles.h gives to `rank`
 * (round-robin t % world; rows rotated when a row holds a multiple of `world` tiles) (tiles of tile_size*4 pixels, numbered
 * row-major as renderer.rs:816-817).  Edge tiles cover all in-bounds pixels, so the union over
 * ranks equals the full-screen dispatch bit for bit.  desc->tile_params is ignored. */
int lupin_hip_pathtrace_scene_tiles(LupinContext *ctx, const LupinPathtraceResources *res,
                                    const LupinScene *scene, LupinTexture *render_target,
                                    uint32_t pathtrace_type, const LupinPathtraceDesc *desc,
                                    uint32_t tile_size, uint32_t rank, uint32_t world);

/* ---- measurement hooks (no reference counterpart; the reference exposes none) ---- */

typedef struct LupinStats {
    uint64_t path_bounces;      /* integrator iterations that issued a closest-hit query (metric unit) */
    uint64_t paths;             /* camera samples started */
    uint64_t extend_launches;   /* launches of the dominant (extend) kernel */
    double extend_ms;           /* summed hipEvent duration of those launches (0 unless timing on) */
    double shade_ms;
    double total_ms;            /* whole pathtrace_scene device time (timing on) */
    /* LUPIN_STATS_WORK_COUNTERS: work done by the tracing kernels in this build's layout, per tracing mode
     * [0] closest hit of the integrator loop, [1] MIS / Direct shadow rays, [2] light-pdf marching:
     * internal-node visits (= the oracle's box tests / 2; one 64-byte node fetch each), triangle tests (one 48-byte
     * record each), instance entries (64 bytes) */
    uint64_t node_visits[3];
    uint64_t tri_tests[3];
    uint64_t instance_entries[3];
    uint64_t wide_node_visits[3];   /* visits of four-wide nodes (one 128-byte fetch each) */
    /* how the persistent tracer's waves spent their scheduling rounds (closest-hit mode, first pass): refill rounds | node
     * rounds, node steps, lanes summed over the node steps | triangle rounds, lanes | instance rounds, lanes | end-of-
     * traversal rounds, lanes.  lanes / (64 x steps or rounds) = lane utilisation of that phase. */
    uint64_t tracer_rounds[10];
    /* shader-clock cycles (s_memtime, summed over the waves) spent in refill | node | triangle | instance | end-of-traversal
     * rounds and in the whole scheduling loop (the work-counting build waits for each round's loads before reading the clock) */
    uint64_t tracer_cycles[6];
    /* The first pass of a two-pass tracer (always counted) -- the four-wide tracer, or the binary tracer on its short stack:
     * queries it took, and how many of them it handed to the full-stack binary tracer (uncertified wide results; stack
     * overflows) -- the fallback rate is wide_retraced / wide_queries. */
    uint64_t wide_queries;
    uint64_t wide_retraced;
    /* LUPIN_VERIFY_WIDE=1: every query also run binary-vs-wide on the device, one ray per lane: rays checked, rays the wide
     * traversal flagged, unflagged rays whose result differed (must be 0), rays that differed flag or not */
    uint64_t verify_checked, verify_flagged, verify_mismatches, verify_raw_mismatches;
    uint64_t verify_reasons[4];     /* flagged rays by reason: second hit within the margin | ill-conditioned hit | triangle outside a box above it | stack bound */
    uint32_t frames_in_flight;      /* lanes the latest pathtrace call could use */
    uint32_t wide_traversal;        /* 1 = the latest pathtrace call ran the four-wide tracer */
    uint32_t frames_per_wavefront;  /* recorded calls the latest wavefront carried (lupin_hip_set_batch_frames) */
    uint32_t short_stack_entries;   /* stack entries per lane of the binary tracer's first pass in the latest wavefront; 0 = one pass on the full stack */
} LupinStats;
enum LupinStatsMode {
    LUPIN_STATS_PLAIN = 0,           /* path-bounce / path counters only (always on) */
    LUPIN_STATS_KERNEL_TIMING = 1,   /* + hipEvents around every extend / shade launch (frames run one at a time) */
    LUPIN_STATS_WORK_COUNTERS = 2    /* + the work-counting instantiation of the tracing kernels */
};
/* reset the counters and choose the mode for the calls that follow */
int lupin_hip_stats_reset(LupinContext *ctx, int mode);
/* synchronises, then reports totals since the last reset */
int lupin_hip_stats_get(LupinContext *ctx, LupinStats *out);

/* Device-to-device copy of `bytes` bytes, `reps` times, on the context's stream: (bytes read + bytes written) / time in
 * GB/s -- the measured HBM peak that bench.py reports next to the nominal 8 TB/s.  Synchronous. */
int lupin_hip_measure_copy_bandwidth(LupinContext *ctx, uint64_t bytes, uint32_t reps, double *out_gb_per_s);

/* Which HIP runtime serves this process: the version the library was built against (HIP_VERSION), the version of the
 * libamdhip64 that is bound, and every distinct libamdhip64 mapped (a PyTorch wheel bundles its own; two in one process
 * make lupin_hip_create_context fail, and LUPIN_GRAPH=1 additionally requires build and runtime major.minor to agree). */
typedef struct LupinRuntimeInfo {
    int32_t  build_hip_version;
    int32_t  runtime_hip_version;
    uint32_t num_hip_runtimes_mapped;
    char     hip_runtime_paths[1012];   /* ';'-separated */
} LupinRuntimeInfo;
int lupin_hip_runtime_info(LupinRuntimeInfo *out);

/* Standalone closest-hit probe over a ray batch: the traversal kernel alone
 * (bvh_custom.wgsl:7-110). Host arrays; n rays; outputs hit(0/1), dst, u, v, instance, tri. */
int lupin_hip_trace_rays(LupinContext *ctx, const LupinScene *scene, uint32_t n,
                         const float *ori_xyz, const float *dir_xyz, float ray_epsilon,
                         uint32_t *out_hit, float *out_dst, float *out_uv,
                         uint32_t *out_instance, uint32_t *out_tri);

/* The same probe through the four-wide traversal the persistent tracer runs by default on scenes traversed from global
 * memory (DESIGN.md 5 "Wide traversal"): out_needs_retrace[i] = 1 when the traversal could not certify that ray's result
 * (the pipeline re-traces such a query with the binary kernel; the other outputs of that ray are then unspecified);
 * every other ray's outputs equal lupin_hip_trace_rays' bit for bit.  Scenes staged in LDS have no wide hierarchy
 * (LUPIN_ERR_INVALID_ARGUMENT). */
int lupin_hip_trace_rays_wide(LupinContext *ctx, const LupinScene *scene, uint32_t n,
                              const float *ori_xyz, const float *dir_xyz, float ray_epsilon,
                              uint32_t *out_hit, float *out_dst, float *out_uv,
                              uint32_t *out_instance, uint32_t *out_tri, uint32_t *out_needs_retrace);

/* Evaluates one function of include/lupin_detmath.h (fn: 0 sin, 1 cos, 2 atan, 3 atan2(x,y), 4 acos,
 * 5 exp, 6 log, 7 pow(x,y), 8 x/y, 9 sqrt) on the device over host arrays; the tests require the
 * result to be bit-identical to the host build of the same header. */
int lupin_hip_detmath_probe(LupinContext *ctx, int fn, uint32_t n, const float *x, const float *y, float *out);

/* tonemapping.rs:106-132  TonemapDesc (+ Viewport :144-151) */
typedef struct LupinTonemapDesc
{
    uint32_t has_viewport;   /* 0 = None: the whole target */
    float    viewport_x, viewport_y, viewport_w, viewport_h;
    float    exposure;       /* color *= 2^exposure */
    uint32_t filmic;         /* bool, default false */
    uint32_t srgb;           /* bool, default true */
    uint32_t clear;          /* bool, default true: target cleared to (0,0,0,1) first */
} LupinTonemapDesc;

/* tonemapping.rs:155-224  lp::tonemap_and_fit_aspect(device, queue, resources, src, dst, desc) with the Rgba8Unorm
 * target held by the host: `dst_rgba8` (dst_width x dst_height x 4 bytes, row 0 = top) is read when !clear and
 * overwritten with the result.  Synchronous.  The reference rasterises a quad and samples through the hardware
 * sampler, whose sub-texel precision is not specified; this is the same mapping evaluated at pixel centres. */
int lupin_hip_tonemap_and_fit_aspect(LupinContext *ctx, const LupinTexture *src, uint8_t *dst_rgba8,
                                     uint32_t dst_width, uint32_t dst_height, const LupinTonemapDesc *desc);

/* Tile-sharded multi-GPU support: pack the pixels of every tile owned by `rank` (include/lupin_tiles.h) (tiles
 * of tile_size*4 pixels, row-major tile order as renderer.rs:816-817) into a dense device
 * buffer / scatter a packed buffer back. Payload layout: tiles in ascending t, each tile
 * row-major, 8 B per pixel. Returns the number of pixels via out_pixels. */
int lupin_hip_pack_tiles(LupinContext *ctx, const LupinTexture *tex, uint32_t tile_size,
                         uint32_t rank, uint32_t world, void *device_dst, uint64_t *out_pixels);
int lupin_hip_unpack_tiles(LupinContext *ctx, LupinTexture *tex, uint32_t tile_size,
                           uint32_t rank, uint32_t world, const void *device_src);
uint64_t lupin_hip_packed_tile_pixels(uint32_t width, uint32_t height, uint32_t tile_size,
                                      uint32_t rank, uint32_t world);
/* The scatter after an all-gather in ONE launch: `device_gathered` holds `world` payloads of `capacity_pixels` pixels each
 * (rank r's at r * capacity_pixels * 8 bytes, padded); every tile NOT owned by `rank` is written into `tex`.  For hosts that
 * run their own collective; lupin_hip_gather_framebuffer does pack + all-gather + this. */
int lupin_hip_unpack_gathered_tiles(LupinContext *ctx, LupinTexture *tex, uint32_t tile_size, uint32_t rank, uint32_t world,
                                    const void *device_gathered, uint64_t capacity_pixels);

/* ---- the one exchange step of tile-sharded rendering: RCCL gather of per-tile framebuffers over xGMI ----
 * No counterpart in the reference (single device; its TileParams sub-dispatch, renderer.rs:807-829, is what the shards
 * are made of).  A host renders its tiles with lupin_hip_pathtrace_scene_tiles for any number of accumulation frames
 * (no communication) and calls lupin_hip_gather_framebuffer once per readback: every rank then holds the whole frame,
 * bit-identical to the single-GPU render.  librccl is dlopen'ed on first use (LUPIN_RCCL_LIB names the one library to load instead of the default sonames;
 * a library that cannot be loaded makes every communicator call return LUPIN_ERR_RCCL).
 *
 * One process per GPU:  rank 0 calls lupin_hip_comm_get_unique_id and hands the 128 bytes to the other ranks by any
 *                       means (file, socket, MPI); every rank then calls lupin_hip_comm_init_rank.
 * One process, n GPUs:  lupin_hip_comm_init_all over n contexts (one per device), gathers through
 *                       lupin_hip_gather_framebuffer_all (the n all-gathers form one RCCL group).
 * A host that already owns an ncclComm_t for the context's device wraps it with lupin_hip_comm_from_nccl. */
#define LUPIN_COMM_ID_BYTES 128
int lupin_hip_comm_get_unique_id(uint8_t *out_id /* LUPIN_COMM_ID_BYTES */);
int lupin_hip_comm_init_rank(LupinContext *ctx, const uint8_t *id /* LUPIN_COMM_ID_BYTES */, uint32_t rank, uint32_t world,
                             LupinComm **out_comm);
int lupin_hip_comm_init_all(LupinContext *const *ctxs, uint32_t n, LupinComm **out_comms /* n entries */);
int lupin_hip_comm_from_nccl(LupinContext *ctx, void *nccl_comm, uint32_t rank, uint32_t world, LupinComm **out_comm);
void lupin_hip_comm_destroy(LupinComm *comm);
uint32_t lupin_hip_comm_rank(const LupinComm *comm);
uint32_t lupin_hip_comm_world(const LupinComm *comm);
/* pack this rank's tiles of `tex` -> ncclAllGather -> scatter the other ranks' tiles into `tex`; enqueued on the context's
 * stream after every frame enqueued so far (asynchronous like pathtrace_scene; download / lupin_hip_sync waits). */
int lupin_hip_gather_framebuffer(LupinComm *comm, LupinTexture *tex, uint32_t tile_size);
int lupin_hip_gather_framebuffer_all(LupinComm *const *comms, LupinTexture *const *texs, uint32_t n, uint32_t tile_size);
/* the readback form: only `root` receives (one grouped ncclSend per peer / ncclRecv per peer on the root, exact payload
 * sizes), so a 3840 x 2160 readback moves 66 MB once instead of to every rank; the other ranks' textures keep their own
 * tiles only.  Same enqueue semantics as lupin_hip_gather_framebuffer. */
int lupin_hip_gather_framebuffer_to(LupinComm *comm, LupinTexture *tex, uint32_t tile_size, uint32_t root);
/* host-side reductions over the ranks for measurement loops (op 0 = sum, 1 = max); synchronous, and every frame this
 * rank enqueued has completed when they return, so lupin_hip_comm_barrier brackets a timed region */
int lupin_hip_comm_allreduce_f64(LupinComm *comm, double *inout, uint32_t n, uint32_t op);
int lupin_hip_comm_barrier(LupinComm *comm);

/* ------------------------------------------------------------------------------------------
 * CPU-side preprocessing that produces the path's inputs (data_structures.rs:20-641).
 * Pure host code, no device needed.
 * ---------------------------------------------------------------------------------------- */

/* Device BLAS builder ("next" row 8f-1; no counterpart in the reference, whose builder is CPU-only,
 * data_structures.rs:196-475): a linear BVH over Morton-sorted triangles -- a complete binary tree with 1-2 triangles
 * per leaf, depth lupin_hip_lbvh_depth(n) <= 22 -- written in the reference's BvhNode format with the reordered index
 * buffer, i.e. a drop-in alternative to lupin_build_bvh for lupin_hip_scene_create.  Returns the node count
 * (= lupin_hip_lbvh_node_count(num_indices / 3)) or a negative status.  Synchronous. */
uint32_t lupin_hip_lbvh_depth(uint32_t num_tris);
uint64_t lupin_hip_lbvh_node_count(uint32_t num_tris);
int64_t lupin_hip_build_bvh_device(LupinContext *ctx, const float *verts_pos4, uint32_t num_verts, uint32_t *indices,
                                   uint32_t num_indices, LupinBvhNode *out_nodes, uint64_t out_capacity);

/* The reference's own BLAS builder run on the device (csrc/sahbvh.hip): level-synchronous binned SAH that reproduces
 * lp::build_bvh's decisions (data_structures.rs:196-475: 5 bins per axis, half-area x count cost, centroid[axis] <= pos
 * partition, depth cap) from min / max / count reductions, so every node's box, split plane and triangle SET equals
 * lupin_build_bvh's bit for bit.  Only bookkeeping differs: nodes are numbered level by level and both sides of a
 * partition keep their input order.  Same signature and return value as lupin_build_bvh (out_nodes must not be NULL;
 * 2 * triangles - 1 nodes always suffice; vertex positions must be finite).  Synchronous. */
int64_t lupin_hip_build_bvh_sah_device(LupinContext *ctx, const float *verts_pos4, uint32_t num_verts, uint32_t *indices,
                                       uint32_t num_indices, LupinBvhNode *out_nodes, uint64_t out_capacity);

/* The four-wide collapse of one mesh's BLAS exactly as lupin_hip_scene_create performs it for the wide tracer (host code,
 * no device needed; DESIGN.md 5 "Wide traversal"): a node's grandchildren are pulled up, largest box first, until it has
 * four children or only leaves; a child whose own children's boxes are not inside its box stays a child.  out_nodes
 * receives 128-byte records of 32 words: lox[4] loy[4] loz[4] hix[4] hiy[4] hiz[4] (f32; NaN in unused slots), ref[4]
 * (u32: bit 31 = leaf, bit 30 = the child's box does not contain every triangle below it (never pruned by distance), low
 * 30 bits = first triangle of the leaf / index into out_nodes; 0xFFFFFFFF = unused slot), 4 zero words.
 * With vertex data (verts_pos4 / indices in BLAS leaf order; may be NULL), out_tri_flags (may be NULL; one byte per
 * triangle) receives bit 0 = last triangle of its leaf, bit 1 = the triangle is not inside every box above it.
 * Returns the node count (out_nodes may be NULL to query it) or < 0; *out_root = the root reference (a single-leaf BLAS
 * has no wide node: the leaf reference passes through). */
int64_t lupin_hip_collapse_bvh4(const LupinBvhNode *nodes, uint32_t num_nodes, const float *verts_pos4, uint32_t num_verts,
                                const uint32_t *indices, uint32_t num_indices, void *out_nodes, uint64_t capacity,
                                uint32_t *out_root, uint8_t *out_tri_flags);

/* build_bvh (data_structures.rs:196-235): reorders `indices` in place; returns node count or <0.
 * out_nodes may be NULL to query the count (indices untouched in that case). */
int64_t lupin_build_bvh(const float *verts_pos4, uint32_t num_verts, uint32_t *indices,
                        uint32_t num_indices, LupinBvhNode *out_nodes, uint64_t out_capacity);
/* build_tlas (data_structures.rs:545-641): out_nodes must hold 2*num_instances entries.
 * model_aabbs: per mesh 6 floats (min xyz, max xyz). */
int64_t lupin_build_tlas(const LupinInstance *instances, uint32_t num_instances,
                         const float *model_aabbs, uint32_t num_meshes, LupinTlasNode *out_nodes);
/* build_alias_table (data_structures.rs:116-193): returns bins written (0 when sum == 0) */
int64_t lupin_build_alias_table(const float *weights, uint64_t n, LupinAliasBin *out_bins);
/* triangle-area weights + total area of build_lights' inner loop (data_structures.rs:40-51,106-112) */
float lupin_mesh_light_weights(const float *verts_pos4, const uint32_t *indices, uint32_t num_indices,
                               float *out_weights);
/* environment texel weights of build_lights (data_structures.rs:65-93); texels = w*h*4 f32 */
void lupin_env_light_weights(const float *texels_rgba_f32, uint32_t width, uint32_t height,
                             const float scale_rgb[3], float *out_weights);
/* Mat3x4 inverse through Mat4::inverse (base.rs:542-578, :708-722) */
void lupin_mat3x4_inverse(const LupinMat3x4 *in, LupinMat3x4 *out);

#ifdef __cplusplus
}
#endif
#endif /* LUPIN_HIP_H */
