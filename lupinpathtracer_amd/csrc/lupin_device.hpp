// lupin_device.hpp -- device-side data layout, traversal and shading for the gfx950 wavefront
// path tracer.  Semantics follow lupin/src/shaders/{bvh_custom,pathtracer}.wgsl of the
// reference (cited per function); the layout, the traversal state machine and the stage split
// are this implementation's own.
//
// Arithmetic contract (see include/lupin_detmath.h and DESIGN.md): every f32 expression keeps
// the WGSL evaluation order, no FMA contraction (-ffp-contract=off), IEEE division and sqrt,
// transcendentals from lupin_detmath.h.  That makes radiance a pure function of
// (scene, pixel, accum_counter), bit-reproducible between this code and the CPU oracle.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lupin_hip.h"
#include "../../include/lupin_detmath.h"

#define LP_DEV __device__ __forceinline__
// Bulky helpers (materials, lights, textures) are force-inlined: as real functions the AMDGPU call ABI spills through
// scratch (measured in round 1: 5.25 -> 1.69 Gsamples/s), and an out-of-line callee that reached its by-reference arguments
// through flat pointers into scratch faulted in round 2 (profiles/r03_direct_fault_forensics.md; tools/isa_guard.py keeps
// the shipped code object free of device calls).
#define LP_FN __device__ __forceinline__
#define LP_BLOCK 256

namespace lpd {

// ------------------------------------------------------------------------------------------------
// Device scene layout
// ------------------------------------------------------------------------------------------------

// Child references used by both levels of the hierarchy.
//   BLAS: bit31 set   -> leaf, low 31 bits = global index of its first triangle; the leaf ends at
//                        the first triangle whose v0.w carries LEAF_END
//         bit31 clear -> index of a WideNode
//   TLAS: bit31 set   -> leaf, low 31 bits = instance index
constexpr uint32_t REF_LEAF = 0x80000000u;
constexpr uint32_t LEAF_END_BITS = 1u;
constexpr uint32_t TRI_LEAKY_BITS = 2u;   // TriVerts.v0.w: the triangle is not inside every box the reference's BLAS stores above it

// One internal node = both children's boxes + references, one 64-byte line.  The reference
// reads the node (32/48 B) and then both children (2 x 32/48 B) per visit
// (bvh_custom.wgsl:47-51, :234-240); here one aligned 64 B fetch carries everything the visit
// needs.
struct WideNode
{
    // the two children's bounds interleaved (left, right per coordinate): a load leaves each (left, right) pair in two
    // consecutive registers, which is what the packed slab arithmetic (slab_pair) consumes
    float4 a;  // l.min.x r.min.x l.min.y r.min.y
    float4 b;  // l.min.z r.min.z l.max.x r.max.x
    float4 c;  // l.max.y r.max.y l.max.z r.max.z
    uint4 d;   // left_ref right_ref flags 0   (flags: bit 0 / 1 = the left / right child's box does not bound its triangles)
};
// slab_dst on the left / right child of a node in that layout
#define LP_NODE_LEFT(nd)  (nd).a.x, (nd).a.z, (nd).b.x, (nd).b.z, (nd).c.x, (nd).c.z
#define LP_NODE_RIGHT(nd) (nd).a.y, (nd).a.w, (nd).b.y, (nd).b.w, (nd).c.y, (nd).c.w

// The same hierarchies collapsed to four children per node (lupin_hip_scene_create: a node's grandchildren are pulled up,
// largest box first, until it has four children or only leaves): one 128-byte line = one L1-miss request, which costs the
// memory system what a 64-byte one does (profiles/r02_gather_probe.jsonl), and a ray needs about half as many of them.
// Child k's box is the box the reference's tree stores for that node; an unused slot holds NaN bounds (its slab test
// misses) and REF_NONE.  TLAS and BLAS nodes share ONE array (no per-lane base select in the traversal step).
// Child references: bit 31 = leaf (as in WideNode.d), bit 30 = REF_LEAKY: the child's stored box does not bound every
// triangle below it (never pruned by distance, see "Wide traversal" below), low 30 bits = node index / first triangle /
// instance.  Traversed by the wide tracer only (k_extend_persistent<.., WIDE>, scene_closest_wide).
constexpr uint32_t REF_NONE = 0xFFFFFFFFu;
constexpr uint32_t REF_LEAKY = 0x40000000u;
constexpr uint32_t REF_INDEX_MASK = 0x3FFFFFFFu;
struct __attribute__((aligned(128))) Wide4
{
    float4 lox, loy, loz;   // children 0..3: min corner, one axis per word
    float4 hix, hiy, hiz;   // max corner
    uint4 ref;              // child references
    uint4 pad;              // not read by the traversal
};

// Triangles pre-gathered in BLAS leaf order: no index indirection during traversal
// (the reference does verts_pos[indices[i*3+k]], bvh_custom.wgsl:217-219).
struct TriVerts
{
    float4 v0;  // xyz + flags (bit0: last triangle of its leaf)
    float4 v1;
    float4 v2;
};

struct InstanceDev
{
    float4 r0, r1, r2;    // rows of the world->local affine (= columns of transpose_inverse_transform)
    uint32_t blas_root;   // child reference of the mesh's BLAS root
    uint32_t mat_idx;
    uint32_t mesh_idx;
    uint32_t flags;       // bit0: opacity may differ from 1 (alpha test needs the material); bits 8..11: material type
};

struct MeshDev
{
    uint32_t tri_offset;      // first global triangle of this mesh
    uint32_t normals_base;    // first vertex in the global normals array, or SENTINEL
    uint32_t texcoords_base;
    uint32_t colors_base;
};

struct TextureDev
{
    uint64_t offset;   // byte offset into the texel pool
    uint32_t width, height;
    uint32_t format;   // LupinTextureFormat
    uint32_t pad;
};

struct AliasRange { uint32_t offset, count; };

struct SceneDev
{
    const WideNode *tlas;        uint32_t tlas_root;   // child reference
    const WideNode *blas;        // ONE array holds both hierarchies ([BLAS nodes of every mesh | TLAS nodes]; tlas == blas): node references are indices into it
    const TriVerts *tris;
    const uint32_t *tri_indices; // 3 per global triangle, mesh-local vertex ids
    const InstanceDev *instances;
    const MeshDev *meshes;
    const MeshDev *inst_meshes;          // per instance: its mesh's record and its material, so that shading fetches them with the
    const LupinMaterial *inst_materials; // instance index it already has instead of one dependent fetch later
    const LupinMaterial *materials;
    const float4 *normals;
    const float2 *texcoords;
    const float4 *colors;
    const TextureDev *textures;
    const uint8_t *texels;
    const LupinEnvironment *environments;
    const LupinLight *lights;
    const float4 *light_bounds;          // per light: conservative world-space sphere (xyz centre, w = padded radius squared)
    const AliasRange *alias_ranges;      // per light
    const AliasRange *env_alias_ranges;  // per environment
    const LupinAliasBin *alias_bins;     // pool
    uint32_t num_lights, num_envs, num_instances;
    uint32_t sort_shade;                 // k_shade sorts each block's paths by material type (scenes with > 1 type)
    // four-wide collapse of the same two hierarchies in one array (wide tracer); inst_root4[i] = wide BLAS root reference of instance i
    const Wide4 *wide4;
    const uint32_t *inst_root4;
    uint32_t tlas4_root;
    // small scenes: [tlas | blas | tris | instances] as one array of 16-byte words that kernels stage in LDS
    const float4 *geo_blob;
    uint32_t geo_blob_words;                       // 0 = scene too large, traverse from global memory
    uint32_t geo_off_blas, geo_off_tris, geo_off_inst;   // offsets in 16-byte words (tlas starts at 0)
};

// ------------------------------------------------------------------------------------------------
// f32 vector helpers (component-wise, left-to-right)
// ------------------------------------------------------------------------------------------------

struct f3 { float x, y, z; };

LP_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
LP_DEV f3 splat(float a) { return mk3(a, a, a); }
LP_DEV f3 add(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
LP_DEV f3 sub(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
LP_DEV f3 mul(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
LP_DEV f3 dvd(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
LP_DEV f3 scale(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
LP_DEV f3 lscale(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
LP_DEV f3 divs(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
LP_DEV f3 adds(f3 a, float s) { return mk3(a.x + s, a.y + s, a.z + s); }
LP_DEV f3 neg(f3 a) { return mk3(-a.x, -a.y, -a.z); }
LP_DEV float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
LP_DEV f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
LP_DEV f3 normalize3(f3 a) { f3 r; lpm_normalize3f(a.x, a.y, a.z, &r.x, &r.y, &r.z); return r; }   // v * (1 / |v|): lupin_detmath.h
LP_DEV float minf(float a, float b) { return (b < a) ? b : a; }
LP_DEV float maxf(float a, float b) { return (a < b) ? b : a; }
LP_DEV float clampf(float x, float lo, float hi) { return minf(maxf(x, lo), hi); }
LP_DEV f3 sqrt3(f3 a) { return mk3(sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)); }
LP_DEV bool is_zero3(f3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
LP_DEV bool none_zero3(f3 a) { return a.x != 0.0f && a.y != 0.0f && a.z != 0.0f; }
LP_DEV bool finite3(f3 a)
{
    return (__float_as_uint(a.x) & 0x7F800000u) != 0x7F800000u &&
           (__float_as_uint(a.y) & 0x7F800000u) != 0x7F800000u &&
           (__float_as_uint(a.z) & 0x7F800000u) != 0x7F800000u;
}
LP_DEV f3 xyz(float4 v) { return mk3(v.x, v.y, v.z); }
// mat3x3 (columns c0 c1 c2) * v
LP_DEV f3 mat3_mul(f3 c0, f3 c1, f3 c2, f3 v) { return add(add(scale(c0, v.x), scale(c1, v.y)), scale(c2, v.z)); }

// WGSL u32(f32)/i32(f32) saturate
LP_DEV uint32_t f2u_sat(float x)
{
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
LP_DEV int32_t f2i_sat(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (int32_t)0x80000000;
    return (int32_t)x;
}

constexpr float LP_F32_MAX = 3.40282346638528859812e+38f;
constexpr float LP_PI = 3.14159265358979323846264338327950288f;
constexpr float LP_MIN_ROUGHNESS = 0.03f * 0.03f;

// ------------------------------------------------------------------------------------------------
// RNG: PCG-RXS-M-XS on a per-pixel u32 stream (pathtracer.wgsl:1561-1629)
// ------------------------------------------------------------------------------------------------

LP_DEV uint32_t hash_u32(uint32_t x)
{
    x ^= x >> 17; x *= 0xed5ad4bbu;
    x ^= x >> 11; x *= 0xac4c1b51u;
    x ^= x >> 15; x *= 0x31848babu;
    x ^= x >> 14;
    return x;
}
LP_DEV uint32_t rng_seed_for(uint32_t pixel_linear, uint32_t accum_counter)
{
    return hash_u32((pixel_linear * 19349663u) ^ (accum_counter * 83492791u) ^ (0u * 73856093u));
}
LP_DEV float rnd(uint32_t &s)
{
    s = s * 747796405u + 2891336453u;
    uint32_t r = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    r = (r >> 22u) ^ r;
    return (float)r / 4294967295.0f;
}
LP_DEV uint32_t rnd_range(uint32_t &s, uint32_t max_exclusive)
{
    uint32_t v = f2u_sat(rnd(s) * (float)max_exclusive);
    uint32_t m = max_exclusive - 1u;
    return v < m ? v : m;
}

// ------------------------------------------------------------------------------------------------
// Intersection primitives (pathtracer.wgsl:2906-2943)
// ------------------------------------------------------------------------------------------------

// Slab test (pathtracer.wgsl:2906-2917).  min/max here are IEEE minNum/maxNum (v_min_f32 / v_max_f32 /
// v_min3 / v_max3 on gfx950, fminf/fmaxf in the oracle): WGSL lets min/max return either operand when one is
// NaN, and the results only feed comparisons, so the sign of a zero is immaterial.
LP_DEV float slab_finish(float tminx, float tmaxx, float tminy, float tmaxy, float tminz, float tmaxz)
{
    float t1x = __builtin_fminf(tminx, tmaxx), t1y = __builtin_fminf(tminy, tmaxy), t1z = __builtin_fminf(tminz, tmaxz);
    float t2x = __builtin_fmaxf(tminx, tmaxx), t2y = __builtin_fmaxf(tminy, tmaxy), t2z = __builtin_fmaxf(tminz, tmaxz);
    float dst_far = __builtin_fminf(__builtin_fminf(t2x, t2y), t2z);
    float dst_near = __builtin_fmaxf(__builtin_fmaxf(t1x, t1y), t1z);
    bool did_hit = dst_far >= dst_near && dst_far > 0.0f;
    return did_hit ? dst_near : LP_F32_MAX;
}
LP_DEV float slab_dst(f3 o, f3 inv_d, float lox, float loy, float loz, float hix, float hiy, float hiz)
{
    float tminx = (lox - o.x) * inv_d.x, tminy = (loy - o.y) * inv_d.y, tminz = (loz - o.z) * inv_d.z;
    float tmaxx = (hix - o.x) * inv_d.x, tmaxy = (hiy - o.y) * inv_d.y, tmaxz = (hiz - o.z) * inv_d.z;
    return slab_finish(tminx, tmaxx, tminy, tmaxy, tminz, tmaxz);
}
// slab_dst of both children of a child-pair node at once: the same IEEE subtractions and multiplications, issued as packed
// instructions (v_pk_add_f32 / v_pk_mul_f32 compute two independent IEEE results), then slab_dst's own tail per child.
typedef float lp_v2 __attribute__((ext_vector_type(2)));
LP_DEV void slab_pair(f3 o, f3 inv_d, float4 a, float4 b, float4 c, float &ld, float &rd)
{
    const lp_v2 ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
    const lp_v2 ix = {inv_d.x, inv_d.x}, iy = {inv_d.y, inv_d.y}, iz = {inv_d.z, inv_d.z};
    const lp_v2 tminx = ((lp_v2){a.x, a.y} - ox) * ix, tminy = ((lp_v2){a.z, a.w} - oy) * iy, tminz = ((lp_v2){b.x, b.y} - oz) * iz;
    const lp_v2 tmaxx = ((lp_v2){b.z, b.w} - ox) * ix, tmaxy = ((lp_v2){c.x, c.y} - oy) * iy, tmaxz = ((lp_v2){c.z, c.w} - oz) * iz;
    ld = slab_finish(tminx.x, tmaxx.x, tminy.x, tmaxy.x, tminz.x, tmaxz.x);
    rd = slab_finish(tminx.y, tmaxx.y, tminy.y, tmaxy.y, tminz.y, tmaxz.y);
}

struct TriHit { float t, u, v; };
LP_DEV TriHit tri_dst(f3 o, f3 d, f3 v0, f3 v1, f3 v2, float eps)
{
    f3 v1v0 = sub(v1, v0), v2v0 = sub(v2, v0), rov0 = sub(o, v0);
    f3 n = cross3(v1v0, v2v0);
    f3 q = cross3(rov0, d);
    float det = dot3(d, n);
    float id = 1.0f / det;
    TriHit h;
    h.u = id * dot3(neg(q), v2v0);
    h.v = id * dot3(q, v1v0);
    h.t = id * dot3(neg(n), rov0);
    if (minf(h.u, h.v) < 0.0f || (h.u + h.v) > 1.0f || h.t < eps) h.t = LP_F32_MAX;
    return h;
}


// ------------------------------------------------------------------------------------------------
// Geometry access: the same traversal code runs over global memory or over a copy of the scene's
// geometry staged in LDS (scenes up to LP_GEO_LDS_LIMIT bytes).  Divergent 64-byte node fetches cost
// one L1 tag lookup per lane and 16 B; from LDS they are four ds_read_b128 per lane group.
// ------------------------------------------------------------------------------------------------

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const v4f *lds_v4p;
constexpr uint32_t LP_GEO_LDS_LIMIT = 24 * 1024;
constexpr uint32_t LP_GEO_LDS_STRIDE = 5;   // 16-byte words per staged node / instance record (80 bytes: see lupin_hip_scene_create)

struct NodeRegs { float4 a, b, c; uint32_t left, right; };

struct GeoGlobal
{
    const WideNode *tlas, *blas;
    const TriVerts *tris;
    const InstanceDev *instances;
    LP_DEV NodeRegs node(bool, uint32_t i) const   // TLAS and BLAS nodes share one array: references are global indices
    {
        const WideNode nd = blas[i];
        NodeRegs r; r.a = nd.a; r.b = nd.b; r.c = nd.c; r.left = nd.d.x; r.right = nd.d.y;
        return r;
    }
    LP_DEV TriVerts tri(uint32_t i) const { return tris[i]; }           // a triangle TEST fetches through tri()
    LP_DEV TriVerts tri_fetch(uint32_t i) const { return tris[i]; }     // shading re-reads vertices through tri_fetch()
    LP_DEV InstanceDev inst(uint32_t i) const { return instances[i]; }
    const Wide4 *wide4;
    const uint32_t *inst_root4;
    struct Node4 { float4 lox, loy, loz, hix, hiy, hiz; uint4 ref; };   // the 112 bytes of a Wide4 the traversal reads
    LP_DEV Node4 node4(uint32_t i) const
    {
        const Wide4 *w = wide4 + i;
        Node4 n; n.lox = w->lox; n.loy = w->loy; n.loz = w->loz; n.hix = w->hix; n.hiy = w->hiy; n.hiz = w->hiz; n.ref = w->ref;
        return n;
    }
    LP_DEV uint32_t root4(uint32_t inst) const { return inst_root4[inst]; }
    static constexpr bool kCounting = false;
};

struct GeoLds
{
    lds_v4p base;
    uint32_t off_blas, off_tris, off_inst;
    static LP_DEV float4 f4(v4f v) { return make_float4(v.x, v.y, v.z, v.w); }
    LP_DEV NodeRegs node(bool, uint32_t i) const
    {
        lds_v4p p = base + i * LP_GEO_LDS_STRIDE;   // [BLAS | TLAS] in the blob: global node indices, like GeoGlobal
        const v4f a = p[0], b = p[1], c = p[2], d = p[3];
        NodeRegs r; r.a = f4(a); r.b = f4(b); r.c = f4(c); r.left = __float_as_uint(d.x); r.right = __float_as_uint(d.y);
        return r;
    }
    LP_DEV TriVerts tri(uint32_t i) const
    {
        lds_v4p p = base + off_tris + i * 3u;
        TriVerts t; t.v0 = f4(p[0]); t.v1 = f4(p[1]); t.v2 = f4(p[2]);
        return t;
    }
    LP_DEV TriVerts tri_fetch(uint32_t i) const { return tri(i); }
    static constexpr bool kCounting = false;
    LP_DEV InstanceDev inst(uint32_t i) const
    {
        lds_v4p p = base + off_inst + i * LP_GEO_LDS_STRIDE;
        const v4f d = p[3];
        InstanceDev in; in.r0 = f4(p[0]); in.r1 = f4(p[1]); in.r2 = f4(p[2]);
        in.blas_root = __float_as_uint(d.x); in.mat_idx = __float_as_uint(d.y); in.mesh_idx = __float_as_uint(d.z); in.flags = __float_as_uint(d.w);
        return in;
    }
};

// pathtrace_scene_debug (renderer.rs:966, pathtracer.wgsl:457-503) counts box and triangle tests per pixel
// (RAY_DEBUG_INFO, bvh_custom.wgsl:54,228,243).  Every traversal fetches internal nodes through node() and tested
// triangles through tri(), so wrapping the accessor counts them without touching the traversal code.
template <typename Base>
struct GeoCounting
{
    Base base;
    uint32_t *aabb_checks, *tri_checks;   // the calling thread's counters
    static constexpr bool kCounting = true;
    LP_DEV NodeRegs node(bool in_blas, uint32_t i) const { *aabb_checks += 2u; return base.node(in_blas, i); }
    LP_DEV TriVerts tri(uint32_t i) const { *tri_checks += 1u; return base.tri(i); }
    LP_DEV TriVerts tri_fetch(uint32_t i) const { return base.tri(i); }
    LP_DEV InstanceDev inst(uint32_t i) const { return base.inst(i); }
};

// Work accounting of the traversal kernels in THIS build's layout (bench.py's roofline numerator): every internal-node
// visit fetches one 64-byte WideNode, every triangle test one 48-byte TriVerts, every instance entry one 64-byte record.
// Same idea as GeoCounting, with the calling thread's own three tallies; the traversal code is untouched.
template <typename Base>
struct GeoTally
{
    Base base;
    uint32_t *tally;   // [0] node visits, [1] triangle tests, [2] instance entries, [3] four-wide node visits (128 bytes each)
    static constexpr bool kCounting = false;   // light culling stays on: the tally is of the work actually done
    LP_DEV NodeRegs node(bool in_blas, uint32_t i) const { tally[0] += 1u; return base.node(in_blas, i); }
    LP_DEV GeoGlobal::Node4 node4(uint32_t i) const { tally[3] += 1u; return base.node4(i); }
    LP_DEV uint32_t root4(uint32_t inst) const { return base.root4(inst); }
    LP_DEV TriVerts tri(uint32_t i) const { tally[1] += 1u; return base.tri(i); }
    LP_DEV TriVerts tri_fetch(uint32_t i) const { return base.tri_fetch(i); }
    LP_DEV InstanceDev inst(uint32_t i) const { tally[2] += 1u; return base.inst(i); }
};

LP_DEV GeoGlobal geo_global(const SceneDev &sc)
{
    GeoGlobal g; g.tlas = sc.tlas; g.blas = sc.blas; g.tris = sc.tris; g.instances = sc.instances;
    g.wide4 = sc.wide4; g.inst_root4 = sc.inst_root4;
    return g;
}
// Cooperative copy of the geometry blob into LDS at `words` (16-byte aligned); caller synchronises.
LP_DEV GeoLds geo_stage_lds(const SceneDev &sc, uint32_t *lds_words)
{
    v4f *dst = (v4f *)lds_words;
    for (uint32_t i = threadIdx.x; i < sc.geo_blob_words; i += LP_BLOCK)
    {
        const float4 t = sc.geo_blob[i];
        dst[i] = (v4f){t.x, t.y, t.z, t.w};
    }
    GeoLds g; g.base = (lds_v4p)lds_words; g.off_blas = sc.geo_off_blas; g.off_tris = sc.geo_off_tris; g.off_inst = sc.geo_off_inst;
    return g;
}

// ------------------------------------------------------------------------------------------------
// Traversal.  Per-lane stack lives in LDS, entry e of lane t at stack[e * LP_BLOCK + t]:
// bank = t mod 32 for every depth, so pushes/pops never conflict inside a wave.
// Only the far child is ever stored (the near one is visited next without a round trip), which
// is the visiting order of the reference's push-far-then-near / pop loop
// (bvh_custom.wgsl:63-94, :252-283).  Like the reference, a child is tested against the best
// hit when it is pushed, not again when it is popped.
// ------------------------------------------------------------------------------------------------

struct Closest
{
    float t, u, v;
    uint32_t tri;   // global triangle index
    uint32_t inst;
};

// Descend one BLAS (bvh_custom.wgsl:195-288) from `root`, updating `best` on strictly closer hits.
// Returns true if any triangle of this mesh replaced the best hit.
template <typename Geo>
LP_DEV bool blas_closest(const Geo &geo, uint32_t *stack, uint32_t sp_base, uint32_t root,
                         f3 o, f3 d, f3 inv_d, float eps, Closest &best)
{
    const uint32_t tid = threadIdx.x;
    uint32_t sp = sp_base;
    uint32_t cur = root;
    bool replaced = false;
    for (;;)
    {
        if (cur & REF_LEAF)
        {
            uint32_t ti = cur & ~REF_LEAF;
            for (;;)
            {
                const TriVerts tv = geo.tri(ti);
                TriHit h = tri_dst(o, d, xyz(tv.v0), xyz(tv.v1), xyz(tv.v2), eps);
                if (h.t < best.t) { best.t = h.t; best.u = h.u; best.v = h.v; best.tri = ti; replaced = true; }
                if (__float_as_uint(tv.v0.w) & LEAF_END_BITS) break;
                ti++;
            }
            if (sp == sp_base) break;
            sp--;
            cur = stack[sp * LP_BLOCK + tid];
        }
        else
        {
            const NodeRegs nd = geo.node(true, cur);
            float ld = slab_dst(o, inv_d, LP_NODE_LEFT(nd));
            float rd = slab_dst(o, inv_d, LP_NODE_RIGHT(nd));
            bool left_first = ld <= rd;
            bool push_l = ld < best.t, push_r = rd < best.t;
            uint32_t near_ref = left_first ? nd.left : nd.right;
            uint32_t far_ref = left_first ? nd.right : nd.left;
            bool push_near = left_first ? push_l : push_r;
            bool push_far = left_first ? push_r : push_l;
            if (push_far) { stack[sp * LP_BLOCK + tid] = far_ref; sp++; }
            if (push_near) { cur = near_ref; }
            else
            {
                if (sp == sp_base) break;
                sp--;
                cur = stack[sp * LP_BLOCK + tid];
            }
        }
    }
    return replaced;
}

// ray_scene_intersection (bvh_custom.wgsl:7-110) as ONE convergent loop over both levels.
//
// The reference nests the BLAS loop inside the TLAS loop; compiled as written, lanes that are between
// instances idle while their neighbours finish a BLAS.  Here TLAS and BLAS internal nodes share one
// code path (same 64-byte node format, only the ray and the node array differ by level), and the
// loop is organised "while-while": every lane first descends through internal nodes of either level
// until it holds a leaf, then the wave handles leaves (instance entry / triangles) together.
// Visiting order per lane is unchanged, so results are identical to the nested form.
template <typename Geo>
LP_DEV Closest scene_closest(const Geo &geo, const SceneDev &sc, uint32_t *stack, f3 o, f3 d, float eps)
{
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t REF_DONE = 0xFFFFFFFFu;   // not a valid leaf reference (leaf payloads are < 2^31 - 1)
    Closest best;
    best.t = LP_F32_MAX; best.u = 0.0f; best.v = 0.0f; best.tri = 0u; best.inst = 0xFFFFFFFFu;
    if (sc.num_instances == 0) return best;
    const f3 inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);

    f3 co = o, cd = d, cinv = inv_d;       // ray of the current level (world, or instance-local)
    uint32_t sp = 0;
    uint32_t blas_base = 0xFFFFFFFFu;      // stack height at instance entry; all-ones = at TLAS level
    uint32_t cur_inst = 0;
    uint32_t cur = sc.tlas_root;

    // pop the next reference; leaving an exhausted BLAS restores the world ray and keeps popping the TLAS part
    auto pop = [&]() {
        if (sp == blas_base) { blas_base = 0xFFFFFFFFu; co = o; cd = d; cinv = inv_d; }
        if (sp == 0) { cur = REF_DONE; return; }
        sp--;
        cur = stack[sp * LP_BLOCK + tid];
    };

    for (;;)
    {
        // ---- phase 1: internal nodes of either level ----
        while (!(cur & REF_LEAF))
        {
            const NodeRegs nd = geo.node(blas_base != 0xFFFFFFFFu, cur);
            float ld = slab_dst(co, cinv, LP_NODE_LEFT(nd));
            float rd = slab_dst(co, cinv, LP_NODE_RIGHT(nd));
            bool left_first = ld <= rd;
            bool push_l = ld < best.t, push_r = rd < best.t;
            uint32_t near_ref = left_first ? nd.left : nd.right;
            uint32_t far_ref = left_first ? nd.right : nd.left;
            bool push_near = left_first ? push_l : push_r;
            bool push_far = left_first ? push_r : push_l;
            if (push_far) { stack[sp * LP_BLOCK + tid] = far_ref; sp++; }
            if (push_near) cur = near_ref; else pop();
        }
        if (cur == REF_DONE) break;

        // ---- phase 2: leaves ----
        if (blas_base == 0xFFFFFFFFu)
        {
            // TLAS leaf: enter the instance (bvh_custom.wgsl:28-37)
            cur_inst = cur & ~REF_LEAF;
            const InstanceDev in = geo.inst(cur_inst);
            co = mk3(o.x * in.r0.x + o.y * in.r0.y + o.z * in.r0.z + 1.0f * in.r0.w,
                     o.x * in.r1.x + o.y * in.r1.y + o.z * in.r1.z + 1.0f * in.r1.w,
                     o.x * in.r2.x + o.y * in.r2.y + o.z * in.r2.z + 1.0f * in.r2.w);
            cd = mk3(d.x * in.r0.x + d.y * in.r0.y + d.z * in.r0.z + 0.0f * in.r0.w,
                     d.x * in.r1.x + d.y * in.r1.y + d.z * in.r1.z + 0.0f * in.r1.w,
                     d.x * in.r2.x + d.y * in.r2.y + d.z * in.r2.z + 0.0f * in.r2.w);
            // the reciprocal direction only feeds box tests: a BLAS whose root is a leaf (quads, small meshes) needs none
            if (!(in.blas_root & REF_LEAF)) cinv = mk3(1.0f / cd.x, 1.0f / cd.y, 1.0f / cd.z);
            blas_base = sp;
            cur = in.blas_root;
        }
        else
        {
            // BLAS leaf: its triangles, first-found wins ties (strict <)
            uint32_t ti = cur & ~REF_LEAF;
            for (;;)
            {
                const TriVerts tv = geo.tri(ti);
                TriHit h = tri_dst(co, cd, xyz(tv.v0), xyz(tv.v1), xyz(tv.v2), eps);
                if (h.t < best.t) { best.t = h.t; best.u = h.u; best.v = h.v; best.tri = ti; best.inst = cur_inst; }
                if (__float_as_uint(tv.v0.w) & LEAF_END_BITS) break;
                ti++;
            }
            pop();
        }
    }
    return best;
}

// ------------------------------------------------------------------------------------------------
// Wide traversal with an exactness certificate.
//
// The reference's closest hit is order-dependent only in its margins: `best` is the FIRST tested triangle that attains the
// minimum t over the triangles the traversal tested, and a subtree is skipped when its box's entry distance is not below
// the best t AT THAT MOMENT (bvh_custom.wgsl:63-94, :221, :252-283).  A traversal that visits the same boxes in another
// order returns the same triangle unless (a) two different triangles are hit at (nearly) the same t -- then visiting order
// decides which one is kept, and which boxes the kept one's t prunes -- or (b) a triangle's computed t is not consistent
// with the boxes around it, so that one order prunes it and another does not: an ill-conditioned intersection, or a
// triangle that is not inside the boxes the reference's tree stores above it (they exist: lupin_hip.hip, collapse_to_wide4).
//
// The four-wide traversal below therefore (1) prunes with a MARGIN: a child is entered while its entry distance is below
// wide_threshold(best.t) = best.t (1 + 2^-10) + abs_margin, so every triangle whose boxes are consistent with a t inside the
// margin above the final best IS tested; (2) raises `flag` when it tests a second triangle hit inside that margin of the
// current best, or a hit whose intersection is ill-conditioned (wide_hit_is_ill_conditioned) or whose triangle is marked as
// outside a box above it (TRI_LEAKY_BITS), or when its bounded stack would overflow; (3) never prunes by distance a child
// whose box does not bound its triangles (REF_LEAKY).  An unflagged ray's (t, u, v, triangle, instance) is the reference's: with m' the conditioning bound and
// (1 + m')^2 < 1 + m, the reference's winner R and this traversal's winner W each pass the other's pruning (their boxes'
// entry distances are below t_W (1 + m) resp. the reference's best at every moment), so both traversals test both; W != R
// would be a second hit inside the margin -> flagged.  Flagged rays are re-traced by the binary kernel, which IS the
// reference's order.  What the certificate assumes is stated where it is used: DESIGN.md 5 "Wide traversal".
// ------------------------------------------------------------------------------------------------

// why a query is handed to the binary tracer (diagnostics: LUPIN_VERIFY_WIDE counts them)
enum : uint32_t { WIDE_WHY_TIE = 1u, WIDE_WHY_CONDITION = 2u, WIDE_WHY_LEAKY = 4u, WIDE_WHY_STACK = 8u };
constexpr float LP_WIDE_REL = 1.0f + 0x1p-10f;
LP_DEV float wide_threshold(float best_t, float abs_margin)
{
    return __builtin_fminf(best_t * LP_WIDE_REL + abs_margin, LP_F32_MAX);   // best_t == MAX -> inf -> MAX: a miss (MAX) never passes
}

// Slab tests of a node's four children + sorting network by entry distance (children failing `d < thr` sort last as +inf).
// Returns the number of children to visit; dk / rk hold them nearest first.  The slab arithmetic is slab_dst's, operation
// for operation (same IEEE subtractions, multiplications, minNum / maxNum), written on pairs of children so that the
// subtractions and multiplications issue as packed instructions (v_pk_add_f32 / v_pk_mul_f32: two IEEE results each).
// slab_dst's verdict from the six per-axis distances, as the child's sort key: dst_near when the slab test hits, +inf when
// it misses (slab_dst returns MAX there; every use below only asks "below the threshold?" / "hit at all?").  Three chained
// selects: each comparison feeds its own select, no mask arithmetic.
LP_DEV float slab_key(float tminx, float tmaxx, float tminy, float tmaxy, float tminz, float tmaxz)
{
    float t1x = __builtin_fminf(tminx, tmaxx), t1y = __builtin_fminf(tminy, tmaxy), t1z = __builtin_fminf(tminz, tmaxz);
    float t2x = __builtin_fmaxf(tminx, tmaxx), t2y = __builtin_fmaxf(tminy, tmaxy), t2z = __builtin_fmaxf(tminz, tmaxz);
    float dst_far = __builtin_fminf(__builtin_fminf(t2x, t2y), t2z);
    float dst_near = __builtin_fmaxf(__builtin_fmaxf(t1x, t1y), t1z);
    float key = dst_far >= dst_near ? dst_near : __builtin_inff();
    key = dst_far > 0.0f ? key : __builtin_inff();
    return key;
}
// Fills dk / rk with the node's children, nearest first; a child that is not to be visited has dk = +inf (and sorts last).
LP_DEV void wide_children(const GeoGlobal::Node4 &nd, f3 co, f3 cinv, float thr, float (&dk)[4], uint32_t (&rk)[4])
{
    const lp_v2 ox = {co.x, co.x}, oy = {co.y, co.y}, oz = {co.z, co.z};
    const lp_v2 ix = {cinv.x, cinv.x}, iy = {cinv.y, cinv.y}, iz = {cinv.z, cinv.z};
    #define LP_PAIR(v, a, b) (lp_v2){(v).a, (v).b}
    const lp_v2 lx01 = (LP_PAIR(nd.lox, x, y) - ox) * ix, lx23 = (LP_PAIR(nd.lox, z, w) - ox) * ix;
    const lp_v2 ly01 = (LP_PAIR(nd.loy, x, y) - oy) * iy, ly23 = (LP_PAIR(nd.loy, z, w) - oy) * iy;
    const lp_v2 lz01 = (LP_PAIR(nd.loz, x, y) - oz) * iz, lz23 = (LP_PAIR(nd.loz, z, w) - oz) * iz;
    const lp_v2 hx01 = (LP_PAIR(nd.hix, x, y) - ox) * ix, hx23 = (LP_PAIR(nd.hix, z, w) - ox) * ix;
    const lp_v2 hy01 = (LP_PAIR(nd.hiy, x, y) - oy) * iy, hy23 = (LP_PAIR(nd.hiy, z, w) - oy) * iy;
    const lp_v2 hz01 = (LP_PAIR(nd.hiz, x, y) - oz) * iz, hz23 = (LP_PAIR(nd.hiz, z, w) - oz) * iz;
    #undef LP_PAIR
    float hit[4];   // entry distance, +inf = the ray misses the box (NaN bounds of an unused slot miss)
    hit[0] = slab_key(lx01.x, hx01.x, ly01.x, hy01.x, lz01.x, hz01.x);
    hit[1] = slab_key(lx01.y, hx01.y, ly01.y, hy01.y, lz01.y, hz01.y);
    hit[2] = slab_key(lx23.x, hx23.x, ly23.x, hy23.x, lz23.x, hz23.x);
    hit[3] = slab_key(lx23.y, hx23.y, ly23.y, hy23.y, lz23.y, hz23.y);
    rk[0] = nd.ref.x; rk[1] = nd.ref.y; rk[2] = nd.ref.z; rk[3] = nd.ref.w;
    #pragma unroll
    for (int k = 0; k < 4; k++) dk[k] = hit[k] < thr ? hit[k] : __builtin_inff();
    // A child whose box does not bound its triangles (REF_LEAKY; one triangle in 4 x 10^5 on the bistro-class meshes) is entered
    // whenever the ray passes the box, however far: a triangle below may lie in front of it; its recorded distance is -inf
    // so that no pop drops it either.  Rare: one test per node, the per-child work only in the lanes it applies to.
    if (((rk[0] | rk[1] | rk[2] | rk[3]) & REF_LEAKY) && (rk[0] & rk[1] & rk[2] & rk[3]) != REF_NONE)
    {
        #pragma unroll
        for (int k = 0; k < 4; k++)
            if (rk[k] != REF_NONE && (rk[k] & REF_LEAKY)) dk[k] = hit[k] < __builtin_inff() ? -__builtin_inff() : __builtin_inff();
    }
    auto cswap = [&](int a, int b) {
        const bool sw = dk[b] < dk[a];
        const float da = dk[a], db = dk[b];
        const uint32_t ra = rk[a], rb = rk[b];
        dk[a] = sw ? db : da; dk[b] = sw ? da : db;
        rk[a] = sw ? rb : ra; rk[b] = sw ? ra : rb;
    };
    cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2);
}

// Is the computed t of this hit too uncertain to be trusted against the boxes around the triangle?  With e1, e2 the edges,
// r = o - v0, n = e1 x e2:  t = -(n.r) / (d.n); the absolute errors of the two dot products are below 12 u |e1||e2||r| and
// 12 u |e1||e2||d| (u = 2^-24), so |t - t_exact| < 12 u (A1 + A2) + 2 u t with A1 = |e1||e2||r| / |d.n|, A2 = t |e1||e2||d| / |d.n|.
// The certificate tolerates an error of m' t + a' (m' = 2^-12: (1 + m')^2 < 1 + 2^-10 with room for the slab test's own
// three roundings; a' = a quarter of the absolute margin): each of A1, A2 must stay below L / 2, L = (m' t + a') / (12 u).
// Squared, so no root or division; the comparisons are written so that zero / NaN / inf flag.
LP_DEV bool wide_hit_is_ill_conditioned(f3 o, f3 d, f3 v0, f3 v1, f3 v2, float t, float abs_margin)
{
    const f3 e1 = sub(v1, v0), e2 = sub(v2, v0), r = sub(o, v0);
    const f3 n = cross3(e1, e2);
    const float det = dot3(d, n);
    const float ee = dot3(e1, e1) * dot3(e2, e2);
    const float L = (0x1p-12f * t + 0.25f * abs_margin) * (0x1p24f / 12.0f);
    const float budget = (L * det) * (L * det);
    const bool ok1 = 4.0f * (ee * dot3(r, r)) <= budget;
    const bool ok2 = 4.0f * ((t * t) * (ee * dot3(d, d))) <= budget;
    return !(ok1 && ok2);
}

// One triangle test of the wide traversal: updates `best` exactly like the reference's test and returns whether the ray
// must be re-traced in the reference's order (second hit within the margin of the best, or an untrustworthy hit).
LP_DEV bool wide_test_triangle(const TriVerts &tv, uint32_t ti, uint32_t inst, f3 co, f3 cd, float eps, float abs_margin, Closest &best, uint32_t *why = nullptr)
{
    const TriHit h = tri_dst(co, cd, xyz(tv.v0), xyz(tv.v1), xyz(tv.v2), eps);
    bool flag = false;
    if (h.t != LP_F32_MAX && h.t == h.t)   // a hit candidate (NaN never wins in any order)
    {
        const float lo = minf(h.t, best.t), hi = maxf(h.t, best.t);
        const bool relevant = hi <= wide_threshold(lo, abs_margin);   // within the margin of the current best (always false while best is MAX)
        if (best.t != LP_F32_MAX && relevant) { flag = true; if (why) *why |= WIDE_WHY_TIE; }
        if (relevant || h.t < best.t)
        {
            if (__float_as_uint(tv.v0.w) & TRI_LEAKY_BITS) { flag = true; if (why) *why |= WIDE_WHY_LEAKY; }
            if (wide_hit_is_ill_conditioned(co, cd, xyz(tv.v0), xyz(tv.v1), xyz(tv.v2), h.t, abs_margin)) { flag = true; if (why) *why |= WIDE_WHY_CONDITION; }
        }
        if (h.t < best.t) { best.t = h.t; best.u = h.u; best.v = h.v; best.tri = ti; best.inst = inst; }
    }
    return flag;
}

// ray_scene_intersection over the four-wide hierarchies, one ray per lane (probes and the verification kernel; the hot
// path runs the same steps phase-scheduled in k_extend_persistent<.., WIDE>).  Stack: `entries` (reference, entry
// distance) pairs per lane at stack[(2 e + {0, 1}) * LP_BLOCK + tid]; a popped entry whose recorded distance is no longer
// below the threshold is dropped without fetching its node.  `flag` = the result may differ from the reference's (re-trace).
template <typename Geo>
LP_DEV Closest scene_closest_wide(const Geo &geo, const SceneDev &sc, uint32_t *stack, uint32_t entries, f3 o, f3 d, float eps, bool &flag, uint32_t *why = nullptr)
{
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t REF_DONE = 0xFFFFFFFFu;
    const float abs_margin = eps;
    Closest best;
    best.t = LP_F32_MAX; best.u = 0.0f; best.v = 0.0f; best.tri = 0u; best.inst = 0xFFFFFFFFu;
    flag = false;
    if (sc.num_instances == 0) return best;
    const f3 inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    f3 co = o, cd = d, cinv = inv_d;
    uint32_t sp = 0, blas_base = 0xFFFFFFFFu, cur_inst = 0, cur = sc.tlas4_root;

    auto pop = [&]() {
        const float thr = wide_threshold(best.t, abs_margin);
        for (;;)
        {
            if (sp == blas_base) { blas_base = 0xFFFFFFFFu; co = o; cd = d; cinv = inv_d; }
            if (sp == 0) { cur = REF_DONE; return; }
            sp--;
            const float sd = __uint_as_float(stack[(2u * sp + 1u) * LP_BLOCK + tid]);
            if (sd < thr) { cur = stack[(2u * sp) * LP_BLOCK + tid]; return; }
        }
    };

    for (;;)
    {
        while (!(cur & REF_LEAF))
        {
            const auto nd = geo.node4(cur & REF_INDEX_MASK);
            float dk[4]; uint32_t rk[4];
            wide_children(nd, co, cinv, wide_threshold(best.t, abs_margin), dk, rk);
            if (sp + 3u > entries) { flag = true; if (why) *why |= WIDE_WHY_STACK; return best; }   // bounded stack (room for three is required): hand the ray to the binary tracer
            #pragma unroll
            for (int k = 3; k >= 1; k--)
                if (dk[k] < __builtin_inff()) { stack[(2u * sp) * LP_BLOCK + tid] = rk[k]; stack[(2u * sp + 1u) * LP_BLOCK + tid] = __float_as_uint(dk[k]); sp++; }
            if (dk[0] < __builtin_inff()) cur = rk[0]; else pop();
        }
        if (cur == REF_DONE) break;
        if (blas_base == 0xFFFFFFFFu)
        {
            cur_inst = cur & REF_INDEX_MASK;
            const InstanceDev in = geo.inst(cur_inst);
            co = mk3(o.x * in.r0.x + o.y * in.r0.y + o.z * in.r0.z + 1.0f * in.r0.w,
                     o.x * in.r1.x + o.y * in.r1.y + o.z * in.r1.z + 1.0f * in.r1.w,
                     o.x * in.r2.x + o.y * in.r2.y + o.z * in.r2.z + 1.0f * in.r2.w);
            cd = mk3(d.x * in.r0.x + d.y * in.r0.y + d.z * in.r0.z + 0.0f * in.r0.w,
                     d.x * in.r1.x + d.y * in.r1.y + d.z * in.r1.z + 0.0f * in.r1.w,
                     d.x * in.r2.x + d.y * in.r2.y + d.z * in.r2.z + 0.0f * in.r2.w);
            const uint32_t root = geo.root4(cur_inst);
            if (!(root & REF_LEAF)) cinv = mk3(1.0f / cd.x, 1.0f / cd.y, 1.0f / cd.z);
            blas_base = sp;
            cur = root;
        }
        else
        {
            uint32_t ti = cur & REF_INDEX_MASK;
            for (;;)
            {
                const TriVerts tv = geo.tri(ti);
                if (wide_test_triangle(tv, ti, cur_inst, co, cd, eps, abs_margin, best, why)) flag = true;
                if (__float_as_uint(tv.v0.w) & LEAF_END_BITS) break;
                ti++;
            }
            pop();
        }
    }
    return best;
}

// ------------------------------------------------------------------------------------------------
// Textures: software bilinear, Repeat addressing (pathtracer.wgsl:1413-1416 + the linear/Repeat
// sampler of wgpu_utils.rs:244-256)
// ------------------------------------------------------------------------------------------------

LP_DEV float half_bits_to_float(uint32_t h) { return __half2float(__ushort_as_half((unsigned short)h)); }

LP_DEV float4 fetch_texel(const SceneDev &sc, const TextureDev &t, int x, int y)
{
    uint64_t i = (uint64_t)y * t.width + (uint64_t)x;
    if (t.format == LUPIN_TEX_RGBA8_UNORM)
    {
        uint32_t p = *reinterpret_cast<const uint32_t *>(sc.texels + t.offset + i * 4);
        return make_float4((float)(p & 0xFFu) / 255.0f, (float)((p >> 8) & 0xFFu) / 255.0f,
                           (float)((p >> 16) & 0xFFu) / 255.0f, (float)(p >> 24) / 255.0f);
    }
    uint2 p = *reinterpret_cast<const uint2 *>(sc.texels + t.offset + i * 8);
    return make_float4(half_bits_to_float(p.x & 0xFFFFu), half_bits_to_float(p.x >> 16),
                       half_bits_to_float(p.y & 0xFFFFu), half_bits_to_float(p.y >> 16));
}

LP_DEV float4 lerp_texels(float4 p, float4 q, float f)
{
    float g = 1.0f - f;
    return make_float4(p.x * g + q.x * f, p.y * g + q.y * f, p.z * g + q.z * f, p.w * g + q.w * f);
}

LP_FN float4 sample_texture(const SceneDev &sc, uint32_t tex_idx, float u, float v)
{
    const TextureDev t = sc.textures[tex_idx];
    int w = (int)t.width, h = (int)t.height;
    float x = u * (float)w - 0.5f;
    float y = v * (float)h - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = f2i_sat(x0f), y0 = f2i_sat(y0f);
    // Repeat addressing: ((i mod n) + n) mod n.  For power-of-two sizes that is i & (n - 1) for every int (two's complement),
    // and (a + 1) mod n is a compare: four integer divisions (~40 instructions each on this hardware) saved per lookup.
    int xa, ya;
    if ((w & (w - 1)) == 0) xa = x0 & (w - 1); else xa = ((x0 % w) + w) % w;
    if ((h & (h - 1)) == 0) ya = y0 & (h - 1); else ya = ((y0 % h) + h) % h;
    int xb = xa + 1 == w ? 0 : xa + 1, yb = ya + 1 == h ? 0 : ya + 1;
    float4 top = lerp_texels(fetch_texel(sc, t, xa, ya), fetch_texel(sc, t, xb, ya), fx);
    float4 bot = lerp_texels(fetch_texel(sc, t, xa, yb), fetch_texel(sc, t, xb, yb), fx);
    return lerp_texels(top, bot, fy);
}

LP_DEV float srgb_to_linear1(float s)
{
    float cutoff = s < 0.04045f ? 1.0f : 0.0f;
    float higher = lpm_powf((s + 0.055f) / 1.055f, 2.4f);
    float lower = s / 12.92f;
    return higher * (1.0f - cutoff) + lower * cutoff;   // mix(higher, lower, cutoff)
}

// ------------------------------------------------------------------------------------------------
// Surface data
// ------------------------------------------------------------------------------------------------

struct MatPoint   // pathtracer.wgsl:1247-1260 (fields the integrators read)
{
    uint32_t type;
    f3 emission, color;
    float opacity, roughness, metallic, ior;
    f3 density, scattering;
    float anisotropy;
};

struct Surface   // a resolved hit: instance, mesh, triangle vertex ids
{
    InstanceDev in;
    MeshDev mesh;
    uint32_t inst;
    uint32_t gtri;
    uint32_t i0, i1, i2;
    float u, v;
};

LP_DEV Surface resolve_surface(const SceneDev &sc, uint32_t inst, uint32_t gtri, float u, float v)
{
    Surface s;
    s.in = sc.instances[inst];
    s.inst = inst;
    s.mesh = sc.inst_meshes[inst];      // (a copy per instance: fetched beside the instance record, not after it)
    s.gtri = gtri;
    s.i0 = sc.tri_indices[(size_t)gtri * 3 + 0];
    s.i1 = sc.tri_indices[(size_t)gtri * 3 + 1];
    s.i2 = sc.tri_indices[(size_t)gtri * 3 + 2];
    s.u = u; s.v = v;
    return s;
}

LP_DEV void interp_texcoords(const SceneDev &sc, const Surface &s, float &tu, float &tv)
{
    float2 a = sc.texcoords[s.mesh.texcoords_base + s.i0];
    float2 b = sc.texcoords[s.mesh.texcoords_base + s.i1];
    float2 c = sc.texcoords[s.mesh.texcoords_base + s.i2];
    float w = 1.0f - s.u - s.v;
    tu = a.x * w + b.x * s.u + c.x * s.v;
    tv = a.y * w + b.y * s.u + c.y * s.v;
}

// get_vert_color (pathtracer.wgsl:1757-1770)
LP_DEV float4 vertex_color(const SceneDev &sc, const Surface &s)
{
    if (s.mesh.colors_base == LUPIN_SENTINEL_IDX) return make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    float4 a = sc.colors[s.mesh.colors_base + s.i0];
    float4 b = sc.colors[s.mesh.colors_base + s.i1];
    float4 c = sc.colors[s.mesh.colors_base + s.i2];
    float w = 1.0f - s.u - s.v;
    return make_float4(a.x * w + b.x * s.u + c.x * s.v, a.y * w + b.y * s.u + c.y * s.v,
                       a.z * w + b.z * s.u + c.z * s.v, a.w * w + b.w * s.u + c.w * s.v);
}

// Opacity alone (the only field ray_skip_alpha_stochastically needs, bvh_custom.wgsl:168-169):
// color_sample.a * mat.color.a * vert_color.a of get_material_point (pathtracer.wgsl:1314).
LP_FN float surface_opacity(const SceneDev &sc, const Surface &s)
{
    const LupinMaterial *m = &sc.inst_materials[s.inst];
    float tex_a = 1.0f;
    if (s.mesh.texcoords_base != LUPIN_SENTINEL_IDX && m->color_tex_idx != LUPIN_SENTINEL_IDX)
    {
        float tu, tv;
        interp_texcoords(sc, s, tu, tv);
        tex_a = sample_texture(sc, m->color_tex_idx, tu, tv).w;
    }
    float4 vc = vertex_color(sc, s);
    return tex_a * m->color[3] * vc.w;
}

// get_material_point (pathtracer.wgsl:1265-1342)
// SIMPLE: the scene is known (at upload) to hold only matte materials without texture references, so the material type
// is a constant and no texture is consulted -- same values, far less code in the kernel that inlines this.
template <bool SIMPLE = false>
LP_FN MatPoint material_point(const SceneDev &sc, const Surface &s)
{
    const LupinMaterial m = sc.inst_materials[s.inst];
    MatPoint r;
    r.type = m.mat_type;
    if (SIMPLE) r.type = LUPIN_MAT_MATTE;   // a compile-time constant: the BSDF switches fold to the matte lobes

    float4 color_s = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    f3 emission_s = splat(1.0f);
    float rough_s = 1.0f, metal_s = 1.0f;
    f3 scatter_s = splat(1.0f);
    if (!SIMPLE && s.mesh.texcoords_base != LUPIN_SENTINEL_IDX)
    {
        float tu, tv;
        interp_texcoords(sc, s, tu, tv);
        if (m.color_tex_idx != LUPIN_SENTINEL_IDX)
        {
            float4 t = sample_texture(sc, m.color_tex_idx, tu, tv);
            color_s = make_float4(srgb_to_linear1(t.x), srgb_to_linear1(t.y), srgb_to_linear1(t.z), t.w);
        }
        if (m.emission_tex_idx != LUPIN_SENTINEL_IDX)
        {
            float4 t = sample_texture(sc, m.emission_tex_idx, tu, tv);
            emission_s = mk3(t.x, t.y, t.z);
        }
        if (m.roughness_tex_idx != LUPIN_SENTINEL_IDX)
        {
            float4 t = sample_texture(sc, m.roughness_tex_idx, tu, tv);
            rough_s = t.y;
            metal_s = t.z;
        }
        if (m.scattering_tex_idx != LUPIN_SENTINEL_IDX)
        {
            float4 t = sample_texture(sc, m.scattering_tex_idx, tu, tv);
            scatter_s = mk3(t.x, t.y, t.z);
        }
    }
    float4 vc = vertex_color(sc, s);

    r.color = mk3(color_s.x * m.color[0] * vc.x, color_s.y * m.color[1] * vc.y, color_s.z * m.color[2] * vc.z);
    r.opacity = color_s.w * m.color[3] * vc.w;
    r.emission = mk3(emission_s.x * m.emission[0], emission_s.y * m.emission[1], emission_s.z * m.emission[2]);
    float rg = rough_s * m.roughness;
    r.roughness = rg * rg;
    r.density = splat(0.0f);
    if (m.mat_type == LUPIN_MAT_REFRACTIVE || m.mat_type == LUPIN_MAT_VOLUMETRIC || m.mat_type == LUPIN_MAT_SUBSURFACE)
    {
        r.density = mk3(-lpm_logf(clampf(r.color.x, 0.0001f, 1.0f)) / m.tr_depth,
                        -lpm_logf(clampf(r.color.y, 0.0001f, 1.0f)) / m.tr_depth,
                        -lpm_logf(clampf(r.color.z, 0.0001f, 1.0f)) / m.tr_depth);
    }
    r.ior = m.ior;
    r.scattering = mk3(scatter_s.x * m.scattering[0], scatter_s.y * m.scattering[1], scatter_s.z * m.scattering[2]);
    r.anisotropy = m.sc_anisotropy;
    r.metallic = metal_s * m.metallic;

    if (r.type == LUPIN_MAT_MATTE || r.type == LUPIN_MAT_GLTFPBR || r.type == LUPIN_MAT_GLOSSY)
        r.roughness = clampf(r.roughness, LP_MIN_ROUGHNESS, 1.0f);
    else if (r.type == LUPIN_MAT_VOLUMETRIC)
        r.roughness = 0.0f;
    else if (r.roughness < LP_MIN_ROUGHNESS)
        r.roughness = 0.0f;
    return r;
}

// normal matrix = mat3x3(T[0].xyz, T[1].xyz, T[2].xyz), T = transpose_inverse_transform
// (pathtracer.wgsl:1748-1750, :2574)
LP_DEV f3 normal_to_world(const InstanceDev &in, f3 n)
{
    return normalize3(mat3_mul(xyz(in.r0), xyz(in.r1), xyz(in.r2), n));
}

// compute_tri_geom_normal (pathtracer.wgsl:2561-2576)
template <typename Geo>
LP_DEV f3 geometric_normal(const Geo &geo, const InstanceDev &in, uint32_t gtri)
{
    const TriVerts tv = geo.tri_fetch(gtri);
    f3 v0 = xyz(tv.v0), v1 = xyz(tv.v1), v2 = xyz(tv.v2);
    f3 local = normalize3(cross3(sub(v2, v0), sub(v1, v0)));
    return normal_to_world(in, local);
}

// compute_shading_normal (pathtracer.wgsl:1344-1384) incl. get_vert_normal (:1730-1755) and
// compute_tangents_from_uv (:1699-1727)
template <typename Geo>
LP_FN f3 shading_normal(const Geo &geo, const SceneDev &sc, const Surface &s)
{
    f3 res;
    float w = 1.0f - s.u - s.v;
    if (s.mesh.normals_base == LUPIN_SENTINEL_IDX)
    {
        res = geometric_normal(geo, s.in, s.gtri);
    }
    else
    {
        f3 n0 = xyz(sc.normals[s.mesh.normals_base + s.i0]);
        f3 n1 = xyz(sc.normals[s.mesh.normals_base + s.i1]);
        f3 n2 = xyz(sc.normals[s.mesh.normals_base + s.i2]);
        f3 local = normalize3(add(add(scale(n0, w), scale(n1, s.u)), scale(n2, s.v)));
        res = normal_to_world(s.in, local);
    }

    if (s.mesh.texcoords_base != LUPIN_SENTINEL_IDX)
    {
        uint32_t ntex = sc.inst_materials[s.inst].normal_tex_idx;
        if (ntex != LUPIN_SENTINEL_IDX)
        {
            float2 uv0 = sc.texcoords[s.mesh.texcoords_base + s.i0];
            float2 uv1 = sc.texcoords[s.mesh.texcoords_base + s.i1];
            float2 uv2 = sc.texcoords[s.mesh.texcoords_base + s.i2];
            float tu = uv0.x * w + uv1.x * s.u + uv2.x * s.v;
            float tv_ = uv0.y * w + uv1.y * s.u + uv2.y * s.v;
            const TriVerts tv = geo.tri_fetch(s.gtri);
            f3 p = sub(xyz(tv.v1), xyz(tv.v0));
            f3 q = sub(xyz(tv.v2), xyz(tv.v0));
            float sx = uv1.x - uv0.x, sy = uv2.x - uv0.x;
            float tx = uv1.y - uv0.y, ty = uv2.y - uv0.y;
            float div = sx * ty - sy * tx;
            f3 tangent_local = mk3(1.0f, 0.0f, 0.0f), bitangent_local = mk3(0.0f, 1.0f, 0.0f);
            if (div != 0.0f)
            {
                tangent_local = divs(mk3(ty * p.x - tx * q.x, ty * p.y - tx * q.y, ty * p.z - tx * q.z), div);
                bitangent_local = divs(mk3(sx * q.x - sy * p.x, sx * q.y - sy * p.y, sx * q.z - sy * p.z), div);
            }
            f3 tangent = normal_to_world(s.in, tangent_local);
            f3 bitangent = normal_to_world(s.in, bitangent_local);

            float4 ns = sample_texture(sc, ntex, tu, tv_);
            f3 nl = mk3(-1.0f + 2.0f * ns.x, -1.0f + 2.0f * ns.y, -1.0f + 2.0f * ns.z);
            f3 fz = res;
            f3 fx = normalize3(sub(tangent, scale(fz, dot3(tangent, fz))));   // orthonormalize (:2774)
            f3 fy = normalize3(cross3(fz, fx));
            if (dot3(fy, bitangent) < 0.0f) nl = scale(nl, -1.0f);
            res = normalize3(mat3_mul(fx, fy, fz, nl));
        }
    }
    return res;
}

// ------------------------------------------------------------------------------------------------
// Environments (pathtracer.wgsl:1386-1410, :2551-2605)
// ------------------------------------------------------------------------------------------------

LP_DEV void dir_to_env_uv(const LupinEnvironment &env, f3 dir, float &u, float &v)
{
    const float (*m)[4] = env.transform.m;
    f3 t = normalize3(mk3(dot3(mk3(m[0][0], m[0][1], m[0][2]), dir),
                          dot3(mk3(m[1][0], m[1][1], m[1][2]), dir),
                          dot3(mk3(m[2][0], m[2][1], m[2][2]), dir)));
    u = lpm_atan2f(t.z, t.x) / (2.0f * LP_PI);
    v = lpm_acosf(clampf(t.y, -1.0f, 1.0f)) / LP_PI;
    if (u < 0.0f) u += 1.0f;
    if (u > 1.0f) u -= 1.0f;
}

LP_FN f3 environment_radiance(const SceneDev &sc, f3 dir)
{
    f3 total = splat(0.0f);
    for (uint32_t i = 0; i < sc.num_envs; i++)
    {
        const LupinEnvironment &env = sc.environments[i];
        float u, v;
        dir_to_env_uv(env, dir, u, v);
        f3 e = mk3(env.emission[0], env.emission[1], env.emission[2]);
        if (env.emission_tex_idx != LUPIN_SENTINEL_IDX)
        {
            float4 t = sample_texture(sc, env.emission_tex_idx, u, v);
            e = mul(e, mk3(t.x, t.y, t.z));
        }
        total = add(total, e);
    }
    return total;
}

LP_FN f3 env_texel_direction(const SceneDev &sc, uint32_t env_i, uint32_t texel)
{
    const LupinEnvironment &env = sc.environments[env_i];
    const TextureDev t = sc.textures[env.emission_tex_idx];
    uint32_t cx = texel % t.width, cy = texel / t.width;
    float u = ((float)cx + 0.5f) / (float)t.width;
    float v = ((float)cy + 0.5f) / (float)t.height;
    float su, cu, sv, cv;
    lpm_sincosf(u * 2.0f * LP_PI, &su, &cu);
    lpm_sincosf(v * LP_PI, &sv, &cv);
    f3 d = mk3(cu * sv, cv, su * sv);
    const float (*m)[4] = env.transform.m;
    // transform_dir: normalize((transform * vec4(dir, 0)).xyz)  (:2656-2660)
    f3 r = mk3(m[0][0] * d.x + m[1][0] * d.y + m[2][0] * d.z + m[3][0] * 0.0f,
               m[0][1] * d.x + m[1][1] * d.y + m[2][1] * d.z + m[3][1] * 0.0f,
               m[0][2] * d.x + m[1][2] * d.y + m[2][2] * d.z + m[3][2] * 0.0f);
    return normalize3(r);
}

// ------------------------------------------------------------------------------------------------
// Fresnel / microfacet (pathtracer.wgsl:1433-1555)
// ------------------------------------------------------------------------------------------------

LP_DEV f3 reflectivity_to_eta(f3 refl)
{
    f3 r = mk3(clampf(refl.x, 0.0f, 0.99f), clampf(refl.y, 0.0f, 0.99f), clampf(refl.z, 0.0f, 0.99f));
    f3 s = sqrt3(r);
    return mk3((1.0f + s.x) / (1.0f - s.x), (1.0f + s.y) / (1.0f - s.y), (1.0f + s.z) / (1.0f - s.z));
}
LP_DEV float eta_to_reflectivity1(float eta) { return ((eta - 1.0f) * (eta - 1.0f)) / ((eta + 1.0f) * (eta + 1.0f)); }

LP_DEV f3 fresnel_schlick3(f3 color, f3 normal, f3 out_dir)
{
    if (is_zero3(color)) return splat(0.0f);
    float cosine = dot3(normal, out_dir);
    float p = lpm_powf(clampf(1.0f - fabsf(cosine), 0.0f, 1.0f), 5.0f);
    return mk3(color.x + (1.0f - color.x) * p, color.y + (1.0f - color.y) * p, color.z + (1.0f - color.z) * p);
}

LP_DEV float fresnel_dielectric(float eta, f3 normal, f3 outgoing)
{
    float cosw = fabsf(dot3(normal, outgoing));
    float sin2 = 1.0f - cosw * cosw;
    float eta2 = eta * eta;
    float cos2t = 1.0f - sin2 / eta2;
    if (cos2t < 0.0f) return 1.0f;
    float t0 = sqrtf(cos2t);
    float t1 = eta * t0;
    float t2 = eta * cosw;
    float rs = (cosw - t1) / (cosw + t1);
    float rp = (t0 - t2) / (t0 + t2);
    return (rs * rs + rp * rp) / 2.0f;
}

LP_DEV float fresnel_conductor1(float eta, float etak, float cosw, float cos2, float sin2)
{
    float eta2 = eta * eta;
    float etak2 = etak * etak;
    float t0 = eta2 - etak2 - sin2;
    float a2plusb2 = sqrtf(t0 * t0 + 4.0f * eta2 * etak2);
    float t1 = a2plusb2 + cos2;
    float a = sqrtf((a2plusb2 + t0) / 2.0f);
    float t2 = 2.0f * a * cosw;
    float rs = (t1 - t2) / (t1 + t2);
    float t3 = cos2 * a2plusb2 + sin2 * sin2;
    float t4 = t2 * sin2;
    float rp = rs * (t3 - t4) / (t3 + t4);
    return (rp + rs) / 2.0f;
}
LP_DEV f3 fresnel_conductor(f3 eta, f3 etak, f3 normal, f3 outgoing)
{
    float cosw = dot3(normal, outgoing);
    if (cosw <= 0.0f) return splat(0.0f);
    cosw = clampf(cosw, -1.0f, 1.0f);
    float cos2 = cosw * cosw;
    float sin2 = clampf(1.0f - cos2, 0.0f, 1.0f);
    return mk3(fresnel_conductor1(eta.x, etak.x, cosw, cos2, sin2),
               fresnel_conductor1(eta.y, etak.y, cosw, cos2, sin2),
               fresnel_conductor1(eta.z, etak.z, cosw, cos2, sin2));
}

// GGX only: every caller in the reference passes ggx = true
LP_DEV float ggx_distribution(float roughness, f3 normal, f3 halfway)
{
    float cosine = dot3(normal, halfway);
    if (cosine <= 0.0f) return 0.0f;
    float r2 = roughness * roughness;
    float c2 = cosine * cosine;
    return r2 / (LP_PI * (c2 * r2 + 1.0f - c2) * (c2 * r2 + 1.0f - c2));
}
LP_DEV float ggx_shadowing1(float roughness, f3 normal, f3 halfway, f3 direction)
{
    float cosine = dot3(normal, direction);
    float cosineh = dot3(halfway, direction);
    if (cosine * cosineh <= 0.0f) return 0.0f;
    float r2 = roughness * roughness;
    float c2 = cosine * cosine;
    return 2.0f * fabsf(cosine) / (fabsf(cosine) + sqrtf(c2 - r2 * c2 + r2));
}
LP_DEV float ggx_shadowing(float roughness, f3 normal, f3 halfway, f3 outgoing, f3 incoming)
{
    return ggx_shadowing1(roughness, normal, halfway, outgoing) * ggx_shadowing1(roughness, normal, halfway, incoming);
}
LP_DEV float ggx_pdf(float roughness, f3 normal, f3 halfway)   // sample_microfacet_pdf (:2209-2214)
{
    float cosine = dot3(normal, halfway);
    if (cosine < 0.0f) return 0.0f;
    return ggx_distribution(roughness, normal, halfway) * cosine;
}

// ------------------------------------------------------------------------------------------------
// Frames and direction sampling (pathtracer.wgsl:1902-1918, :2216-2229, :2424-2463)
// ------------------------------------------------------------------------------------------------

// basis_fromz(v) * local  (Pixar orthonormal basis, :2424-2434)
LP_DEV f3 from_z_frame(f3 v, f3 local)
{
    f3 z = normalize3(v);
    float sign = (z.z < 0.0f) ? -1.0f : 1.0f;   // copysignf(1, z.z) as defined at :2436
    float a = -1.0f / (sign + z.z);
    float b = z.x * z.y * a;
    f3 x = mk3(1.0f + sign * z.x * z.x * a, sign * b, -sign * z.x);
    f3 y = mk3(b, sign + z.y * z.y * a, -z.y);
    return mat3_mul(x, y, z, local);
}
LP_DEV f3 reflect_about(f3 w, f3 n) { return add(neg(w), scale(n, 2.0f * dot3(n, w))); }   // reflect_ (:2439)
LP_DEV f3 refract_through(f3 w, f3 n, float inv_eta)                                          // refract_ (:2444)
{
    float cosine = dot3(n, w);
    float k = 1.0f + inv_eta * inv_eta * (cosine * cosine - 1.0f);
    if (k < 0.0f) return splat(0.0f);
    return add(scale(neg(w), inv_eta), scale(n, inv_eta * cosine - sqrtf(k)));
}
LP_DEV bool same_hemisphere(f3 normal, f3 outgoing, f3 incoming) { return dot3(normal, outgoing) * dot3(normal, incoming) >= 0.0f; }
LP_DEV f3 face_forward(f3 normal, f3 outgoing) { return (dot3(normal, outgoing) <= 0.0f) ? neg(normal) : normal; }

LP_DEV f3 sample_cos_hemisphere(f3 normal, float r0, float r1)
{
    float z = sqrtf(r1);
    float r = sqrtf(1.0f - z * z);
    float phi = 2.0f * LP_PI * r0;
    float s, c;
    lpm_sincosf(phi, &s, &c);
    return normalize3(from_z_frame(normal, mk3(r * c, r * s, z)));
}
LP_DEV float cos_hemisphere_pdf(f3 normal, f3 direction)
{
    float cosw = dot3(normal, direction);
    return (cosw <= 0.0f) ? 0.0f : cosw / LP_PI;
}
LP_DEV f3 sample_ggx_halfway(float roughness, f3 normal, float r0, float r1)
{
    float phi = 2.0f * LP_PI * r0;
    float theta = lpm_atanf(roughness * sqrtf(r1 / (1.0f - r1)));
    float sp, cp, st, ct;
    lpm_sincosf(phi, &sp, &cp);
    lpm_sincosf(theta, &st, &ct);
    return normalize3(from_z_frame(normal, mk3(cp * st, sp * st, ct)));
}
LP_DEV f3 sample_unit_sphere(float r0, float r1)
{
    float z = 2.0f * r1 - 1.0f;
    float r = sqrtf(clampf(1.0f - z * z, 0.0f, 1.0f));
    float phi = 2.0f * LP_PI * r0;
    float s, c;
    lpm_sincosf(phi, &s, &c);
    return mk3(r * c, r * s, z);
}

LP_DEV bool mat_is_delta(const MatPoint &m)
{
    return ((m.type == LUPIN_MAT_REFLECTIVE || m.type == LUPIN_MAT_REFRACTIVE || m.type == LUPIN_MAT_TRANSPARENT) && m.roughness == 0.0f) ||
           m.type == LUPIN_MAT_VOLUMETRIC;
}
LP_DEV bool mat_is_volumetric(const MatPoint &m)
{
    return m.type == LUPIN_MAT_REFRACTIVE || m.type == LUPIN_MAT_VOLUMETRIC || m.type == LUPIN_MAT_SUBSURFACE;
}

// ------------------------------------------------------------------------------------------------
// BSDF sample / eval / pdf (pathtracer.wgsl:1789-1900, :1951-2090, :2097-2207)
// ------------------------------------------------------------------------------------------------

LP_DEV f3 gltf_reflectivity(const MatPoint &m)
{
    float base = eta_to_reflectivity1(m.ior);
    float g = 1.0f - m.metallic;
    return mk3(base * g + m.color.x * m.metallic, base * g + m.color.y * m.metallic, base * g + m.color.z * m.metallic);
}

LP_FN f3 bsdf_sample(const MatPoint &m, f3 normal, f3 outgoing, float rnl, float r0, float r1)
{
    if (m.roughness == 0.0f) return splat(0.0f);
    switch (m.type)
    {
    case LUPIN_MAT_MATTE:
        return sample_cos_hemisphere(face_forward(normal, outgoing), r0, r1);
    case LUPIN_MAT_GLOSSY:
    {
        f3 up = face_forward(normal, outgoing);
        if (rnl < fresnel_dielectric(m.ior, up, outgoing))
        {
            f3 h = sample_ggx_halfway(m.roughness, up, r0, r1);
            f3 in = reflect_about(outgoing, h);
            return same_hemisphere(up, outgoing, in) ? in : splat(0.0f);
        }
        return sample_cos_hemisphere(up, r0, r1);
    }
    case LUPIN_MAT_REFLECTIVE:
    {
        f3 up = face_forward(normal, outgoing);
        f3 h = sample_ggx_halfway(m.roughness, up, r0, r1);
        f3 in = reflect_about(outgoing, h);
        return same_hemisphere(up, outgoing, in) ? in : splat(0.0f);
    }
    case LUPIN_MAT_TRANSPARENT:
    {
        f3 up = face_forward(normal, outgoing);
        f3 h = sample_ggx_halfway(m.roughness, up, r0, r1);
        if (rnl < fresnel_dielectric(m.ior, h, outgoing))
        {
            f3 in = reflect_about(outgoing, h);
            return same_hemisphere(up, outgoing, in) ? in : splat(0.0f);
        }
        f3 reflected = reflect_about(outgoing, h);
        f3 in = neg(reflect_about(reflected, up));
        return same_hemisphere(up, outgoing, in) ? splat(0.0f) : in;
    }
    case LUPIN_MAT_REFRACTIVE:
    case LUPIN_MAT_SUBSURFACE:
    {
        bool entering = dot3(normal, outgoing) >= 0.0f;
        f3 up = entering ? normal : neg(normal);
        f3 h = sample_ggx_halfway(m.roughness, up, r0, r1);
        if (rnl < fresnel_dielectric(entering ? m.ior : 1.0f / m.ior, h, outgoing))
        {
            f3 in = reflect_about(outgoing, h);
            return same_hemisphere(up, outgoing, in) ? in : splat(0.0f);
        }
        f3 in = refract_through(outgoing, h, entering ? 1.0f / m.ior : m.ior);
        return same_hemisphere(up, outgoing, in) ? splat(0.0f) : in;
    }
    case LUPIN_MAT_GLTFPBR:
    {
        f3 up = face_forward(normal, outgoing);
        f3 fs = fresnel_schlick3(gltf_reflectivity(m), up, outgoing);
        if (rnl < (fs.x + fs.y + fs.z) / 3.0f)
        {
            f3 h = sample_ggx_halfway(m.roughness, up, r0, r1);
            f3 in = reflect_about(outgoing, h);
            return same_hemisphere(up, outgoing, in) ? in : splat(0.0f);
        }
        return sample_cos_hemisphere(up, r0, r1);
    }
    default:
        return splat(0.0f);
    }
}

LP_FN f3 bsdf_eval(const MatPoint &m, f3 normal, f3 outgoing, f3 incoming)
{
    if (m.roughness == 0.0f) return splat(0.0f);
    float ndi = dot3(normal, incoming), ndo = dot3(normal, outgoing);
    switch (m.type)
    {
    case LUPIN_MAT_MATTE:
    {
        if (ndi * ndo <= 0.0f) return splat(0.0f);
        float a = fabsf(dot3(normal, incoming));
        return mk3(m.color.x / LP_PI * a, m.color.y / LP_PI * a, m.color.z / LP_PI * a);
    }
    case LUPIN_MAT_GLOSSY:
    {
        if (ndi * ndo <= 0.0f) return splat(0.0f);
        f3 up = face_forward(normal, outgoing);
        float F1 = fresnel_dielectric(m.ior, up, outgoing);
        f3 h = normalize3(add(incoming, outgoing));
        float F = fresnel_dielectric(m.ior, h, incoming);
        float D = ggx_distribution(m.roughness, up, h);
        float G = ggx_shadowing(m.roughness, up, h, outgoing, incoming);
        float ai = fabsf(dot3(up, incoming));
        float spec = 1.0f * F * D * G / (4.0f * dot3(up, outgoing) * dot3(up, incoming)) * ai;
        return mk3(m.color.x * (1.0f - F1) / LP_PI * ai + spec,
                   m.color.y * (1.0f - F1) / LP_PI * ai + spec,
                   m.color.z * (1.0f - F1) / LP_PI * ai + spec);
    }
    case LUPIN_MAT_REFLECTIVE:
    {
        if (ndi * ndo <= 0.0f) return splat(0.0f);
        f3 up = face_forward(normal, outgoing);
        f3 h = normalize3(add(incoming, outgoing));
        f3 F = fresnel_conductor(reflectivity_to_eta(m.color), splat(0.0f), h, incoming);
        float D = ggx_distribution(m.roughness, up, h);
        float G = ggx_shadowing(m.roughness, up, h, outgoing, incoming);
        float den = 4.0f * dot3(up, outgoing) * dot3(up, incoming);
        float ai = fabsf(dot3(up, incoming));
        return mk3(F.x * D * G / den * ai, F.y * D * G / den * ai, F.z * D * G / den * ai);
    }
    case LUPIN_MAT_TRANSPARENT:
    {
        f3 up = face_forward(normal, outgoing);
        if (ndi * ndo >= 0.0f)
        {
            f3 h = normalize3(add(incoming, outgoing));
            float F = fresnel_dielectric(m.ior, h, outgoing);
            float D = ggx_distribution(m.roughness, up, h);
            float G = ggx_shadowing(m.roughness, up, h, outgoing, incoming);
            float val = 1.0f * F * D * G / (4.0f * dot3(up, outgoing) * dot3(up, incoming)) * fabsf(dot3(up, incoming));
            return splat(val);
        }
        f3 reflected = reflect_about(neg(incoming), up);
        f3 h = normalize3(add(reflected, outgoing));
        float F = fresnel_dielectric(m.ior, h, outgoing);
        float D = ggx_distribution(m.roughness, up, h);
        float G = ggx_shadowing(m.roughness, up, h, outgoing, reflected);
        float den = 4.0f * dot3(up, outgoing) * dot3(up, reflected);
        float ar = fabsf(dot3(up, reflected));
        return mk3(m.color.x * (1.0f - F) * D * G / den * ar,
                   m.color.y * (1.0f - F) * D * G / den * ar,
                   m.color.z * (1.0f - F) * D * G / den * ar);
    }
    case LUPIN_MAT_REFRACTIVE:
    case LUPIN_MAT_SUBSURFACE:
    {
        bool entering = ndo >= 0.0f;
        f3 up = entering ? normal : neg(normal);
        float rel_ior = entering ? m.ior : 1.0f / m.ior;
        if (ndi * ndo >= 0.0f)
        {
            f3 h = normalize3(add(incoming, outgoing));
            float F = fresnel_dielectric(rel_ior, h, outgoing);
            float D = ggx_distribution(m.roughness, up, h);
            float G = ggx_shadowing(m.roughness, up, h, outgoing, incoming);
            return splat(1.0f * F * D * G / fabsf(4.0f * dot3(normal, outgoing) * dot3(normal, incoming)) * fabsf(dot3(normal, incoming)));
        }
        f3 h = scale(neg(normalize3(add(lscale(rel_ior, incoming), outgoing))), entering ? 1.0f : -1.0f);
        float F = fresnel_dielectric(rel_ior, h, outgoing);
        float D = ggx_distribution(m.roughness, up, h);
        float G = ggx_shadowing(m.roughness, up, h, outgoing, incoming);
        float pw = rel_ior * dot3(h, incoming) + dot3(h, outgoing);
        float val = 1.0f * fabsf((dot3(outgoing, h) * dot3(incoming, h)) / (dot3(outgoing, normal) * dot3(incoming, normal))) *
                    (1.0f - F) * D * G / (pw * pw) * fabsf(dot3(normal, incoming));
        return splat(val);
    }
    case LUPIN_MAT_GLTFPBR:
    {
        if (ndi * ndo <= 0.0f) return splat(0.0f);
        f3 refl = gltf_reflectivity(m);
        f3 up = face_forward(normal, outgoing);
        f3 F1 = fresnel_schlick3(refl, up, outgoing);
        f3 h = normalize3(add(incoming, outgoing));
        f3 F = fresnel_schlick3(refl, h, incoming);
        float D = ggx_distribution(m.roughness, up, h);
        float G = ggx_shadowing(m.roughness, up, h, outgoing, incoming);
        float den = 4.0f * dot3(up, outgoing) * dot3(up, incoming);
        float ai = fabsf(dot3(up, incoming));
        float dm = 1.0f - m.metallic;
        return mk3(m.color.x * dm * (1.0f - F1.x) / LP_PI * ai + F.x * D * G / den * ai,
                   m.color.y * dm * (1.0f - F1.y) / LP_PI * ai + F.y * D * G / den * ai,
                   m.color.z * dm * (1.0f - F1.z) / LP_PI * ai + F.z * D * G / den * ai);
    }
    default:
        return splat(0.0f);
    }
}

LP_FN float bsdf_pdf(const MatPoint &m, f3 normal, f3 outgoing, f3 incoming)
{
    if (m.roughness == 0.0f) return 0.0f;
    float ndi = dot3(normal, incoming), ndo = dot3(normal, outgoing);
    switch (m.type)
    {
    case LUPIN_MAT_MATTE:
        if (ndi * ndo <= 0.0f) return 0.0f;
        return cos_hemisphere_pdf(face_forward(normal, outgoing), incoming);
    case LUPIN_MAT_GLOSSY:
    {
        if (ndi * ndo <= 0.0f) return 0.0f;
        f3 up = face_forward(normal, outgoing);
        f3 h = normalize3(add(outgoing, incoming));
        float F = fresnel_dielectric(m.ior, up, outgoing);
        return F * ggx_pdf(m.roughness, up, h) / (4.0f * fabsf(dot3(outgoing, h))) + (1.0f - F) * cos_hemisphere_pdf(up, incoming);
    }
    case LUPIN_MAT_REFLECTIVE:
    {
        if (ndi * ndo <= 0.0f) return 0.0f;
        f3 up = face_forward(normal, outgoing);
        f3 h = normalize3(add(outgoing, incoming));
        return ggx_pdf(m.roughness, up, h) / (4.0f * fabsf(dot3(outgoing, h)));
    }
    case LUPIN_MAT_TRANSPARENT:
    {
        f3 up = face_forward(normal, outgoing);
        if (ndi * ndo >= 0.0f)
        {
            f3 h = normalize3(add(incoming, outgoing));
            return fresnel_dielectric(m.ior, h, outgoing) * ggx_pdf(m.roughness, up, h) / (4.0f * fabsf(dot3(outgoing, h)));
        }
        f3 reflected = reflect_about(neg(incoming), up);
        f3 h = normalize3(add(reflected, outgoing));
        float dd = (1.0f - fresnel_dielectric(m.ior, h, outgoing)) * ggx_pdf(m.roughness, up, h);
        return dd / (4.0f * fabsf(dot3(outgoing, h)));
    }
    case LUPIN_MAT_REFRACTIVE:
    case LUPIN_MAT_SUBSURFACE:
    {
        bool entering = ndo >= 0.0f;
        f3 up = entering ? normal : neg(normal);
        float rel_ior = entering ? m.ior : 1.0f / m.ior;
        if (ndi * ndo >= 0.0f)
        {
            f3 h = normalize3(add(incoming, outgoing));
            return fresnel_dielectric(rel_ior, h, outgoing) * ggx_pdf(m.roughness, up, h) / (4.0f * fabsf(dot3(outgoing, h)));
        }
        f3 h = scale(neg(normalize3(add(lscale(rel_ior, incoming), outgoing))), entering ? 1.0f : -1.0f);
        float pw = rel_ior * dot3(h, incoming) + dot3(h, outgoing);
        return (1.0f - fresnel_dielectric(rel_ior, h, outgoing)) * ggx_pdf(m.roughness, up, h) * fabsf(dot3(h, incoming)) / (pw * pw);
    }
    case LUPIN_MAT_GLTFPBR:
    {
        if (ndi * ndo <= 0.0f) return 0.0f;
        f3 up = face_forward(normal, outgoing);
        f3 h = normalize3(add(outgoing, incoming));
        f3 fs = fresnel_schlick3(gltf_reflectivity(m), up, outgoing);
        float F = (fs.x + fs.y + fs.z) / 3.0f;
        return F * ggx_pdf(m.roughness, up, h) / (4.0f * fabsf(dot3(outgoing, h))) + (1.0f - F) * cos_hemisphere_pdf(up, incoming);
    }
    default:
        return 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// Delta lobes (pathtracer.wgsl:2231-2404)
// ------------------------------------------------------------------------------------------------

LP_FN f3 delta_sample(const MatPoint &m, f3 normal, f3 outgoing, float rnl)
{
    if (m.roughness != 0.0f) return splat(0.0f);
    switch (m.type)
    {
    case LUPIN_MAT_REFLECTIVE:
        return reflect_about(outgoing, face_forward(normal, outgoing));
    case LUPIN_MAT_TRANSPARENT:
    {
        f3 up = face_forward(normal, outgoing);
        return (rnl < fresnel_dielectric(m.ior, up, outgoing)) ? reflect_about(outgoing, up) : neg(outgoing);
    }
    case LUPIN_MAT_REFRACTIVE:
    {
        if (fabsf(m.ior - 1.0f) < 1e-3f) return neg(outgoing);
        bool entering = dot3(normal, outgoing) >= 0.0f;
        f3 up = entering ? normal : neg(normal);
        float rel_ior = entering ? m.ior : 1.0f / m.ior;
        if (rnl < fresnel_dielectric(rel_ior, up, outgoing)) return reflect_about(outgoing, up);
        return refract_through(outgoing, up, 1.0f / rel_ior);
    }
    case LUPIN_MAT_VOLUMETRIC:
        return neg(outgoing);
    default:
        return splat(0.0f);
    }
}

LP_FN f3 delta_eval(const MatPoint &m, f3 normal, f3 outgoing, f3 incoming)
{
    if (m.roughness != 0.0f) return splat(0.0f);
    float side = dot3(normal, incoming) * dot3(normal, outgoing);
    switch (m.type)
    {
    case LUPIN_MAT_REFLECTIVE:
        if (side <= 0.0f) return splat(0.0f);
        return fresnel_conductor(reflectivity_to_eta(m.color), splat(0.0f), face_forward(normal, outgoing), outgoing);
    case LUPIN_MAT_TRANSPARENT:
    {
        f3 up = face_forward(normal, outgoing);
        float F = fresnel_dielectric(m.ior, up, outgoing);
        if (side >= 0.0f) return splat(1.0f * F);
        return scale(m.color, 1.0f - F);
    }
    case LUPIN_MAT_REFRACTIVE:
    {
        if (fabsf(m.ior - 1.0f) < 1e-3f) return (side <= 0.0f) ? splat(1.0f) : splat(0.0f);
        bool entering = dot3(normal, outgoing) >= 0.0f;
        f3 up = entering ? normal : neg(normal);
        float rel_ior = entering ? m.ior : 1.0f / m.ior;
        float F = fresnel_dielectric(rel_ior, up, outgoing);
        if (side >= 0.0f) return splat(1.0f * F);
        return splat(1.0f * (1.0f / (rel_ior * rel_ior)) * (1.0f - F));
    }
    case LUPIN_MAT_VOLUMETRIC:
        return (side >= 0.0f) ? splat(0.0f) : splat(1.0f);
    default:
        return splat(0.0f);
    }
}

LP_FN float delta_pdf(const MatPoint &m, f3 normal, f3 outgoing, f3 incoming)
{
    if (m.roughness != 0.0f) return 0.0f;
    float side = dot3(normal, incoming) * dot3(normal, outgoing);
    switch (m.type)
    {
    case LUPIN_MAT_REFLECTIVE:
        return (side <= 0.0f) ? 0.0f : 1.0f;
    case LUPIN_MAT_TRANSPARENT:
    {
        float F = fresnel_dielectric(m.ior, face_forward(normal, outgoing), outgoing);
        return (side >= 0.0f) ? F : 1.0f - F;
    }
    case LUPIN_MAT_REFRACTIVE:
    {
        if (fabsf(m.ior - 1.0f) < 1e-3f) return (side < 0.0f) ? 1.0f : 0.0f;
        bool entering = dot3(normal, outgoing) >= 0.0f;
        f3 up = entering ? normal : neg(normal);
        float rel_ior = entering ? m.ior : 1.0f / m.ior;
        float F = fresnel_dielectric(rel_ior, up, outgoing);
        return (side >= 0.0f) ? F : (1.0f - F);
    }
    case LUPIN_MAT_VOLUMETRIC:
        return (side >= 0.0f) ? 0.0f : 1.0f;
    default:
        return 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// Homogeneous media (pathtracer.wgsl:1920-1949, :2092-2095, :2339-2347, :2406-2422)
// ------------------------------------------------------------------------------------------------

struct Medium { f3 density, scattering; float anisotropy; };

LP_DEV float f3_at(f3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

LP_DEV float medium_sample_distance(f3 density, float max_distance, float rl, float rd)
{
    int ch = f2i_sat(rl * 3.0f);
    ch = ch < 0 ? 0 : (ch > 2 ? 2 : ch);
    float dc = f3_at(density, ch);
    float distance = (dc == 0.0f) ? LP_F32_MAX : -lpm_logf(1.0f - rd) / dc;
    return minf(distance, max_distance);
}
LP_DEV f3 medium_transmittance(f3 density, float distance)
{
    return mk3(lpm_expf(-density.x * distance), lpm_expf(-density.y * distance), lpm_expf(-density.z * distance));
}
LP_DEV float medium_distance_pdf(f3 density, float distance, float max_distance)
{
    if (distance < max_distance)
    {
        f3 e = medium_transmittance(density, distance);
        return dot3(mul(density, e), splat(1.0f)) / 3.0f;
    }
    return dot3(medium_transmittance(density, max_distance), splat(1.0f)) / 3.0f;
}
LP_DEV f3 phase_sample(const Medium &md, f3 outgoing, float r0, float r1)
{
    if (is_zero3(md.density)) return splat(0.0f);
    float g = md.anisotropy;
    float cos_theta;
    if (fabsf(g) < 1e-3f) cos_theta = 1.0f - 2.0f * r1;
    else
    {
        float square = (1.0f - g * g) / (1.0f + g - 2.0f * g * r1);
        cos_theta = (1.0f + g * g - square * square) / (2.0f * g);
    }
    float sin_theta = sqrtf(maxf(0.0f, 1.0f - cos_theta * cos_theta));
    float phi = 2.0f * LP_PI * r0;
    float s, c;
    lpm_sincosf(phi, &s, &c);
    return from_z_frame(neg(outgoing), mk3(sin_theta * c, sin_theta * s, cos_theta));
}
LP_DEV float phase_pdf(const Medium &md, f3 outgoing, f3 incoming)
{
    if (is_zero3(md.density)) return 0.0f;
    float g = md.anisotropy;
    float cosine = -dot3(outgoing, incoming);
    float denom = 1.0f + g * g - 2.0f * g * cosine;
    return (1.0f - g * g) / (4.0f * LP_PI * denom * sqrtf(denom));
}
LP_DEV f3 phase_eval(const Medium &md, f3 outgoing, f3 incoming)
{
    if (is_zero3(md.density)) return splat(0.0f);
    float p = phase_pdf(md, outgoing, incoming);
    return scale(mul(md.scattering, md.density), p);
}

// ------------------------------------------------------------------------------------------------
// Light sampling (pathtracer.wgsl:2468-2549, :2610-2638) and its pdf (bvh_custom.wgsl:112-152)
// ------------------------------------------------------------------------------------------------

LP_DEV uint32_t alias_pick(const SceneDev &sc, AliasRange rg, uint32_t &rng)
{
    uint32_t slot = rnd_range(rng, rg.count);
    const LupinAliasBin bin = sc.alias_bins[rg.offset + slot];
    return (rnd(rng) >= bin.alias_threshold) ? bin.alias : slot;
}

LP_FN f3 lights_sample(const SceneDev &sc, f3 pos, uint32_t &rng)
{
    uint32_t nl = sc.num_lights, ne = sc.num_envs;
    if (nl + ne == 0) return splat(0.0f);
    uint32_t pick = rnd_range(rng, nl + ne);
    if (pick < nl)
    {
        uint32_t ltri = alias_pick(sc, sc.alias_ranges[pick], rng);
        const InstanceDev in = sc.instances[sc.lights[pick].instance_idx];
        float ra = rnd(rng), rb = rnd(rng);
        float sq = sqrtf(ra);
        float tu = 1.0f - sq, tv = rb * sq;     // random_tri_uv (:1675-1679)

        // local->world = inverse of the stored world->local affine (mat4x3f_inverse, :2790-2802)
        f3 a0 = mk3(in.r0.x, in.r1.x, in.r2.x), a1 = mk3(in.r0.y, in.r1.y, in.r2.y);
        f3 a2 = mk3(in.r0.z, in.r1.z, in.r2.z), a3 = mk3(in.r0.w, in.r1.w, in.r2.w);
        f3 cyz = cross3(a1, a2), czx = cross3(a2, a0), cxy = cross3(a0, a1);
        float idet = 1.0f / dot3(a0, cyz);
        f3 m0 = scale(mk3(cyz.x, czx.x, cxy.x), idet);
        f3 m1 = scale(mk3(cyz.y, czx.y, cxy.y), idet);
        f3 m2 = scale(mk3(cyz.z, czx.z, cxy.z), idet);
        f3 m3 = neg(mat3_mul(m0, m1, m2, a3));

        const TriVerts tv3 = sc.tris[sc.meshes[in.mesh_idx].tri_offset + ltri];
        float w = 1.0f - tu - tv;
        f3 lp = add(add(scale(xyz(tv3.v0), w), scale(xyz(tv3.v1), tu)), scale(xyz(tv3.v2), tv));
        f3 wp = add(add(add(scale(m0, lp.x), scale(m1, lp.y)), scale(m2, lp.z)), scale(m3, 1.0f));
        return normalize3(sub(wp, pos));
    }
    uint32_t ei = pick - nl;
    if (sc.environments[ei].emission_tex_idx == LUPIN_SENTINEL_IDX)
    {
        float ra = rnd(rng), rb = rnd(rng);
        return sample_unit_sphere(ra, rb);
    }
    uint32_t texel = alias_pick(sc, sc.env_alias_ranges[ei], rng);
    return env_texel_direction(sc, ei, texel);
}

template <typename Geo>
LP_FN float lights_pdf(const Geo &geo, const SceneDev &sc, uint32_t *stack, f3 pos, f3 incoming, float eps)
{
    float pdf = 0.0f;
    // every emissive instance: march the ray through its BLAS (<= 100 crossings), no occlusion test.
    // Phase 1 (wave-uniform loop, scalar loads): which lights can this ray reach at all?  Conservative sphere test (the
    // radius is padded at upload, the test's own rounding by an explicit error term).  Phase 2: each lane walks its own
    // candidates in increasing light order, so the sum below has the reference's order and the skipped terms are +0.0f.
    float mesh_pdf = 0.0f;
    // (the cull is not part of the arithmetic contract -- it only has to be conservative -- so it uses fused multiply-adds)
    const float inv_dd = 1.0f / dot3(incoming, incoming);
    // the cull of one group of up to 32 lights starting at `base`: bit k = light base + k may be reached
    auto cull32 = [&](uint32_t base) -> uint32_t {
        if (base >= sc.num_lights) return 0u;
        const uint32_t cnt = (sc.num_lights - base) < 32u ? (sc.num_lights - base) : 32u;
        uint32_t mask = 0u;
        for (uint32_t k4 = 0; k4 < cnt; k4 += 2u)
        {
            // two bounds per scalar fetch (32 aligned bytes; the array is padded to whole groups of four at upload) and two
            // lights in flight: the packed straight-line form of four costs k_shade its last registers
            struct alignas(32) Bounds2 { float4 b[2]; };
            const Bounds2 q = *reinterpret_cast<const Bounds2 *>(sc.light_bounds + base + k4);
            #pragma unroll
            for (uint32_t j = 0; j < 2u; j++)
            {
                const uint32_t k = k4 + j;
                const float4 b = q.b[j];                  // centre, padded radius SQUARED
                const f3 v = mk3(b.x - pos.x, b.y - pos.y, b.z - pos.z);
                const float vv = __builtin_fmaf(v.x, v.x, __builtin_fmaf(v.y, v.y, v.z * v.z));
                const float vd = __builtin_fmaf(v.x, incoming.x, __builtin_fmaf(v.y, incoming.y, v.z * incoming.z));
                // distance^2 from the centre to the ray's line:  vv - vd^2 / dd.  Its rounding error stays below 1e-6 vv (three
                // fused steps per dot product, one division per ray), the right-hand side allows eight times that; hits need
                // t >= eps > 0, so a sphere behind the origin only counts if the origin is inside it.  The comparisons are
                // written so that NaN / inf keep the light.
                const float line_d2 = __builtin_fmaf(-(vd * inv_dd), vd, vv);
                const bool behind = vd < 0.0f && eps > 0.0f;
                // (selects, not branches: the loop is wave-uniform and stays straight-line code)
                const float lhs = behind ? vv : line_d2;
                const float rhs = behind ? b.w : __builtin_fmaf(8e-6f, vv, b.w);
                const bool out_of_reach = lhs > rhs;
                // "not provably out of reach": NaN / inf operands keep the light
                // (the debug heat maps count every light's tests, so the counting accessor keeps them all)
                if (k < cnt && (Geo::kCounting || !out_of_reach)) mask |= 1u << k;
            }
        }
        return mask;
    };
    // Four groups are culled before any light is marched: the march loop below runs as long as the lane with the most
    // candidates, so one loop over a lane's candidates of 128 lights costs the wave max(sum) iterations where a loop per group
    // cost sum(max).  Each lane still takes its candidates in increasing light order.
    for (uint32_t base = 0; base < sc.num_lights; base += 128u)
    {
        uint32_t m0 = cull32(base), m1 = cull32(base + 32u), m2 = cull32(base + 64u), m3 = cull32(base + 96u);
        while (m0 | m1 | m2 | m3)
        {
        const uint32_t g = m0 ? 0u : (m1 ? 1u : (m2 ? 2u : 3u));
        const uint32_t mg = m0 ? m0 : (m1 ? m1 : (m2 ? m2 : m3));
        const uint32_t i = base + 32u * g + (uint32_t)__builtin_ctz(mg);
        const uint32_t rest = mg & (mg - 1u);
        if (g == 0u) m0 = rest; else if (g == 1u) m1 = rest; else if (g == 2u) m2 = rest; else m3 = rest;
        const LupinLight light = sc.lights[i];
        const InstanceDev in = geo.inst(light.instance_idx);
        float light_pdf = 0.0f;
        f3 next_pos = pos;
        for (uint32_t crossing = 0; crossing < 100u; crossing++)
        {
            // transform_ray_without_normalizing_direction with the instance's world->local (:2671-2680)
            f3 lo = mk3(in.r0.x * next_pos.x + in.r0.y * next_pos.y + in.r0.z * next_pos.z + in.r0.w * 1.0f,
                        in.r1.x * next_pos.x + in.r1.y * next_pos.y + in.r1.z * next_pos.z + in.r1.w * 1.0f,
                        in.r2.x * next_pos.x + in.r2.y * next_pos.y + in.r2.z * next_pos.z + in.r2.w * 1.0f);
            f3 ld = mk3(in.r0.x * incoming.x + in.r0.y * incoming.y + in.r0.z * incoming.z + in.r0.w * 0.0f,
                        in.r1.x * incoming.x + in.r1.y * incoming.y + in.r1.z * incoming.z + in.r1.w * 0.0f,
                        in.r2.x * incoming.x + in.r2.y * incoming.y + in.r2.z * incoming.z + in.r2.w * 0.0f);
            Closest c;
            c.t = LP_F32_MAX; c.u = 0.0f; c.v = 0.0f; c.tri = 0u; c.inst = 0u;
            // emissive meshes are mostly quads whose BLAS is a single leaf: no box test, so no reciprocal direction
            f3 linv = splat(0.0f);
            if (!(in.blas_root & REF_LEAF)) linv = mk3(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z);
            blas_closest(geo, stack, 0u, in.blas_root, lo, ld, linv, eps, c);
            if (c.t == LP_F32_MAX) break;
            f3 ln = geometric_normal(geo, in, c.tri);
            f3 light_pos = add(next_pos, scale(incoming, c.t));
            f3 dl = sub(light_pos, pos);
            float dist2 = dot3(dl, dl);
            float cos_theta = fabsf(dot3(ln, incoming));
            light_pdf += dist2 / (cos_theta * light.area);
            next_pos = add(light_pos, incoming);
        }
        mesh_pdf += light_pdf;
        }
    }
    pdf += mesh_pdf;

    for (uint32_t i = 0; i < sc.num_envs; i++)
    {
        const LupinEnvironment &env = sc.environments[i];
        if (env.emission_tex_idx == LUPIN_SENTINEL_IDX) { pdf += 1.0f / (4.0f * LP_PI); continue; }
        const TextureDev t = sc.textures[env.emission_tex_idx];
        float u, v;
        dir_to_env_uv(env, incoming, u, v);
        uint32_t cx = f2u_sat(u * (float)t.width), cy = f2u_sat(v * (float)t.height);
        cx = cx < t.width - 1 ? cx : t.width - 1;
        cy = cy < t.height - 1 ? cy : t.height - 1;
        float prob = sc.alias_bins[sc.env_alias_ranges[i].offset + cy * t.width + cx].prob;
        float solid_angle = (2.0f * LP_PI / (float)t.width) * (LP_PI / (float)t.height) *
                            lpm_sinf(LP_PI * ((float)cy + 0.5f) / (float)t.height);
        pdf += prob / solid_angle;
    }
    pdf /= (float)(sc.num_lights + sc.num_envs);
    return pdf;
}

}  // namespace lpd
