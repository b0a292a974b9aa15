// lupin_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of LupinPathTracer's software-BVH megakernel: `pathtrace_main` and everything
// it calls in lupin/src/shaders/pathtracer.wgsl + lupin/src/shaders/bvh_custom.wgsl.  One C++
// function per WGSL function, same names, same statement order, f32 arithmetic, each citing the
// WGSL lines it follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// may load this library; the shipped HIP path never links or calls it.
//
// Parity status: the reference path (WGSL through wgpu/naga/Vulkan, driven from Rust) can be
// neither compiled nor run in this environment, so this restatement is pinned by the
// reference's own fixtures only: the furnace1 known-answer golden, the alias-table unit-test
// vectors (data_structures.rs:1086-1157) and the surviving golden renders (statistical).
//
// Choices where WGSL leaves behaviour to the driver (stated in DESIGN.md):
//   * transcendentals = include/lupin_detmath.h (correctly rounded f32);
//   * dot/cross/matrix products are summed left to right, no FMA contraction;
//   * min(a,b) = b<a ? b : a, max(a,b) = a<b ? b : a (GLSL.std.450 FMin/FMax on ordered input), except in
//     the ray/box slab test, where they are IEEE minNum/maxNum (the hardware min/max of the target);
//   * normalize(v) = v * (1 / sqrt(dot(v,v))) (lpm_normalize3f, one division instead of three); mix(a,b,t) = a*(1-t) + b*t;
//   * pow(x, 2.0) is x*x (what every Vulkan compiler folds it to; pow of a negative base is
//     otherwise undefined in WGSL) -- pathtracer.wgsl:2067,2190;
//   * textureSampleLevel = software bilinear, Repeat addressing, level 0, texel centres at +0.5;
//   * textureStore to rgba16float: WGSL/Vulkan leave the f32->f16 rounding to the device (RTE or
//     RTZ).  The reference's golden renders pin it: replaying lupin_tests' 101-frame protocol with
//     round-toward-zero stores reproduces their image means to 1e-4, round-to-nearest-even comes out
//     1.8 % brighter (truncation error accumulates in the running average).  Default = RTZ;
//     RTE selectable (store_rounding = 1).
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off -fopenmp).

#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/lupin_hip.h"
#include "../include/lupin_detmath.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Small vector algebra with WGSL semantics
// ---------------------------------------------------------------------------------------------

struct vec2f { float x, y; };
struct vec3f {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct vec4f { float x, y, z, w; };

inline vec3f v3(float a) { return {a, a, a}; }
inline vec3f v3(float x, float y, float z) { return {x, y, z}; }
inline vec3f operator+(vec3f a, vec3f b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3f operator-(vec3f a, vec3f b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3f operator*(vec3f a, vec3f b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3f operator/(vec3f a, vec3f b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline vec3f operator*(vec3f a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3f operator*(float s, vec3f a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3f operator/(vec3f a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3f operator/(float s, vec3f a) { return {s / a.x, s / a.y, s / a.z}; }
inline vec3f operator+(vec3f a, float s) { return {a.x + s, a.y + s, a.z + s}; }
inline vec3f operator+(float s, vec3f a) { return {s + a.x, s + a.y, s + a.z}; }
inline vec3f operator-(vec3f a, float s) { return {a.x - s, a.y - s, a.z - s}; }
inline vec3f operator-(float s, vec3f a) { return {s - a.x, s - a.y, s - a.z}; }
inline vec3f operator-(vec3f a) { return {-a.x, -a.y, -a.z}; }
inline vec3f &operator+=(vec3f &a, vec3f b) { a = a + b; return a; }
inline vec3f &operator*=(vec3f &a, vec3f b) { a = a * b; return a; }
inline vec3f &operator*=(vec3f &a, float s) { a = a * s; return a; }

inline vec2f operator+(vec2f a, vec2f b) { return {a.x + b.x, a.y + b.y}; }
inline vec2f operator*(vec2f a, float s) { return {a.x * s, a.y * s}; }
inline vec4f operator+(vec4f a, vec4f b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4f operator*(vec4f a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }

inline float fmin_(float a, float b) { return (b < a) ? b : a; }
inline float fmax_(float a, float b) { return (a < b) ? b : a; }
inline float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
inline vec3f min3(vec3f a, vec3f b) { return {fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)}; }
inline vec3f max3(vec3f a, vec3f b) { return {fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)}; }
inline vec3f clamp3(vec3f v, vec3f lo, vec3f hi) { return min3(max3(v, lo), hi); }
inline float dot(vec3f a, vec3f b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3f cross(vec3f a, vec3f b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(vec3f a) { return sqrtf(dot(a, a)); }
inline vec3f normalize(vec3f a) { vec3f r; lpm_normalize3f(a.x, a.y, a.z, &r.x, &r.y, &r.z); return r; }   // v * (1 / |v|): lupin_detmath.h
inline vec3f sqrt3(vec3f a) { return {sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)}; }
inline vec3f exp3(vec3f a) { return {lpm_expf(a.x), lpm_expf(a.y), lpm_expf(a.z)}; }
inline vec3f log3(vec3f a) { return {lpm_logf(a.x), lpm_logf(a.y), lpm_logf(a.z)}; }
inline vec3f mix3(vec3f a, vec3f b, float t) { return a * (1.0f - t) + b * t; }
inline vec3f mix3(vec3f a, vec3f b, vec3f t) { return a * (1.0f - t) + b * t; }
inline bool all_eq0(vec3f a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
inline bool all_ne0(vec3f a) { return a.x != 0.0f && a.y != 0.0f && a.z != 0.0f; }

// WGSL u32(f32) / i32(f32) saturate; C's conversion is undefined out of range.
inline uint32_t f2u(float x)
{
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
inline int32_t f2i(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (int32_t)0x80000000;
    return (int32_t)x;
}
inline uint32_t bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }

struct mat3x3f { vec3f c[3]; };  // columns
inline vec3f operator*(const mat3x3f &m, vec3f v) { return m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z; }

// pathtracer.wgsl:2644-2646
const float F32_MAX = 3.40282346638528859812e+38f;
const float PI = 3.14159265358979323846264338327950288f;
const uint32_t SENTINEL_IDX = 0xFFFFFFFFu;  // :66

// f16 <-> f32 (texture storage formats)
inline float half_to_float(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1F;
    uint32_t man = h & 0x3FFu;
    uint32_t u;
    if (exp == 0) {
        if (man == 0) { u = sign; }
        else {
            int e = -1;
            do { man <<= 1; e++; } while ((man & 0x400u) == 0);
            man &= 0x3FFu;
            u = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        u = sign | 0x7F800000u | (man << 13);
    } else {
        u = sign | ((exp + 112) << 23) | (man << 13);
    }
    float f; memcpy(&f, &u, 4); return f;
}
inline uint16_t float_to_half_rne(float f)
{
    uint32_t u = bits(f);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t a = u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) {  // inf / nan
        return (uint16_t)(sign | 0x7C00u | ((a > 0x7F800000u) ? 0x200u : 0u));
    }
    if (a >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);  // rounds to >= 65520 -> inf
    if (a < 0x33000001u) return (uint16_t)sign;                 // <= 2^-25 -> 0 (tie to even)
    int exp = (int)(a >> 23) - 127;
    uint32_t man = (a & 0x7FFFFFu) | 0x800000u;
    int shift;
    uint32_t hexp;
    if (exp < -14) { shift = 13 + (-14 - exp); hexp = 0; }
    else { shift = 13; hexp = (uint32_t)(exp + 15); }
    uint32_t q = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    // q includes the implicit bit for normals (0x400); adding hexp<<10 handles mantissa carry.
    uint32_t h = (hexp == 0) ? q : (((hexp - 1) << 10) + q);
    return (uint16_t)(sign | h);
}

inline uint16_t float_to_half_rtz(float f)
{
    uint16_t h = float_to_half_rne(f);
    if ((h & 0x7FFFu) > 0x7C00u) return h;   // NaN
    if (fabsf(half_to_float(h)) > fabsf(f)) h = (uint16_t)(h - 1);   // rounded away from zero (incl. to inf): step back
    return h;
}

// ---------------------------------------------------------------------------------------------
// Per-invocation state (WGSL `var<private>`), scene bindings and counters
// ---------------------------------------------------------------------------------------------

struct Ray { vec3f ori, dir, inv_dir; };  // pathtracer.wgsl:2897-2902

struct HitInfo  // :2953-2961
{
    bool hit = false;
    float dst = 0.0f;
    vec2f uv = {0.0f, 0.0f};
    uint32_t instance_idx = 0;
    uint32_t tri_idx = 0;
    bool hit_backside = false;
};

struct MaterialPoint  // :1247-1260
{
    uint32_t mat_type = 0;
    vec3f emission = {0, 0, 0};
    vec3f color = {0, 0, 0};
    float opacity = 0;
    float roughness = 0;
    float metallic = 0;
    float ior = 0;
    vec3f density = {0, 0, 0};
    vec3f scattering = {0, 0, 0};
    float sc_anisotropy = 0;
    float tr_depth = 0;
};

// Work accounting.  Traversal work is split by who asked for it, because the HIP
// build runs the three kinds in different kernels:
//   ctx 0: the integrator's closest-hit query (ray_skip_alpha_stochastically)   -> extend stage
//   ctx 1: light-pdf marching (compute_instance_lights_pdf)                      -> shade stage
//   ctx 2: shadow / MIS closest-hit queries issued from inside the loop body    -> shade stage
struct Counters
{
    uint64_t path_bounces = 0, paths = 0;
    uint64_t tlas_aabb[3] = {0, 0, 0}, instances_entered[3] = {0, 0, 0}, blas_aabb[3] = {0, 0, 0}, tri_tests[3] = {0, 0, 0};
    uint64_t material_points = 0, tex_ldr = 0, tex_hdr = 0, light_mesh = 0, light_env = 0;
    uint64_t closest_hit_queries = 0, light_pdf_queries = 0, surface_hits = 0;
    uint64_t normal_fetches = 0, uv_fetches = 0, color_fetches = 0;   // 3-vertex attribute gathers
    uint64_t debug_num_bounces = 0;   // DEBUG_NUM_BOUNCES (pathtracer.wgsl:583,606-608): surface hits of pathtrace_standard
};

const int MAX_VOLUMES = 10;                 // :582
const uint32_t MAX_OPACITY_BOUNCES = 128;   // :1263
const float MIN_ROUGHNESS = 0.03f * 0.03f;  // :1262
const uint32_t TLAS_STACK = 51, BVH_STACK = 26;  // bvh_custom.wgsl:4,11,182,197

struct Inv
{
    const LupinSceneDesc *s;
    LupinPushConstants constants;
    uint32_t MAX_BOUNCES, SAMPLES_PER_PIXEL;
    uint32_t RNG_STATE = 0;
    Counters n;
    int wctx = 2;   // which traversal context is charged (see Counters)

    // ---- RNG (pathtracer.wgsl:1561-1629, :1675-1679) ----
    static uint32_t hash_u32(uint32_t seed)  // :1569-1580
    {
        uint32_t x = seed;
        x ^= x >> 17; x *= 0xed5ad4bbu;
        x ^= x >> 11; x *= 0xac4c1b51u;
        x ^= x >> 15; x *= 0x31848babu;
        x ^= x >> 14;
        return x;
    }
    void init_rng(uint32_t global_id)  // :1563-1567 (seed is the literal 0, constants.rng_seed unused)
    {
        const uint32_t seed = 0u;
        RNG_STATE = hash_u32((global_id * 19349663u) ^ (constants.accum_counter * 83492791u) ^ (seed * 73856093u));
    }
    float random_f32()  // :1594-1600
    {
        RNG_STATE = RNG_STATE * 747796405u + 2891336453u;
        uint32_t result = ((RNG_STATE >> ((RNG_STATE >> 28u) + 4u)) ^ RNG_STATE) * 277803737u;
        result = (result >> 22u) ^ result;
        return (float)result / 4294967295.0f;
    }
    uint32_t random_u32_range_unsafe(uint32_t max_exclusive)  // :1603-1606
    {
        uint32_t v = f2u(random_f32() * (float)max_exclusive);
        uint32_t m = max_exclusive - 1u;
        return v < m ? v : m;
    }
    vec2f random_vec2f()  // :1608-1614
    {
        float rnd0 = random_f32();
        float rnd1 = random_f32();
        return {rnd0, rnd1};
    }
    vec2f random_in_disk()  // :1623-1629
    {
        vec2f rnd = random_vec2f();
        float r = sqrtf(rnd.y);
        float phi = 2.0f * PI * rnd.x;
        return {lpm_cosf(phi) * r, lpm_sinf(phi) * r};
    }
    vec2f random_tri_uv()  // :1675-1679
    {
        vec2f rnd = random_vec2f();
        return {1.0f - sqrtf(rnd.x), rnd.y * sqrtf(rnd.x)};
    }

    // ---- buffer accessors ----
    vec3f vert_pos(uint32_t mesh, uint32_t i) const
    {
        const float *p = s->meshes[mesh].verts_pos + (size_t)i * 4;
        return {p[0], p[1], p[2]};
    }
    uint32_t index(uint32_t mesh, uint32_t i) const { return s->meshes[mesh].indices[i]; }
    vec3f vert_normal(uint32_t buf, uint32_t i) const
    {
        const float *p = s->verts_normal_array[buf].data + (size_t)i * 4;
        return {p[0], p[1], p[2]};
    }
    vec2f vert_uv(uint32_t buf, uint32_t i) const
    {
        const float *p = s->verts_texcoord_array[buf].data + (size_t)i * 2;
        return {p[0], p[1]};
    }
    vec4f vert_color(uint32_t buf, uint32_t i) const
    {
        const float *p = s->verts_color_array[buf].data + (size_t)i * 4;
        return {p[0], p[1], p[2], p[3]};
    }
    uint32_t num_lights() const { return (constants.flags & LUPIN_FLAG_LIGHTS_EMPTY) ? 0u : s->num_lights; }
    uint32_t num_envs() const { return (constants.flags & LUPIN_FLAG_ENVS_EMPTY) ? 0u : s->num_environments; }
    // instance.transpose_inverse_transform[i] as vec4 (column i of the WGSL mat3x4f)
    static vec4f tit(const LupinInstance &in, int i)
    {
        const float *c = in.transpose_inverse_transform.m[i];
        return {c[0], c[1], c[2], c[3]};
    }

    // ---- intersection primitives (pathtracer.wgsl:2906-2943) ----
    static float ray_aabb_dst(const Ray &ray, vec3f aabb_min, vec3f aabb_max)  // :2906-2917
    {
        vec3f t_min = (aabb_min - ray.ori) * ray.inv_dir;
        vec3f t_max = (aabb_max - ray.ori) * ray.inv_dir;
        // min/max of the slab test are IEEE minNum/maxNum (fminf/fmaxf): a NaN operand (0 * inf when a ray
        // with a zero direction component starts on a box plane) yields the other operand
        vec3f t1 = {fminf(t_min.x, t_max.x), fminf(t_min.y, t_max.y), fminf(t_min.z, t_max.z)};
        vec3f t2 = {fmaxf(t_min.x, t_max.x), fmaxf(t_min.y, t_max.y), fmaxf(t_min.z, t_max.z)};
        float dst_far = fminf(fminf(t2.x, t2.y), t2.z);
        float dst_near = fmaxf(fmaxf(t1.x, t1.y), t1.z);
        bool did_hit = dst_far >= dst_near && dst_far > 0.0f;
        return did_hit ? dst_near : F32_MAX;
    }
    vec4f ray_tri_dst(const Ray &ray, vec3f v0, vec3f v1, vec3f v2) const  // :2922-2943
    {
        vec3f v1v0 = v1 - v0;
        vec3f v2v0 = v2 - v0;
        vec3f rov0 = ray.ori - v0;
        vec3f nn = cross(v1v0, v2v0);
        vec3f q = cross(rov0, ray.dir);
        float det = dot(ray.dir, nn);
        float d = 1.0f / det;
        float u = d * dot(-q, v2v0);
        float v = d * dot(q, v1v0);
        float t = d * dot(-nn, rov0);
        if (fmin_(u, v) < 0.0f || (u + v) > 1.0f || t < constants.ray_epsilon) { t = F32_MAX; }
        return {t, u, v, det};
    }

    // ---- bvh_custom.wgsl:195-288 ----
    struct RayMeshIntersectionResult { vec4f hit; uint32_t tri_idx; };
    RayMeshIntersectionResult _ray_mesh_intersection(const Ray &ray, float cur_min_hit_dst, uint32_t mesh_idx)
    {
        uint32_t bvh_stack[BVH_STACK];
        uint32_t stack_idx = 2u;
        bvh_stack[0] = 0u;
        bvh_stack[1] = 0u;
        const LupinBvhNode *nodes = s->meshes[mesh_idx].bvh_nodes;

        vec4f min_hit = {cur_min_hit_dst, 0.0f, 0.0f, 0.0f};
        uint32_t tri_idx = 0u;
        while (stack_idx > 1)
        {
            stack_idx--;
            const LupinBvhNode &node = nodes[bvh_stack[stack_idx]];
            if (node.tri_count > 0u)  // leaf
            {
                uint32_t tri_begin = node.tri_begin_or_first_child;
                uint32_t tri_count = node.tri_count;
                for (uint32_t i = tri_begin; i < tri_begin + tri_count; i++)
                {
                    vec3f v0 = vert_pos(mesh_idx, index(mesh_idx, i * 3 + 0));
                    vec3f v1 = vert_pos(mesh_idx, index(mesh_idx, i * 3 + 1));
                    vec3f v2 = vert_pos(mesh_idx, index(mesh_idx, i * 3 + 2));
                    vec4f hit = ray_tri_dst(ray, v0, v1, v2);
                    if (hit.x < min_hit.x) { min_hit = hit; tri_idx = i; }
                    n.tri_tests[wctx]++;
                }
            }
            else
            {
                uint32_t left_child = node.tri_begin_or_first_child;
                uint32_t right_child = left_child + 1;
                const LupinBvhNode &l = nodes[left_child];
                const LupinBvhNode &r = nodes[right_child];
                float left_dst = ray_aabb_dst(ray, v3(l.aabb_min[0], l.aabb_min[1], l.aabb_min[2]), v3(l.aabb_max[0], l.aabb_max[1], l.aabb_max[2]));
                float right_dst = ray_aabb_dst(ray, v3(r.aabb_min[0], r.aabb_min[1], r.aabb_min[2]), v3(r.aabb_max[0], r.aabb_max[1], r.aabb_max[2]));
                n.blas_aabb[wctx] += 2;

                bool visit_left_first = left_dst <= right_dst;
                bool push_left = left_dst < min_hit.x;
                bool push_right = right_dst < min_hit.x;
                if (visit_left_first)
                {
                    if (push_right) { bvh_stack[stack_idx] = right_child; stack_idx++; }
                    if (push_left) { bvh_stack[stack_idx] = left_child; stack_idx++; }
                }
                else
                {
                    if (push_left) { bvh_stack[stack_idx] = left_child; stack_idx++; }
                    if (push_right) { bvh_stack[stack_idx] = right_child; stack_idx++; }
                }
            }
        }
        return {min_hit, tri_idx};
    }

    // bvh_custom.wgsl:7-110
    HitInfo ray_scene_intersection(const Ray &ray)
    {
        if ((constants.flags & LUPIN_FLAG_INSTANCES_EMPTY) != 0) { return HitInfo(); }
        n.closest_hit_queries++;

        uint32_t tlas_stack[TLAS_STACK];
        uint32_t stack_idx = 2;
        tlas_stack[0] = 0u;
        tlas_stack[1] = 0u;

        vec4f min_hit = {F32_MAX, 0.0f, 0.0f, 0.0f};
        uint32_t tri_idx = 0;
        uint32_t instance_idx = 0;
        while (stack_idx > 1)
        {
            stack_idx--;
            const LupinTlasNode &node = s->tlas_nodes[tlas_stack[stack_idx]];
            if (node.left == 0u)  // leaf
            {
                const LupinInstance &instance = s->instances[node.instance_idx];
                n.instances_entered[wctx]++;
                Ray ray_trans = ray;
                vec4f c0 = tit(instance, 0), c1 = tit(instance, 1), c2 = tit(instance, 2);
                // vec4f(ori, 1) * mat3x4 : dot with each column (:32)
                vec3f o = ray.ori, d = ray.dir;
                ray_trans.ori = {o.x * c0.x + o.y * c0.y + o.z * c0.z + 1.0f * c0.w,
                                 o.x * c1.x + o.y * c1.y + o.z * c1.z + 1.0f * c1.w,
                                 o.x * c2.x + o.y * c2.y + o.z * c2.z + 1.0f * c2.w};
                ray_trans.dir = {d.x * c0.x + d.y * c0.y + d.z * c0.z + 0.0f * c0.w,
                                 d.x * c1.x + d.y * c1.y + d.z * c1.z + 0.0f * c1.w,
                                 d.x * c2.x + d.y * c2.y + d.z * c2.z + 0.0f * c2.w};
                ray_trans.inv_dir = 1.0f / ray_trans.dir;

                RayMeshIntersectionResult result = _ray_mesh_intersection(ray_trans, min_hit.x, instance.mesh_idx);
                if (result.hit.x < min_hit.x)
                {
                    min_hit = result.hit;
                    tri_idx = result.tri_idx;
                    instance_idx = node.instance_idx;
                }
            }
            else
            {
                const LupinTlasNode &l = s->tlas_nodes[node.left];
                const LupinTlasNode &r = s->tlas_nodes[node.right];
                float left_dst = ray_aabb_dst(ray, v3(l.aabb_min[0], l.aabb_min[1], l.aabb_min[2]), v3(l.aabb_max[0], l.aabb_max[1], l.aabb_max[2]));
                float right_dst = ray_aabb_dst(ray, v3(r.aabb_min[0], r.aabb_min[1], r.aabb_min[2]), v3(r.aabb_max[0], r.aabb_max[1], r.aabb_max[2]));
                n.tlas_aabb[wctx] += 2;

                bool visit_left_first = left_dst <= right_dst;
                bool push_left = left_dst < min_hit.x;
                bool push_right = right_dst < min_hit.x;
                if (visit_left_first)
                {
                    if (push_right) { tlas_stack[stack_idx] = node.right; stack_idx++; }
                    if (push_left) { tlas_stack[stack_idx] = node.left; stack_idx++; }
                }
                else
                {
                    if (push_left) { tlas_stack[stack_idx] = node.left; stack_idx++; }
                    if (push_right) { tlas_stack[stack_idx] = node.right; stack_idx++; }
                }
            }
        }

        HitInfo hit_info;
        if (min_hit.x != F32_MAX)
        {
            hit_info.hit = true;
            hit_info.dst = min_hit.x;
            hit_info.uv = {min_hit.y, min_hit.z};
            hit_info.instance_idx = instance_idx;
            hit_info.tri_idx = tri_idx;
            hit_info.hit_backside = min_hit.w > 0.0f;
        }
        return hit_info;
    }

    // pathtracer.wgsl:2648-2680 transform helpers on a mat4 given as 4 columns
    struct mat4 { vec4f c[4]; };
    static vec3f mat4_mul_xyz(const mat4 &m, vec4f v)
    {
        return {m.c[0].x * v.x + m.c[1].x * v.y + m.c[2].x * v.z + m.c[3].x * v.w,
                m.c[0].y * v.x + m.c[1].y * v.y + m.c[2].y * v.z + m.c[3].y * v.w,
                m.c[0].z * v.x + m.c[1].z * v.y + m.c[2].z * v.z + m.c[3].z * v.w};
    }
    static vec3f transform_point(vec3f p, const mat4 &t) { return mat4_mul_xyz(t, {p.x, p.y, p.z, 1.0f}); }  // :2648
    static vec3f transform_dir(vec3f d, const mat4 &t) { return normalize(mat4_mul_xyz(t, {d.x, d.y, d.z, 0.0f})); }  // :2656
    static Ray transform_ray(const Ray &ray, const mat4 &t)  // :2662-2669
    {
        Ray res = ray;
        res.ori = transform_point(res.ori, t);
        res.dir = transform_dir(res.dir, t);
        res.inv_dir = 1.0f / res.dir;
        return res;
    }
    static Ray transform_ray_without_normalizing_direction(const Ray &ray, const mat4 &t)  // :2671-2680
    {
        Ray res = ray;
        res.ori = transform_point(res.ori, t);
        res.dir = mat4_mul_xyz(t, {res.dir.x, res.dir.y, res.dir.z, 0.0f});
        res.inv_dir = 1.0f / res.dir;
        return res;
    }
    // transpose(instance.transpose_inverse_transform) widened to mat4 (bvh_custom.wgsl:294-295, pathtracer.wgsl:2566-2567)
    static mat4 instance_inv_mat4(const LupinInstance &in)
    {
        vec4f r0 = tit(in, 0), r1 = tit(in, 1), r2 = tit(in, 2);
        mat4 m;
        m.c[0] = {r0.x, r1.x, r2.x, 0.0f};
        m.c[1] = {r0.y, r1.y, r2.y, 0.0f};
        m.c[2] = {r0.z, r1.z, r2.z, 0.0f};
        m.c[3] = {r0.w, r1.w, r2.w, 1.0f};
        return m;
    }

    // bvh_custom.wgsl:290-300
    RayMeshIntersectionResult _ray_instance_intersection(const Ray &ray, float cur_min_hit_dst, uint32_t instance_idx)
    {
        const LupinInstance &instance = s->instances[instance_idx];
        mat4 trans_mat4 = instance_inv_mat4(instance);
        Ray ray_trans = transform_ray_without_normalizing_direction(ray, trans_mat4);
        return _ray_mesh_intersection(ray_trans, cur_min_hit_dst, instance.mesh_idx);
    }

    // pathtracer.wgsl:2561-2576
    vec3f compute_tri_geom_normal(uint32_t instance_idx, uint32_t tri_idx) const
    {
        const LupinInstance &instance = s->instances[instance_idx];
        uint32_t mesh_idx = instance.mesh_idx;
        vec3f v0 = vert_pos(mesh_idx, index(mesh_idx, tri_idx * 3 + 0));
        vec3f v1 = vert_pos(mesh_idx, index(mesh_idx, tri_idx * 3 + 1));
        vec3f v2 = vert_pos(mesh_idx, index(mesh_idx, tri_idx * 3 + 2));
        vec3f local_normal = normalize(cross(v2 - v0, v1 - v0));
        // transpose(mat3x3(inv_trans[0].xyz, ...)) has columns tit[i].xyz
        vec4f r0 = tit(instance, 0), r1 = tit(instance, 1), r2 = tit(instance, 2);
        mat3x3f normal_mat = {{{r0.x, r0.y, r0.z}, {r1.x, r1.y, r1.z}, {r2.x, r2.y, r2.z}}};
        return normalize(normal_mat * local_normal);
    }

    // bvh_custom.wgsl:112-152
    float compute_instance_lights_pdf(const Ray &ray_in)
    {
        uint32_t nl = num_lights();
        vec3f pos = ray_in.ori;
        vec3f incoming = ray_in.dir;
        float pdf = 0.0f;
        for (uint32_t i = 0u; i < nl; i++)
        {
            LupinLight light = s->lights[i];
            uint32_t instance_idx = light.instance_idx;
            float light_pdf = 0.0f;
            vec3f next_pos = pos;
            for (uint32_t bounce = 0u; bounce < 100; bounce++)
            {
                Ray ray = {next_pos, incoming, 1.0f / incoming};
                n.light_pdf_queries++;
                const int saved_ctx = wctx;
                wctx = 1;
                n.instances_entered[1]++;
                RayMeshIntersectionResult hit_info = _ray_instance_intersection(ray, F32_MAX, instance_idx);
                wctx = saved_ctx;
                float hit_dst = hit_info.hit.x;
                if (hit_dst == F32_MAX) { break; }
                vec3f light_normal = compute_tri_geom_normal(instance_idx, hit_info.tri_idx);
                vec3f light_pos = ray.ori + ray.dir * hit_dst;
                float dist2 = dot(light_pos - pos, light_pos - pos);
                float cos_theta = fabsf(dot(light_normal, incoming));
                light_pdf += dist2 / (cos_theta * light.area);
                next_pos = light_pos + incoming;
            }
            pdf += light_pdf;
        }
        return pdf;
    }

    // ---- textures (pathtracer.wgsl:1413-1416 + wgpu_utils.rs:244-256 sampler) ----
    vec4f texel(const LupinTextureDesc &t, int x, int y)
    {
        size_t i = (size_t)y * t.width + (size_t)x;
        if (t.format == LUPIN_TEX_RGBA8_UNORM)
        {
            const uint8_t *p = (const uint8_t *)t.pixels + i * 4;
            return {(float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, (float)p[3] / 255.0f};
        }
        const uint16_t *p = (const uint16_t *)t.pixels + i * 4;
        return {half_to_float(p[0]), half_to_float(p[1]), half_to_float(p[2]), half_to_float(p[3])};
    }
    vec4f sample_texture(uint32_t tex_idx, vec2f uv)
    {
        const LupinTextureDesc &t = s->textures[tex_idx];
        if (t.format == LUPIN_TEX_RGBA8_UNORM) n.tex_ldr++; else n.tex_hdr++;
        int w = (int)t.width, h = (int)t.height;
        float x = uv.x * (float)w - 0.5f;
        float y = uv.y * (float)h - 0.5f;
        float x0f = floorf(x), y0f = floorf(y);
        float fx = x - x0f, fy = y - y0f;
        int x0 = f2i(x0f), y0 = f2i(y0f);
        int xa = ((x0 % w) + w) % w, ya = ((y0 % h) + h) % h;
        int xb = (xa + 1) % w, yb = (ya + 1) % h;
        vec4f t00 = texel(t, xa, ya), t10 = texel(t, xb, ya), t01 = texel(t, xa, yb), t11 = texel(t, xb, yb);
        vec4f top = t00 * (1.0f - fx) + t10 * fx;
        vec4f bot = t01 * (1.0f - fx) + t11 * fx;
        return top * (1.0f - fy) + bot * fy;
    }

    static vec3f vec3f_srgb_to_linear(vec3f srgb)  // :2729-2736
    {
        vec3f cutoff = {srgb.x < 0.04045f ? 1.0f : 0.0f, srgb.y < 0.04045f ? 1.0f : 0.0f, srgb.z < 0.04045f ? 1.0f : 0.0f};
        vec3f b = (srgb + v3(0.055f)) / v3(1.055f);
        vec3f higher = {lpm_powf(b.x, 2.4f), lpm_powf(b.y, 2.4f), lpm_powf(b.z, 2.4f)};
        vec3f lower = srgb / v3(12.92f);
        return mix3(higher, lower, cutoff);
    }

    // :1757-1770
    vec4f get_vert_color(uint32_t instance_idx, uint32_t tri_idx, vec2f uv) const
    {
        uint32_t mesh_idx = s->instances[instance_idx].mesh_idx;
        const LupinMeshInfo &mesh_info = s->mesh_infos[mesh_idx];
        if (mesh_info.colors_buf_idx == SENTINEL_IDX) { return {1.0f, 1.0f, 1.0f, 1.0f}; }
        const_cast<Inv *>(this)->n.color_fetches++;
        vec4f c0 = vert_color(mesh_info.colors_buf_idx, index(mesh_idx, tri_idx * 3 + 0));
        vec4f c1 = vert_color(mesh_info.colors_buf_idx, index(mesh_idx, tri_idx * 3 + 1));
        vec4f c2 = vert_color(mesh_info.colors_buf_idx, index(mesh_idx, tri_idx * 3 + 2));
        float w = 1.0f - uv.x - uv.y;
        return c0 * w + c1 * uv.x + c2 * uv.y;
    }

    // :1265-1342
    MaterialPoint get_material_point(const HitInfo &hit)
    {
        n.material_points++;
        uint32_t instance_idx = hit.instance_idx;
        const LupinInstance &instance = s->instances[instance_idx];
        uint32_t mesh_idx = instance.mesh_idx;
        const LupinMaterial &mat = s->materials[instance.mat_idx];
        const LupinMeshInfo &mesh_info = s->mesh_infos[mesh_idx];
        uint32_t tri_idx = hit.tri_idx;

        MaterialPoint res;
        res.mat_type = mat.mat_type;

        vec4f color_sample = {1.0f, 1.0f, 1.0f, 1.0f};
        vec3f emission_sample = v3(1.0f);
        float roughness_sample = 1.0f;
        float metallic_sample = 1.0f;
        vec3f scattering_sample = v3(1.0f);
        if (mesh_info.texcoords_buf_idx != SENTINEL_IDX)
        {
            n.uv_fetches++;
            vec2f uv0 = vert_uv(mesh_info.texcoords_buf_idx, index(mesh_idx, tri_idx * 3 + 0));
            vec2f uv1 = vert_uv(mesh_info.texcoords_buf_idx, index(mesh_idx, tri_idx * 3 + 1));
            vec2f uv2 = vert_uv(mesh_info.texcoords_buf_idx, index(mesh_idx, tri_idx * 3 + 2));
            float w = 1.0f - hit.uv.x - hit.uv.y;
            vec2f texcoords = uv0 * w + uv1 * hit.uv.x + uv2 * hit.uv.y;

            if (mat.color_tex_idx != SENTINEL_IDX)
            {
                color_sample = sample_texture(mat.color_tex_idx, texcoords);
                vec3f lin = vec3f_srgb_to_linear({color_sample.x, color_sample.y, color_sample.z});
                color_sample = {lin.x, lin.y, lin.z, color_sample.w};
            }
            if (mat.emission_tex_idx != SENTINEL_IDX)
            {
                vec4f t = sample_texture(mat.emission_tex_idx, texcoords);
                emission_sample = {t.x, t.y, t.z};
            }
            if (mat.roughness_tex_idx != SENTINEL_IDX)
            {
                vec4f t = sample_texture(mat.roughness_tex_idx, texcoords);
                roughness_sample = t.y;
                metallic_sample = t.z;
            }
            if (mat.scattering_tex_idx != SENTINEL_IDX)
            {
                vec4f t = sample_texture(mat.scattering_tex_idx, texcoords);
                scattering_sample = {t.x, t.y, t.z};
            }
        }

        vec4f vert_color = get_vert_color(instance_idx, tri_idx, hit.uv);

        res.color = v3(color_sample.x, color_sample.y, color_sample.z) * v3(mat.color[0], mat.color[1], mat.color[2]) * v3(vert_color.x, vert_color.y, vert_color.z);
        res.opacity = color_sample.w * mat.color[3] * vert_color.w;
        res.emission = emission_sample * v3(mat.emission[0], mat.emission[1], mat.emission[2]);
        res.roughness = roughness_sample * mat.roughness;
        res.roughness *= res.roughness;
        res.density = v3(0.0f);
        if (mat.mat_type == LUPIN_MAT_REFRACTIVE || mat.mat_type == LUPIN_MAT_VOLUMETRIC || mat.mat_type == LUPIN_MAT_SUBSURFACE)
        {
            res.density = -log3(clamp3(res.color, v3(0.0001f), v3(1.0f))) / mat.tr_depth;
        }
        res.ior = mat.ior;
        res.scattering = scattering_sample * v3(mat.scattering[0], mat.scattering[1], mat.scattering[2]);
        res.sc_anisotropy = mat.sc_anisotropy;
        res.tr_depth = mat.tr_depth;
        res.metallic = metallic_sample * mat.metallic;

        if (res.mat_type == LUPIN_MAT_MATTE || res.mat_type == LUPIN_MAT_GLTFPBR || res.mat_type == LUPIN_MAT_GLOSSY) {
            res.roughness = clamp_(res.roughness, MIN_ROUGHNESS, 1.0f);
        } else if (res.mat_type == LUPIN_MAT_VOLUMETRIC) {
            res.roughness = 0.0f;
        } else {
            if (res.roughness < MIN_ROUGHNESS) { res.roughness = 0.0f; }
        }
        return res;
    }

    // bvh_custom.wgsl:154-180
    HitInfo ray_skip_alpha_stochastically(const Ray &start_ray)
    {
        if ((constants.flags & LUPIN_FLAG_INSTANCES_EMPTY) != 0) { return HitInfo(); }
        HitInfo hit;
        Ray ray = start_ray;
        float dst = 0.0f;
        for (uint32_t opacity_bounce = 0u; opacity_bounce < MAX_OPACITY_BOUNCES; opacity_bounce++)
        {
            wctx = 0;
            hit = ray_scene_intersection(ray);
            wctx = 2;
            if (!hit.hit) { break; }
            dst += hit.dst;
            MaterialPoint mat_point = get_material_point(hit);
            if (mat_point.opacity < 1.0f && random_f32() >= mat_point.opacity) {
                ray.ori = ray.ori + ray.dir * hit.dst;
            } else {
                break;
            }
        }
        hit.dst = dst;
        return hit;
    }

    // :1699-1727
    struct Tangents { vec3f tangent, bitangent; };
    Tangents compute_tangents_from_uv(uint32_t instance_idx, vec3f p0, vec3f p1, vec3f p2, vec2f uv0, vec2f uv1, vec2f uv2) const
    {
        vec3f p = p1 - p0;
        vec3f q = p2 - p0;
        vec2f sv = {uv1.x - uv0.x, uv2.x - uv0.x};
        vec2f tv = {uv1.y - uv0.y, uv2.y - uv0.y};
        float div = sv.x * tv.y - sv.y * tv.x;
        vec3f tangent_local = {1.0f, 0.0f, 0.0f};
        vec3f bitangent_local = {0.0f, 1.0f, 0.0f};
        if (div != 0)
        {
            tangent_local = v3(tv.y * p.x - tv.x * q.x, tv.y * p.y - tv.x * q.y, tv.y * p.z - tv.x * q.z) / div;
            bitangent_local = v3(sv.x * q.x - sv.y * p.x, sv.x * q.y - sv.y * p.y, sv.x * q.z - sv.y * p.z) / div;
        }
        const LupinInstance &instance = s->instances[instance_idx];
        vec4f r0 = tit(instance, 0), r1 = tit(instance, 1), r2 = tit(instance, 2);
        mat3x3f normal_mat = {{{r0.x, r0.y, r0.z}, {r1.x, r1.y, r1.z}, {r2.x, r2.y, r2.z}}};
        return {normalize(normal_mat * tangent_local), normalize(normal_mat * bitangent_local)};
    }

    // :1730-1755
    vec3f get_vert_normal(uint32_t instance_idx, uint32_t tri_idx, vec2f uv, bool /*hit_backside*/) const
    {
        uint32_t mesh_idx = s->instances[instance_idx].mesh_idx;
        const LupinMeshInfo &mesh_info = s->mesh_infos[mesh_idx];
        vec3f normal;
        if (mesh_info.normals_buf_idx == SENTINEL_IDX)
        {
            normal = compute_tri_geom_normal(instance_idx, tri_idx);
        }
        else
        {
            const_cast<Inv *>(this)->n.normal_fetches++;
            vec3f n0 = vert_normal(mesh_info.normals_buf_idx, index(mesh_idx, tri_idx * 3 + 0));
            vec3f n1 = vert_normal(mesh_info.normals_buf_idx, index(mesh_idx, tri_idx * 3 + 1));
            vec3f n2 = vert_normal(mesh_info.normals_buf_idx, index(mesh_idx, tri_idx * 3 + 2));
            float w = 1.0f - uv.x - uv.y;
            vec3f normal_local = normalize(n0 * w + n1 * uv.x + n2 * uv.y);
            const LupinInstance &instance = s->instances[instance_idx];
            vec4f r0 = tit(instance, 0), r1 = tit(instance, 1), r2 = tit(instance, 2);
            mat3x3f normal_mat = {{{r0.x, r0.y, r0.z}, {r1.x, r1.y, r1.z}, {r2.x, r2.y, r2.z}}};
            normal = normalize(normal_mat * normal_local);
        }
        return normal;
    }

    static vec3f orthonormalize(vec3f a, vec3f b) { return normalize(a - b * dot(a, b)); }  // :2774-2777

    // :1344-1384
    vec3f compute_shading_normal(const HitInfo &hit)
    {
        n.surface_hits++;
        const LupinInstance &instance = s->instances[hit.instance_idx];
        uint32_t mesh_idx = instance.mesh_idx;
        const LupinMeshInfo &mesh_info = s->mesh_infos[mesh_idx];
        const LupinMaterial &mat = s->materials[instance.mat_idx];
        vec2f uv = hit.uv;
        float w = 1.0f - hit.uv.x - hit.uv.y;
        uint32_t tri_idx = hit.tri_idx;

        vec3f res = get_vert_normal(hit.instance_idx, hit.tri_idx, hit.uv, hit.hit_backside);
        if (mesh_info.texcoords_buf_idx != SENTINEL_IDX)
        {
            if (mat.normal_tex_idx != SENTINEL_IDX)
            {
                vec2f uv0 = vert_uv(mesh_info.texcoords_buf_idx, index(mesh_idx, tri_idx * 3 + 0));
                vec2f uv1 = vert_uv(mesh_info.texcoords_buf_idx, index(mesh_idx, tri_idx * 3 + 1));
                vec2f uv2 = vert_uv(mesh_info.texcoords_buf_idx, index(mesh_idx, tri_idx * 3 + 2));
                vec2f texcoords = uv0 * w + uv1 * uv.x + uv2 * uv.y;
                vec3f p0 = vert_pos(mesh_idx, index(mesh_idx, tri_idx * 3 + 0));
                vec3f p1 = vert_pos(mesh_idx, index(mesh_idx, tri_idx * 3 + 1));
                vec3f p2 = vert_pos(mesh_idx, index(mesh_idx, tri_idx * 3 + 2));
                Tangents tangents = compute_tangents_from_uv(hit.instance_idx, p0, p1, p2, uv0, uv1, uv2);

                vec4f ns = sample_texture(mat.normal_tex_idx, texcoords);
                vec3f normal_local = -1.0f + 2.0f * v3(ns.x, ns.y, ns.z);
                mat3x3f frame = {{tangents.tangent, tangents.bitangent, res}};
                frame.c[0] = orthonormalize(frame.c[0], frame.c[2]);
                frame.c[1] = normalize(cross(frame.c[2], frame.c[0]));
                bool should_flip_v = dot(frame.c[1], tangents.bitangent) < 0.0f;
                if (should_flip_v) { normal_local *= -1.0f; }
                res = normalize(frame * normal_local);
            }
        }
        return res;
    }

    // ---- environments (:1386-1410, :2551-2605) ----
    static vec3f transform_direction_inverse(const mat3x3f &a, vec3f b)  // :2779-2787
    {
        return normalize(v3(dot(a.c[0], b), dot(a.c[1], b), dot(a.c[2], b)));
    }
    vec2f dir_to_env_uv(vec3f dir, uint32_t env_idx) const  // :2579-2587
    {
        const LupinEnvironment &env = s->environments[env_idx];
        const float (*m)[4] = env.transform.m;
        mat3x3f a = {{{m[0][0], m[0][1], m[0][2]}, {m[1][0], m[1][1], m[1][2]}, {m[2][0], m[2][1], m[2][2]}}};
        vec3f trans_dir = transform_direction_inverse(a, dir);
        vec2f uv = {lpm_atan2f(trans_dir.z, trans_dir.x) / (2.0f * PI), lpm_acosf(clamp_(trans_dir.y, -1.0f, 1.0f)) / PI};
        if (uv.x < 0.0f) { uv.x += 1.0f; }
        if (uv.x > 1.0f) { uv.x -= 1.0f; }
        return uv;
    }
    vec3f sample_environment(vec3f dir, uint32_t env_idx)  // :1399-1410
    {
        const LupinEnvironment &env = s->environments[env_idx];
        vec2f uv = dir_to_env_uv(dir, env_idx);
        vec3f res = {env.emission[0], env.emission[1], env.emission[2]};
        if (env.emission_tex_idx != SENTINEL_IDX) {
            vec4f t = sample_texture(env.emission_tex_idx, uv);
            res *= v3(t.x, t.y, t.z);
        }
        return res;
    }
    vec3f sample_environments(vec3f dir)  // :1386-1397
    {
        if ((constants.flags & LUPIN_FLAG_ENVS_EMPTY) != 0) { return v3(0.0f); }
        vec3f emission = v3(0.0f);
        for (uint32_t i = 0u; i < s->num_environments; i++) { emission += sample_environment(dir, i); }
        return emission;
    }
    void env_tex_size(uint32_t env, uint32_t &w, uint32_t &h) const
    {
        const LupinTextureDesc &t = s->textures[s->environments[env].emission_tex_idx];
        w = t.width; h = t.height;
    }
    void dir_to_env_coords(vec3f dir, uint32_t env, uint32_t &cx, uint32_t &cy) const  // :2551-2559
    {
        uint32_t w, h; env_tex_size(env, w, h);
        vec2f uv = dir_to_env_uv(dir, env);
        uint32_t ux = f2u(uv.x * (float)w), uy = f2u(uv.y * (float)h);
        cx = ux < w - 1 ? ux : w - 1;
        cy = uy < h - 1 ? uy : h - 1;
    }
    vec3f env_uv_to_dir(vec2f uv, uint32_t env) const  // :2599-2605
    {
        vec3f dir = {lpm_cosf(uv.x * 2.0f * PI) * lpm_sinf(uv.y * PI),
                     lpm_cosf(uv.y * PI),
                     lpm_sinf(uv.x * 2.0f * PI) * lpm_sinf(uv.y * PI)};
        const float (*m)[4] = s->environments[env].transform.m;
        mat4 t;
        for (int c = 0; c < 4; c++) t.c[c] = {m[c][0], m[c][1], m[c][2], m[c][3]};
        return transform_dir(dir, t);
    }
    vec3f env_idx_to_dir(uint32_t idx, uint32_t env) const  // :2589-2597
    {
        uint32_t w, h; env_tex_size(env, w, h);
        uint32_t cx = idx % w, cy = idx / w;
        vec2f uv = {((float)cx + 0.5f) / (float)w, ((float)cy + 0.5f) / (float)h};
        return env_uv_to_dir(uv, env);
    }

    // ---- material predicates and Fresnel (:1418-1504) ----
    static bool is_mat_delta(const MaterialPoint &m)
    {
        return (m.mat_type == LUPIN_MAT_REFLECTIVE && m.roughness == 0.0f) ||
               (m.mat_type == LUPIN_MAT_REFRACTIVE && m.roughness == 0.0f) ||
               (m.mat_type == LUPIN_MAT_TRANSPARENT && m.roughness == 0.0f) ||
               (m.mat_type == LUPIN_MAT_VOLUMETRIC);
    }
    static bool is_mat_volumetric(const MaterialPoint &m)
    {
        return (m.mat_type == LUPIN_MAT_REFRACTIVE) || (m.mat_type == LUPIN_MAT_VOLUMETRIC) || (m.mat_type == LUPIN_MAT_SUBSURFACE);
    }
    static vec3f reflectivity_to_eta(vec3f reflectivity)  // :1433-1437
    {
        vec3f r = clamp3(reflectivity, v3(0.0f), v3(0.99f));
        return (1.0f + sqrt3(r)) / (1.0f - sqrt3(r));
    }
    static vec3f eta_to_reflectivity(vec3f eta) { return ((eta - 1.0f) * (eta - 1.0f)) / ((eta + 1.0f) * (eta + 1.0f)); }  // :1439
    static vec3f fresnel_schlick_vec3f(vec3f color, vec3f normal, vec3f out_dir)  // :1445-1451
    {
        if (all_eq0(color)) { return v3(0.0f); }
        float cosine = dot(normal, out_dir);
        return color + (1.0f - color) * lpm_powf(clamp_(1.0f - fabsf(cosine), 0.0f, 1.0f), 5.0f);
    }
    static float fresnel_dielectric(float eta, vec3f normal, vec3f outgoing)  // :1461-1479
    {
        float cosw = fabsf(dot(normal, outgoing));
        float sin2 = 1.0f - cosw * cosw;
        float eta2 = eta * eta;
        float cos2t = 1.0f - sin2 / eta2;
        if (cos2t < 0.0f) { return 1.0f; }
        float t0 = sqrtf(cos2t);
        float t1 = eta * t0;
        float t2 = eta * cosw;
        float rs = (cosw - t1) / (cosw + t1);
        float rp = (t0 - t2) / (t0 + t2);
        return (rs * rs + rp * rp) / 2.0f;
    }
    static vec3f fresnel_conductor(vec3f eta, vec3f etak, vec3f normal, vec3f outgoing)  // :1481-1504
    {
        float cosw = dot(normal, outgoing);
        if (cosw <= 0.0f) { return v3(0.0f); }
        cosw = clamp_(cosw, -1.0f, 1.0f);
        float cos2 = cosw * cosw;
        float sin2 = clamp_(1.0f - cos2, 0.0f, 1.0f);
        vec3f eta2 = eta * eta;
        vec3f etak2 = etak * etak;
        vec3f t0 = eta2 - etak2 - sin2;
        vec3f a2plusb2 = sqrt3(t0 * t0 + 4.0f * eta2 * etak2);
        vec3f t1 = a2plusb2 + cos2;
        vec3f a = sqrt3((a2plusb2 + t0) / 2.0f);
        vec3f t2 = 2.0f * a * cosw;
        vec3f rs = (t1 - t2) / (t1 + t2);
        vec3f t3 = cos2 * a2plusb2 + sin2 * sin2;
        vec3f t4 = t2 * sin2;
        vec3f rp = rs * (t3 - t4) / (t3 + t4);
        return (rp + rs) / 2.0f;
    }
    static float microfacet_distribution(float roughness, vec3f normal, vec3f halfway, bool ggx)  // :1506-1521
    {
        float cosine = dot(normal, halfway);
        if (cosine <= 0.0f) { return 0.0f; }
        float roughness2 = roughness * roughness;
        float cosine2 = cosine * cosine;
        if (ggx) {
            return roughness2 / (PI * (cosine2 * roughness2 + 1 - cosine2) * (cosine2 * roughness2 + 1 - cosine2));
        } else {
            return lpm_expf((cosine2 - 1) / (roughness2 * cosine2)) / (PI * roughness2 * cosine2 * cosine2);
        }
    }
    static float microfacet_shadowing1(float roughness, vec3f normal, vec3f halfway, vec3f direction, bool ggx)  // :1523-1549
    {
        float cosine = dot(normal, direction);
        float cosineh = dot(halfway, direction);
        if (cosine * cosineh <= 0.0f) { return 0.0f; }
        float roughness2 = roughness * roughness;
        float cosine2 = cosine * cosine;
        if (ggx) {
            return 2.0f * fabsf(cosine) / (fabsf(cosine) + sqrtf(cosine2 - roughness2 * cosine2 + roughness2));
        } else {
            float ci = fabsf(cosine) / (roughness * sqrtf(1 - cosine2));
            if (ci < 1.6f) { return (3.535f * ci + 2.181f * ci * ci) / (1.0f + 2.276f * ci + 2.577f * ci * ci); }
            return 1.0f;
        }
    }
    static float microfacet_shadowing(float roughness, vec3f normal, vec3f halfway, vec3f outgoing, vec3f incoming, bool ggx)  // :1551-1555
    {
        return microfacet_shadowing1(roughness, normal, halfway, outgoing, ggx) *
               microfacet_shadowing1(roughness, normal, halfway, incoming, ggx);
    }

    // ---- geometry helpers (:2424-2450, :2682-2685) ----
    static float copysignf_(float mag, float sgn) { return sgn < 0.0f ? -mag : mag; }  // :2436
    static mat3x3f basis_fromz(vec3f v)  // :2424-2434
    {
        vec3f z = normalize(v);
        float sign = copysignf_(1.0f, z.z);
        float a = -1.0f / (sign + z.z);
        float b = z.x * z.y * a;
        vec3f x = {1.0f + sign * z.x * z.x * a, sign * b, -sign * z.x};
        vec3f y = {b, sign + z.y * z.y * a, -z.y};
        return {{x, y, z}};
    }
    static vec3f reflect_(vec3f w, vec3f nrm) { return -w + 2 * dot(nrm, w) * nrm; }  // :2439-2442
    static vec3f refract_(vec3f w, vec3f nrm, float inv_eta)  // :2444-2450
    {
        float cosine = dot(nrm, w);
        float k = 1 + inv_eta * inv_eta * (cosine * cosine - 1);
        if (k < 0.0f) { return v3(0.0f); }
        return -w * inv_eta + (inv_eta * cosine - sqrtf(k)) * nrm;
    }
    static bool same_hemisphere(vec3f normal, vec3f outgoing, vec3f incoming) { return dot(normal, outgoing) * dot(normal, incoming) >= 0; }
    static vec3f up(vec3f normal, vec3f outgoing) { return dot(normal, outgoing) <= 0.0f ? -normal : normal; }  // select(normal,-normal, dot<=0)

    // ---- sampling (:1789-1949, :2216-2229, :2452-2463) ----
    static vec3f sample_hemisphere_cos(vec3f normal, vec2f ruv)  // :2216-2223
    {
        float z = sqrtf(ruv.y);
        float r = sqrtf(1 - z * z);
        float phi = 2 * PI * ruv.x;
        vec3f local_direction = {r * lpm_cosf(phi), r * lpm_sinf(phi), z};
        return normalize(basis_fromz(normal) * local_direction);
    }
    static float sample_hemisphere_cos_pdf(vec3f normal, vec3f direction)  // :2225-2229
    {
        float cosw = dot(normal, direction);
        return cosw <= 0.0f ? 0.0f : cosw / PI;
    }
    static vec3f sample_microfacet(float roughness, vec3f normal, vec2f rn, bool ggx)  // :1902-1918
    {
        float phi = 2.0f * PI * rn.x;
        float theta = 0.0f;
        if (ggx) {
            theta = lpm_atanf(roughness * sqrtf(rn.y / (1 - rn.y)));
        } else {
            float roughness2 = roughness * roughness;
            theta = lpm_atanf(sqrtf(-roughness2 * lpm_logf(1 - rn.y)));
        }
        vec3f local_half_vector = {lpm_cosf(phi) * lpm_sinf(theta), lpm_sinf(phi) * lpm_sinf(theta), lpm_cosf(theta)};
        return normalize(basis_fromz(normal) * local_half_vector);
    }
    static float sample_microfacet_pdf(float roughness, vec3f normal, vec3f halfway, bool ggx)  // :2209-2214
    {
        float cosine = dot(normal, halfway);
        if (cosine < 0) { return 0.0f; }
        return microfacet_distribution(roughness, normal, halfway, ggx) * cosine;
    }
    static vec3f sample_sphere(vec2f ruv)  // :2452-2458
    {
        float z = 2.0f * ruv.y - 1.0f;
        float r = sqrtf(clamp_(1.0f - z * z, 0.0f, 1.0f));
        float phi = 2.0f * PI * ruv.x;
        return {r * lpm_cosf(phi), r * lpm_sinf(phi), z};
    }
    static float sample_sphere_pdf() { return 1.0f / (4.0f * PI); }  // :2460-2463

    static vec3f sample_matte(vec3f, vec3f normal, vec3f outgoing, vec2f rn)  // :1808-1812
    {
        return sample_hemisphere_cos(up(normal, outgoing), rn);
    }
    static vec3f sample_glossy(vec3f, float ior, float roughness, vec3f normal, vec3f outgoing, float rnl, vec2f rn)  // :1814-1829
    {
        vec3f up_normal = up(normal, outgoing);
        if (rnl < fresnel_dielectric(ior, up_normal, outgoing))
        {
            vec3f halfway = sample_microfacet(roughness, up_normal, rn, true);
            vec3f incoming = reflect_(outgoing, halfway);
            if (!same_hemisphere(up_normal, outgoing, incoming)) { return v3(0.0f); }
            return incoming;
        }
        return sample_hemisphere_cos(up_normal, rn);
    }
    static vec3f sample_reflective(vec3f, float roughness, vec3f normal, vec3f outgoing, vec2f rn)  // :1831-1839
    {
        vec3f up_normal = up(normal, outgoing);
        vec3f halfway = sample_microfacet(roughness, up_normal, rn, true);
        vec3f incoming = reflect_(outgoing, halfway);
        if (!same_hemisphere(up_normal, outgoing, incoming)) { return v3(0.0f); }
        return incoming;
    }
    static vec3f sample_transparent(vec3f, float ior, float roughness, vec3f normal, vec3f outgoing, float rnl, vec2f rn)  // :1841-1859
    {
        vec3f up_normal = up(normal, outgoing);
        vec3f halfway = sample_microfacet(roughness, up_normal, rn, true);
        if (rnl < fresnel_dielectric(ior, halfway, outgoing))
        {
            vec3f incoming = reflect_(outgoing, halfway);
            if (!same_hemisphere(up_normal, outgoing, incoming)) { return v3(0.0f); }
            return incoming;
        }
        vec3f reflected = reflect_(outgoing, halfway);
        vec3f incoming = -reflect_(reflected, up_normal);
        if (same_hemisphere(up_normal, outgoing, incoming)) { return v3(0.0f); }
        return incoming;
    }
    static vec3f sample_refractive(vec3f, float ior, float roughness, vec3f normal, vec3f outgoing, float rnl, vec2f rn)  // :1861-1880
    {
        bool entering = dot(normal, outgoing) >= 0;
        vec3f up_normal = entering ? normal : -normal;
        vec3f halfway = sample_microfacet(roughness, up_normal, rn, true);
        if (rnl < fresnel_dielectric(entering ? ior : 1.0f / ior, halfway, outgoing))
        {
            vec3f incoming = reflect_(outgoing, halfway);
            if (!same_hemisphere(up_normal, outgoing, incoming)) { return v3(0.0f); }
            return incoming;
        }
        vec3f incoming = refract_(outgoing, halfway, entering ? 1.0f / ior : ior);
        if (same_hemisphere(up_normal, outgoing, incoming)) { return v3(0.0f); }
        return incoming;
    }
    static vec3f sample_gltfpbr(vec3f color, float ior, float roughness, float metallic, vec3f normal, vec3f outgoing, float rnl, vec2f rn)  // :1882-1900
    {
        vec3f up_normal = up(normal, outgoing);
        vec3f reflectivity = mix3(eta_to_reflectivity(v3(ior)), color, metallic);
        vec3f fs = fresnel_schlick_vec3f(reflectivity, up_normal, outgoing);
        if (rnl < (fs.x + fs.y + fs.z) / 3.0f)
        {
            vec3f halfway = sample_microfacet(roughness, up_normal, rn, true);
            vec3f incoming = reflect_(outgoing, halfway);
            if (!same_hemisphere(up_normal, outgoing, incoming)) { return v3(0.0f); }
            return incoming;
        }
        return sample_hemisphere_cos(up_normal, rn);
    }
    static vec3f sample_bsdfcos(const MaterialPoint &m, vec3f normal, vec3f outgoing, float rnl, vec2f rn)  // :1789-1806
    {
        if (m.roughness == 0.0f) { return v3(0.0f); }
        switch (m.mat_type)
        {
            case LUPIN_MAT_MATTE: return sample_matte(m.color, normal, outgoing, rn);
            case LUPIN_MAT_GLOSSY: return sample_glossy(m.color, m.ior, m.roughness, normal, outgoing, rnl, rn);
            case LUPIN_MAT_REFLECTIVE: return sample_reflective(m.color, m.roughness, normal, outgoing, rn);
            case LUPIN_MAT_TRANSPARENT: return sample_transparent(m.color, m.ior, m.roughness, normal, outgoing, rnl, rn);
            case LUPIN_MAT_REFRACTIVE: return sample_refractive(m.color, m.ior, m.roughness, normal, outgoing, rnl, rn);
            case LUPIN_MAT_SUBSURFACE: return sample_refractive(m.color, m.ior, m.roughness, normal, outgoing, rnl, rn);
            case LUPIN_MAT_GLTFPBR: return sample_gltfpbr(m.color, m.ior, m.roughness, m.metallic, normal, outgoing, rnl, rn);
            default: return v3(0.0f);
        }
    }

    // ---- volumes (:1920-1949, :2092-2095, :2339-2347, :2406-2422) ----
    static float sample_transmittance(vec3f density, float max_distance, float rl, float rd)  // :1920-1926
    {
        int32_t channel = f2i(rl * 3);
        channel = channel < 0 ? 0 : (channel > 2 ? 2 : channel);
        float distance = density[channel] == 0 ? F32_MAX : -lpm_logf(1.0f - rd) / density[channel];
        return fmin_(distance, max_distance);
    }
    static vec3f eval_transmittance(vec3f density, float distance) { return exp3(-density * distance); }  // :2092-2095
    static float sample_transmittance_pdf(vec3f density, float distance, float max_distance)  // :2406-2413
    {
        if (distance < max_distance) {
            return dot(density * exp3(-density * distance), v3(1.0f)) / 3.0f;
        } else {
            return dot(exp3(-density * max_distance), v3(1.0f)) / 3.0f;
        }
    }
    static vec3f sample_scattering(const MaterialPoint &mat, vec3f outgoing, vec2f rn)  // :1928-1949
    {
        if (all_eq0(mat.density)) { return v3(0.0f); }
        float cos_theta = 0.0f;
        if (fabsf(mat.sc_anisotropy) < 1e-3f) {
            cos_theta = 1.0f - 2.0f * rn.y;
        } else {
            float square = (1.0f - mat.sc_anisotropy * mat.sc_anisotropy) / (1.0f + mat.sc_anisotropy - 2.0f * mat.sc_anisotropy * rn.y);
            cos_theta = (1.0f + mat.sc_anisotropy * mat.sc_anisotropy - square * square) / (2.0f * mat.sc_anisotropy);
        }
        float sin_theta = sqrtf(fmax_(0.0f, 1.0f - cos_theta * cos_theta));
        float phi = 2.0f * PI * rn.x;
        vec3f local_incoming = {sin_theta * lpm_cosf(phi), sin_theta * lpm_sinf(phi), cos_theta};
        return basis_fromz(-outgoing) * local_incoming;
    }
    static vec3f eval_scattering(const MaterialPoint &mat, vec3f outgoing, vec3f incoming)  // :2339-2347
    {
        if (all_eq0(mat.density)) { return v3(0.0f); }
        float cosine = -dot(outgoing, incoming);
        float denom = 1.0f + mat.sc_anisotropy * mat.sc_anisotropy - 2.0f * mat.sc_anisotropy * cosine;
        float phasefunction = (1.0f - mat.sc_anisotropy * mat.sc_anisotropy) / (4.0f * PI * denom * sqrtf(denom));
        return mat.scattering * mat.density * phasefunction;
    }
    static float sample_scattering_pdf(const MaterialPoint &mat, vec3f outgoing, vec3f incoming)  // :2415-2422
    {
        if (all_eq0(mat.density)) { return 0.0f; }
        float cosine = -dot(outgoing, incoming);
        float denom = 1.0f + mat.sc_anisotropy * mat.sc_anisotropy - 2.0f * mat.sc_anisotropy * cosine;
        return (1.0f - mat.sc_anisotropy * mat.sc_anisotropy) / (4.0f * PI * denom * sqrtf(denom));
    }

    // ---- BSDF evaluation (:1951-2090) ----
    static vec3f eval_matte(vec3f color, vec3f normal, vec3f outgoing, vec3f incoming)  // :1970-1974
    {
        if (dot(normal, incoming) * dot(normal, outgoing) <= 0) { return v3(0.0f); }
        return color / PI * fabsf(dot(normal, incoming));
    }
    static vec3f eval_glossy(vec3f color, float ior, float roughness, vec3f normal, vec3f outgoing, vec3f incoming)  // :1976-1991
    {
        if (dot(normal, incoming) * dot(normal, outgoing) <= 0) { return v3(0.0f); }
        vec3f up_normal = up(normal, outgoing);
        float F1 = fresnel_dielectric(ior, up_normal, outgoing);
        vec3f halfway = normalize(incoming + outgoing);
        float F = fresnel_dielectric(ior, halfway, incoming);
        float D = microfacet_distribution(roughness, up_normal, halfway, true);
        float G = microfacet_shadowing(roughness, up_normal, halfway, outgoing, incoming, true);
        return color * (1.0f - F1) / PI * fabsf(dot(up_normal, incoming)) +
               v3(1.0f) * F * D * G / (4.0f * dot(up_normal, outgoing) * dot(up_normal, incoming)) * fabsf(dot(up_normal, incoming));
    }
    static vec3f eval_reflective(vec3f color, float roughness, vec3f normal, vec3f outgoing, vec3f incoming)  // :1993-2006
    {
        if (dot(normal, incoming) * dot(normal, outgoing) <= 0) { return v3(0.0f); }
        vec3f up_normal = up(normal, outgoing);
        vec3f halfway = normalize(incoming + outgoing);
        vec3f F = fresnel_conductor(reflectivity_to_eta(color), v3(0.0f), halfway, incoming);
        float D = microfacet_distribution(roughness, up_normal, halfway, true);
        float G = microfacet_shadowing(roughness, up_normal, halfway, outgoing, incoming, true);
        return F * D * G / (4 * dot(up_normal, outgoing) * dot(up_normal, incoming)) * fabsf(dot(up_normal, incoming));
    }
    static vec3f eval_transparent(vec3f color, float ior, float roughness, vec3f normal, vec3f outgoing, vec3f incoming)  // :2008-2035
    {
        vec3f up_normal = up(normal, outgoing);
        if (dot(normal, incoming) * dot(normal, outgoing) >= 0.0f)
        {
            vec3f halfway = normalize(incoming + outgoing);
            float F = fresnel_dielectric(ior, halfway, outgoing);
            float D = microfacet_distribution(roughness, up_normal, halfway, true);
            float G = microfacet_shadowing(roughness, up_normal, halfway, outgoing, incoming, true);
            return v3(1.0f) * F * D * G / (4 * dot(up_normal, outgoing) * dot(up_normal, incoming)) * fabsf(dot(up_normal, incoming));
        }
        vec3f reflected = reflect_(-incoming, up_normal);
        vec3f halfway = normalize(reflected + outgoing);
        float F = fresnel_dielectric(ior, halfway, outgoing);
        float D = microfacet_distribution(roughness, up_normal, halfway, true);
        float G = microfacet_shadowing(roughness, up_normal, halfway, outgoing, reflected, true);
        return color * (1.0f - F) * D * G / (4.0f * dot(up_normal, outgoing) * dot(up_normal, reflected)) * (fabsf(dot(up_normal, reflected)));
    }
    static vec3f eval_refractive(vec3f, float ior, float roughness, vec3f normal, vec3f outgoing, vec3f incoming)  // :2037-2071
    {
        bool entering = dot(normal, outgoing) >= 0;
        vec3f up_normal = entering ? normal : -normal;
        float rel_ior = entering ? ior : 1.0f / ior;
        if (dot(normal, incoming) * dot(normal, outgoing) >= 0)
        {
            vec3f halfway = normalize(incoming + outgoing);
            float F = fresnel_dielectric(rel_ior, halfway, outgoing);
            float D = microfacet_distribution(roughness, up_normal, halfway, true);
            float G = microfacet_shadowing(roughness, up_normal, halfway, outgoing, incoming, true);
            return v3(1.0f * F * D * G / fabsf(4.0f * dot(normal, outgoing) * dot(normal, incoming)) * fabsf(dot(normal, incoming)));
        }
        vec3f halfway = -normalize(rel_ior * incoming + outgoing) * (entering ? 1.0f : -1.0f);
        float F = fresnel_dielectric(rel_ior, halfway, outgoing);
        float D = microfacet_distribution(roughness, up_normal, halfway, true);
        float G = microfacet_shadowing(roughness, up_normal, halfway, outgoing, incoming, true);
        float pw = rel_ior * dot(halfway, incoming) + dot(halfway, outgoing);
        return v3(1.0f) *
               fabsf((dot(outgoing, halfway) * dot(incoming, halfway)) / (dot(outgoing, normal) * dot(incoming, normal))) *
               (1 - F) * D * G / (pw * pw) * fabsf(dot(normal, incoming));
    }
    static vec3f eval_gltfpbr(vec3f color, float ior, float roughness, float metallic, vec3f normal, vec3f outgoing, vec3f incoming)  // :2073-2090
    {
        if (dot(normal, incoming) * dot(normal, outgoing) <= 0.0f) { return v3(0.0f); }
        vec3f reflectivity = mix3(eta_to_reflectivity(v3(ior)), color, metallic);
        vec3f up_normal = up(normal, outgoing);
        vec3f F1 = fresnel_schlick_vec3f(reflectivity, up_normal, outgoing);
        vec3f halfway = normalize(incoming + outgoing);
        vec3f F = fresnel_schlick_vec3f(reflectivity, halfway, incoming);
        float D = microfacet_distribution(roughness, up_normal, halfway, true);
        float G = microfacet_shadowing(roughness, up_normal, halfway, outgoing, incoming, true);
        return color * (1 - metallic) * (1 - F1) / PI * fabsf(dot(up_normal, incoming)) +
               F * D * G / (4 * dot(up_normal, outgoing) * dot(up_normal, incoming)) * fabsf(dot(up_normal, incoming));
    }
    static vec3f eval_bsdfcos(const MaterialPoint &m, vec3f normal, vec3f outgoing, vec3f incoming)  // :1951-1968
    {
        if (m.roughness == 0.0f) { return v3(0.0f); }
        switch (m.mat_type)
        {
            case LUPIN_MAT_MATTE: return eval_matte(m.color, normal, outgoing, incoming);
            case LUPIN_MAT_GLOSSY: return eval_glossy(m.color, m.ior, m.roughness, normal, outgoing, incoming);
            case LUPIN_MAT_REFLECTIVE: return eval_reflective(m.color, m.roughness, normal, outgoing, incoming);
            case LUPIN_MAT_TRANSPARENT: return eval_transparent(m.color, m.ior, m.roughness, normal, outgoing, incoming);
            case LUPIN_MAT_REFRACTIVE: return eval_refractive(m.color, m.ior, m.roughness, normal, outgoing, incoming);
            case LUPIN_MAT_SUBSURFACE: return eval_refractive(m.color, m.ior, m.roughness, normal, outgoing, incoming);
            case LUPIN_MAT_GLTFPBR: return eval_gltfpbr(m.color, m.ior, m.roughness, m.metallic, normal, outgoing, incoming);
            default: return v3(0.0f);
        }
    }

    // ---- BSDF pdfs (:2097-2207) ----
    static float sample_matte_pdf(vec3f, vec3f normal, vec3f outgoing, vec3f incoming)  // :2117-2122
    {
        if (dot(normal, incoming) * dot(normal, outgoing) <= 0.0f) { return 0.0f; }
        return sample_hemisphere_cos_pdf(up(normal, outgoing), incoming);
    }
    static float sample_glossy_pdf(vec3f, float ior, float roughness, vec3f normal, vec3f outgoing, vec3f incoming)  // :2124-2134
    {
        if (dot(normal, incoming) * dot(normal, outgoing) <= 0.0f) { return 0.0f; }
        vec3f up_normal = up(normal, outgoing);
        vec3f halfway = normalize(outgoing + incoming);
        float F = fresnel_dielectric(ior, up_normal, outgoing);
        return F * sample_microfacet_pdf(roughness, up_normal, halfway, true) / (4 * fabsf(dot(outgoing, halfway))) +
               (1 - F) * sample_hemisphere_cos_pdf(up_normal, incoming);
    }
    static float sample_reflective_pdf(vec3f, float roughness, vec3f normal, vec3f outgoing, vec3f incoming)  // :2136-2144
    {
        if (dot(normal, incoming) * dot(normal, outgoing) <= 0) { return 0.0f; }
        vec3f up_normal = up(normal, outgoing);
        vec3f halfway = normalize(outgoing + incoming);
        return sample_microfacet_pdf(roughness, up_normal, halfway, true) / (4 * fabsf(dot(outgoing, halfway)));
    }
    static float sample_transparent_pdf(vec3f, float ior, float roughness, vec3f normal, vec3f outgoing, vec3f incoming)  // :2146-2165
    {
        vec3f up_normal = up(normal, outgoing);
        if (dot(normal, incoming) * dot(normal, outgoing) >= 0)
        {
            vec3f halfway = normalize(incoming + outgoing);
            return fresnel_dielectric(ior, halfway, outgoing) * sample_microfacet_pdf(roughness, up_normal, halfway, true) / (4 * fabsf(dot(outgoing, halfway)));
        }
        vec3f reflected = reflect_(-incoming, up_normal);
        vec3f halfway = normalize(reflected + outgoing);
        float d = (1 - fresnel_dielectric(ior, halfway, outgoing)) * sample_microfacet_pdf(roughness, up_normal, halfway, true);
        return d / (4 * fabsf(dot(outgoing, halfway)));
    }
    static float sample_refractive_pdf(vec3f, float ior, float roughness, vec3f normal, vec3f outgoing, vec3f incoming)  // :2167-2192
    {
        bool entering = dot(normal, outgoing) >= 0;
        vec3f up_normal = entering ? normal : -normal;
        float rel_ior = entering ? ior : 1.0f / ior;
        if (dot(normal, incoming) * dot(normal, outgoing) >= 0.0f)
        {
            vec3f halfway = normalize(incoming + outgoing);
            return fresnel_dielectric(rel_ior, halfway, outgoing) * sample_microfacet_pdf(roughness, up_normal, halfway, true) / (4 * fabsf(dot(outgoing, halfway)));
        }
        vec3f halfway = -normalize(rel_ior * incoming + outgoing) * (entering ? 1.0f : -1.0f);
        float pw = rel_ior * dot(halfway, incoming) + dot(halfway, outgoing);
        return (1 - fresnel_dielectric(rel_ior, halfway, outgoing)) * sample_microfacet_pdf(roughness, up_normal, halfway, true) *
               fabsf(dot(halfway, incoming)) / (pw * pw);
    }
    static float sample_gltfpbr_pdf(vec3f color, float ior, float roughness, float metallic, vec3f normal, vec3f outgoing, vec3f incoming)  // :2194-2207
    {
        if (dot(normal, incoming) * dot(normal, outgoing) <= 0) { return 0.0f; }
        vec3f up_normal = up(normal, outgoing);
        vec3f halfway = normalize(outgoing + incoming);
        vec3f reflectivity = mix3(eta_to_reflectivity(v3(ior)), color, metallic);
        vec3f fs = fresnel_schlick_vec3f(reflectivity, up_normal, outgoing);
        float F = (fs.x + fs.y + fs.z) / 3.0f;
        return F * sample_microfacet_pdf(roughness, up_normal, halfway, true) / (4 * fabsf(dot(outgoing, halfway))) +
               (1 - F) * sample_hemisphere_cos_pdf(up_normal, incoming);
    }
    static float sample_bsdfcos_pdf(const MaterialPoint &m, vec3f normal, vec3f outgoing, vec3f incoming)  // :2097-2115
    {
        if (m.roughness == 0.0f) { return 0.0f; }
        switch (m.mat_type)
        {
            case LUPIN_MAT_MATTE: return sample_matte_pdf(m.color, normal, outgoing, incoming);
            case LUPIN_MAT_GLOSSY: return sample_glossy_pdf(m.color, m.ior, m.roughness, normal, outgoing, incoming);
            case LUPIN_MAT_REFLECTIVE: return sample_reflective_pdf(m.color, m.roughness, normal, outgoing, incoming);
            case LUPIN_MAT_TRANSPARENT: return sample_transparent_pdf(m.color, m.ior, m.roughness, normal, outgoing, incoming);
            case LUPIN_MAT_REFRACTIVE: return sample_refractive_pdf(m.color, m.ior, m.roughness, normal, outgoing, incoming);
            case LUPIN_MAT_SUBSURFACE: return sample_refractive_pdf(m.color, m.ior, m.roughness, normal, outgoing, incoming);
            case LUPIN_MAT_GLTFPBR: return sample_gltfpbr_pdf(m.color, m.ior, m.roughness, m.metallic, normal, outgoing, incoming);
            default: return 0.0f;
        }
    }

    // ---- delta lobes (:2231-2404) ----
    static vec3f sample_delta(const MaterialPoint &m, vec3f normal, vec3f outgoing, float rnl)  // :2231-2279
    {
        if (m.roughness != 0.0f) { return v3(0.0f); }
        switch (m.mat_type)
        {
            case LUPIN_MAT_REFLECTIVE: return reflect_(outgoing, up(normal, outgoing));
            case LUPIN_MAT_TRANSPARENT:
            {
                vec3f up_normal = up(normal, outgoing);
                if (rnl < fresnel_dielectric(m.ior, up_normal, outgoing)) { return reflect_(outgoing, up_normal); }
                return -outgoing;
            }
            case LUPIN_MAT_REFRACTIVE:
            {
                if (fabsf(m.ior - 1) < 1e-3f) { return -outgoing; }
                bool entering = dot(normal, outgoing) >= 0;
                vec3f up_normal = entering ? normal : -normal;
                float rel_ior = entering ? m.ior : 1.0f / m.ior;
                if (rnl < fresnel_dielectric(rel_ior, up_normal, outgoing)) { return reflect_(outgoing, up_normal); }
                return refract_(outgoing, up_normal, 1 / rel_ior);
            }
            case LUPIN_MAT_VOLUMETRIC: return -outgoing;
            default: return v3(0.0f);
        }
    }
    static vec3f eval_delta(const MaterialPoint &m, vec3f normal, vec3f outgoing, vec3f incoming)  // :2281-2337
    {
        if (m.roughness != 0.0f) { return v3(0.0f); }
        switch (m.mat_type)
        {
            case LUPIN_MAT_REFLECTIVE:
            {
                if (dot(normal, incoming) * dot(normal, outgoing) <= 0.0f) { return v3(0.0f); }
                return fresnel_conductor(reflectivity_to_eta(m.color), v3(0.0f), up(normal, outgoing), outgoing);
            }
            case LUPIN_MAT_TRANSPARENT:
            {
                vec3f up_normal = up(normal, outgoing);
                if (dot(normal, incoming) * dot(normal, outgoing) >= 0) { return v3(1.0f) * fresnel_dielectric(m.ior, up_normal, outgoing); }
                return m.color * (1 - fresnel_dielectric(m.ior, up_normal, outgoing));
            }
            case LUPIN_MAT_REFRACTIVE:
            {
                if (fabsf(m.ior - 1.0f) < 1e-3f) {
                    return (dot(normal, incoming) * dot(normal, outgoing) <= 0) ? v3(1.0f) : v3(0.0f);
                }
                bool entering = dot(normal, outgoing) >= 0;
                vec3f up_normal = entering ? normal : -normal;
                float rel_ior = entering ? m.ior : 1.0f / m.ior;
                if (dot(normal, incoming) * dot(normal, outgoing) >= 0.0f) { return v3(1.0f) * fresnel_dielectric(rel_ior, up_normal, outgoing); }
                return v3(1.0f) * (1 / (rel_ior * rel_ior)) * (1 - fresnel_dielectric(rel_ior, up_normal, outgoing));
            }
            case LUPIN_MAT_VOLUMETRIC:
                return (dot(normal, incoming) * dot(normal, outgoing) >= 0.0f) ? v3(0.0f) : v3(1.0f);
            default: return v3(0.0f);
        }
    }
    static float sample_delta_pdf(const MaterialPoint &m, vec3f normal, vec3f outgoing, vec3f incoming)  // :2349-2404
    {
        if (m.roughness != 0.0f) { return 0.0f; }
        switch (m.mat_type)
        {
            case LUPIN_MAT_REFLECTIVE:
                return (dot(normal, incoming) * dot(normal, outgoing) <= 0.0f) ? 0.0f : 1.0f;
            case LUPIN_MAT_TRANSPARENT:
            {
                vec3f up_normal = up(normal, outgoing);
                if (dot(normal, incoming) * dot(normal, outgoing) >= 0.0f) { return fresnel_dielectric(m.ior, up_normal, outgoing); }
                return 1.0f - fresnel_dielectric(m.ior, up_normal, outgoing);
            }
            case LUPIN_MAT_REFRACTIVE:
            {
                if (fabsf(m.ior - 1) < 1e-3f) { return (dot(normal, incoming) * dot(normal, outgoing) < 0.0f) ? 1.0f : 0.0f; }
                bool entering = dot(normal, outgoing) >= 0;
                vec3f up_normal = entering ? normal : -normal;
                float rel_ior = entering ? m.ior : 1.0f / m.ior;
                if (dot(normal, incoming) * dot(normal, outgoing) >= 0.0f) { return fresnel_dielectric(rel_ior, up_normal, outgoing); }
                return (1 - fresnel_dielectric(rel_ior, up_normal, outgoing));
            }
            case LUPIN_MAT_VOLUMETRIC:
                return (dot(normal, incoming) * dot(normal, outgoing) >= 0.0f) ? 0.0f : 1.0f;
            default: return 0.0f;
        }
    }

    // ---- light sampling (:2468-2549, :2610-2638, :2790-2802) ----
    uint32_t sample_alias_table(const LupinAliasTableDesc &table)  // :2610-2638
    {
        uint32_t rnd_idx = random_u32_range_unsafe(table.num_bins);
        LupinAliasBin bin = table.bins[rnd_idx];
        if (random_f32() >= bin.alias_threshold) { return bin.alias; }
        return rnd_idx;
    }
    vec3f sample_lights(vec3f pos, vec3f /*outgoing*/)  // :2468-2514
    {
        uint32_t nl = num_lights();
        uint32_t ne = num_envs();
        if (nl + ne <= 0) { return v3(0.0f); }

        uint32_t light_idx = random_u32_range_unsafe(nl + ne);
        if (light_idx < nl)
        {
            n.light_mesh++;
            uint32_t tri_idx = sample_alias_table(s->alias_tables[light_idx]);
            uint32_t instance_idx = s->lights[light_idx].instance_idx;
            const LupinInstance &instance = s->instances[instance_idx];
            uint32_t mesh_idx = instance.mesh_idx;
            vec2f uv = random_tri_uv();

            // mat4x3f_inverse(transpose(tit)) (:2790-2802); a[j] = column j of the world->local affine
            vec4f r0 = tit(instance, 0), r1 = tit(instance, 1), r2 = tit(instance, 2);
            vec3f a0 = {r0.x, r1.x, r2.x}, a1 = {r0.y, r1.y, r2.y}, a2 = {r0.z, r1.z, r2.z}, a3 = {r0.w, r1.w, r2.w};
            vec3f cross_yz = cross(a1, a2);
            vec3f cross_zx = cross(a2, a0);
            vec3f cross_xy = cross(a0, a1);
            // adjoint = transpose(mat3x3(cross_yz, cross_zx, cross_xy)): column i = (cross_yz[i], cross_zx[i], cross_xy[i])
            float determinant = dot(a0, cross_yz);
            float idet = 1.0f / determinant;
            mat3x3f minv = {{v3(cross_yz.x, cross_zx.x, cross_xy.x) * idet,
                             v3(cross_yz.y, cross_zx.y, cross_xy.y) * idet,
                             v3(cross_yz.z, cross_zx.z, cross_xy.z) * idet}};
            vec3f tcol = -(minv * a3);

            vec3f v0 = vert_pos(mesh_idx, index(mesh_idx, tri_idx * 3 + 0));
            vec3f v1 = vert_pos(mesh_idx, index(mesh_idx, tri_idx * 3 + 1));
            vec3f v2 = vert_pos(mesh_idx, index(mesh_idx, tri_idx * 3 + 2));
            float w = 1.0f - uv.x - uv.y;
            vec3f local_tri_pos = v0 * w + v1 * uv.x + v2 * uv.y;
            vec3f world_tri_pos = minv.c[0] * local_tri_pos.x + minv.c[1] * local_tri_pos.y + minv.c[2] * local_tri_pos.z + tcol * 1.0f;
            return normalize(world_tri_pos - pos);
        }
        else
        {
            n.light_env++;
            uint32_t env_idx = light_idx - nl;
            uint32_t env_tex_idx = s->environments[env_idx].emission_tex_idx;
            if (env_tex_idx == SENTINEL_IDX) { return sample_sphere(random_vec2f()); }
            uint32_t sample = sample_alias_table(s->env_alias_tables[env_idx]);
            return env_idx_to_dir(sample, env_idx);
        }
    }
    float sample_lights_pdf(vec3f pos, vec3f incoming)  // :2516-2549
    {
        float pdf = 0.0f;
        uint32_t nl = num_lights();
        uint32_t ne = num_envs();
        pdf += compute_instance_lights_pdf({pos, incoming, 1.0f / incoming});
        for (uint32_t i = 0u; i < ne; i++)
        {
            uint32_t env_tex_idx = s->environments[i].emission_tex_idx;
            if (env_tex_idx == SENTINEL_IDX) { pdf += sample_sphere_pdf(); }
            else
            {
                uint32_t w, h; env_tex_size(i, w, h);
                uint32_t cx, cy; dir_to_env_coords(incoming, i, cx, cy);
                uint32_t pixel_idx = cy * w + cx;
                float prob = s->env_alias_tables[i].bins[pixel_idx].prob;
                float solid_angle = (2.0f * PI / (float)w) * (PI / (float)h) * lpm_sinf(PI * ((float)cy + 0.5f) / (float)h);
                pdf += prob / solid_angle;
            }
        }
        pdf /= (float)(nl + ne);
        return pdf;
    }

    // ---- camera (:505-542) ----
    Ray compute_camera_ray(uint32_t gx, uint32_t gy, uint32_t dim_x, uint32_t dim_y, vec2f pixel_offset)
    {
        vec2f resolution = {(float)dim_x, (float)dim_y};
        vec2f pixel_coord = {(float)gx + 0.5f, (resolution.y - (float)gy) + 0.5f};
        vec2f nudged_uv = {(pixel_coord.x + pixel_offset.x) / resolution.x, (pixel_coord.y + pixel_offset.y) / resolution.y};

        float camera_lens = constants.camera_lens, camera_film = constants.camera_film, camera_aspect = constants.camera_aspect;
        float camera_focus = constants.camera_focus, camera_aperture = constants.camera_aperture;
        vec2f film_size = camera_aspect >= 1 ? vec2f{camera_film, camera_film / camera_aspect} : vec2f{camera_film * camera_aspect, camera_film};
        vec2f lens_uv = random_in_disk();

        mat4 cam;
        for (int c = 0; c < 4; c++) cam.c[c] = {constants.camera_transform.m[c][0], constants.camera_transform.m[c][1], constants.camera_transform.m[c][2], constants.camera_transform.m[c][3]};

        if ((constants.flags & LUPIN_FLAG_CAMERA_ORTHO) != 0)
        {
            float scale = 1.0f / camera_lens;
            vec3f q = {film_size.x * (0.5f - nudged_uv.x) * scale, film_size.y * (0.5f - nudged_uv.y) * scale, camera_lens};
            vec3f e = v3(-q.x, -q.y, 0) + v3(lens_uv.x * camera_aperture / 2.0f, lens_uv.y * camera_aperture / 2.0f, 0);
            vec3f p = {-q.x, -q.y, -camera_focus};
            vec3f d = normalize(p - e) * v3(1.0f, 1.0f, -1.0f);
            Ray res = {e, d, 1.0f / d};
            return transform_ray(res, cam);
        }
        vec3f q = {film_size.x * (0.5f - nudged_uv.x), film_size.y * (0.5f - nudged_uv.y), camera_lens};
        vec3f look_at = -normalize(q);
        vec3f lens_point = {lens_uv.x * (camera_aperture / 2.0f), lens_uv.y * (camera_aperture / 2.0f), 0.0f};
        vec3f focus_point = look_at * camera_focus / fabsf(look_at.z);
        vec3f final_dir = normalize(focus_point - lens_point) * v3(1.0f, 1.0f, -1.0f);
        Ray res = {lens_point, final_dir, 1.0f / final_dir};
        return transform_ray(res, cam);
    }

    static bool vec3f_is_finite(vec3f v)  // :2769-2772
    {
        return (bits(v.x) & 0x7F800000u) != 0x7F800000u && (bits(v.y) & 0x7F800000u) != 0x7F800000u && (bits(v.z) & 0x7F800000u) != 0x7F800000u;
    }
    vec3f clamp_radiance(vec3f radiance) const  // :1774-1783
    {
        vec3f res = radiance;
        if (!vec3f_is_finite(res)) { res = v3(0.0f); }
        float mr = constants.max_radiance;
        if (res.x > mr || res.y > mr || res.z > mr) {
            res *= mr / fmax_(res.x, fmax_(res.y, res.z));
        }
        return res;
    }

    // shared tail of every integrator's loop body (:720-729 and copies)
    // returns false when the path terminates
    bool check_weight_and_roulette(vec3f &weight, int bounce)
    {
        if (all_eq0(weight) || !vec3f_is_finite(weight)) { return false; }
        if (bounce > 3)
        {
            float survive_prob = fmin_(0.99f, fmax_(weight.x, fmax_(weight.y, weight.z)));
            if (random_f32() >= survive_prob) { return false; }
            weight *= 1.0f / survive_prob;
        }
        return true;
    }
    // volume-stack update (:667-681 and copies)
    static void update_volume_stack(MaterialPoint *volume_stack, int &volume_stack_len, const MaterialPoint &mat_point, vec3f normal, vec3f outgoing, vec3f incoming)
    {
        if (is_mat_volumetric(mat_point) && dot(normal, outgoing) * dot(normal, incoming) < 0.0f)
        {
            if (volume_stack_len == 0)
            {
                if (volume_stack_len < MAX_VOLUMES) { volume_stack[volume_stack_len] = mat_point; }
                volume_stack_len++;
            }
            else { volume_stack_len--; }
        }
    }

    // ---- pathtrace_standard (:588-733) ----
    vec3f pathtrace_standard(const Ray &start_ray)
    {
        Ray ray = start_ray;
        vec3f weight = v3(1.0f);
        vec3f radiance = v3(0.0f);
        MaterialPoint volume_stack[MAX_VOLUMES];
        int volume_stack_len = 0;

        for (int bounce = 0; bounce <= (int)MAX_BOUNCES; bounce++)
        {
            n.path_bounces++;
            HitInfo hit = ray_skip_alpha_stochastically(ray);
            if (!hit.hit)
            {
                radiance += weight * sample_environments(ray.dir);
                break;
            }
            n.debug_num_bounces++;   // `if DEBUG { DEBUG_NUM_BOUNCES++; }` (:606-608)

            bool in_volume = false;
            float volume_dst = hit.dst;
            if (volume_stack_len > 0 && volume_stack_len < MAX_VOLUMES)
            {
                MaterialPoint vsdf = volume_stack[volume_stack_len - 1];
                float rnd1 = random_f32();
                float rnd2 = random_f32();
                volume_dst = sample_transmittance(vsdf.density, hit.dst, rnd1, rnd2);
                weight *= eval_transmittance(vsdf.density, volume_dst) / sample_transmittance_pdf(vsdf.density, volume_dst, hit.dst);
                in_volume = volume_dst < hit.dst;
            }

            vec3f outgoing = -ray.dir;
            if (!in_volume)
            {
                vec3f hit_pos = ray.ori + ray.dir * hit.dst;
                MaterialPoint mat_point = get_material_point(hit);
                vec3f normal = compute_shading_normal(hit);

                radiance += weight * mat_point.emission;

                vec3f incoming = v3(0.0f);
                if (!is_mat_delta(mat_point))
                {
                    const float light_prob = 0.5f;
                    const float bsdf_prob = 1.0f - light_prob;
                    if (random_f32() < bsdf_prob)
                    {
                        float rnd0 = random_f32();
                        vec2f rnd1 = random_vec2f();
                        incoming = sample_bsdfcos(mat_point, normal, outgoing, rnd0, rnd1);
                    }
                    else
                    {
                        incoming = sample_lights(hit_pos, outgoing);
                    }
                    if (all_eq0(incoming)) { break; }
                    float prob = bsdf_prob * sample_bsdfcos_pdf(mat_point, normal, outgoing, incoming) +
                                 light_prob * sample_lights_pdf(hit_pos, incoming);
                    weight *= eval_bsdfcos(mat_point, normal, outgoing, incoming) / prob;
                }
                else
                {
                    incoming = sample_delta(mat_point, normal, outgoing, random_f32());
                    if (all_eq0(incoming)) { break; }
                    weight *= eval_delta(mat_point, normal, outgoing, incoming) / sample_delta_pdf(mat_point, normal, outgoing, incoming);
                }

                update_volume_stack(volume_stack, volume_stack_len, mat_point, normal, outgoing, incoming);

                ray.ori = hit_pos;
                ray.dir = incoming;
                ray.inv_dir = 1.0f / ray.dir;
            }
            else
            {
                vec3f hit_pos = ray.ori + ray.dir * volume_dst;
                const MaterialPoint &vsdf = volume_stack[volume_stack_len - 1];
                vec3f incoming = v3(0.0f);
                const float light_prob = 0.5f;
                const float scatter_prob = 1.0f - light_prob;
                if (random_f32() < scatter_prob)
                {
                    float rnd0 = random_f32(); (void)rnd0;
                    vec2f rnd1 = random_vec2f();
                    incoming = sample_scattering(vsdf, outgoing, rnd1);
                }
                else
                {
                    incoming = sample_lights(hit_pos, outgoing);
                }
                if (all_eq0(incoming)) { break; }
                float prob = scatter_prob * sample_scattering_pdf(vsdf, outgoing, incoming) +
                             light_prob * sample_lights_pdf(hit_pos, incoming);
                weight *= eval_scattering(vsdf, outgoing, incoming) / prob;

                ray.ori = hit_pos;
                ray.dir = incoming;
                ray.inv_dir = 1.0f / ray.dir;
            }

            if (!check_weight_and_roulette(weight, bounce)) { break; }
        }
        return radiance;
    }

    static float mis_heuristic(float this_pdf, float other_pdf)  // :935-938
    {
        return (this_pdf * this_pdf) / (this_pdf * this_pdf + other_pdf * other_pdf);
    }

    // ---- pathtrace_mis (:737-933) ----
    vec3f pathtrace_mis(const Ray &start_ray)
    {
        Ray ray = start_ray;
        vec3f weight = v3(1.0f);
        vec3f radiance = v3(0.0f);
        MaterialPoint volume_stack[MAX_VOLUMES];
        int volume_stack_len = 0;
        bool next_emission = true;
        HitInfo next_intersection;

        for (int bounce = 0; bounce <= (int)MAX_BOUNCES; bounce++)
        {
            n.path_bounces++;
            HitInfo hit;
            if (next_emission) { hit = ray_skip_alpha_stochastically(ray); }
            else { hit = next_intersection; }

            if (!hit.hit)
            {
                radiance += weight * sample_environments(ray.dir);
                break;
            }

            bool in_volume = false;
            float volume_dst = hit.dst;
            if (volume_stack_len > 0 && volume_stack_len < MAX_VOLUMES)
            {
                MaterialPoint vsdf = volume_stack[volume_stack_len - 1];
                float rnd1 = random_f32();
                float rnd2 = random_f32();
                volume_dst = sample_transmittance(vsdf.density, hit.dst, rnd1, rnd2);
                weight *= eval_transmittance(vsdf.density, volume_dst) / sample_transmittance_pdf(vsdf.density, volume_dst, hit.dst);
                in_volume = volume_dst < hit.dst;
            }

            vec3f outgoing = -ray.dir;
            if (!in_volume)
            {
                vec3f hit_pos = ray.ori + ray.dir * hit.dst;
                MaterialPoint mat_point = get_material_point(hit);
                vec3f normal = compute_shading_normal(hit);

                if (next_emission) { radiance += weight * mat_point.emission; }

                vec3f incoming = v3(0.0f);
                if (!is_mat_delta(mat_point))
                {
                    for (int i = 0; i < 2; i++)
                    {
                        bool do_sample_light = (i != 0);
                        vec3f mis_incoming = v3(0.0f);
                        if (do_sample_light) {
                            mis_incoming = sample_lights(hit_pos, outgoing);
                        } else {
                            float rnd0 = random_f32();
                            vec2f rnd1 = random_vec2f();
                            mis_incoming = sample_bsdfcos(mat_point, normal, outgoing, rnd0, rnd1);
                        }
                        if (all_eq0(mis_incoming)) { break; }
                        if (!do_sample_light) { incoming = mis_incoming; }

                        vec3f bsdfcos = eval_bsdfcos(mat_point, normal, outgoing, mis_incoming);
                        float light_pdf = sample_lights_pdf(hit_pos, mis_incoming);
                        float bsdf_pdf = sample_bsdfcos_pdf(mat_point, normal, outgoing, mis_incoming);

                        float mis_weight = 0.0f;
                        if (do_sample_light) { mis_weight = mis_heuristic(light_pdf, bsdf_pdf) / light_pdf; }
                        else { mis_weight = mis_heuristic(bsdf_pdf, light_pdf) / bsdf_pdf; }

                        if (all_ne0(bsdfcos) && mis_weight != 0)
                        {
                            Ray mis_ray = {hit_pos, mis_incoming, 1.0f / mis_incoming};
                            HitInfo mis_hit = ray_scene_intersection(mis_ray);
                            if (!do_sample_light) { next_intersection = mis_hit; }
                            vec3f emission = v3(0.0f);
                            if (mis_hit.hit) {
                                MaterialPoint mis_mat_point = get_material_point(mis_hit);
                                emission = mis_mat_point.emission;
                            } else {
                                emission = sample_environments(mis_incoming);
                            }
                            radiance += weight * bsdfcos * emission * mis_weight;
                        }
                    }
                    weight *= eval_bsdfcos(mat_point, normal, outgoing, incoming) / sample_bsdfcos_pdf(mat_point, normal, outgoing, incoming);
                    next_emission = false;
                }
                else
                {
                    incoming = sample_delta(mat_point, normal, outgoing, random_f32());
                    if (all_eq0(incoming)) { break; }
                    weight *= eval_delta(mat_point, normal, outgoing, incoming) / sample_delta_pdf(mat_point, normal, outgoing, incoming);
                    next_emission = true;
                }

                update_volume_stack(volume_stack, volume_stack_len, mat_point, normal, outgoing, incoming);

                ray.ori = hit_pos;
                ray.dir = incoming;
                ray.inv_dir = 1.0f / ray.dir;
            }
            else
            {
                vec3f hit_pos = ray.ori + ray.dir * volume_dst;
                const MaterialPoint &vsdf = volume_stack[volume_stack_len - 1];
                vec3f incoming = v3(0.0f);
                const float light_prob = 0.5f;
                const float scatter_prob = 1.0f - light_prob;
                if (random_f32() < scatter_prob)
                {
                    float rnd0 = random_f32(); (void)rnd0;
                    vec2f rnd1 = random_vec2f();
                    incoming = sample_scattering(vsdf, outgoing, rnd1);
                    next_emission = true;
                }
                else
                {
                    incoming = sample_lights(hit_pos, outgoing);
                    next_emission = true;
                }
                if (all_eq0(incoming)) { break; }
                float prob = scatter_prob * sample_scattering_pdf(vsdf, outgoing, incoming) +
                             light_prob * sample_lights_pdf(hit_pos, incoming);
                weight *= eval_scattering(vsdf, outgoing, incoming) / prob;

                ray.ori = hit_pos;
                ray.dir = incoming;
                ray.inv_dir = 1.0f / ray.dir;
            }

            if (!check_weight_and_roulette(weight, bounce)) { break; }
        }
        return radiance;
    }

    // ---- pathtrace_naive (:942-1059) ----
    vec3f pathtrace_naive(const Ray &start_ray)
    {
        Ray ray = start_ray;
        vec3f weight = v3(1.0f);
        vec3f radiance = v3(0.0f);
        MaterialPoint volume_stack[MAX_VOLUMES];
        int volume_stack_len = 0;

        for (int bounce = 0; bounce <= (int)MAX_BOUNCES; bounce++)
        {
            n.path_bounces++;
            HitInfo hit = ray_skip_alpha_stochastically(ray);
            if (!hit.hit)
            {
                radiance += weight * sample_environments(ray.dir);
                break;
            }

            bool in_volume = false;
            float volume_dst = hit.dst;
            if (volume_stack_len > 0 && volume_stack_len < MAX_VOLUMES)
            {
                MaterialPoint vsdf = volume_stack[volume_stack_len - 1];
                float rnd1 = random_f32();
                float rnd2 = random_f32();
                volume_dst = sample_transmittance(vsdf.density, hit.dst, rnd1, rnd2);
                weight *= eval_transmittance(vsdf.density, volume_dst) / sample_transmittance_pdf(vsdf.density, volume_dst, hit.dst);
                in_volume = volume_dst < hit.dst;
            }

            vec3f outgoing = -ray.dir;
            MaterialPoint mat_point = get_material_point(hit);

            if (!in_volume)
            {
                vec3f hit_pos = ray.ori + ray.dir * hit.dst;
                vec3f normal = compute_shading_normal(hit);
                radiance += weight * mat_point.emission;

                vec3f incoming = v3(0.0f);
                if (!is_mat_delta(mat_point))
                {
                    float rnd0 = random_f32();
                    vec2f rnd1 = random_vec2f();
                    incoming = sample_bsdfcos(mat_point, normal, outgoing, rnd0, rnd1);
                    if (all_eq0(incoming)) { break; }
                    weight *= eval_bsdfcos(mat_point, normal, outgoing, incoming) / sample_bsdfcos_pdf(mat_point, normal, outgoing, incoming);
                }
                else
                {
                    incoming = sample_delta(mat_point, normal, outgoing, random_f32());
                    if (all_eq0(incoming)) { break; }
                    weight *= eval_delta(mat_point, normal, outgoing, incoming) / sample_delta_pdf(mat_point, normal, outgoing, incoming);
                }

                update_volume_stack(volume_stack, volume_stack_len, mat_point, normal, outgoing, incoming);

                ray.ori = hit_pos;
                ray.dir = incoming;
                ray.inv_dir = 1.0f / ray.dir;
            }
            else
            {
                vec3f hit_pos = ray.ori + ray.dir * volume_dst;
                const MaterialPoint &vsdf = volume_stack[volume_stack_len - 1];
                float rnd0 = random_f32(); (void)rnd0;
                vec2f rnd1 = random_vec2f();
                vec3f incoming = sample_scattering(vsdf, outgoing, rnd1);
                if (all_eq0(incoming)) { break; }
                float prob = sample_scattering_pdf(vsdf, outgoing, incoming);
                weight *= eval_scattering(vsdf, outgoing, incoming) / prob;

                ray.ori = hit_pos;
                ray.dir = incoming;
                ray.inv_dir = 1.0f / ray.dir;
            }

            if (!check_weight_and_roulette(weight, bounce)) { break; }
        }
        return radiance;
    }

    // ---- pathtrace_direct (:1062-1245) ----
    vec3f pathtrace_direct(const Ray &start_ray)
    {
        Ray ray = start_ray;
        vec3f weight = v3(1.0f);
        vec3f radiance = v3(0.0f);
        MaterialPoint volume_stack[MAX_VOLUMES];
        int volume_stack_len = 0;
        bool next_emission = true;

        for (int bounce = 0; bounce <= (int)MAX_BOUNCES; bounce++)
        {
            n.path_bounces++;
            HitInfo hit = ray_skip_alpha_stochastically(ray);
            if (!hit.hit)
            {
                if (next_emission) { radiance += weight * sample_environments(ray.dir); }
                break;
            }

            bool in_volume = false;
            float volume_dst = hit.dst;
            if (volume_stack_len > 0 && volume_stack_len < MAX_VOLUMES)
            {
                MaterialPoint vsdf = volume_stack[volume_stack_len - 1];
                float rnd1 = random_f32();
                float rnd2 = random_f32();
                volume_dst = sample_transmittance(vsdf.density, hit.dst, rnd1, rnd2);
                weight *= eval_transmittance(vsdf.density, volume_dst) / sample_transmittance_pdf(vsdf.density, volume_dst, hit.dst);
                in_volume = volume_dst < hit.dst;
            }

            vec3f outgoing = -ray.dir;
            if (!in_volume)
            {
                vec3f hit_pos = ray.ori + ray.dir * hit.dst;
                MaterialPoint mat_point = get_material_point(hit);
                vec3f normal = compute_shading_normal(hit);

                if (next_emission) { radiance += weight * mat_point.emission; }

                if (!is_mat_delta(mat_point))
                {
                    vec3f incoming = sample_lights(hit_pos, outgoing);
                    float pdf = sample_lights_pdf(hit_pos, incoming);
                    vec3f bsdfcos = eval_bsdfcos(mat_point, normal, outgoing, incoming);
                    if (all_ne0(bsdfcos) && pdf > 0.0f)
                    {
                        Ray light_ray = {hit_pos, incoming, 1.0f / incoming};
                        HitInfo light_hit = ray_scene_intersection(light_ray);
                        vec3f emission = v3(0.0f);
                        if (light_hit.hit) {
                            MaterialPoint light_mat_point = get_material_point(light_hit);
                            emission = light_mat_point.emission;
                        } else {
                            emission = sample_environments(incoming);
                        }
                        radiance += weight * bsdfcos * emission / pdf;
                    }
                    next_emission = false;
                }
                else
                {
                    next_emission = true;
                }

                vec3f incoming = v3(0.0f);
                if (!is_mat_delta(mat_point))
                {
                    const float light_prob = 0.5f;
                    const float bsdf_prob = 1.0f - light_prob;
                    if (random_f32() < bsdf_prob)
                    {
                        float rnd0 = random_f32();
                        vec2f rnd1 = random_vec2f();
                        incoming = sample_bsdfcos(mat_point, normal, outgoing, rnd0, rnd1);
                    }
                    else
                    {
                        incoming = sample_lights(hit_pos, outgoing);
                    }
                    if (all_eq0(incoming)) { break; }
                    float prob = bsdf_prob * sample_bsdfcos_pdf(mat_point, normal, outgoing, incoming) +
                                 light_prob * sample_lights_pdf(hit_pos, incoming);
                    weight *= eval_bsdfcos(mat_point, normal, outgoing, incoming) / prob;
                }
                else
                {
                    incoming = sample_delta(mat_point, normal, outgoing, random_f32());
                    if (all_eq0(incoming)) { break; }
                    weight *= eval_delta(mat_point, normal, outgoing, incoming) / sample_delta_pdf(mat_point, normal, outgoing, incoming);
                }

                update_volume_stack(volume_stack, volume_stack_len, mat_point, normal, outgoing, incoming);

                ray.ori = hit_pos;
                ray.dir = incoming;
                ray.inv_dir = 1.0f / ray.dir;
            }
            else
            {
                vec3f hit_pos = ray.ori + ray.dir * volume_dst;
                const MaterialPoint &vsdf = volume_stack[volume_stack_len - 1];
                vec3f incoming = v3(0.0f);
                const float light_prob = 0.5f;
                const float scatter_prob = 1.0f - light_prob;
                if (random_f32() < scatter_prob)
                {
                    float rnd0 = random_f32(); (void)rnd0;
                    vec2f rnd1 = random_vec2f();
                    incoming = sample_scattering(vsdf, outgoing, rnd1);
                }
                else
                {
                    incoming = sample_lights(hit_pos, outgoing);
                }
                if (all_eq0(incoming)) { break; }
                float prob = scatter_prob * sample_scattering_pdf(vsdf, outgoing, incoming) +
                             light_prob * sample_lights_pdf(hit_pos, incoming);
                weight *= eval_scattering(vsdf, outgoing, incoming) / prob;

                ray.ori = hit_pos;
                ray.dir = incoming;
                ray.inv_dir = 1.0f / ray.dir;
            }

            if (!check_weight_and_roulette(weight, bounce)) { break; }
        }
        return radiance;
    }

    // hash_color (:544-573): three PCG draws from a local state seeded with the id
    static vec3f hash_color(uint32_t id)
    {
        uint32_t local_state = id;
        float c[3];
        for (int k = 0; k < 3; k++)
        {
            local_state = local_state * 747796405u + 2891336453u;
            uint32_t result = ((local_state >> ((local_state >> 28u) + 4u)) ^ local_state) * 277803737u;
            result = (result >> 22u) ^ result;
            c[k] = (float)result / 4294967295.0f;
        }
        return {c[0], c[1], c[2]};
    }

    // ---- pathtrace_falsecolor_main (:296-452), one invocation ----
    bool pathtrace_falsecolor_main(uint32_t gx, uint32_t gy, uint32_t dim_x, uint32_t dim_y, const uint16_t *prev_frame, float out_rgb[3])
    {
        init_rng(gy * dim_x + gx);
        vec3f color = v3(0.0f);
        for (uint32_t sample = 0; sample < SAMPLES_PER_PIXEL; sample++)
        {
            vec2f ro = random_vec2f();
            vec2f pixel_offset = {ro.x - 0.5f, ro.y - 0.5f};
            Ray camera_ray = compute_camera_ray(gx, gy, dim_x, dim_y, pixel_offset);
            const uint32_t t = constants.falsecolor_type;
            if (t <= 6)   // ALBEDO .. METALLIC use the alpha-skipping query
            {
                HitInfo hit = ray_skip_alpha_stochastically(camera_ray);
                if (hit.hit)
                {
                    if (t == 0) color += get_material_point(hit).color;
                    else if (t == 1) color += compute_shading_normal(hit);
                    else if (t == 2) color += compute_shading_normal(hit) * 0.5f + 0.5f;
                    else if (t == 3) color += v3(hit.hit_backside ? 0.0f : 1.0f);
                    else if (t == 4) color += get_material_point(hit).emission;
                    else if (t == 5) color += v3(get_material_point(hit).roughness);
                    else color += v3(get_material_point(hit).metallic);
                }
            }
            else if (t <= 11)   // OPACITY .. TRI use the plain closest hit
            {
                HitInfo hit = ray_scene_intersection(camera_ray);
                if (hit.hit)
                {
                    if (t == 7) color += v3(get_material_point(hit).opacity);
                    else if (t == 8) color += hash_color(s->instances[hit.instance_idx].mat_idx);
                    else if (t == 9) color += v3(is_mat_delta(get_material_point(hit)) ? 1.0f : 0.0f);
                    else if (t == 10) color += hash_color(hit.instance_idx);
                    else color += hash_color(hit.tri_idx);
                }
            }
        }
        color = color / (float)SAMPLES_PER_PIXEL;
        color = max3(color, v3(0.0f));
        bool in_bounds = gx < dim_x && gy < dim_y;
        if (constants.accum_counter != 0 && in_bounds)
        {
            float weight = 1.0f / (float)constants.accum_counter;
            const uint16_t *p = prev_frame + ((size_t)gy * dim_x + gx) * 4;
            vec3f prev_color = {half_to_float(p[0]), half_to_float(p[1]), half_to_float(p[2])};
            color = prev_color * (1.0f - weight) + color * weight;
            color = max3(color, v3(0.0f));
        }
        out_rgb[0] = color.x; out_rgb[1] = color.y; out_rgb[2] = color.z;
        return in_bounds;
    }

    // ---- pathtrace_main (:220-292), one invocation ----
    // gx, gy already include constants.id_offset. Returns false when the texel is out of bounds
    // (the invocation still runs in the reference; its result is discarded).
    // get_heatmap_color (:2806-2872)
    static vec3f get_heatmap_color(float val, float min_, float max_)
    {
        const float wavelength = 380.0f + 370.0f * fmax_(val - min_, 0.0f) / fmax_(max_ - min_, 0.0f);
        vec3f color = v3(0.0f);
        if (wavelength <= 380.0f) { color.x = 0.0f; color.y = 0.0f; color.z = 0.0f; }
        else if (wavelength > 380.0f && wavelength <= 440.0f)
        {
            color.x = -(wavelength - 440.0f) / (440.0f - 380.0f) / 3.0f;
            color.y = 0.0f;
            color.z = 0.8f;
        }
        else if (wavelength >= 440.0f && wavelength <= 490.0f)
        {
            color.x = 0.0f;
            color.y = (wavelength - 440.0f) / (490.0f - 440.0f);
            color.z = 1.0f;
        }
        else if (wavelength >= 490.0f && wavelength <= 510.0f)
        {
            color.x = 0.0f;
            color.y = 1.0f;
            color.z = -(wavelength - 510.0f) / (510.0f - 490.0f);
        }
        else if (wavelength >= 510.0f && wavelength <= 580.0f)
        {
            color.x = (wavelength - 510.0f) / (580.0f - 510.0f);
            color.y = 1.0f;
            color.z = 0.0f;
        }
        else if (wavelength >= 580.0f && wavelength <= 645.0f)
        {
            color.x = 1.0f;
            color.y = -(wavelength - 645.0f) / (645.0f - 580.0f);
            color.z = 0.0f;
        }
        else if (wavelength >= 645.0f && wavelength <= 780.0f) { color.x = 1.0f; color.y = 0.0f; color.z = 0.0f; }
        else color = v3(1.0f);

        // Gamma correct.
        const float gamma = 0.8f;
        float factor = 1.0f;
        const vec3f white = v3(1.0f);
        if (wavelength >= 380.0f && wavelength < 420.0f) factor = 0.3f + 0.7f * (wavelength - 380.0f) / (float)(420 - 380);
        else if (wavelength >= 420.0f && wavelength < 701.0f) factor = 1.0f;
        else if (wavelength >= 701.0f && wavelength < 781.0f)
        {
            factor = 0.3f + 0.7f * (780.0f - wavelength) / (float)(780 - 700);
            vec3f b = color + white * factor;
            return {lpm_powf(b.x, gamma), lpm_powf(b.y, gamma), lpm_powf(b.z, gamma)};
        }
        else factor = 1.0f;
        vec3f b = color * factor;
        return {lpm_powf(b.x, gamma), lpm_powf(b.y, gamma), lpm_powf(b.z, gamma)};
    }

    // ---- pathtrace_debug_main (:457-503), one invocation (DEBUG = true) ----
    bool pathtrace_debug_main(uint32_t gx, uint32_t gy, uint32_t dim_x, uint32_t dim_y, const uint16_t *prev_frame, float out_rgb[3])
    {
        init_rng(gy * dim_x + gx);
        vec2f ro = random_vec2f();
        vec2f pixel_offset = {ro.x - 0.5f, ro.y - 0.5f};
        Ray camera_ray = compute_camera_ray(gx, gy, dim_x, dim_y, pixel_offset);

        const bool first_hit_only = (constants.flags & LUPIN_FLAG_DEBUG_FIRST_HIT_ONLY) != 0;
        const bool debug_num_bounces = (constants.flags & LUPIN_FLAG_DEBUG_NUM_BOUNCES) != 0;
        if (first_hit_only && !debug_num_bounces) ray_scene_intersection(camera_ray);
        else pathtrace_standard(camera_ray);

        // RAY_DEBUG_INFO (bvh_custom.wgsl:54,228,243): every box pair / triangle tested on behalf of this invocation
        uint64_t num_tri_checks = 0, num_aabb_checks = 0;
        for (int k = 0; k < 3; k++) { num_tri_checks += n.tri_tests[k]; num_aabb_checks += n.tlas_aabb[k] + n.blas_aabb[k]; }
        float val = 0.0f;
        if ((constants.flags & LUPIN_FLAG_DEBUG_TRI_CHECKS) != 0) val = (float)(uint32_t)num_tri_checks;
        else if ((constants.flags & LUPIN_FLAG_DEBUG_AABB_CHECKS) != 0) val = (float)(uint32_t)num_aabb_checks;
        else if (debug_num_bounces) val = (float)(uint32_t)n.debug_num_bounces;

        vec3f color = get_heatmap_color(val, constants.heatmap_min, constants.heatmap_max);
        bool in_bounds = gx < dim_x && gy < dim_y;
        if (constants.accum_counter != 0 && in_bounds)
        {
            float weight = 1.0f / (float)constants.accum_counter;
            const uint16_t *p = prev_frame + ((size_t)gy * dim_x + gx) * 4;
            vec3f prev_color = {half_to_float(p[0]), half_to_float(p[1]), half_to_float(p[2])};
            color = prev_color * (1.0f - weight) + color * weight;
            color = max3(color, v3(0.0f));
        }
        out_rgb[0] = color.x; out_rgb[1] = color.y; out_rgb[2] = color.z;
        return in_bounds;
    }

    // prev_f32 (optional, W*H*3): the f32-accumulate mode of SURVEY 8d -- the same blend with an unquantised prev_frame
    const float *prev_f32 = nullptr;
    bool pathtrace_main(uint32_t gx, uint32_t gy, uint32_t dim_x, uint32_t dim_y, const uint16_t *prev_frame, float out_rgb[3])
    {
        init_rng(gy * dim_x + gx);
        vec3f color = v3(0.0f);
        for (uint32_t sample = 0; sample < SAMPLES_PER_PIXEL; sample++)
        {
            vec2f ro = random_vec2f();
            vec2f pixel_offset = {ro.x - 0.5f, ro.y - 0.5f};
            Ray camera_ray = compute_camera_ray(gx, gy, dim_x, dim_y, pixel_offset);
            n.paths++;
            vec3f r;
            switch (constants.pathtrace_type)
            {
                case LUPIN_PATHTRACE_STANDARD: r = pathtrace_standard(camera_ray); break;
                case LUPIN_PATHTRACE_MIS: r = pathtrace_mis(camera_ray); break;
                case LUPIN_PATHTRACE_NAIVE: r = pathtrace_naive(camera_ray); break;
                case LUPIN_PATHTRACE_DIRECT: r = pathtrace_direct(camera_ray); break;
                default: r = v3(0.0f); break;  // the WGSL switch's default leaves color = 0
            }
            color += clamp_radiance(r);
        }
        color = color / (float)SAMPLES_PER_PIXEL;
        color = max3(color, v3(0.0f));

        bool in_bounds = gx < dim_x && gy < dim_y;
        if (constants.accum_counter != 0 && in_bounds)
        {
            float weight = 1.0f / (float)constants.accum_counter;
            const uint16_t *p = prev_frame + ((size_t)gy * dim_x + gx) * 4;
            vec3f prev_color = {half_to_float(p[0]), half_to_float(p[1]), half_to_float(p[2])};
            if (prev_f32) { const float *q = prev_f32 + ((size_t)gy * dim_x + gx) * 3; prev_color = {q[0], q[1], q[2]}; }
            color = prev_color * (1.0f - weight) + color * weight;
            color = max3(color, v3(0.0f));
        }
        out_rgb[0] = color.x; out_rgb[1] = color.y; out_rgb[2] = color.z;
        return in_bounds;
    }
};

void accumulate(Counters &a, const Counters &b)
{
    a.path_bounces += b.path_bounces; a.paths += b.paths;
    for (int k = 0; k < 3; k++)
    {
        a.tlas_aabb[k] += b.tlas_aabb[k]; a.instances_entered[k] += b.instances_entered[k];
        a.blas_aabb[k] += b.blas_aabb[k]; a.tri_tests[k] += b.tri_tests[k];
    }
    a.surface_hits += b.surface_hits;
    a.normal_fetches += b.normal_fetches; a.uv_fetches += b.uv_fetches; a.color_fetches += b.color_fetches;
    a.material_points += b.material_points; a.tex_ldr += b.tex_ldr; a.tex_hdr += b.tex_hdr;
    a.light_mesh += b.light_mesh; a.light_env += b.light_env;
    a.closest_hit_queries += b.closest_hit_queries; a.light_pdf_queries += b.light_pdf_queries;
}

}  // namespace

extern "C" {

// Counter record returned to the harness (SURVEY 8d work accounting).
struct OracleCounters
{
    uint64_t path_bounces, paths;
    uint64_t tlas_aabb[3], instances_entered[3], blas_aabb[3], tri_tests[3];   // [extend, light-pdf, shadow]
    uint64_t material_points, tex_ldr, tex_hdr, light_mesh, light_env, closest_hit_queries, light_pdf_queries, surface_hits;
    uint64_t normal_fetches, uv_fetches, color_fetches;
};

// One dispatch of pathtrace_main (pathtracer.wgsl:220-292) over `groups_x` x `groups_y`
// 4x4-pixel workgroups starting at constants.id_offset, exactly as renderer.rs:807-838 issues it.
// prev_frame / out_rgba16f: W*H*4 half floats (row 0 = top); prev_frame may be NULL when
// accum_counter == 0.  out_rgb_f32 (optional, W*H*3) receives the unquantised colour.
// store_rounding: 0 = toward zero (what the reference's goldens show), 1 = nearest even.
// falsecolor == 1 runs pathtrace_falsecolor_main (pathtracer.wgsl:296-452) with constants->falsecolor_type instead,
// falsecolor == 2 pathtrace_debug_main (:457-503) with the DEBUG flags / heatmap range of the push constants.
// Texels outside the dispatch are left untouched.  Returns 0 on success.
int oracle_pathtrace_f32prev(const LupinSceneDesc *scene, const LupinPushConstants *constants,
                             uint32_t max_bounces, uint32_t samples_per_pixel,
                             uint32_t width, uint32_t height, uint32_t groups_x, uint32_t groups_y,
                             const uint16_t *prev_frame, uint16_t *out_rgba16f, float *out_rgb_f32,
                             OracleCounters *counters, int num_threads, int store_rounding, int falsecolor, const float *prev_rgb_f32);
int oracle_pathtrace(const LupinSceneDesc *scene, const LupinPushConstants *constants,
                     uint32_t max_bounces, uint32_t samples_per_pixel,
                     uint32_t width, uint32_t height, uint32_t groups_x, uint32_t groups_y,
                     const uint16_t *prev_frame, uint16_t *out_rgba16f, float *out_rgb_f32,
                     OracleCounters *counters, int num_threads, int store_rounding, int falsecolor)
{
    return oracle_pathtrace_f32prev(scene, constants, max_bounces, samples_per_pixel, width, height, groups_x, groups_y, prev_frame, out_rgba16f,
                                    out_rgb_f32, counters, num_threads, store_rounding, falsecolor, nullptr);
}
// the same with an optional unquantised prev_frame (W*H*3 f32): f32-accumulate mode; prev_frame (f16) may then be NULL
int oracle_pathtrace_f32prev(const LupinSceneDesc *scene, const LupinPushConstants *constants,
                             uint32_t max_bounces, uint32_t samples_per_pixel,
                             uint32_t width, uint32_t height, uint32_t groups_x, uint32_t groups_y,
                             const uint16_t *prev_frame, uint16_t *out_rgba16f, float *out_rgb_f32,
                             OracleCounters *counters, int num_threads, int store_rounding, int falsecolor, const float *prev_rgb_f32)
{
    auto to_half = [store_rounding](float f) { return store_rounding == 1 ? float_to_half_rne(f) : float_to_half_rtz(f); };
    if (!scene || !constants || !out_rgba16f) return -1;
    if (constants->accum_counter != 0 && !prev_frame && !prev_rgb_f32) return -1;
    std::vector<uint16_t> zero_prev;
    if (!prev_frame && prev_rgb_f32) { zero_prev.assign((size_t)width * height * 4, 0); prev_frame = zero_prev.data(); }
    Counters total;
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
    const uint32_t px_y = groups_y * LUPIN_WORKGROUP_SIZE, px_x = groups_x * LUPIN_WORKGROUP_SIZE;
    #pragma omp parallel
    {
        Counters local;
        #pragma omp for schedule(dynamic, 1)
        for (int64_t ly = 0; ly < (int64_t)px_y; ly++)
        {
            for (uint32_t lx = 0; lx < px_x; lx++)
            {
                uint32_t gx = lx + constants->id_offset[0];
                uint32_t gy = (uint32_t)ly + constants->id_offset[1];
                if (!(gx < width && gy < height)) continue;  // result would be discarded (:287)
                Inv inv;
                inv.s = scene; inv.constants = *constants;
                inv.MAX_BOUNCES = max_bounces; inv.SAMPLES_PER_PIXEL = samples_per_pixel;
                inv.prev_f32 = (falsecolor == 0) ? prev_rgb_f32 : nullptr;
                float rgb[3];
                if (falsecolor == 2) inv.pathtrace_debug_main(gx, gy, width, height, prev_frame, rgb);
                else if (falsecolor) inv.pathtrace_falsecolor_main(gx, gy, width, height, prev_frame, rgb);
                else inv.pathtrace_main(gx, gy, width, height, prev_frame, rgb);
                size_t o = (size_t)gy * width + gx;
                out_rgba16f[o * 4 + 0] = to_half(rgb[0]);
                out_rgba16f[o * 4 + 1] = to_half(rgb[1]);
                out_rgba16f[o * 4 + 2] = to_half(rgb[2]);
                out_rgba16f[o * 4 + 3] = 0x3C00;  // 1.0
                if (out_rgb_f32) { out_rgb_f32[o * 3 + 0] = rgb[0]; out_rgb_f32[o * 3 + 1] = rgb[1]; out_rgb_f32[o * 3 + 2] = rgb[2]; }
                accumulate(local, inv.n);
            }
        }
        #pragma omp critical
        accumulate(total, local);
    }
    if (counters)
    {
        counters->path_bounces = total.path_bounces; counters->paths = total.paths;
        for (int k = 0; k < 3; k++)
        {
            counters->tlas_aabb[k] = total.tlas_aabb[k]; counters->instances_entered[k] = total.instances_entered[k];
            counters->blas_aabb[k] = total.blas_aabb[k]; counters->tri_tests[k] = total.tri_tests[k];
        }
        counters->surface_hits = total.surface_hits;
        counters->normal_fetches = total.normal_fetches; counters->uv_fetches = total.uv_fetches; counters->color_fetches = total.color_fetches;
        counters->material_points = total.material_points; counters->tex_ldr = total.tex_ldr; counters->tex_hdr = total.tex_hdr;
        counters->light_mesh = total.light_mesh; counters->light_env = total.light_env;
        counters->closest_hit_queries = total.closest_hit_queries; counters->light_pdf_queries = total.light_pdf_queries;
    }
    return 0;
}

// ray_scene_intersection (bvh_custom.wgsl:7-110) over a batch of rays.
int oracle_trace_rays(const LupinSceneDesc *scene, uint32_t n, const float *ori_xyz, const float *dir_xyz,
                      float ray_epsilon, uint32_t flags, uint32_t *out_hit, float *out_dst, float *out_uv,
                      uint32_t *out_instance, uint32_t *out_tri)
{
    if (!scene) return -1;
    #pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < (int64_t)n; i++)
    {
        Inv inv;
        inv.s = scene; memset(&inv.constants, 0, sizeof(inv.constants));
        inv.constants.ray_epsilon = ray_epsilon; inv.constants.flags = flags;
        inv.MAX_BOUNCES = 0; inv.SAMPLES_PER_PIXEL = 0;
        Ray r;
        r.ori = {ori_xyz[i * 3 + 0], ori_xyz[i * 3 + 1], ori_xyz[i * 3 + 2]};
        r.dir = {dir_xyz[i * 3 + 0], dir_xyz[i * 3 + 1], dir_xyz[i * 3 + 2]};
        r.inv_dir = 1.0f / r.dir;
        HitInfo h = inv.ray_scene_intersection(r);
        out_hit[i] = h.hit ? 1u : 0u;
        out_dst[i] = h.dst; out_uv[i * 2 + 0] = h.uv.x; out_uv[i * 2 + 1] = h.uv.y;
        out_instance[i] = h.instance_idx; out_tri[i] = h.tri_idx;
    }
    return 0;
}

// RNG stream probe: first `count` random_f32() outputs for (pixel linear index, accum_counter)
// (pathtracer.wgsl:1563-1600).
void oracle_rng_stream(uint32_t global_id, uint32_t accum_counter, uint32_t count, float *out)
{
    Inv inv; inv.s = nullptr; memset(&inv.constants, 0, sizeof(inv.constants));
    inv.constants.accum_counter = accum_counter;
    inv.init_rng(global_id);
    for (uint32_t i = 0; i < count; i++) out[i] = inv.random_f32();
}

// BSDF probes for fixture minting: sample / eval / pdf triples of the non-delta and delta lobes
// for an explicit MaterialPoint (mat_type, color, roughness (already squared), metallic, ior).
void oracle_bsdf_probe(uint32_t mat_type, const float color[3], float roughness, float metallic, float ior,
                       const float normal[3], const float outgoing[3], float rnl, const float rn[2],
                       float out_incoming[3], float out_eval[3], float *out_pdf)
{
    MaterialPoint m;
    m.mat_type = mat_type; m.color = {color[0], color[1], color[2]};
    m.roughness = roughness; m.metallic = metallic; m.ior = ior;
    vec3f nrm = {normal[0], normal[1], normal[2]}, o = {outgoing[0], outgoing[1], outgoing[2]};
    vec3f inc, ev; float pdf;
    if (!Inv::is_mat_delta(m)) {
        inc = Inv::sample_bsdfcos(m, nrm, o, rnl, {rn[0], rn[1]});
        ev = Inv::eval_bsdfcos(m, nrm, o, inc);
        pdf = Inv::sample_bsdfcos_pdf(m, nrm, o, inc);
    } else {
        inc = Inv::sample_delta(m, nrm, o, rnl);
        ev = Inv::eval_delta(m, nrm, o, inc);
        pdf = Inv::sample_delta_pdf(m, nrm, o, inc);
    }
    out_incoming[0] = inc.x; out_incoming[1] = inc.y; out_incoming[2] = inc.z;
    out_eval[0] = ev.x; out_eval[1] = ev.y; out_eval[2] = ev.z;
    *out_pdf = pdf;
}

// tonemap_and_fit_aspect (tonemapping.rs:155-224 + tonemapping.wgsl) evaluated at target pixel centres: viewport /
// scissor (:163-167,:216-217), aspect-fit scale of the quad (:168-174), vertex positions pos * scale with tex coords
// 0..1 (wgsl:24-48), fragment: clamp-to-edge linear sample, max(.,0), * exp2(exposure), tonemap_filmic,
// linear_to_srgb (wgsl:51-79), Rgba8Unorm store.  viewport = NULL means the whole target.
int oracle_tonemap(const uint16_t *src_rgba16f, uint32_t src_w, uint32_t src_h, uint8_t *dst_rgba8, uint32_t dst_w, uint32_t dst_h,
                   const float *viewport_xywh, float exposure, int filmic, int srgb, int clear)
{
    if (!src_rgba16f || !dst_rgba8 || !src_w || !src_h || !dst_w || !dst_h) return -1;
    float vx = 0.0f, vy = 0.0f, vw = (float)dst_w, vh = (float)dst_h;
    if (viewport_xywh) { vx = viewport_xywh[0]; vy = viewport_xywh[1]; vw = viewport_xywh[2]; vh = viewport_xywh[3]; }
    const float src_aspect = (float)src_w / (float)src_h;
    const float dst_aspect = vw / vh;
    float scale_x, scale_y;
    if (src_aspect > dst_aspect) { scale_x = 1.0f; scale_y = dst_aspect / src_aspect; }
    else { scale_x = src_aspect / dst_aspect; scale_y = 1.0f; }
    if (clear)
        for (size_t i = 0; i < (size_t)dst_w * dst_h; i++) { dst_rgba8[i * 4 + 0] = 0; dst_rgba8[i * 4 + 1] = 0; dst_rgba8[i * 4 + 2] = 0; dst_rgba8[i * 4 + 3] = 255; }
    const uint32_t x0 = std::min((uint32_t)vx, dst_w), y0 = std::min((uint32_t)vy, dst_h);
    const uint32_t x1 = (uint32_t)std::min<uint64_t>((uint64_t)x0 + (uint32_t)vw, dst_w), y1 = (uint32_t)std::min<uint64_t>((uint64_t)y0 + (uint32_t)vh, dst_h);
    auto texel = [&](int x, int y, int c) { return half_to_float(src_rgba16f[((size_t)y * src_w + x) * 4 + c]); };
    auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    const float gain = lpm_powf(2.0f, exposure);
    for (uint32_t y = y0; y < y1; y++)
        for (uint32_t x = x0; x < x1; x++)
        {
            const float fx = ((float)x + 0.5f - vx) / vw, fy = ((float)y + 0.5f - vy) / vh;
            const float nx = 2.0f * fx - 1.0f, ny = 1.0f - 2.0f * fy;
            if (!(fabsf(nx) <= scale_x && fabsf(ny) <= scale_y)) continue;
            const float u = (nx / scale_x + 1.0f) * 0.5f, v = (1.0f - ny / scale_y) * 0.5f;
            const float sx = u * (float)src_w - 0.5f, sy = v * (float)src_h - 0.5f;
            const float x0f = floorf(sx), y0f = floorf(sy);
            const float tx = sx - x0f, ty = sy - y0f;
            const int xa = clampi(f2i(x0f), 0, (int)src_w - 1), xb = clampi(f2i(x0f) + 1, 0, (int)src_w - 1);
            const int ya = clampi(f2i(y0f), 0, (int)src_h - 1), yb = clampi(f2i(y0f) + 1, 0, (int)src_h - 1);
            const float gx = 1.0f - tx, gy = 1.0f - ty;
            uint8_t *o = dst_rgba8 + ((size_t)y * dst_w + x) * 4;
            for (int c = 0; c < 3; c++)
            {
                float col = (texel(xa, ya, c) * gx + texel(xb, ya, c) * tx) * gy + (texel(xa, yb, c) * gx + texel(xb, yb, c) * tx) * ty;
                col = fmax_(col, 0.0f);
                if (exposure != 0.0f) col *= gain;
                if (filmic)
                {
                    const float hdr = col * 0.6f;
                    const float ldr = (hdr * hdr * 2.51f + hdr * 0.03f) / (hdr * hdr * 2.43f + hdr * 0.59f + 0.14f);
                    col = fmax_(ldr, 0.0f);
                }
                if (srgb)
                {
                    const float cutoff = col <= 0.0031308f ? 1.0f : 0.0f;
                    const float higher = 1.055f * lpm_powf(col, 1.0f / 2.4f) - 0.055f;
                    const float lower = col * 12.92f;
                    col = higher * (1.0f - cutoff) + lower * cutoff;
                }
                float q = clamp_(col, 0.0f, 1.0f) * 255.0f;
                if (!(q == q)) q = 0.0f;
                o[c] = (uint8_t)rintf(q);
            }
            o[3] = 255;
        }
    return 0;
}

uint16_t oracle_float_to_half(float f) { return float_to_half_rne(f); }
uint16_t oracle_float_to_half_rtz(float f) { return float_to_half_rtz(f); }
float oracle_half_to_float(uint16_t h) { return half_to_float(h); }
int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
