// lupin_hip.hip -- wavefront path-tracing kernels for gfx950 + the C ABI of include/lupin_hip.h.
//
// What the reference runs as ONE megakernel invocation per pixel (`pathtrace_main`,
// pathtracer.wgsl:220-292) is split here into stages over compacted queues of live paths:
//
//   k_begin      RNG seeding, first camera ray of every pixel of the dispatch      (:224-237, :505-542)
//   k_extend     closest hit with stochastic alpha skipping                       (bvh_custom.wgsl:154-180)
//   k_shade      the rest of one integrator-loop iteration: medium sampling, material fetch,
//                emission, BSDF / light sampling + pdfs, volume stack, Russian roulette, and --
//                when a path ends -- clamp_radiance, next camera sample of the same pixel
//                (the per-pixel RNG stream continues across samples, :234-239), or retirement
//   k_resolve    /spp, progressive blend with prev_frame, Rgba16Float store          (:275-289)
//
// One thread owns one pixel for the whole call, so radiance accumulation needs no atomics; the
// only atomics are the per-wave queue appends (ballot + one atomicAdd per wave).
//
// THERE IS NO CPU FALLBACK: without a HIP device every entry point that needs one fails with
// LUPIN_ERR_NO_DEVICE.

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

#include "lupin_device.hpp"

using namespace lpd;

// ------------------------------------------------------------------------------------------------
// Path state (SoA over the pixels of one dispatch; slot = pixel of the dispatch region)
// ------------------------------------------------------------------------------------------------

// meta word: bounce [0,12) | flags [12,16) | sample [16,32)
constexpr uint32_t META_BOUNCE_MASK = 0xFFFu;
constexpr uint32_t META_VOLUME = 1u << 12;         // volume_stack_len == 1
constexpr uint32_t META_NEXT_EMISSION = 1u << 13;  // MIS / Direct `next_emission`
constexpr uint32_t META_TERMINATED = 1u << 14;     // MIS / Direct: the path ended in k_shade; k_shadow folds it after its shadow rays
constexpr uint32_t META_SAMPLE_SHIFT = 16;
constexpr uint32_t HIT_MISS = 0xFFFFFFFFu;

struct PathBuffers
{
    float4 *ori_rng;    // ori.xyz | rng state
    float4 *dir_meta;   // dir.xyz | meta
    float4 *weight;     // weight.xyz
    float4 *radiance;   // radiance.xyz
    float4 *color;      // per-pixel sum over samples
    float4 *hit;        // dst u v | instance (HIT_MISS = no hit)
    uint32_t *hit_tri;  // global triangle
    float4 *vol0;       // medium density.xyz | anisotropy
    float4 *vol1;       // medium scattering.xyz
    float4 *next_hit;   // MIS: hit of the BSDF-sampled shadow ray, reused as next vertex
    uint32_t *next_tri;
    // MIS / Direct shadow rays, recorded by k_shade and traced by k_shadow (radiance += factor * emission (*|/) scalar)
    float4 *sh_org;     // origin.xyz | flags (bit0: ray 0 valid, bit1: ray 1 valid)
    float4 *sh_d0;      // ray 0 direction | scalar 0
    float4 *sh_f0;      // ray 0 factor = weight * bsdfcos
    float4 *sh_d1;      // ray 1 direction | scalar 1
    float4 *sh_f1;      // ray 1 factor (.w: triangle of the pre-traced hit, see sh_hit1)
    float4 *sh_hit1;    // large scenes trace the shadow rays in the persistent kernel: hit record of ray 1 (ray 0 -> next_hit)
    // Live-path queues, sharded: one global counter per iteration would serialise every wave's append on a
    // single L2 atomic (~88 per microsecond chip-wide -- measured: 186 us per 1M-path iteration, more than the
    // shading itself).  Each of LP_SHARDS shards owns a fixed segment of the queue and its own counter; block b
    // always reads and appends shard b % LP_SHARDS, so a shard never grows beyond its initial size.
    uint32_t *queue[2];   // [parity][shard * shard_cap + i]
    uint32_t *counts;     // counts[k * LP_SHARDS + s] = live paths of shard s entering iteration k
    uint32_t shard_cap;   // slots per shard (multiple of LP_BLOCK)
};

#ifndef LP_NUM_SHARDS
#define LP_NUM_SHARDS 256
#endif
constexpr uint32_t LP_SHARDS = LP_NUM_SHARDS;

struct FrameParams
{
    LupinPushConstants pc;
    uint32_t width, height;      // full image (RNG seeding uses the full width, :226)
    uint32_t off_x, off_y;       // dispatch origin (id_offset)
    uint32_t reg_w, reg_h;       // in-bounds pixels of the dispatch
    uint32_t max_bounces, spp;
    // tile-set dispatch (multi-GPU sharding): slot -> pixel goes through the list of owned tiles
    uint32_t store_rne;          // f32 -> f16 store rounding: 0 = toward zero (reference goldens), 1 = nearest even
    uint32_t tile_px;            // 0 = rectangular dispatch
    uint32_t tiles_x, rank, world;
};

// ------------------------------------------------------------------------------------------------
// Camera (compute_camera_ray, pathtracer.wgsl:505-542) -- draws 2 (jitter) + 2 (lens) numbers
// ------------------------------------------------------------------------------------------------

LP_FN void camera_ray(const FrameParams &fp, uint32_t gx, uint32_t gy, uint32_t &rng, f3 &ori, f3 &dir)
{
    const LupinPushConstants &pc = fp.pc;
    float j0 = rnd(rng), j1 = rnd(rng);
    float offx = j0 - 0.5f, offy = j1 - 0.5f;
    float resx = (float)fp.width, resy = (float)fp.height;
    float pcx = (float)gx + 0.5f, pcy = (resy - (float)gy) + 0.5f;
    float uvx = (pcx + offx) / resx, uvy = (pcy + offy) / resy;

    float lens = pc.camera_lens, film = pc.camera_film, aspect = pc.camera_aspect;
    float focus = pc.camera_focus, aperture = pc.camera_aperture;
    float fsx = (aspect >= 1.0f) ? film : film * aspect;
    float fsy = (aspect >= 1.0f) ? film / aspect : film;
    // random_in_disk (:1623-1629)
    float d0 = rnd(rng), d1 = rnd(rng);
    float dr = sqrtf(d1);
    float ds, dc;
    lpm_sincosf(2.0f * LP_PI * d0, &ds, &dc);
    float lux = dc * dr, luy = ds * dr;

    f3 e, d;
    if (pc.flags & LUPIN_FLAG_CAMERA_ORTHO)
    {
        float sc = 1.0f / lens;
        f3 q = mk3(fsx * (0.5f - uvx) * sc, fsy * (0.5f - uvy) * sc, lens);
        e = add(mk3(-q.x, -q.y, 0.0f), mk3(lux * aperture / 2.0f, luy * aperture / 2.0f, 0.0f));
        f3 p = mk3(-q.x, -q.y, -focus);
        d = mul(normalize3(sub(p, e)), mk3(1.0f, 1.0f, -1.0f));
    }
    else
    {
        f3 q = mk3(fsx * (0.5f - uvx), fsy * (0.5f - uvy), lens);
        f3 look_at = neg(normalize3(q));
        e = mk3(lux * (aperture / 2.0f), luy * (aperture / 2.0f), 0.0f);
        f3 focus_point = divs(scale(look_at, focus), fabsf(look_at.z));
        d = mul(normalize3(sub(focus_point, e)), mk3(1.0f, 1.0f, -1.0f));
    }
    // transform_ray by camera_transform (:2662-2669): point without w-divide, direction normalised
    const float (*m)[4] = pc.camera_transform.m;
    ori = mk3(m[0][0] * e.x + m[1][0] * e.y + m[2][0] * e.z + m[3][0] * 1.0f,
              m[0][1] * e.x + m[1][1] * e.y + m[2][1] * e.z + m[3][1] * 1.0f,
              m[0][2] * e.x + m[1][2] * e.y + m[2][2] * e.z + m[3][2] * 1.0f);
    dir = normalize3(mk3(m[0][0] * d.x + m[1][0] * d.y + m[2][0] * d.z + m[3][0] * 0.0f,
                         m[0][1] * d.x + m[1][1] * d.y + m[2][1] * d.z + m[3][1] * 0.0f,
                         m[0][2] * d.x + m[1][2] * d.y + m[2][2] * d.z + m[3][2] * 0.0f));
}

__device__ __forceinline__ void slot_to_pixel(const FrameParams &fp, uint32_t slot, uint32_t &gx, uint32_t &gy)
{
    if (fp.tile_px)
    {
        const uint32_t per_tile = fp.tile_px * fp.tile_px;
        const uint32_t t = fp.rank + (slot / per_tile) * fp.world;
        const uint32_t r = slot % per_tile;
        gx = (t % fp.tiles_x) * fp.tile_px + r % fp.tile_px;
        gy = (t / fp.tiles_x) * fp.tile_px + r / fp.tile_px;
        return;
    }
    gx = fp.off_x + slot % fp.reg_w;
    gy = fp.off_y + slot / fp.reg_w;
}

// append the lanes with `alive` to a queue: one atomic per wave
__device__ __forceinline__ void queue_append(bool alive, uint32_t slot, uint32_t *queue, uint32_t *counter)
{
    const unsigned long long mask = __ballot(alive);
    if (mask)
    {
        const int lane = threadIdx.x & 63;
        const int leader = __ffsll((long long)mask) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
        base = __shfl(base, leader);
        if (alive) queue[base + __popcll(mask & ((1ull << lane) - 1ull))] = slot;
    }
}

// ------------------------------------------------------------------------------------------------
// Stage kernels
// ------------------------------------------------------------------------------------------------

// publishes one call's parameters to the lane's device copy (kernel arguments are captured at launch, so the host
// struct may go out of scope)
__global__ void k_set_params(FrameParams fp, FrameParams *dst) { *dst = fp; }

__global__ void __launch_bounds__(LP_BLOCK) k_begin(const FrameParams *__restrict__ fpp, PathBuffers pb, uint32_t n)
{
    const FrameParams fp = *fpp;   // per-frame parameters live in device memory so that a captured graph can be replayed
    uint32_t slot = blockIdx.x * LP_BLOCK + threadIdx.x;
    uint32_t gx = 0, gy = 0;
    bool live = slot < n;
    if (live)
    {
        slot_to_pixel(fp, slot, gx, gy);
        live = gx < fp.width && gy < fp.height;   // edge tiles: texels outside the image are never stored (:287)
    }
    const uint32_t shard = blockIdx.x % LP_SHARDS;
    queue_append(live, slot, pb.queue[0] + (size_t)shard * pb.shard_cap, &pb.counts[shard]);
    if (!live) return;
    uint32_t rng = rng_seed_for(gy * fp.width + gx, fp.pc.accum_counter);
    f3 o, d;
    camera_ray(fp, gx, gy, rng, o, d);
    pb.ori_rng[slot] = make_float4(o.x, o.y, o.z, __uint_as_float(rng));
    pb.dir_meta[slot] = make_float4(d.x, d.y, d.z, __uint_as_float(META_NEXT_EMISSION));
    pb.weight[slot] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
    pb.radiance[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    pb.color[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    pb.next_hit[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(HIT_MISS));   // `var next_intersection = HitInfo()` (:746)
    pb.next_tri[slot] = 0u;
}

// ray_skip_alpha_stochastically (bvh_custom.wgsl:154-180).  The material is consulted only for
// instances whose opacity can differ from 1 (flag set at upload); for all others
// opacity == 1 exactly, so `opacity < 1` is false and no random number is drawn -- identical to
// the reference, which evaluates get_material_point for every hit.
#ifndef LP_EXTEND_WAVES
#define LP_EXTEND_WAVES 4
#endif
#ifndef LP_SHADE_WAVES
#define LP_SHADE_WAVES 3
#endif
#ifndef LP_SIMPLE_SHADE_WAVES
#define LP_SIMPLE_SHADE_WAVES 4
#endif
#ifndef LP_MIS_SHADE_WAVES
#define LP_MIS_SHADE_WAVES 2
#endif
// dynamic LDS layout of the stage kernels: [traversal stacks: stack_entries * LP_BLOCK words][geometry blob, if staged]
template <bool LDSGEO> struct GeoOf { typedef GeoGlobal type; };
template <> struct GeoOf<true> { typedef GeoLds type; };
template <bool LDSGEO>
__device__ __forceinline__ typename GeoOf<LDSGEO>::type make_geo(const SceneDev &sc, uint32_t *lds, uint32_t stack_words);
template <>
__device__ __forceinline__ GeoGlobal make_geo<false>(const SceneDev &sc, uint32_t *, uint32_t) { return geo_global(sc); }
template <>
__device__ __forceinline__ GeoLds make_geo<true>(const SceneDev &sc, uint32_t *lds, uint32_t stack_words)
{
    GeoLds g = geo_stage_lds(sc, lds + stack_words);
    __syncthreads();
    return g;
}

// One closest-hit query of the integrator loop, with stochastic alpha skipping (bvh_custom.wgsl:154-180).
// Returns the hit record (dst accumulated over skipped surfaces | u | v | instance or HIT_MISS) and the triangle.
template <typename Geo, bool OPAQUE = false>   // OPAQUE: no instance of the scene can have opacity != 1 (LupinScene::all_opaque)
__device__ __forceinline__ void trace_alpha(const Geo &geo, const SceneDev &sc, uint32_t *stack, f3 o, f3 d, uint32_t &rng, float eps,
                                            float4 &hitrec, uint32_t &hit_tri)
{
    float total = 0.0f;
    Closest c;
    c.t = LP_F32_MAX; c.u = 0.0f; c.v = 0.0f; c.tri = 0u; c.inst = HIT_MISS;
    bool hit = false;
    for (uint32_t k = 0; k < 128u; k++)   // MAX_OPACITY_BOUNCES (pathtracer.wgsl:1263)
    {
        c = scene_closest(geo, sc, stack, o, d, eps);
        hit = (c.t != LP_F32_MAX);
        if (!hit) break;
        total += c.t;
        if (OPAQUE || !(sc.instances[c.inst].flags & 1u)) break;
        Surface s = resolve_surface(sc, c.inst, c.tri, c.u, c.v);
        float opacity = surface_opacity(sc, s);
        if (opacity < 1.0f && rnd(rng) >= opacity) o = add(o, scale(d, c.t));
        else break;
    }
    hitrec = make_float4(total, c.u, c.v, __uint_as_float(hit ? c.inst : HIT_MISS));
    hit_tri = c.tri;
}

template <int TYPE, bool LDSGEO, bool OPAQUE>
__global__ void __attribute__((amdgpu_waves_per_eu(LP_EXTEND_WAVES, 8))) __launch_bounds__(LP_BLOCK) k_extend(SceneDev sc, const FrameParams *__restrict__ fpp, PathBuffers pb, uint32_t iter,
                                                     unsigned long long *shard_stats, uint32_t stack_words)
{
    const FrameParams fp = *fpp;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    const auto geo = make_geo<LDSGEO>(sc, lds_stack, stack_words);
    const uint32_t shard = blockIdx.x % LP_SHARDS;
    const uint32_t count = pb.counts[iter * LP_SHARDS + shard];
    const uint32_t i = (blockIdx.x / LP_SHARDS) * LP_BLOCK + threadIdx.x;
    if (i == 0 && count) shard_stats[shard * 2 + 0] += count;   // one writer per shard per launch: no atomic needed
    if (i >= count) return;
    const uint32_t slot = pb.queue[iter & 1][(size_t)shard * pb.shard_cap + i];

    float4 orr = pb.ori_rng[slot];
    float4 dm = pb.dir_meta[slot];
    if (TYPE == LUPIN_PATHTRACE_MIS)
    {
        if (!(__float_as_uint(dm.w) & META_NEXT_EMISSION))
        {
            pb.hit[slot] = pb.next_hit[slot];
            pb.hit_tri[slot] = pb.next_tri[slot];
            return;
        }
    }
    uint32_t rng = __float_as_uint(orr.w);
    const uint32_t rng_in = rng;
    float4 hitrec;
    uint32_t hit_tri;
    trace_alpha<typename GeoOf<LDSGEO>::type, OPAQUE>(geo, sc, lds_stack, mk3(orr.x, orr.y, orr.z), mk3(dm.x, dm.y, dm.z), rng, fp.pc.ray_epsilon, hitrec, hit_tri);
    pb.hit[slot] = hitrec;
    pb.hit_tri[slot] = hit_tri;
    if (rng != rng_in) pb.ori_rng[slot].w = __uint_as_float(rng);
}

// Persistent, phase-scheduled form of k_extend -- the default for scenes traversed from global memory.  Rays of one
// wave need very different numbers of traversal steps and sit in different phases of the traversal, so the
// one-ray-per-lane kernel keeps 11 % of the VALU lanes busy on the bistro-class scene (39 % on the Cornell box; PMC:
// SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64)).  Here each wave owns its lanes for the whole launch, refills empty
// lanes whenever at least `refill_min` of them are free, and executes per round the one phase most lanes wait for.
// Work is partitioned statically, so refilling needs no atomics: the grid holds `wps` waves per shard, and wave j of
// shard s owns the 64-entry chunks j, j + wps, j + 2 wps, ... of that shard's queue.  Every ray is still traced by
// exactly the same sequence of operations as in k_extend, only by a different lane.
#ifndef LP_REFILL_MIN
#define LP_REFILL_MIN 16
#endif

// MODE 0: the integrator's closest-hit queries (one per queue entry, stochastic alpha skipping).
// MODE 1: the shadow rays k_shade recorded for MIS / Direct (two jobs per queue entry, plain closest hit); their hits go
//         to next_hit / next_tri (MIS ray 0, which doubles as the next vertex) or sh_hit1 / sh_f1.w, and
//         k_shadow<.., PRETRACED> folds them into the radiance.
template <int TYPE, bool LDSGEO, int MODE>
__global__ void __launch_bounds__(LP_BLOCK) k_extend_persistent(SceneDev sc, const FrameParams *__restrict__ fpp, PathBuffers pb, uint32_t iter,
                                                                unsigned long long *shard_stats, uint32_t refill_min, uint32_t stack_words, uint32_t nsteps)
{
    const FrameParams fp = *fpp;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    const auto geo = make_geo<LDSGEO>(sc, lds_stack, stack_words);
    static_assert(LP_SHARDS <= LP_BLOCK && 256 % LP_SHARDS == 0, "block 0 books one shard per thread; 64-block grids hold whole waves per shard");
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t *counts = pb.counts + (size_t)iter * LP_SHARDS;
    const uint32_t *queue = pb.queue[iter & 1];
    if (MODE == 0 && blockIdx.x == 0 && tid < LP_SHARDS) { const uint32_t c = counts[tid]; if (c) shard_stats[tid * 2 + 0] += c; }   // one writer per shard per launch

    // this wave's share of the work
    const uint32_t wave = blockIdx.x * (LP_BLOCK / 64) + tid / 64;         // wave-uniform
    const uint32_t wps = (gridDim.x * (LP_BLOCK / 64)) / LP_SHARDS;         // waves per shard (grid is a multiple of 64 blocks)
    const uint32_t shard = wave % LP_SHARDS, j = wave / LP_SHARDS;
    const uint32_t cnt = counts[shard] * (MODE == 1 ? 2u : 1u);   // jobs
    const uint32_t full_chunks = cnt / 64u;
    uint32_t n_mine = (full_chunks > j) ? ((full_chunks - j - 1u) / wps + 1u) * 64u : 0u;
    if ((cnt % 64u) && (full_chunks % wps) == j) n_mine += cnt % 64u;       // the partial last chunk
    if (n_mine == 0) return;
    const size_t shard_base = (size_t)shard * pb.shard_cap;
    uint32_t next_pos = 0;                                                  // position in this wave's private sequence

    const float eps = fp.pc.ray_epsilon;
    constexpr uint32_t REF_DONE = 0xFFFFFFFFu;

    // per-lane ray + traversal state
    bool active = false;
    uint32_t slot = 0, rng = 0, rng_in = 0, alpha_k = 0, ray_k = 0;
    float total_dst = 0.0f;
    f3 o = splat(0.0f), d = splat(0.0f), inv_d = splat(0.0f);
    f3 co = o, cd = d, cinv = inv_d;
    uint32_t sp = 0, blas_base = 0xFFFFFFFFu, cur_inst = 0, cur = REF_DONE;
    Closest best;
    best.t = LP_F32_MAX; best.u = 0.0f; best.v = 0.0f; best.tri = 0u; best.inst = HIT_MISS;

    auto start_traversal = [&]() {
        inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        co = o; cd = d; cinv = inv_d;
        sp = 0; blas_base = 0xFFFFFFFFu;
        cur = sc.num_instances ? sc.tlas_root : REF_DONE;
        best.t = LP_F32_MAX; best.u = 0.0f; best.v = 0.0f; best.tri = 0u; best.inst = HIT_MISS;
    };
    auto pop = [&]() {
        if (sp == blas_base) { blas_base = 0xFFFFFFFFu; co = o; cd = d; cinv = inv_d; }
        if (sp == 0) { cur = REF_DONE; return; }
        sp--;
        cur = lds_stack[sp * LP_BLOCK + tid];
    };

    // Phase scheduling: a lane is at an internal node (N), a TLAS leaf = instance entry (I), a triangle of a BLAS leaf (T),
    // at the end of a traversal (F) or empty (E).  Every round the wave executes the ONE phase most of its lanes wait for
    // (wave-uniform branch), so each instruction runs with as many lanes as possible; lanes of other phases just wait.
    for (;;)
    {
        const bool isN = active && !(cur & REF_LEAF);
        const bool isF = active && cur == REF_DONE;
        const bool isLeaf = active && (cur & REF_LEAF) && cur != REF_DONE;
        const bool isI = isLeaf && blas_base == 0xFFFFFFFFu;
        const bool isT = isLeaf && !isI;
        const unsigned long long idle = __ballot(!active);
        const uint32_t cE = (uint32_t)__popcll(idle);
        const uint32_t cN = (uint32_t)__popcll(__ballot(isN)), cI = (uint32_t)__popcll(__ballot(isI));
        const uint32_t cT = (uint32_t)__popcll(__ballot(isT)), cF = (uint32_t)__popcll(__ballot(isF));

        if (cE >= refill_min && next_pos < n_mine)
        {
            // ---- refill empty lanes ----
            const uint32_t my_rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            const uint32_t take = min(cE, n_mine - next_pos);
            const bool got = !active && my_rank < take;
            const uint32_t pos = next_pos + my_rank;
            const size_t q_index = shard_base + (size_t)((pos / 64u) * wps + j) * 64u + pos % 64u;
            next_pos += take;
            if (got && MODE == 0)
            {
                slot = queue[q_index];
                const float4 orr = pb.ori_rng[slot];
                const float4 dm = pb.dir_meta[slot];
                bool trace = true;
                if (TYPE == LUPIN_PATHTRACE_MIS)
                {
                    if (!(__float_as_uint(dm.w) & META_NEXT_EMISSION))   // reuse the BSDF-sampled hit (pathtracer.wgsl:751-755)
                    {
                        pb.hit[slot] = pb.next_hit[slot];
                        pb.hit_tri[slot] = pb.next_tri[slot];
                        trace = false;
                    }
                }
                if (trace)
                {
                    o = mk3(orr.x, orr.y, orr.z);
                    d = mk3(dm.x, dm.y, dm.z);
                    rng = rng_in = __float_as_uint(orr.w);
                    total_dst = 0.0f;
                    alpha_k = 0;
                    start_traversal();
                    active = true;
                }
            }
            if (got && MODE == 1)
            {
                const size_t job = q_index - shard_base;
                slot = queue[shard_base + (job >> 1)];
                ray_k = (uint32_t)(job & 1u);
                const float4 so = pb.sh_org[slot];
                if (__float_as_uint(so.w) & (1u << ray_k))
                {
                    const float4 dd = ray_k ? pb.sh_d1[slot] : pb.sh_d0[slot];
                    o = mk3(so.x, so.y, so.z);
                    d = mk3(dd.x, dd.y, dd.z);
                    start_traversal();
                    active = true;
                }
            }
            continue;
        }
        if (cE == 64u) break;   // nothing in flight and (see above) nothing left to fetch

        if (cN >= cT && cN >= cI && cN >= cF)
        {
            // ---- N: internal nodes of either level; keeps stepping while at least half of the voters are still at one ----
            for (uint32_t r = 0;; r++)
            {
                const bool n = active && !(cur & REF_LEAF);
                if (r > 0 && (r >= nsteps || (uint32_t)__popcll(__ballot(n)) * 2u < cN)) break;
                if (n)
                {
                    const NodeRegs nd = geo.node(blas_base != 0xFFFFFFFFu, cur);
                    float ld = slab_dst(co, cinv, nd.a.x, nd.a.y, nd.a.z, nd.a.w, nd.b.x, nd.b.y);
                    float rd = slab_dst(co, cinv, nd.b.z, nd.b.w, nd.c.x, nd.c.y, nd.c.z, nd.c.w);
                    bool left_first = ld <= rd;
                    bool push_l = ld < best.t, push_r = rd < best.t;
                    uint32_t near_ref = left_first ? nd.left : nd.right;
                    uint32_t far_ref = left_first ? nd.right : nd.left;
                    bool push_near = left_first ? push_l : push_r;
                    bool push_far = left_first ? push_r : push_l;
                    if (push_far) { lds_stack[sp * LP_BLOCK + tid] = far_ref; sp++; }
                    if (push_near) cur = near_ref; else pop();
                }
            }
        }
        else if (cT >= cI && cT >= cF)
        {
            // ---- T: one triangle of a BLAS leaf, first-found wins ties (strict <) ----
            if (isT)
            {
                const uint32_t ti = cur & ~REF_LEAF;
                const TriVerts tv = geo.tri(ti);
                TriHit h = tri_dst(co, cd, xyz(tv.v0), xyz(tv.v1), xyz(tv.v2), eps);
                if (h.t < best.t) { best.t = h.t; best.u = h.u; best.v = h.v; best.tri = ti; best.inst = cur_inst; }
                if (__float_as_uint(tv.v0.w) & LEAF_END_BITS) pop(); else cur++;
            }
        }
        else if (cI >= cF)
        {
            // ---- I: enter an instance (bvh_custom.wgsl:28-37) ----
            if (isI)
            {
                cur_inst = cur & ~REF_LEAF;
                const InstanceDev in = geo.inst(cur_inst);
                co = mk3(o.x * in.r0.x + o.y * in.r0.y + o.z * in.r0.z + 1.0f * in.r0.w,
                         o.x * in.r1.x + o.y * in.r1.y + o.z * in.r1.z + 1.0f * in.r1.w,
                         o.x * in.r2.x + o.y * in.r2.y + o.z * in.r2.z + 1.0f * in.r2.w);
                cd = mk3(d.x * in.r0.x + d.y * in.r0.y + d.z * in.r0.z + 0.0f * in.r0.w,
                         d.x * in.r1.x + d.y * in.r1.y + d.z * in.r1.z + 0.0f * in.r1.w,
                         d.x * in.r2.x + d.y * in.r2.y + d.z * in.r2.z + 0.0f * in.r2.w);
                if (!(in.blas_root & REF_LEAF)) cinv = mk3(1.0f / cd.x, 1.0f / cd.y, 1.0f / cd.z);
                blas_base = sp;
                cur = in.blas_root;
            }
        }
        else
        {
            // ---- F: end of a traversal = one iteration of ray_skip_alpha_stochastically (bvh_custom.wgsl:154-180) ----
            if (isF && MODE == 1)
            {
                const bool hit = best.t != LP_F32_MAX;
                const float4 rec = make_float4(hit ? best.t : 0.0f, hit ? best.u : 0.0f, hit ? best.v : 0.0f, __uint_as_float(hit ? best.inst : HIT_MISS));
                if (TYPE == LUPIN_PATHTRACE_MIS && ray_k == 0) { pb.next_hit[slot] = rec; pb.next_tri[slot] = best.tri; }
                else { pb.sh_hit1[slot] = rec; pb.sh_f1[slot].w = __uint_as_float(best.tri); }
                active = false;
            }
            if (isF && MODE == 0)
            {
                const bool hit = best.t != LP_F32_MAX;
                bool again = false;
                if (hit)
                {
                    total_dst += best.t;
                    if (sc.instances[best.inst].flags & 1u)
                    {
                        Surface sf = resolve_surface(sc, best.inst, best.tri, best.u, best.v);
                        float opacity = surface_opacity(sc, sf);
                        if (opacity < 1.0f && rnd(rng) >= opacity)
                        {
                            o = add(o, scale(d, best.t));
                            alpha_k++;
                            again = alpha_k < 128u;   // MAX_OPACITY_BOUNCES (pathtracer.wgsl:1263)
                        }
                    }
                }
                if (again)
                {
                    start_traversal();
                }
                else
                {
                    pb.hit[slot] = make_float4(total_dst, best.u, best.v, __uint_as_float(hit ? best.inst : HIT_MISS));
                    pb.hit_tri[slot] = best.tri;
                    if (rng != rng_in) pb.ori_rng[slot].w = __uint_as_float(rng);
                    active = false;
                }
            }
        }
    }
}

// clamp_radiance (pathtracer.wgsl:1774-1783)
__device__ __forceinline__ f3 clamp_radiance(f3 r, float max_radiance)
{
    if (!finite3(r)) r = splat(0.0f);
    if (r.x > max_radiance || r.y > max_radiance || r.z > max_radiance)
        r = scale(r, max_radiance / maxf(r.x, maxf(r.y, r.z)));
    return r;
}

// shadow rays a vertex wants traced (MIS: BSDF- and light-sampled directions; Direct: the light ray)
struct ShadowRays
{
    f3 org;
    f3 d0, f0; float s0; bool v0;
    f3 d1, f1; float s1; bool v1;
};

struct PathRegs
{
    f3 ori, dir, weight, radiance;
    uint32_t rng;
    int bounce;
    bool in_medium;       // volume_stack_len == 1
    bool next_emission;
    Medium medium;
};

// One iteration of the integrator loop body after the closest-hit query.  Returns true when the
// path continues with (ori, dir) set for the next bounce, false on `break`.
// TYPE 0: pathtrace_standard (:588-733)   1: pathtrace_mis (:737-933)
//      2: pathtrace_naive (:942-1059)     3: pathtrace_direct (:1062-1245)
template <int TYPE, typename Geo, bool SIMPLE = false>
__device__ bool integrate_vertex(const Geo &geo, const SceneDev &sc, uint32_t *stack, const FrameParams &fp, PathRegs &p,
                                 float4 hitrec, uint32_t hit_tri, ShadowRays &sh)
{
    const float eps = fp.pc.ray_epsilon;
    const uint32_t hit_inst = __float_as_uint(hitrec.w);
    if (hit_inst == HIT_MISS)
    {
        if (TYPE != LUPIN_PATHTRACE_DIRECT || p.next_emission)
            p.radiance = add(p.radiance, mul(p.weight, environment_radiance(sc, p.dir)));
        return false;
    }
    const float hit_dst = hitrec.x;

    // transmission inside a medium (:611-621)
    bool in_volume = false;
    float volume_dst = hit_dst;
    if (p.in_medium)
    {
        float r1 = rnd(p.rng);
        float r2 = rnd(p.rng);
        volume_dst = medium_sample_distance(p.medium.density, hit_dst, r1, r2);
        f3 tr = medium_transmittance(p.medium.density, volume_dst);
        float tp = medium_distance_pdf(p.medium.density, volume_dst, hit_dst);
        p.weight = mul(p.weight, divs(tr, tp));
        in_volume = volume_dst < hit_dst;
    }

    const f3 outgoing = neg(p.dir);
    f3 incoming = splat(0.0f);
    f3 hit_pos;
    if (!in_volume)
    {
        hit_pos = add(p.ori, scale(p.dir, hit_dst));
        const Surface s = resolve_surface(sc, hit_inst, hit_tri, hitrec.y, hitrec.z);
        const MatPoint mp = material_point<SIMPLE>(sc, s);
        const f3 normal = shading_normal(geo, sc, s);

        if (TYPE == LUPIN_PATHTRACE_STANDARD || TYPE == LUPIN_PATHTRACE_NAIVE || p.next_emission)
            p.radiance = add(p.radiance, mul(p.weight, mp.emission));

        const bool delta = mat_is_delta(mp);

        if (TYPE == LUPIN_PATHTRACE_DIRECT)   // light ray before choosing the continuation (:1117-1146)
        {
            if (!delta)
            {
                f3 li = lights_sample(sc, hit_pos, p.rng);
                float pdf = lights_pdf(geo, sc, stack, hit_pos, li, eps);
                f3 bsdfcos = bsdf_eval(mp, normal, outgoing, li);
                if (none_zero3(bsdfcos) && pdf > 0.0f)
                {
                    // radiance += weight * bsdfcos * emission(light_ray) / pdf   -- traced by k_shadow (:1125-1138)
                    sh.org = hit_pos; sh.d1 = li; sh.f1 = mul(p.weight, bsdfcos); sh.s1 = pdf; sh.v1 = true;
                }
                p.next_emission = false;
            }
            else p.next_emission = true;
        }

        if (!delta)
        {
            if (TYPE == LUPIN_PATHTRACE_STANDARD || TYPE == LUPIN_PATHTRACE_DIRECT)
            {
                // one-sample mixture of BSDF and light sampling (:640-657)
                if (rnd(p.rng) < 0.5f)
                {
                    float rnl = rnd(p.rng);
                    float ra = rnd(p.rng), rb = rnd(p.rng);
                    incoming = bsdf_sample(mp, normal, outgoing, rnl, ra, rb);
                }
                else incoming = lights_sample(sc, hit_pos, p.rng);
                if (is_zero3(incoming)) return false;
                float prob = 0.5f * bsdf_pdf(mp, normal, outgoing, incoming) + 0.5f * lights_pdf(geo, sc, stack, hit_pos, incoming, eps);
                p.weight = mul(p.weight, divs(bsdf_eval(mp, normal, outgoing, incoming), prob));
            }
            else if (TYPE == LUPIN_PATHTRACE_NAIVE)
            {
                float rnl = rnd(p.rng);
                float ra = rnd(p.rng), rb = rnd(p.rng);
                incoming = bsdf_sample(mp, normal, outgoing, rnl, ra, rb);
                if (is_zero3(incoming)) return false;
                p.weight = mul(p.weight, divs(bsdf_eval(mp, normal, outgoing, incoming), bsdf_pdf(mp, normal, outgoing, incoming)));
            }
            else   // MIS: BSDF sample then light sample, power heuristic (:802-855)
            {
                #pragma unroll 1
                for (int k = 0; k < 2; k++)
                {
                    const bool light_turn = (k != 0);
                    f3 mi;
                    if (light_turn) mi = lights_sample(sc, hit_pos, p.rng);
                    else
                    {
                        float rnl = rnd(p.rng);
                        float ra = rnd(p.rng), rb = rnd(p.rng);
                        mi = bsdf_sample(mp, normal, outgoing, rnl, ra, rb);
                    }
                    if (is_zero3(mi)) break;
                    if (!light_turn) incoming = mi;

                    f3 bsdfcos = bsdf_eval(mp, normal, outgoing, mi);
                    float light_pdf = lights_pdf(geo, sc, stack, hit_pos, mi, eps);
                    float b_pdf = bsdf_pdf(mp, normal, outgoing, mi);
                    float mis_w;
                    if (light_turn) mis_w = (light_pdf * light_pdf) / (light_pdf * light_pdf + b_pdf * b_pdf) / light_pdf;
                    else            mis_w = (b_pdf * b_pdf) / (b_pdf * b_pdf + light_pdf * light_pdf) / b_pdf;

                    if (none_zero3(bsdfcos) && mis_w != 0.0f)
                    {
                        // radiance += weight * bsdfcos * emission(mis_ray) * mis_weight   -- traced by k_shadow (:831-849);
                        // the BSDF-sampled ray's hit also becomes `next_intersection`
                        sh.org = hit_pos;
                        if (!light_turn) { sh.d0 = mi; sh.f0 = mul(p.weight, bsdfcos); sh.s0 = mis_w; sh.v0 = true; }
                        else             { sh.d1 = mi; sh.f1 = mul(p.weight, bsdfcos); sh.s1 = mis_w; sh.v1 = true; }
                    }
                }
                p.weight = mul(p.weight, divs(bsdf_eval(mp, normal, outgoing, incoming), bsdf_pdf(mp, normal, outgoing, incoming)));
                p.next_emission = false;
            }
        }
        else
        {
            incoming = delta_sample(mp, normal, outgoing, rnd(p.rng));
            if (is_zero3(incoming)) return false;
            p.weight = mul(p.weight, divs(delta_eval(mp, normal, outgoing, incoming), delta_pdf(mp, normal, outgoing, incoming)));
            if (TYPE == LUPIN_PATHTRACE_MIS) p.next_emission = true;
        }

        // volume stack: push when empty, otherwise pop (:667-681) -- depth never exceeds 1
        if (mat_is_volumetric(mp) && dot3(normal, outgoing) * dot3(normal, incoming) < 0.0f)
        {
            if (!p.in_medium)
            {
                p.medium.density = mp.density;
                p.medium.scattering = mp.scattering;
                p.medium.anisotropy = mp.anisotropy;
                p.in_medium = true;
            }
            else p.in_medium = false;
        }
    }
    else
    {
        hit_pos = add(p.ori, scale(p.dir, volume_dst));
        if (TYPE == LUPIN_PATHTRACE_NAIVE)
        {
            float unused0 = rnd(p.rng); (void)unused0;
            float ra = rnd(p.rng), rb = rnd(p.rng);
            incoming = phase_sample(p.medium, outgoing, ra, rb);
            if (is_zero3(incoming)) return false;
            float prob = phase_pdf(p.medium, outgoing, incoming);
            p.weight = mul(p.weight, divs(phase_eval(p.medium, outgoing, incoming), prob));
        }
        else
        {
            if (rnd(p.rng) < 0.5f)
            {
                float unused0 = rnd(p.rng); (void)unused0;   // rnd0 is drawn and dropped (:700)
                float ra = rnd(p.rng), rb = rnd(p.rng);
                incoming = phase_sample(p.medium, outgoing, ra, rb);
            }
            else incoming = lights_sample(sc, hit_pos, p.rng);
            if (TYPE == LUPIN_PATHTRACE_MIS) p.next_emission = true;
            if (is_zero3(incoming)) return false;
            float prob = 0.5f * phase_pdf(p.medium, outgoing, incoming) + 0.5f * lights_pdf(geo, sc, stack, hit_pos, incoming, eps);
            p.weight = mul(p.weight, divs(phase_eval(p.medium, outgoing, incoming), prob));
        }
    }

    p.ori = hit_pos;
    p.dir = incoming;

    // weight check and Russian roulette (:720-729)
    if (is_zero3(p.weight) || !finite3(p.weight)) return false;
    if (p.bounce > 3)
    {
        float survive = minf(0.99f, maxf(p.weight.x, maxf(p.weight.y, p.weight.z)));
        if (rnd(p.rng) >= survive) return false;
        p.weight = scale(p.weight, 1.0f / survive);
    }
    return true;
}

// Everything of one integrator-loop iteration after the closest-hit query, for one path; writes the path state
// back and returns whether the pixel still has work (the path continues, or its next camera sample was started).
template <int TYPE, typename Geo, bool SIMPLE = false>
__device__ __forceinline__ bool shade_path(const Geo &geo, const SceneDev &sc, uint32_t *stack, const FrameParams &fp, PathBuffers &pb,
                                           uint32_t slot, float4 orr, float4 dm, uint32_t rng, float4 hitrec, uint32_t hit_tri)
{
    bool alive = false;
    float4 w4 = pb.weight[slot];
    float4 r4 = pb.radiance[slot];
    uint32_t meta = __float_as_uint(dm.w);

    PathRegs p;
    p.ori = mk3(orr.x, orr.y, orr.z);
    p.dir = mk3(dm.x, dm.y, dm.z);
    p.weight = mk3(w4.x, w4.y, w4.z);
    p.radiance = mk3(r4.x, r4.y, r4.z);
    p.rng = rng;
    p.bounce = (int)(meta & META_BOUNCE_MASK);
    p.in_medium = SIMPLE ? false : (meta & META_VOLUME) != 0;   // matte surfaces never open a medium
    p.next_emission = (meta & META_NEXT_EMISSION) != 0;
    uint32_t sample = meta >> META_SAMPLE_SHIFT;
    const bool was_in_medium = p.in_medium;
    if (p.in_medium)
    {
        float4 a = pb.vol0[slot], b = pb.vol1[slot];
        p.medium.density = mk3(a.x, a.y, a.z);
        p.medium.anisotropy = a.w;
        p.medium.scattering = mk3(b.x, b.y, b.z);
    }
    else { p.medium.density = splat(0.0f); p.medium.scattering = splat(0.0f); p.medium.anisotropy = 0.0f; }

    ShadowRays sh;
    sh.v0 = sh.v1 = false;
    bool cont = integrate_vertex<TYPE, Geo, SIMPLE>(geo, sc, stack, fp, p, hitrec, hit_tri, sh);
    if (cont)
    {
        p.bounce++;
        if (p.bounce > (int)fp.max_bounces) cont = false;   // loop condition `bounce <= MAX_BOUNCES` (:596)
    }

    if (TYPE == LUPIN_PATHTRACE_MIS || TYPE == LUPIN_PATHTRACE_DIRECT)
    {
        // hand the vertex to k_shadow: it adds the shadow-ray terms to `radiance` (the order of the additions is the
        // reference's) and only then folds a finished path into the pixel / starts the next sample
        if (p.in_medium && !was_in_medium)
        {
            pb.vol0[slot] = make_float4(p.medium.density.x, p.medium.density.y, p.medium.density.z, p.medium.anisotropy);
            pb.vol1[slot] = make_float4(p.medium.scattering.x, p.medium.scattering.y, p.medium.scattering.z, 0.0f);
        }
        pb.weight[slot] = make_float4(p.weight.x, p.weight.y, p.weight.z, 0.0f);
        pb.radiance[slot] = make_float4(p.radiance.x, p.radiance.y, p.radiance.z, 0.0f);
        const uint32_t nm = ((uint32_t)p.bounce & META_BOUNCE_MASK) | (p.in_medium ? META_VOLUME : 0u) |
                            (p.next_emission ? META_NEXT_EMISSION : 0u) | (cont ? 0u : META_TERMINATED) | (sample << META_SAMPLE_SHIFT);
        pb.ori_rng[slot] = make_float4(p.ori.x, p.ori.y, p.ori.z, __uint_as_float(p.rng));
        pb.dir_meta[slot] = make_float4(p.dir.x, p.dir.y, p.dir.z, __uint_as_float(nm));
        const uint32_t flags = (sh.v0 ? 1u : 0u) | (sh.v1 ? 2u : 0u);
        pb.sh_org[slot] = make_float4(sh.org.x, sh.org.y, sh.org.z, __uint_as_float(flags));
        if (sh.v0) { pb.sh_d0[slot] = make_float4(sh.d0.x, sh.d0.y, sh.d0.z, sh.s0); pb.sh_f0[slot] = make_float4(sh.f0.x, sh.f0.y, sh.f0.z, 0.0f); }
        if (sh.v1) { pb.sh_d1[slot] = make_float4(sh.d1.x, sh.d1.y, sh.d1.z, sh.s1); pb.sh_f1[slot] = make_float4(sh.f1.x, sh.f1.y, sh.f1.z, 0.0f); }
        return true;
    }

    if (cont)
    {
        alive = true;
        if (p.in_medium && !was_in_medium)
        {
            pb.vol0[slot] = make_float4(p.medium.density.x, p.medium.density.y, p.medium.density.z, p.medium.anisotropy);
            pb.vol1[slot] = make_float4(p.medium.scattering.x, p.medium.scattering.y, p.medium.scattering.z, 0.0f);
        }
        pb.weight[slot] = make_float4(p.weight.x, p.weight.y, p.weight.z, 0.0f);
        if (p.radiance.x != r4.x || p.radiance.y != r4.y || p.radiance.z != r4.z)   // only emitters touch it
            pb.radiance[slot] = make_float4(p.radiance.x, p.radiance.y, p.radiance.z, 0.0f);
    }
    else
    {
        // path finished: fold its radiance into the pixel, start the pixel's next sample (:234-239)
        float4 c4 = pb.color[slot];
        f3 cr = clamp_radiance(p.radiance, fp.pc.max_radiance);
        pb.color[slot] = make_float4(c4.x + cr.x, c4.y + cr.y, c4.z + cr.z, 0.0f);
        sample++;
        if (sample < fp.spp)
        {
            alive = true;
            uint32_t gx, gy;
            slot_to_pixel(fp, slot, gx, gy);
            camera_ray(fp, gx, gy, p.rng, p.ori, p.dir);
            p.bounce = 0;
            p.in_medium = false;
            p.next_emission = true;
            pb.weight[slot] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
            pb.radiance[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (TYPE == LUPIN_PATHTRACE_MIS)
            {
                pb.next_hit[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(HIT_MISS));
                pb.next_tri[slot] = 0u;
            }
        }
    }
    if (alive)
    {
        uint32_t nm = ((uint32_t)p.bounce & META_BOUNCE_MASK) | (p.in_medium ? META_VOLUME : 0u) |
                      (p.next_emission ? META_NEXT_EMISSION : 0u) | (sample << META_SAMPLE_SHIFT);
        pb.ori_rng[slot] = make_float4(p.ori.x, p.ori.y, p.ori.z, __uint_as_float(p.rng));
        pb.dir_meta[slot] = make_float4(p.dir.x, p.dir.y, p.dir.z, __uint_as_float(nm));
    }
    return alive;
}

// SIMPLE: scenes of untextured matte surfaces without environments (LupinScene::simple_matte, decided at upload) get a
// k_shade in which those facts are compile-time constants: same arithmetic on the paths that exist, none of the code
// for the ones that cannot.
template <int TYPE, bool LDSGEO, bool SIMPLE>
__global__ void __attribute__((amdgpu_waves_per_eu(TYPE == 1 ? LP_MIS_SHADE_WAVES : (SIMPLE ? LP_SIMPLE_SHADE_WAVES : LP_SHADE_WAVES), 8))) __launch_bounds__(LP_BLOCK) k_shade(SceneDev sc, const FrameParams *__restrict__ fpp, PathBuffers pb, uint32_t iter,
                                                    unsigned long long *shard_stats, uint32_t stack_words)
{
    const FrameParams fp = *fpp;
    if (SIMPLE) { sc.num_envs = 0; sc.sort_shade = 0; }   // facts of a simple_matte scene, constant from here on
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    const auto geo = make_geo<LDSGEO>(sc, lds_stack, stack_words);
    const uint32_t shard = blockIdx.x % LP_SHARDS;
    const uint32_t count = pb.counts[iter * LP_SHARDS + shard];
    const uint32_t i = (blockIdx.x / LP_SHARDS) * LP_BLOCK + threadIdx.x;
    bool alive = false;
    bool mine = i < count;
    uint32_t slot = 0;
    if (mine) slot = pb.queue[iter & 1][(size_t)shard * pb.shard_cap + i];
    if (sc.sort_shade && (blockIdx.x / LP_SHARDS) * LP_BLOCK < count)   // block-uniform
    {
        // Scenes with several material types: counting-sort the block's 256 paths by what they will execute (material
        // type of the hit | miss | inside a medium) so that a wave runs one or two BSDF families instead of all of them.
        // Which thread shades which path does not matter: all path state lives in the path's slot.
        __shared__ uint32_t bins[16];
        if (threadIdx.x < 16) bins[threadIdx.x] = 0u;
        __syncthreads();
        uint32_t key = 15u;
        if (mine)
        {
            const uint32_t inst = __float_as_uint(pb.hit[slot].w);
            const uint32_t meta = __float_as_uint(pb.dir_meta[slot].w);
            key = (meta & META_VOLUME) ? 9u : (inst == HIT_MISS ? 8u : ((sc.instances[inst].flags >> 8) & 7u));
        }
        const uint32_t rank = atomicAdd(&bins[key], 1u);
        __syncthreads();
        uint32_t base = 0;
        for (uint32_t k = 0; k < key; k++) base += bins[k];
        lds_stack[base + rank] = mine ? slot : 0xFFFFFFFFu;   // the traversal stacks are not in use yet
        __syncthreads();
        slot = lds_stack[threadIdx.x];
        mine = slot != 0xFFFFFFFFu;
        __syncthreads();
    }
    if (mine)
    {
        const float4 orr = pb.ori_rng[slot];
        alive = shade_path<TYPE, typename GeoOf<LDSGEO>::type, SIMPLE>(geo, sc, lds_stack, fp, pb, slot, orr, pb.dir_meta[slot], __float_as_uint(orr.w), pb.hit[slot], pb.hit_tri[slot]);
    }
    if (i == 0 && iter == 0) shard_stats[shard * 2 + 1] += (unsigned long long)count * fp.spp;
    if (TYPE == LUPIN_PATHTRACE_MIS || TYPE == LUPIN_PATHTRACE_DIRECT) return;   // k_shadow appends
    queue_append(alive, slot, pb.queue[(iter + 1) & 1] + (size_t)shard * pb.shard_cap, &pb.counts[(iter + 1) * LP_SHARDS + shard]);
}

// emission of a surface point: emission_sample * mat.emission of get_material_point (pathtracer.wgsl:1295-1298,1315)
__device__ __forceinline__ f3 surface_emission(const SceneDev &sc, const Surface &s)
{
    const LupinMaterial *m = &sc.materials[s.in.mat_idx];
    f3 es = splat(1.0f);
    if (s.mesh.texcoords_base != LUPIN_SENTINEL_IDX && m->emission_tex_idx != LUPIN_SENTINEL_IDX)
    {
        float tu, tv;
        interp_texcoords(sc, s, tu, tv);
        float4 t = sample_texture(sc, m->emission_tex_idx, tu, tv);
        es = mk3(t.x, t.y, t.z);
    }
    return mk3(es.x * m->emission[0], es.y * m->emission[1], es.z * m->emission[2]);
}

// Shadow-ray stage of the MIS and Direct integrators: traces the rays k_shade recorded (plain closest hit, no alpha
// skipping -- pathtracer.wgsl:834,1126), adds their terms to the path radiance in the reference's order, keeps the
// BSDF-sampled hit as MIS `next_intersection`, and finishes paths that ended at this vertex.
template <int TYPE, bool LDSGEO, bool PRETRACED>
__global__ void __attribute__((amdgpu_waves_per_eu(LP_EXTEND_WAVES, 8))) __launch_bounds__(LP_BLOCK) k_shadow(SceneDev sc, const FrameParams *__restrict__ fpp, PathBuffers pb, uint32_t iter,
                                                     uint32_t stack_words)
{
    const FrameParams fp = *fpp;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    const auto geo = make_geo<LDSGEO>(sc, lds_stack, stack_words);
    const uint32_t shard = blockIdx.x % LP_SHARDS;
    const uint32_t count = pb.counts[iter * LP_SHARDS + shard];
    const uint32_t i = (blockIdx.x / LP_SHARDS) * LP_BLOCK + threadIdx.x;
    bool alive = false;
    uint32_t slot = 0;
    if (i < count)
    {
        slot = pb.queue[iter & 1][(size_t)shard * pb.shard_cap + i];
        const float eps = fp.pc.ray_epsilon;
        const float4 so = pb.sh_org[slot];
        const uint32_t flags = __float_as_uint(so.w);
        const f3 org = mk3(so.x, so.y, so.z);
        const float4 r4 = pb.radiance[slot];
        f3 radiance = mk3(r4.x, r4.y, r4.z);
        for (int k = 0; k < 2; k++)
        {
            if (!(flags & (1u << k))) continue;
            const float4 dd = k ? pb.sh_d1[slot] : pb.sh_d0[slot];
            const float4 ff = k ? pb.sh_f1[slot] : pb.sh_f0[slot];
            const f3 dir = mk3(dd.x, dd.y, dd.z);
            Closest c;
            if (PRETRACED)   // k_extend_persistent<.., 1> traced the ray
            {
                const bool first = (TYPE == LUPIN_PATHTRACE_MIS && k == 0);
                const float4 rec = first ? pb.next_hit[slot] : pb.sh_hit1[slot];
                c.inst = __float_as_uint(rec.w);
                c.t = c.inst != HIT_MISS ? rec.x : LP_F32_MAX; c.u = rec.y; c.v = rec.z;
                c.tri = first ? pb.next_tri[slot] : __float_as_uint(ff.w);
            }
            else c = scene_closest(geo, sc, lds_stack, org, dir, eps);
            const bool hit = c.t != LP_F32_MAX;
            if (!PRETRACED && TYPE == LUPIN_PATHTRACE_MIS && k == 0)
            {
                pb.next_hit[slot] = make_float4(hit ? c.t : 0.0f, hit ? c.u : 0.0f, hit ? c.v : 0.0f, __uint_as_float(hit ? c.inst : HIT_MISS));
                pb.next_tri[slot] = c.tri;
            }
            f3 emission;
            if (hit) emission = surface_emission(sc, resolve_surface(sc, c.inst, c.tri, c.u, c.v));
            else emission = environment_radiance(sc, dir);
            const f3 term = mul(mk3(ff.x, ff.y, ff.z), emission);
            if (TYPE == LUPIN_PATHTRACE_MIS) radiance = add(radiance, scale(term, dd.w));
            else radiance = add(radiance, divs(term, dd.w));
        }

        const float4 dm = pb.dir_meta[slot];
        uint32_t meta = __float_as_uint(dm.w);
        if (!(meta & META_TERMINATED))
        {
            alive = true;
            if (flags) pb.radiance[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
        }
        else
        {
            float4 c4 = pb.color[slot];
            f3 cr = clamp_radiance(radiance, fp.pc.max_radiance);
            pb.color[slot] = make_float4(c4.x + cr.x, c4.y + cr.y, c4.z + cr.z, 0.0f);
            uint32_t sample = (meta >> META_SAMPLE_SHIFT) + 1u;
            if (sample < fp.spp)
            {
                alive = true;
                uint32_t rng = __float_as_uint(pb.ori_rng[slot].w);
                uint32_t gx, gy;
                slot_to_pixel(fp, slot, gx, gy);
                f3 o, d;
                camera_ray(fp, gx, gy, rng, o, d);
                pb.weight[slot] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
                pb.radiance[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                pb.next_hit[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(HIT_MISS));
                pb.next_tri[slot] = 0u;
                pb.ori_rng[slot] = make_float4(o.x, o.y, o.z, __uint_as_float(rng));
                pb.dir_meta[slot] = make_float4(d.x, d.y, d.z, __uint_as_float(META_NEXT_EMISSION | (sample << META_SAMPLE_SHIFT)));
            }
        }
    }
    queue_append(alive, slot, pb.queue[(iter + 1) & 1] + (size_t)shard * pb.shard_cap, &pb.counts[(iter + 1) * LP_SHARDS + shard]);
}

// pathtrace_main tail (pathtracer.wgsl:275-289)
__global__ void __launch_bounds__(LP_BLOCK) k_resolve(FrameParams fp, PathBuffers pb, uint32_t n,
                                                      const __half *prev, __half *out)
{
    uint32_t slot = blockIdx.x * LP_BLOCK + threadIdx.x;
    if (slot >= n) return;
    uint32_t gx, gy;
    slot_to_pixel(fp, slot, gx, gy);
    if (gx >= fp.width || gy >= fp.height) return;
    float4 c4 = pb.color[slot];
    float spp = (float)fp.spp;
    f3 c = mk3(maxf(c4.x / spp, 0.0f), maxf(c4.y / spp, 0.0f), maxf(c4.z / spp, 0.0f));
    size_t px = ((size_t)gy * fp.width + gx) * 4;
    if (fp.pc.accum_counter != 0)
    {
        float w = 1.0f / (float)fp.pc.accum_counter;
        f3 pc = mk3(__half2float(prev[px + 0]), __half2float(prev[px + 1]), __half2float(prev[px + 2]));
        c = mk3(maxf(pc.x * (1.0f - w) + c.x * w, 0.0f), maxf(pc.y * (1.0f - w) + c.y * w, 0.0f), maxf(pc.z * (1.0f - w) + c.z * w, 0.0f));
    }
    if (fp.store_rne)
    {
        out[px + 0] = __float2half_rn(c.x);
        out[px + 1] = __float2half_rn(c.y);
        out[px + 2] = __float2half_rn(c.z);
    }
    else
    {
        out[px + 0] = __float2half_rz(c.x);
        out[px + 1] = __float2half_rz(c.y);
        out[px + 2] = __float2half_rz(c.z);
    }
    out[px + 3] = __float2half_rn(1.0f);
}

// pathtrace_falsecolor_main (pathtracer.wgsl:296-452): G-buffer style visualisations, one thread per pixel, no bounces.
__device__ __forceinline__ f3 hash_color(uint32_t id)   // :544-573
{
    uint32_t st = id;
    float c[3];
    for (int k = 0; k < 3; k++)
    {
        st = st * 747796405u + 2891336453u;
        uint32_t r = ((st >> ((st >> 28u) + 4u)) ^ st) * 277803737u;
        r = (r >> 22u) ^ r;
        c[k] = (float)r / 4294967295.0f;
    }
    return mk3(c[0], c[1], c[2]);
}

template <bool LDSGEO>
__global__ void __launch_bounds__(LP_BLOCK) k_falsecolor(SceneDev sc, FrameParams fp, uint32_t n, const __half *prev, __half *out, uint32_t stack_words)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    const auto geo = make_geo<LDSGEO>(sc, lds_stack, stack_words);
    const uint32_t slot = blockIdx.x * LP_BLOCK + threadIdx.x;
    if (slot >= n) return;
    uint32_t gx, gy;
    slot_to_pixel(fp, slot, gx, gy);
    if (gx >= fp.width || gy >= fp.height) return;
    const float eps = fp.pc.ray_epsilon;
    const uint32_t type = fp.pc.falsecolor_type;
    uint32_t rng = rng_seed_for(gy * fp.width + gx, fp.pc.accum_counter);
    f3 color = splat(0.0f);
    for (uint32_t sample = 0; sample < fp.spp; sample++)
    {
        f3 o, d;
        camera_ray(fp, gx, gy, rng, o, d);
        float4 hitrec;
        uint32_t hit_tri;
        if (type <= 6) trace_alpha(geo, sc, lds_stack, o, d, rng, eps, hitrec, hit_tri);
        else if (type <= 11)
        {
            const Closest c = scene_closest(geo, sc, lds_stack, o, d, eps);
            const bool hit = c.t != LP_F32_MAX;
            hitrec = make_float4(c.t, c.u, c.v, __uint_as_float(hit ? c.inst : HIT_MISS));
            hit_tri = c.tri;
        }
        else continue;
        const uint32_t inst = __float_as_uint(hitrec.w);
        if (inst == HIT_MISS) continue;
        const Surface s = resolve_surface(sc, inst, hit_tri, hitrec.y, hitrec.z);
        f3 add_c;
        switch (type)
        {
        case 0: add_c = material_point(sc, s).color; break;
        case 1: add_c = shading_normal(geo, sc, s); break;
        case 2: { f3 nn = shading_normal(geo, sc, s); add_c = mk3(nn.x * 0.5f + 0.5f, nn.y * 0.5f + 0.5f, nn.z * 0.5f + 0.5f); break; }
        case 3:
        {
            // hit_backside = det > 0 with det = dot(local dir, cross(v1 - v0, v2 - v0)) of the winning triangle
            // (bvh_custom.wgsl:106, pathtracer.wgsl:2933-2935); recomputed from the instance-local direction
            const TriVerts tv = geo.tri_fetch(hit_tri);
            const f3 ld = mk3(d.x * s.in.r0.x + d.y * s.in.r0.y + d.z * s.in.r0.z + 0.0f * s.in.r0.w,
                              d.x * s.in.r1.x + d.y * s.in.r1.y + d.z * s.in.r1.z + 0.0f * s.in.r1.w,
                              d.x * s.in.r2.x + d.y * s.in.r2.y + d.z * s.in.r2.z + 0.0f * s.in.r2.w);
            const float det = dot3(ld, cross3(sub(xyz(tv.v1), xyz(tv.v0)), sub(xyz(tv.v2), xyz(tv.v0))));
            add_c = splat(det > 0.0f ? 0.0f : 1.0f);
            break;
        }
        case 4: add_c = material_point(sc, s).emission; break;
        case 5: add_c = splat(material_point(sc, s).roughness); break;
        case 6: add_c = splat(material_point(sc, s).metallic); break;
        case 7: add_c = splat(material_point(sc, s).opacity); break;
        case 8: add_c = hash_color(s.in.mat_idx); break;
        case 9: add_c = splat(mat_is_delta(material_point(sc, s)) ? 1.0f : 0.0f); break;
        case 10: add_c = hash_color(inst); break;
        default: add_c = hash_color(hit_tri - s.mesh.tri_offset); break;
        }
        color = add(color, add_c);
    }
    const float spp = (float)fp.spp;
    f3 c = mk3(maxf(color.x / spp, 0.0f), maxf(color.y / spp, 0.0f), maxf(color.z / spp, 0.0f));
    const size_t px = ((size_t)gy * fp.width + gx) * 4;
    if (fp.pc.accum_counter != 0)
    {
        float w = 1.0f / (float)fp.pc.accum_counter;
        f3 pc = mk3(__half2float(prev[px + 0]), __half2float(prev[px + 1]), __half2float(prev[px + 2]));
        c = mk3(maxf(pc.x * (1.0f - w) + c.x * w, 0.0f), maxf(pc.y * (1.0f - w) + c.y * w, 0.0f), maxf(pc.z * (1.0f - w) + c.z * w, 0.0f));
    }
    if (fp.store_rne) { out[px + 0] = __float2half_rn(c.x); out[px + 1] = __float2half_rn(c.y); out[px + 2] = __float2half_rn(c.z); }
    else { out[px + 0] = __float2half_rz(c.x); out[px + 1] = __float2half_rz(c.y); out[px + 2] = __float2half_rz(c.z); }
    out[px + 3] = __float2half_rn(1.0f);
}

// get_heatmap_color (pathtracer.wgsl:2806-2872): value -> wavelength 380..750 nm -> rgb, gamma 0.8
__device__ __forceinline__ f3 heatmap_color(float val, float lo, float hi)
{
    const float wavelength = 380.0f + 370.0f * maxf(val - lo, 0.0f) / maxf(hi - lo, 0.0f);
    f3 color = splat(0.0f);
    if (wavelength <= 380.0f) color = mk3(0.0f, 0.0f, 0.0f);
    else if (wavelength > 380.0f && wavelength <= 440.0f) color = mk3(-(wavelength - 440.0f) / 60.0f / 3.0f, 0.0f, 0.8f);
    else if (wavelength >= 440.0f && wavelength <= 490.0f) color = mk3(0.0f, (wavelength - 440.0f) / 50.0f, 1.0f);
    else if (wavelength >= 490.0f && wavelength <= 510.0f) color = mk3(0.0f, 1.0f, -(wavelength - 510.0f) / 20.0f);
    else if (wavelength >= 510.0f && wavelength <= 580.0f) color = mk3((wavelength - 510.0f) / 70.0f, 1.0f, 0.0f);
    else if (wavelength >= 580.0f && wavelength <= 645.0f) color = mk3(1.0f, -(wavelength - 645.0f) / 65.0f, 0.0f);
    else if (wavelength >= 645.0f && wavelength <= 780.0f) color = mk3(1.0f, 0.0f, 0.0f);
    else color = splat(1.0f);

    const float gamma = 0.8f;
    float factor = 1.0f;
    if (wavelength >= 380.0f && wavelength < 420.0f) factor = 0.3f + 0.7f * (wavelength - 380.0f) / 40.0f;
    else if (wavelength >= 420.0f && wavelength < 701.0f) factor = 1.0f;
    else if (wavelength >= 701.0f && wavelength < 781.0f)
    {
        factor = 0.3f + 0.7f * (780.0f - wavelength) / 80.0f;
        return mk3(lpm_powf(color.x + factor * 1.0f, gamma), lpm_powf(color.y + factor * 1.0f, gamma), lpm_powf(color.z + factor * 1.0f, gamma));
    }
    else factor = 1.0f;
    return mk3(lpm_powf(factor * color.x, gamma), lpm_powf(factor * color.y, gamma), lpm_powf(factor * color.z, gamma));
}

// pathtrace_debug_main (pathtracer.wgsl:457-503): one sample per pixel of either the first closest-hit query or the whole
// Standard path, as ONE thread (the view is a diagnostic, not a hot path), with the box / triangle tests and the surface
// hits counted exactly where the reference counts them; the count becomes a heat-map colour.
template <bool LDSGEO>
__global__ void __launch_bounds__(LP_BLOCK) k_debug(SceneDev sc, FrameParams fp, uint32_t n, const __half *prev, __half *out, uint32_t stack_words)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    const auto base_geo = make_geo<LDSGEO>(sc, lds_stack, stack_words);
    const uint32_t slot = blockIdx.x * LP_BLOCK + threadIdx.x;
    if (slot >= n) return;
    uint32_t gx, gy;
    slot_to_pixel(fp, slot, gx, gy);
    if (gx >= fp.width || gy >= fp.height) return;
    const float eps = fp.pc.ray_epsilon;
    uint32_t aabb_checks = 0, tri_checks = 0, num_bounces = 0;
    GeoCounting<typename GeoOf<LDSGEO>::type> geo;
    geo.base = base_geo; geo.aabb_checks = &aabb_checks; geo.tri_checks = &tri_checks;

    PathRegs p;
    p.rng = rng_seed_for(gy * fp.width + gx, fp.pc.accum_counter);
    camera_ray(fp, gx, gy, p.rng, p.ori, p.dir);
    const bool first_hit_only = (fp.pc.flags & LUPIN_FLAG_DEBUG_FIRST_HIT_ONLY) != 0;
    const bool debug_num_bounces = (fp.pc.flags & LUPIN_FLAG_DEBUG_NUM_BOUNCES) != 0;
    if (first_hit_only && !debug_num_bounces)
    {
        scene_closest(geo, sc, lds_stack, p.ori, p.dir, eps);
    }
    else
    {
        p.weight = splat(1.0f); p.radiance = splat(0.0f);
        p.bounce = 0; p.in_medium = false; p.next_emission = true;
        p.medium.density = splat(0.0f); p.medium.scattering = splat(0.0f); p.medium.anisotropy = 0.0f;
        for (;;)
        {
            float4 hitrec;
            uint32_t hit_tri;
            trace_alpha(geo, sc, lds_stack, p.ori, p.dir, p.rng, eps, hitrec, hit_tri);
            if (__float_as_uint(hitrec.w) != HIT_MISS) num_bounces++;   // DEBUG_NUM_BOUNCES++ (:606-608)
            ShadowRays sh;
            sh.v0 = sh.v1 = false;
            if (!integrate_vertex<LUPIN_PATHTRACE_STANDARD>(geo, sc, lds_stack, fp, p, hitrec, hit_tri, sh)) break;
            p.bounce++;
            if (p.bounce > (int)fp.max_bounces) break;
        }
    }

    float val = 0.0f;
    if (fp.pc.flags & LUPIN_FLAG_DEBUG_TRI_CHECKS) val = (float)tri_checks;
    else if (fp.pc.flags & LUPIN_FLAG_DEBUG_AABB_CHECKS) val = (float)aabb_checks;
    else if (debug_num_bounces) val = (float)num_bounces;
    f3 c = heatmap_color(val, fp.pc.heatmap_min, fp.pc.heatmap_max);
    const size_t px = ((size_t)gy * fp.width + gx) * 4;
    if (fp.pc.accum_counter != 0)
    {
        float w = 1.0f / (float)fp.pc.accum_counter;
        f3 pc = mk3(__half2float(prev[px + 0]), __half2float(prev[px + 1]), __half2float(prev[px + 2]));
        c = mk3(maxf(pc.x * (1.0f - w) + c.x * w, 0.0f), maxf(pc.y * (1.0f - w) + c.y * w, 0.0f), maxf(pc.z * (1.0f - w) + c.z * w, 0.0f));
    }
    if (fp.store_rne) { out[px + 0] = __float2half_rn(c.x); out[px + 1] = __float2half_rn(c.y); out[px + 2] = __float2half_rn(c.z); }
    else { out[px + 0] = __float2half_rz(c.x); out[px + 1] = __float2half_rz(c.y); out[px + 2] = __float2half_rz(c.z); }
    out[px + 3] = __float2half_rn(1.0f);
}

// tonemap_and_fit_aspect (tonemapping.rs:155-224, tonemapping.wgsl): the reference draws a quad scaled to the source
// aspect inside a viewport of an Rgba8Unorm target.  As a compute kernel: one thread per target pixel of the scissor
// rectangle; pixel centres inside the quad sample the source (linear filter, clamp-to-edge), the rest keep the clear
// colour / the previous contents.  max(.,0) -> * 2^exposure -> filmic (ACES fit) -> linear-to-sRGB -> unorm8.
struct TonemapArgs
{
    uint32_t src_w, src_h, dst_w, dst_h;
    float vp_x, vp_y, vp_w, vp_h;
    float scale_x, scale_y, exposure;
    uint32_t filmic, srgb;
    uint32_t sc_x0, sc_y0, sc_x1, sc_y1;   // scissor rectangle clipped to the target
};

__device__ __forceinline__ float3 tonemap_texel(const __half *src, uint32_t w, uint32_t x, uint32_t y)
{
    const size_t i = ((size_t)y * w + x) * 4;
    return make_float3(__half2float(src[i + 0]), __half2float(src[i + 1]), __half2float(src[i + 2]));
}
__device__ __forceinline__ float linear_to_srgb1(float c)   // tonemapping.wgsl:73-79
{
    const float cutoff = c <= 0.0031308f ? 1.0f : 0.0f;
    const float higher = 1.055f * lpm_powf(c, 1.0f / 2.4f) - 0.055f;
    const float lower = c * 12.92f;
    return higher * (1.0f - cutoff) + lower * cutoff;
}
__device__ __forceinline__ float filmic1(float c)            // tonemapping.wgsl:63-71
{
    const float hdr = c * 0.6f;
    const float ldr = (hdr * hdr * 2.51f + hdr * 0.03f) / (hdr * hdr * 2.43f + hdr * 0.59f + 0.14f);
    return maxf(ldr, 0.0f);
}
__device__ __forceinline__ uint32_t unorm8(float c)
{
    const float v = clampf(c, 0.0f, 1.0f) * 255.0f;
    return (uint32_t)rintf(v == v ? v : 0.0f);
}

__global__ void __launch_bounds__(LP_BLOCK) k_tonemap(TonemapArgs a, const __half *src, uint32_t *dst)
{
    const uint32_t x = a.sc_x0 + blockIdx.x * LP_BLOCK + threadIdx.x, y = a.sc_y0 + blockIdx.y;
    if (x >= a.sc_x1 || y >= a.sc_y1) return;
    const float fx = ((float)x + 0.5f - a.vp_x) / a.vp_w, fy = ((float)y + 0.5f - a.vp_y) / a.vp_h;
    const float nx = 2.0f * fx - 1.0f, ny = 1.0f - 2.0f * fy;
    if (!(fabsf(nx) <= a.scale_x && fabsf(ny) <= a.scale_y)) return;   // outside the quad
    const float u = (nx / a.scale_x + 1.0f) * 0.5f, v = (1.0f - ny / a.scale_y) * 0.5f;
    // linear filter, clamp to edge
    const float sx = u * (float)a.src_w - 0.5f, sy = v * (float)a.src_h - 0.5f;
    const float x0f = floorf(sx), y0f = floorf(sy);
    const float tx = sx - x0f, ty = sy - y0f;
    const int xa = min(max(f2i_sat(x0f), 0), (int)a.src_w - 1), xb = min(max(f2i_sat(x0f) + 1, 0), (int)a.src_w - 1);
    const int ya = min(max(f2i_sat(y0f), 0), (int)a.src_h - 1), yb = min(max(f2i_sat(y0f) + 1, 0), (int)a.src_h - 1);
    const float3 p00 = tonemap_texel(src, a.src_w, xa, ya), p10 = tonemap_texel(src, a.src_w, xb, ya);
    const float3 p01 = tonemap_texel(src, a.src_w, xa, yb), p11 = tonemap_texel(src, a.src_w, xb, yb);
    const float gx = 1.0f - tx, gy = 1.0f - ty;
    float c[3] = {(p00.x * gx + p10.x * tx) * gy + (p01.x * gx + p11.x * tx) * ty,
                  (p00.y * gx + p10.y * tx) * gy + (p01.y * gx + p11.y * tx) * ty,
                  (p00.z * gx + p10.z * tx) * gy + (p01.z * gx + p11.z * tx) * ty};
    const float gain = lpm_powf(2.0f, a.exposure);   // exp2(exposure)
    uint32_t packed = 0xFF000000u;
    for (int k = 0; k < 3; k++)
    {
        float v1 = maxf(c[k], 0.0f);
        if (a.exposure != 0.0f) v1 *= gain;
        if (a.filmic) v1 = filmic1(v1);
        if (a.srgb) v1 = linear_to_srgb1(v1);
        packed |= unorm8(v1) << (8 * k);
    }
    dst[(size_t)y * a.dst_w + x] = packed;
}

// standalone closest-hit probe (bvh_custom.wgsl:7-110)
__global__ void __launch_bounds__(LP_BLOCK) k_trace(SceneDev sc, uint32_t n, const float *ori, const float *dir, float eps,
                                                    uint32_t *out_hit, float *out_dst, float *out_uv, uint32_t *out_inst, uint32_t *out_tri)
{
    extern __shared__ uint32_t lds_stack[];
    uint32_t i = blockIdx.x * LP_BLOCK + threadIdx.x;
    if (i >= n) return;
    f3 o = mk3(ori[i * 3 + 0], ori[i * 3 + 1], ori[i * 3 + 2]);
    f3 d = mk3(dir[i * 3 + 0], dir[i * 3 + 1], dir[i * 3 + 2]);
    Closest c = scene_closest(geo_global(sc), sc, lds_stack, o, d, eps);
    bool hit = c.t != LP_F32_MAX;
    out_hit[i] = hit ? 1u : 0u;
    out_dst[i] = hit ? c.t : 0.0f;
    out_uv[i * 2 + 0] = hit ? c.u : 0.0f;
    out_uv[i * 2 + 1] = hit ? c.v : 0.0f;
    out_inst[i] = hit ? c.inst : 0u;
    out_tri[i] = hit ? (c.tri - sc.meshes[sc.instances[c.inst].mesh_idx].tri_offset) : 0u;
}

// lupin_detmath.h evaluated on the device (tests compare it bit for bit with the host build)
__global__ void __launch_bounds__(LP_BLOCK) k_detmath(int fn, uint32_t n, const float *x, const float *y, float *out)
{
    uint32_t i = blockIdx.x * LP_BLOCK + threadIdx.x;
    if (i >= n) return;
    float a = x[i], b = y[i], r;
    switch (fn)
    {
    case 0: r = lpm_sinf(a); break;
    case 1: r = lpm_cosf(a); break;
    case 2: r = lpm_atanf(a); break;
    case 3: r = lpm_atan2f(a, b); break;
    case 4: r = lpm_acosf(a); break;
    case 5: r = lpm_expf(a); break;
    case 6: r = lpm_logf(a); break;
    case 7: r = lpm_powf(a, b); break;
    case 8: r = a / b; break;
    case 9: r = sqrtf(a); break;
    default: r = 0.0f; break;
    }
    out[i] = r;
}

// tile pack / unpack for the multi-GPU gather: tiles t = rank, rank+world, ... in row-major tile order
__global__ void __launch_bounds__(LP_BLOCK) k_pack_tiles(const uint2 *tex, uint2 *packed, uint32_t width, uint32_t height,
                                                         uint32_t tile_px, uint32_t rank, uint32_t world, int unpack)
{
    uint32_t ntx = (width - 1) / tile_px + 1;
    uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t y = blockIdx.y;
    if (x >= width || y >= height) return;
    uint32_t tx = x / tile_px, ty = y / tile_px;
    uint32_t t = ty * ntx + tx;
    if (t % world != rank) return;
    // pixels in owned tiles before tile t
    unsigned long long before = 0;
    uint32_t nty = (height - 1) / tile_px + 1;
    (void)nty;
    // full rows of tiles above: count owned tiles per row analytically would need care at edges;
    // tiles are few (<= a few thousand), a loop is fine.
    for (uint32_t q = rank; q < t; q += world)
    {
        uint32_t qx = (q % ntx) * tile_px, qy = (q / ntx) * tile_px;
        uint32_t w = min(tile_px, width - qx), h = min(tile_px, height - qy);
        before += (unsigned long long)w * h;
    }
    uint32_t ox = tx * tile_px, oy = ty * tile_px;
    uint32_t w = min(tile_px, width - ox);
    unsigned long long pi = before + (unsigned long long)(y - oy) * w + (x - ox);
    if (unpack) const_cast<uint2 *>(tex)[(size_t)y * width + x] = packed[pi];
    else packed[pi] = tex[(size_t)y * width + x];
}

// ------------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------------

static inline float host_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static thread_local std::string g_last_error;
static int fail(int code, const std::string &msg) { g_last_error = msg; return code; }
#define HIP_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return fail(LUPIN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); } while (0)

#define LP_MAX_LANES 4
// "Lanes" (stream + path buffers + counters) let consecutive pathtrace_scene calls overlap: the wavefront of
// frame k+1 starts while the thin tail of frame k is still draining.  Frames only meet at k_resolve (frame k+1 blends
// with frame k's output), which waits on the previous call's completion event.
struct Lane
{
    hipStream_t stream = nullptr;
    PathBuffers pb{};
    uint64_t capacity = 0;          // slots the path buffers hold
    uint32_t counts_capacity = 0;
    unsigned long long *stat_counters = nullptr;   // per shard: [2s] path bounces, [2s+1] paths
    hipEvent_t done = nullptr;      // recorded after the last kernel of the lane's latest call
    bool used = false;
    FrameParams *d_fp = nullptr;    // this lane's per-call parameters (k_set_params writes, the stage kernels read)
    uint64_t pb_generation = 0;     // bumped when the path buffers are reallocated
    // the lane's wavefront (memset + k_begin + all iterations) as a replayable graph
    struct GraphKey
    {
        uint64_t scene_id = 0, pb_generation = 0;
        uint32_t n = 0, blocks = 0, type = 0, iterations = 0, stack_words = 0, lds = 0, pblocks = 0, refill_min = 0, node_steps = 0;
        int persistent = 0, persistent_shadow = 0, lds_geometry = 0;
        bool operator==(const GraphKey &o) const { return memcmp(this, &o, sizeof(GraphKey)) == 0; }
    } graph_key, seen_key;           // key of graph_exec | key of the lane's previous call
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
};

struct LupinContext
{
    int device = 0;
    hipStream_t stream = nullptr;   // primary stream (= lanes[0].stream): every non-pathtrace operation runs here
    Lane lanes[LP_MAX_LANES];
    int num_lanes = 3;              // LUPIN_LANES=1..4 (LUPIN_OVERLAP=0 == 1 lane)
    uint64_t call_index = 0;
    int last_lane = -1;
    hipEvent_t marker = nullptr;
    bool timing = false;
    int store_rounding = 0;        // LUPIN_STORE_ROUND_TOWARD_ZERO
    bool lds_geometry = true;       // LUPIN_LDS_GEOMETRY=0 keeps small scenes in global memory (A/B runs)
    int persistent_extend = 2;      // LUPIN_EXTEND: "simple" 0 | "persistent" 1 (always) | default 2: persistent for scenes traversed from global memory
    uint32_t num_cus = 256;
    int blocks_per_cu_override = 0; // LUPIN_EXTEND_BLOCKS_PER_CU
    uint32_t refill_min = LP_REFILL_MIN;   // LUPIN_REFILL_MIN
    bool specialize_simple = true;          // LUPIN_SIMPLE_SHADE=0: always the general k_shade
    bool use_graph = false;                 // LUPIN_GRAPH=1: replay the lane-private wavefront as a HIP graph (opt-in)
    bool persistent_shadow = true;          // LUPIN_SHADOW=simple: MIS / Direct shadow rays stay in k_shadow even on large scenes
    uint32_t node_steps = 4;               // LUPIN_NODE_STEPS: node visits per scheduling round of k_extend_persistent
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_extend, ev_shade, ev_total;
    std::vector<hipEvent_t> ev_pool;
    uint64_t extend_launches = 0;
};

struct LupinPathtraceResources
{
    LupinContext *ctx;
    LupinBakedPathtraceParams params;
};

struct LupinTexture
{
    LupinContext *ctx;
    uint32_t width, height;
    __half *data;
};

struct LupinDoubleBufferedTexture
{
    LupinContext *ctx;
    LupinTexture *tex[2];
    int front_idx, back_idx;
};

struct LupinScene
{
    LupinContext *ctx;
    SceneDev dev{};
    std::vector<void *> allocations;
    uint32_t stack_entries = 1;
    uint32_t persistent_blocks[4] = {0, 0, 0, 0};   // grid of k_extend_persistent per integrator (lazy)
    uint64_t id = 0;                                // unique per created scene (graph cache key)
    bool all_opaque = false;                        // no instance can have opacity != 1: k_extend<.., OPAQUE> drops the alpha test
    bool simple_matte = false;                      // only untextured matte materials, no vertex colours, no environments: k_shade<.., SIMPLE>
    bool has_sw_bvh = false;
    bool envs_empty = true, lights_empty = true, instances_empty = true;
};

template <typename T>
static int upload(LupinScene *sc, const std::vector<T> &host, const T **out)
{
    *out = nullptr;
    size_t bytes = std::max<size_t>(host.size() * sizeof(T), sizeof(T) > 16 ? sizeof(T) : 16);
    void *d = nullptr;
    HIP_TRY(hipMalloc(&d, bytes));
    sc->allocations.push_back(d);
    HIP_TRY(hipMemsetAsync(d, 0, bytes, sc->ctx->stream));
    if (!host.empty()) HIP_TRY(hipMemcpyAsync(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice, sc->ctx->stream));
    *out = reinterpret_cast<const T *>(d);
    return LUPIN_OK;
}

static hipEvent_t get_event(LupinContext *ctx)
{
    if (!ctx->ev_pool.empty()) { hipEvent_t e = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

static int ensure_path_buffers(LupinContext *ctx0, Lane *ctx, uint64_t slots, uint32_t iterations)
{
    (void)ctx0;
    // shard segments are whole blocks: round the queue length up to LP_SHARDS * LP_BLOCK
    const uint64_t per_round = (uint64_t)LP_SHARDS * LP_BLOCK;
    slots = (slots + per_round - 1) / per_round * per_round;
    if (slots > ctx->capacity)
    {
        PathBuffers &pb = ctx->pb;
        void **ptrs[] = {(void **)&pb.ori_rng, (void **)&pb.dir_meta, (void **)&pb.weight, (void **)&pb.radiance, (void **)&pb.color,
                         (void **)&pb.hit, (void **)&pb.hit_tri, (void **)&pb.vol0, (void **)&pb.vol1, (void **)&pb.next_hit,
                         (void **)&pb.next_tri, (void **)&pb.queue[0], (void **)&pb.queue[1],
                         (void **)&pb.sh_org, (void **)&pb.sh_d0, (void **)&pb.sh_f0, (void **)&pb.sh_d1, (void **)&pb.sh_f1, (void **)&pb.sh_hit1};
        size_t elem[] = {16, 16, 16, 16, 16, 16, 4, 16, 16, 16, 4, 4, 4, 16, 16, 16, 16, 16, 16};
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < 19; k++)
        {
            if (*ptrs[k]) { hipFree(*ptrs[k]); *ptrs[k] = nullptr; }
            HIP_TRY(hipMalloc(ptrs[k], (size_t)slots * elem[k]));
        }
        ctx->capacity = slots;
        ctx->pb_generation++;
    }
    if (iterations + 2 > ctx->counts_capacity)
    {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->pb.counts) hipFree(ctx->pb.counts);
        ctx->pb.counts = nullptr;
        HIP_TRY(hipMalloc((void **)&ctx->pb.counts, (size_t)(iterations + 2) * LP_SHARDS * sizeof(uint32_t)));
        ctx->counts_capacity = iterations + 2;
        ctx->pb_generation++;
    }
    return LUPIN_OK;
}

// depth of a hierarchy in internal levels = worst-case number of parked far children
static uint32_t blas_depth(const LupinBvhNode *nodes, uint32_t count)
{
    if (count == 0) return 0;
    uint32_t best = 0;
    std::vector<std::pair<uint32_t, uint32_t>> st;
    st.push_back({0u, 0u});
    while (!st.empty())
    {
        auto [n, d] = st.back();
        st.pop_back();
        if (nodes[n].tri_count == 0)
        {
            best = std::max(best, d + 1);
            st.push_back({nodes[n].tri_begin_or_first_child, d + 1});
            st.push_back({nodes[n].tri_begin_or_first_child + 1, d + 1});
        }
    }
    return best;
}

// grid of the persistent tracer: as many blocks as the device keeps resident with this scene's traversal-stack size
// (whole waves per shard); queried once per scene and integrator, outside any stream capture
template <int TYPE, bool LDSGEO>
static uint32_t persistent_grid_t(LupinContext *ctx, const LupinScene *scene, size_t lds)
{
    uint32_t &cached = const_cast<LupinScene *>(scene)->persistent_blocks[TYPE];
    if (cached == 0)
    {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_extend_persistent<TYPE, LDSGEO, 0>, LP_BLOCK, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        if (ctx->blocks_per_cu_override > 0) per_cu = ctx->blocks_per_cu_override;
        cached = std::max(64u, ctx->num_cus * (uint32_t)per_cu / 64u * 64u);
    }
    return cached;
}
static bool use_persistent(const LupinContext *ctx, bool lds_geo) { return ctx->persistent_extend == 1 || (ctx->persistent_extend == 2 && !lds_geo); }
static uint32_t persistent_grid(LupinContext *ctx, const LupinScene *scene, uint32_t type, bool lds_geo, size_t lds)
{
    if (!use_persistent(ctx, lds_geo)) return 0;
    switch (type)
    {
    case LUPIN_PATHTRACE_STANDARD: return lds_geo ? persistent_grid_t<LUPIN_PATHTRACE_STANDARD, true>(ctx, scene, lds) : persistent_grid_t<LUPIN_PATHTRACE_STANDARD, false>(ctx, scene, lds);
    case LUPIN_PATHTRACE_MIS: return lds_geo ? persistent_grid_t<LUPIN_PATHTRACE_MIS, true>(ctx, scene, lds) : persistent_grid_t<LUPIN_PATHTRACE_MIS, false>(ctx, scene, lds);
    case LUPIN_PATHTRACE_NAIVE: return lds_geo ? persistent_grid_t<LUPIN_PATHTRACE_NAIVE, true>(ctx, scene, lds) : persistent_grid_t<LUPIN_PATHTRACE_NAIVE, false>(ctx, scene, lds);
    default: return lds_geo ? persistent_grid_t<LUPIN_PATHTRACE_DIRECT, true>(ctx, scene, lds) : persistent_grid_t<LUPIN_PATHTRACE_DIRECT, false>(ctx, scene, lds);
    }
}

template <int TYPE, bool LDSGEO>
static void launch_iteration_t(LupinContext *ctx, Lane *ln, const LupinScene *scene, uint32_t blocks, uint32_t pblocks, size_t lds, uint32_t stack_words, uint32_t iter)
{
    hipStream_t st = ln->stream;
    const FrameParams *fp = ln->d_fp;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    if (ctx->timing) { e0 = get_event(ctx); e1 = get_event(ctx); e2 = get_event(ctx); hipEventRecord(e0, st); }
    const bool persistent = pblocks != 0;
    if (persistent)
        hipLaunchKernelGGL((k_extend_persistent<TYPE, LDSGEO, 0>), dim3(pblocks), dim3(LP_BLOCK), lds, st,
                           scene->dev, fp, ln->pb, iter, ln->stat_counters, ctx->refill_min, stack_words, ctx->node_steps);
    else
    {
        if (scene->all_opaque && ctx->specialize_simple)
            hipLaunchKernelGGL((k_extend<TYPE, LDSGEO, true>), dim3(blocks), dim3(LP_BLOCK), lds, st, scene->dev, fp, ln->pb, iter, ln->stat_counters, stack_words);
        else
            hipLaunchKernelGGL((k_extend<TYPE, LDSGEO, false>), dim3(blocks), dim3(LP_BLOCK), lds, st, scene->dev, fp, ln->pb, iter, ln->stat_counters, stack_words);
    }
    if (ctx->timing) hipEventRecord(e1, st);
    if (scene->simple_matte && ctx->specialize_simple)
        hipLaunchKernelGGL((k_shade<TYPE, LDSGEO, true>), dim3(blocks), dim3(LP_BLOCK), lds, st, scene->dev, fp, ln->pb, iter, ln->stat_counters, stack_words);
    else
        hipLaunchKernelGGL((k_shade<TYPE, LDSGEO, false>), dim3(blocks), dim3(LP_BLOCK), lds, st, scene->dev, fp, ln->pb, iter, ln->stat_counters, stack_words);
    if constexpr (TYPE == LUPIN_PATHTRACE_MIS || TYPE == LUPIN_PATHTRACE_DIRECT)   // shadow rays + path finish (booked with "shade" in the timing)
    {
        if (persistent && ctx->persistent_shadow)
        {
            // large scenes: the shadow rays go through the phase-scheduled persistent tracer as well, then a light finish pass
            hipLaunchKernelGGL((k_extend_persistent<TYPE, LDSGEO, 1>), dim3(pblocks), dim3(LP_BLOCK), lds, st,
                               scene->dev, fp, ln->pb, iter, ln->stat_counters, ctx->refill_min, stack_words, ctx->node_steps);
            hipLaunchKernelGGL((k_shadow<TYPE, LDSGEO, true>), dim3(blocks), dim3(LP_BLOCK), lds, st, scene->dev, fp, ln->pb, iter, stack_words);
        }
        else
            hipLaunchKernelGGL((k_shadow<TYPE, LDSGEO, false>), dim3(blocks), dim3(LP_BLOCK), lds, st, scene->dev, fp, ln->pb, iter, stack_words);
    }
    if (ctx->timing)
    {
        hipEventRecord(e2, st);
        ctx->ev_extend.push_back({e0, e1});
        ctx->ev_shade.push_back({e1, e2});
    }
    ctx->extend_launches++;
}

template <int TYPE>
static void launch_iteration(LupinContext *ctx, Lane *ln, const LupinScene *scene, bool lds_geo, uint32_t blocks, uint32_t pblocks, size_t lds, uint32_t stack_words, uint32_t iter)
{
    if (lds_geo) launch_iteration_t<TYPE, true>(ctx, ln, scene, blocks, pblocks, lds, stack_words, iter);
    else launch_iteration_t<TYPE, false>(ctx, ln, scene, blocks, pblocks, lds, stack_words, iter);
}

// the lane-private part of one call: clear the queue counters, first rays, every iteration of the wavefront
static hipError_t enqueue_wavefront(LupinContext *ctx, Lane *ln, const LupinScene *scene, uint32_t pathtrace_type, bool lds_geo, uint32_t n,
                                    uint32_t blocks, uint32_t pblocks, size_t lds, uint32_t stack_words, uint32_t iterations)
{
    hipStream_t st = ln->stream;
    hipError_t e = hipMemsetAsync(ln->pb.counts, 0, (size_t)ln->counts_capacity * LP_SHARDS * sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_begin, dim3(blocks), dim3(LP_BLOCK), 0, st, (const FrameParams *)ln->d_fp, ln->pb, n);
    for (uint32_t it = 0; it < iterations; it++)
    {
        switch (pathtrace_type)
        {
        case LUPIN_PATHTRACE_STANDARD: launch_iteration<LUPIN_PATHTRACE_STANDARD>(ctx, ln, scene, lds_geo, blocks, pblocks, lds, stack_words, it); break;
        case LUPIN_PATHTRACE_MIS: launch_iteration<LUPIN_PATHTRACE_MIS>(ctx, ln, scene, lds_geo, blocks, pblocks, lds, stack_words, it); break;
        case LUPIN_PATHTRACE_NAIVE: launch_iteration<LUPIN_PATHTRACE_NAIVE>(ctx, ln, scene, lds_geo, blocks, pblocks, lds, stack_words, it); break;
        default: launch_iteration<LUPIN_PATHTRACE_DIRECT>(ctx, ln, scene, lds_geo, blocks, pblocks, lds, stack_words, it); break;
        }
    }
    return hipSuccess;
}

// The resolves form a chain (each waits for the previous call's), so the latest call's event covers all lanes' texture writes.
static void join_primary(LupinContext *ctx)
{
    if (ctx->last_lane > 0) hipStreamWaitEvent(ctx->stream, ctx->lanes[ctx->last_lane].done, 0);
}
static hipError_t sync_all(LupinContext *ctx)
{
    hipError_t e = hipSuccess;
    for (int k = 0; k < LP_MAX_LANES && e == hipSuccess; k++)
        if (ctx->lanes[k].stream) e = hipStreamSynchronize(ctx->lanes[k].stream);
    return e;
}

#include "lupin_internal.hpp"
int lupin_internal_fail(int code, const char *msg) { return fail(code, msg); }
int lupin_internal_ctx_device(const LupinContext *ctx) { return ctx->device; }
hipStream_t lupin_internal_ctx_stream(const LupinContext *ctx) { return ctx->stream; }

extern "C" {

const char *lupin_hip_last_error(void) { return g_last_error.c_str(); }

int lupin_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lupin_hip_create_context(int device_ordinal, LupinContext **out_ctx)
{
    if (!out_ctx) return fail(LUPIN_ERR_INVALID_ARGUMENT, "out_ctx is null");
    int n = lupin_hip_device_count();
    if (n <= 0) return fail(LUPIN_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    if (device_ordinal < 0 || device_ordinal >= n) return fail(LUPIN_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device_ordinal));
    LupinContext *ctx = new LupinContext();
    ctx->device = device_ordinal;
    hipError_t e = hipSuccess;
    const char *ov = getenv("LUPIN_OVERLAP");
    const char *nl = getenv("LUPIN_LANES");
    if (nl) ctx->num_lanes = std::min(LP_MAX_LANES, std::max(1, atoi(nl)));
    if (ov && strcmp(ov, "0") == 0) ctx->num_lanes = 1;
    for (int k = 0; k < ctx->num_lanes && e == hipSuccess; k++)
    {
        Lane &ln = ctx->lanes[k];
        e = hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipMalloc((void **)&ln.d_fp, sizeof(FrameParams));
        if (e == hipSuccess) e = hipMalloc((void **)&ln.stat_counters, 2 * LP_SHARDS * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemsetAsync(ln.stat_counters, 0, 2 * LP_SHARDS * sizeof(unsigned long long), ln.stream);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->marker, hipEventDisableTiming);
    if (e != hipSuccess) { delete ctx; return fail(LUPIN_ERR_HIP, std::string("context setup: ") + hipGetErrorString(e)); }
    ctx->stream = ctx->lanes[0].stream;
    const char *ext = getenv("LUPIN_EXTEND");
    if (ext && strcmp(ext, "persistent") == 0) ctx->persistent_extend = 1;
    else if (ext && strcmp(ext, "simple") == 0) ctx->persistent_extend = 0;
    const char *lg = getenv("LUPIN_LDS_GEOMETRY");
    ctx->lds_geometry = !(lg && strcmp(lg, "0") == 0);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0)
    {
        const char *bpc = getenv("LUPIN_EXTEND_BLOCKS_PER_CU");
        if (bpc) ctx->blocks_per_cu_override = std::max(0, atoi(bpc));
        ctx->num_cus = (uint32_t)prop.multiProcessorCount;
    }
    const char *ssh = getenv("LUPIN_SIMPLE_SHADE");
    if (ssh && strcmp(ssh, "0") == 0) ctx->specialize_simple = false;
    const char *gr = getenv("LUPIN_GRAPH");
    if (gr) ctx->use_graph = strcmp(gr, "0") != 0;
    const char *shd = getenv("LUPIN_SHADOW");
    if (shd && strcmp(shd, "simple") == 0) ctx->persistent_shadow = false;
    const char *ns = getenv("LUPIN_NODE_STEPS");
    if (ns) ctx->node_steps = (uint32_t)std::min(16, std::max(1, atoi(ns)));
    const char *rm = getenv("LUPIN_REFILL_MIN");
    if (rm) ctx->refill_min = (uint32_t)std::min(64, std::max(1, atoi(rm)));
    *out_ctx = ctx;
    return LUPIN_OK;
}

void lupin_hip_destroy_context(LupinContext *ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    sync_all(ctx);
    for (int k = 0; k < LP_MAX_LANES; k++)
    {
        PathBuffers &pb = ctx->lanes[k].pb;
        void *ptrs[] = {pb.ori_rng, pb.dir_meta, pb.weight, pb.radiance, pb.color, pb.hit, pb.hit_tri, pb.vol0, pb.vol1,
                        pb.next_hit, pb.next_tri, pb.queue[0], pb.queue[1], pb.counts, ctx->lanes[k].stat_counters,
                        pb.sh_org, pb.sh_d0, pb.sh_f0, pb.sh_d1, pb.sh_f1, pb.sh_hit1};
        for (void *p : ptrs) if (p) hipFree(p);
        if (ctx->lanes[k].done) hipEventDestroy(ctx->lanes[k].done);
        if (ctx->lanes[k].graph_exec) hipGraphExecDestroy(ctx->lanes[k].graph_exec);
        if (ctx->lanes[k].graph) hipGraphDestroy(ctx->lanes[k].graph);
        if (ctx->lanes[k].d_fp) hipFree(ctx->lanes[k].d_fp);
    }
    if (ctx->marker) hipEventDestroy(ctx->marker);
    for (auto &pr : ctx->ev_extend) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto &pr : ctx->ev_shade) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto &pr : ctx->ev_total) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto e : ctx->ev_pool) hipEventDestroy(e);
    for (int k = 0; k < LP_MAX_LANES; k++) if (ctx->lanes[k].stream) hipStreamDestroy(ctx->lanes[k].stream);
    delete ctx;
}

int lupin_hip_sync(LupinContext *ctx)
{
    if (!ctx) return fail(LUPIN_ERR_INVALID_ARGUMENT, "ctx is null");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(sync_all(ctx));
    return LUPIN_OK;
}

int lupin_hip_set_f16_store_rounding(LupinContext *ctx, int mode)
{
    if (!ctx || (mode != 0 && mode != 1)) return fail(LUPIN_ERR_INVALID_ARGUMENT, "mode must be 0 (toward zero) or 1 (nearest even)");
    ctx->store_rounding = mode;
    return LUPIN_OK;
}

int lupin_hip_build_pathtrace_resources(LupinContext *ctx, const LupinBakedPathtraceParams *params, LupinPathtraceResources **out_res)
{
    if (!ctx || !params || !out_res) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    if (params->samples_per_pixel == 0 || params->samples_per_pixel > 0xFFFFu) return fail(LUPIN_ERR_INVALID_ARGUMENT, "samples_per_pixel must be in [1, 65535]");
    if (params->max_bounces >= META_BOUNCE_MASK) return fail(LUPIN_ERR_INVALID_ARGUMENT, "max_bounces must be < 4095");
    LupinPathtraceResources *r = new LupinPathtraceResources();
    r->ctx = ctx;
    r->params = *params;
    *out_res = r;
    return LUPIN_OK;
}
void lupin_hip_destroy_pathtrace_resources(LupinPathtraceResources *res) { delete res; }

// ---- scene upload ----

int lupin_hip_scene_create(LupinContext *ctx, const LupinSceneDesc *desc, LupinScene **out_scene)
{
    if (!ctx || !desc || !out_scene) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const LupinSceneDesc &s = *desc;

    // ---- validation (validate_scene, data_structures.rs:876-928, plus what the kernels index) ----
    for (uint32_t i = 0; i < s.num_instances; i++)
    {
        if (s.instances[i].mesh_idx >= s.num_meshes) return fail(LUPIN_ERR_INVALID_ARGUMENT, "instance mesh_idx out of range");
        if (s.instances[i].mat_idx >= s.num_materials) return fail(LUPIN_ERR_INVALID_ARGUMENT, "instance mat_idx out of range");
    }
    auto tex_ok = [&](uint32_t t) { return t == LUPIN_SENTINEL_IDX || t < s.num_textures; };
    for (uint32_t i = 0; i < s.num_materials; i++)
    {
        const LupinMaterial &m = s.materials[i];
        if (!tex_ok(m.color_tex_idx) || !tex_ok(m.emission_tex_idx) || !tex_ok(m.roughness_tex_idx) || !tex_ok(m.scattering_tex_idx) || !tex_ok(m.normal_tex_idx))
            return fail(LUPIN_ERR_INVALID_ARGUMENT, "material texture index out of range");
    }
    if (s.num_environments > LUPIN_MAX_ENVS) return fail(LUPIN_ERR_INVALID_ARGUMENT, "too many environments");
    for (uint32_t i = 0; i < s.num_environments; i++)
    {
        const LupinEnvironment &e = s.environments[i];
        if (!tex_ok(e.emission_tex_idx)) return fail(LUPIN_ERR_INVALID_ARGUMENT, "environment texture index out of range");
        if (e.emission_tex_idx != LUPIN_SENTINEL_IDX)
        {
            const LupinTextureDesc &t = s.textures[e.emission_tex_idx];
            if (!s.env_alias_tables || s.env_alias_tables[i].num_bins != t.width * t.height)
                return fail(LUPIN_ERR_INVALID_ARGUMENT, "environment alias table must have one bin per texel");
        }
    }
    for (uint32_t i = 0; i < s.num_lights; i++)
    {
        if (s.lights[i].instance_idx >= s.num_instances) return fail(LUPIN_ERR_INVALID_ARGUMENT, "light instance_idx out of range");
        if (!s.alias_tables || s.alias_tables[i].num_bins == 0) return fail(LUPIN_ERR_INVALID_ARGUMENT, "light without alias table");
        uint32_t mesh = s.instances[s.lights[i].instance_idx].mesh_idx;
        if (s.alias_tables[i].num_bins != s.meshes[mesh].num_indices / 3) return fail(LUPIN_ERR_INVALID_ARGUMENT, "light alias table size != triangle count");
    }
    for (uint32_t i = 0; i < s.num_textures; i++)
        if (s.textures[i].width == 0 || s.textures[i].height == 0 || !s.textures[i].pixels || s.textures[i].format > LUPIN_TEX_RGBA16_FLOAT)
            return fail(LUPIN_ERR_INVALID_ARGUMENT, "bad texture descriptor");

    LupinScene *sc = new LupinScene();
    sc->ctx = ctx;
    sc->instances_empty = s.num_instances == 0;
    sc->lights_empty = s.num_lights == 0;
    sc->envs_empty = s.num_environments == 0;
    sc->has_sw_bvh = s.num_tlas_nodes > 0 || s.num_instances == 0;
    if (s.num_instances > 0 && s.num_tlas_nodes == 0) { delete sc; return fail(LUPIN_ERR_NO_SW_BVH, "scene has instances but no TLAS (software BVH required)"); }

    // ---- vertex attribute pools ----
    std::vector<uint32_t> normal_base(s.num_normal_buffers), uv_base(s.num_texcoord_buffers), color_base(s.num_color_buffers);
    std::vector<float4> normals;
    std::vector<float2> texcoords;
    std::vector<float4> colors;
    for (uint32_t b = 0; b < s.num_normal_buffers; b++)
    {
        normal_base[b] = (uint32_t)normals.size();
        for (uint32_t v = 0; v < s.verts_normal_array[b].num_verts; v++) { const float *p = s.verts_normal_array[b].data + (size_t)v * 4; normals.push_back(make_float4(p[0], p[1], p[2], 0.0f)); }
    }
    for (uint32_t b = 0; b < s.num_texcoord_buffers; b++)
    {
        uv_base[b] = (uint32_t)texcoords.size();
        for (uint32_t v = 0; v < s.verts_texcoord_array[b].num_verts; v++) { const float *p = s.verts_texcoord_array[b].data + (size_t)v * 2; texcoords.push_back(make_float2(p[0], p[1])); }
    }
    for (uint32_t b = 0; b < s.num_color_buffers; b++)
    {
        color_base[b] = (uint32_t)colors.size();
        for (uint32_t v = 0; v < s.verts_color_array[b].num_verts; v++) { const float *p = s.verts_color_array[b].data + (size_t)v * 4; colors.push_back(make_float4(p[0], p[1], p[2], p[3])); }
    }

    // ---- meshes: triangles in leaf order, BLAS as 64-byte child-pair nodes ----
    std::vector<TriVerts> tris;
    std::vector<uint32_t> tri_indices;
    std::vector<WideNode> blas;
    std::vector<MeshDev> meshes(s.num_meshes);
    std::vector<uint32_t> mesh_root(s.num_meshes);
    uint32_t max_blas_depth = 0;
    for (uint32_t mi = 0; mi < s.num_meshes; mi++)
    {
        const LupinMeshDesc &m = s.meshes[mi];
        const LupinMeshInfo &info = s.mesh_infos[mi];
        MeshDev md;
        md.tri_offset = (uint32_t)tris.size();
        auto attr_base = [&](uint32_t idx, const std::vector<uint32_t> &bases, uint32_t nbuf, const LupinVertexBufferDesc *bufs, bool &ok) -> uint32_t {
            if (idx == LUPIN_SENTINEL_IDX) return LUPIN_SENTINEL_IDX;
            if (idx >= nbuf || bufs[idx].num_verts != m.num_verts) { ok = false; return LUPIN_SENTINEL_IDX; }
            return bases[idx];
        };
        bool ok = true;
        md.normals_base = attr_base(info.normals_buf_idx, normal_base, s.num_normal_buffers, s.verts_normal_array, ok);
        md.texcoords_base = attr_base(info.texcoords_buf_idx, uv_base, s.num_texcoord_buffers, s.verts_texcoord_array, ok);
        md.colors_base = attr_base(info.colors_buf_idx, color_base, s.num_color_buffers, s.verts_color_array, ok);
        if (!ok) { lupin_hip_scene_destroy(sc); return fail(LUPIN_ERR_INVALID_ARGUMENT, "mesh attribute buffer index / size mismatch"); }
        meshes[mi] = md;

        uint32_t ntris = m.num_indices / 3;
        for (uint32_t i = 0; i < ntris * 3; i++)
            if (m.indices[i] >= m.num_verts) { lupin_hip_scene_destroy(sc); return fail(LUPIN_ERR_INVALID_ARGUMENT, "vertex index out of range"); }
        for (uint32_t t = 0; t < ntris; t++)
        {
            TriVerts tv;
            const float *p0 = m.verts_pos + (size_t)m.indices[t * 3 + 0] * 4;
            const float *p1 = m.verts_pos + (size_t)m.indices[t * 3 + 1] * 4;
            const float *p2 = m.verts_pos + (size_t)m.indices[t * 3 + 2] * 4;
            tv.v0 = make_float4(p0[0], p0[1], p0[2], 0.0f);
            tv.v1 = make_float4(p1[0], p1[1], p1[2], 0.0f);
            tv.v2 = make_float4(p2[0], p2[1], p2[2], 0.0f);
            tris.push_back(tv);
            tri_indices.push_back(m.indices[t * 3 + 0]);
            tri_indices.push_back(m.indices[t * 3 + 1]);
            tri_indices.push_back(m.indices[t * 3 + 2]);
        }
        if (ntris == 0 || m.num_bvh_nodes == 0)
        {
            // degenerate mesh: one never-hit triangle so that traversal has a well-formed leaf
            TriVerts tv;
            tv.v0 = make_float4(0, 0, 0, host_u2f(LEAF_END_BITS));
            tv.v1 = tv.v2 = make_float4(0, 0, 0, 0);
            mesh_root[mi] = REF_LEAF | (uint32_t)tris.size();
            tris.push_back(tv);
            tri_indices.push_back(0); tri_indices.push_back(0); tri_indices.push_back(0);
            continue;
        }
        // node index -> child reference
        std::vector<uint32_t> ref(m.num_bvh_nodes);
        uint32_t wide_base = (uint32_t)blas.size(), wide_count = 0;
        for (uint32_t n = 0; n < m.num_bvh_nodes; n++)
        {
            const LupinBvhNode &nd = m.bvh_nodes[n];
            if (nd.tri_count > 0)
            {
                if ((uint64_t)nd.tri_begin_or_first_child + nd.tri_count > ntris) { lupin_hip_scene_destroy(sc); return fail(LUPIN_ERR_INVALID_ARGUMENT, "BLAS leaf range out of bounds"); }
                ref[n] = REF_LEAF | (md.tri_offset + nd.tri_begin_or_first_child);
                uint32_t last = md.tri_offset + nd.tri_begin_or_first_child + nd.tri_count - 1;
                tris[last].v0.w = host_u2f(LEAF_END_BITS);
            }
            else
            {
                if ((uint64_t)nd.tri_begin_or_first_child + 1 >= m.num_bvh_nodes) { lupin_hip_scene_destroy(sc); return fail(LUPIN_ERR_INVALID_ARGUMENT, "BLAS child index out of bounds"); }
                ref[n] = wide_base + wide_count++;
            }
        }
        blas.resize(wide_base + wide_count);
        for (uint32_t n = 0; n < m.num_bvh_nodes; n++)
        {
            const LupinBvhNode &nd = m.bvh_nodes[n];
            if (nd.tri_count > 0) continue;
            const LupinBvhNode &l = m.bvh_nodes[nd.tri_begin_or_first_child];
            const LupinBvhNode &r = m.bvh_nodes[nd.tri_begin_or_first_child + 1];
            WideNode w;
            w.a = make_float4(l.aabb_min[0], l.aabb_min[1], l.aabb_min[2], l.aabb_max[0]);
            w.b = make_float4(l.aabb_max[1], l.aabb_max[2], r.aabb_min[0], r.aabb_min[1]);
            w.c = make_float4(r.aabb_min[2], r.aabb_max[0], r.aabb_max[1], r.aabb_max[2]);
            w.d = make_uint4(ref[nd.tri_begin_or_first_child], ref[nd.tri_begin_or_first_child + 1], 0u, 0u);
            blas[ref[n]] = w;
        }
        mesh_root[mi] = ref[0];
        max_blas_depth = std::max(max_blas_depth, blas_depth(m.bvh_nodes, m.num_bvh_nodes));
    }

    // ---- TLAS ----
    std::vector<WideNode> tlas;
    uint32_t tlas_root = REF_LEAF;
    uint32_t tlas_depth = 0;
    if (s.num_tlas_nodes > 0)
    {
        std::vector<uint32_t> ref(s.num_tlas_nodes);
        uint32_t wide_count = 0;
        for (uint32_t n = 0; n < s.num_tlas_nodes; n++)
        {
            const LupinTlasNode &nd = s.tlas_nodes[n];
            if (nd.left == 0)
            {
                if (nd.instance_idx >= s.num_instances) { lupin_hip_scene_destroy(sc); return fail(LUPIN_ERR_INVALID_ARGUMENT, "TLAS leaf instance out of range"); }
                ref[n] = REF_LEAF | nd.instance_idx;
            }
            else
            {
                if (nd.left >= s.num_tlas_nodes || nd.right >= s.num_tlas_nodes) { lupin_hip_scene_destroy(sc); return fail(LUPIN_ERR_INVALID_ARGUMENT, "TLAS child out of range"); }
                ref[n] = wide_count++;
            }
        }
        tlas.resize(wide_count);
        for (uint32_t n = 0; n < s.num_tlas_nodes; n++)
        {
            const LupinTlasNode &nd = s.tlas_nodes[n];
            if (nd.left == 0) continue;
            const LupinTlasNode &l = s.tlas_nodes[nd.left];
            const LupinTlasNode &r = s.tlas_nodes[nd.right];
            WideNode w;
            w.a = make_float4(l.aabb_min[0], l.aabb_min[1], l.aabb_min[2], l.aabb_max[0]);
            w.b = make_float4(l.aabb_max[1], l.aabb_max[2], r.aabb_min[0], r.aabb_min[1]);
            w.c = make_float4(r.aabb_min[2], r.aabb_max[0], r.aabb_max[1], r.aabb_max[2]);
            w.d = make_uint4(ref[nd.left], ref[nd.right], 0u, 0u);
            tlas[ref[n]] = w;
        }
        tlas_root = ref[0];
        // depth from the root (bounded walk: a malformed cyclic TLAS is rejected)
        std::vector<std::pair<uint32_t, uint32_t>> st;
        st.push_back({0u, 0u});
        uint64_t visited = 0;
        while (!st.empty())
        {
            auto [n, d] = st.back();
            st.pop_back();
            if (++visited > (uint64_t)s.num_tlas_nodes * 2 + 2) { lupin_hip_scene_destroy(sc); return fail(LUPIN_ERR_INVALID_ARGUMENT, "TLAS is not a tree"); }
            if (s.tlas_nodes[n].left != 0)
            {
                tlas_depth = std::max(tlas_depth, d + 1);
                st.push_back({s.tlas_nodes[n].left, d + 1});
                st.push_back({s.tlas_nodes[n].right, d + 1});
            }
        }
    }
    sc->stack_entries = tlas_depth + max_blas_depth + 1;

    // ---- instances ----
    std::vector<InstanceDev> instances(s.num_instances);
    uint32_t mat_types_seen = 0;
    bool any_alpha = false;
    for (uint32_t i = 0; i < s.num_instances; i++)
    {
        const LupinInstance &in = s.instances[i];
        const float (*m)[4] = in.transpose_inverse_transform.m;
        InstanceDev d;
        d.r0 = make_float4(m[0][0], m[0][1], m[0][2], m[0][3]);
        d.r1 = make_float4(m[1][0], m[1][1], m[1][2], m[1][3]);
        d.r2 = make_float4(m[2][0], m[2][1], m[2][2], m[2][3]);
        d.blas_root = mesh_root[in.mesh_idx];
        d.mat_idx = in.mat_idx;
        d.mesh_idx = in.mesh_idx;
        const LupinMaterial &mat = s.materials[in.mat_idx];
        bool maybe_alpha = !(mat.color[3] == 1.0f) ||
                           (mat.color_tex_idx != LUPIN_SENTINEL_IDX && meshes[in.mesh_idx].texcoords_base != LUPIN_SENTINEL_IDX) ||
                           meshes[in.mesh_idx].colors_base != LUPIN_SENTINEL_IDX;
        any_alpha = any_alpha || maybe_alpha;
        d.flags = (maybe_alpha ? 1u : 0u) | ((mat.mat_type & 0xFu) << 8);   // bits 8..11: material type = k_shade's sort key
        mat_types_seen |= 1u << (mat.mat_type & 0xFu);
        instances[i] = d;
    }

    // ---- textures ----
    std::vector<TextureDev> textures(s.num_textures);
    std::vector<uint8_t> texels;
    for (uint32_t i = 0; i < s.num_textures; i++)
    {
        const LupinTextureDesc &t = s.textures[i];
        size_t bpp = (t.format == LUPIN_TEX_RGBA8_UNORM) ? 4 : 8;
        size_t bytes = (size_t)t.width * t.height * bpp;
        size_t off = (texels.size() + 15) & ~(size_t)15;
        texels.resize(off + bytes);
        memcpy(texels.data() + off, t.pixels, bytes);
        textures[i].offset = off;
        textures[i].width = t.width;
        textures[i].height = t.height;
        textures[i].format = t.format;
        textures[i].pad = 0;
    }

    // ---- lights ----
    std::vector<AliasRange> alias_ranges(s.num_lights), env_alias_ranges(s.num_environments);
    std::vector<LupinAliasBin> alias_bins;
    for (uint32_t i = 0; i < s.num_lights; i++)
    {
        alias_ranges[i] = {(uint32_t)alias_bins.size(), s.alias_tables[i].num_bins};
        alias_bins.insert(alias_bins.end(), s.alias_tables[i].bins, s.alias_tables[i].bins + s.alias_tables[i].num_bins);
    }
    for (uint32_t i = 0; i < s.num_environments; i++)
    {
        uint32_t nb = s.env_alias_tables ? s.env_alias_tables[i].num_bins : 0;
        env_alias_ranges[i] = {(uint32_t)alias_bins.size(), nb};
        if (nb) alias_bins.insert(alias_bins.end(), s.env_alias_tables[i].bins, s.env_alias_tables[i].bins + nb);
    }

    // Conservative world-space bounding sphere of every light instance (all mesh vertices through the inverse of the
    // stored world->local rows, in double, radius padded): lights_pdf skips lights whose sphere the ray cannot reach.
    // A skipped light contributes exactly +0.0f in the reference's sum, so results do not change.
    std::vector<float4> light_bounds(s.num_lights);
    for (uint32_t i = 0; i < s.num_lights; i++)
    {
        const LupinInstance &in = s.instances[s.lights[i].instance_idx];
        const LupinMeshDesc &m = s.meshes[in.mesh_idx];
        const float (*r)[4] = in.transpose_inverse_transform.m;   // 3 rows x 4: world -> local
        const double a = r[0][0], b = r[0][1], c = r[0][2], d = r[1][0], e = r[1][1], f = r[1][2], g = r[2][0], h = r[2][1], k = r[2][2];
        const double det = a * (e * k - f * h) - b * (d * k - f * g) + c * (d * h - e * g);
        float4 bound = make_float4(0.0f, 0.0f, 0.0f, INFINITY);   // singular / non-finite transform: never culled
        if (std::isfinite(det) && det != 0.0 && m.num_verts > 0)
        {
            const double inv[3][3] = {{(e * k - f * h) / det, (c * h - b * k) / det, (b * f - c * e) / det},
                                      {(f * g - d * k) / det, (a * k - c * g) / det, (c * d - a * f) / det},
                                      {(d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det}};
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            bool finite = true;
            for (uint32_t v = 0; v < m.num_verts; v++)
            {
                const double q[3] = {m.verts_pos[4 * v + 0] - (double)r[0][3], m.verts_pos[4 * v + 1] - (double)r[1][3], m.verts_pos[4 * v + 2] - (double)r[2][3]};
                for (int ax = 0; ax < 3; ax++)
                {
                    const double w = inv[ax][0] * q[0] + inv[ax][1] * q[1] + inv[ax][2] * q[2];
                    finite = finite && std::isfinite(w);
                    lo[ax] = std::min(lo[ax], w); hi[ax] = std::max(hi[ax], w);
                }
            }
            if (finite)
            {
                const double cx = 0.5 * (lo[0] + hi[0]), cy = 0.5 * (lo[1] + hi[1]), cz = 0.5 * (lo[2] + hi[2]);
                const double rad = 0.5 * std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
                const double cmag = std::fabs(cx) + std::fabs(cy) + std::fabs(cz);
                bound = make_float4((float)cx, (float)cy, (float)cz, (float)(rad * 1.02 + 1e-4 * (cmag + rad) + 1e-6));
            }
        }
        light_bounds[i] = bound;
    }

    // small scenes: one blob [tlas | blas | tris | instances] in 16-byte words for LDS staging
    std::vector<float4> geo_blob;
    uint32_t off_blas = 0, off_tris = 0, off_inst = 0;
    {
        const size_t bytes = tlas.size() * 64 + blas.size() * 64 + tris.size() * 48 + instances.size() * 64;
        if (bytes > 0 && bytes <= LP_GEO_LDS_LIMIT)
        {
            auto append = [&](const void *p, size_t nbytes) {
                const float4 *f = reinterpret_cast<const float4 *>(p);
                geo_blob.insert(geo_blob.end(), f, f + nbytes / 16);
            };
            append(tlas.data(), tlas.size() * 64);
            off_blas = (uint32_t)geo_blob.size();
            append(blas.data(), blas.size() * 64);
            off_tris = (uint32_t)geo_blob.size();
            append(tris.data(), tris.size() * 48);
            off_inst = (uint32_t)geo_blob.size();
            append(instances.data(), instances.size() * 64);
        }
    }

    SceneDev &dv = sc->dev;
    int rc = LUPIN_OK;
    std::vector<LupinMaterial> materials(s.materials, s.materials + s.num_materials);
    std::vector<LupinEnvironment> envs(s.environments, s.environments + s.num_environments);
    std::vector<LupinLight> lights(s.lights, s.lights + s.num_lights);
    if ((rc = upload(sc, tlas, &dv.tlas)) || (rc = upload(sc, blas, &dv.blas)) || (rc = upload(sc, tris, &dv.tris)) ||
        (rc = upload(sc, tri_indices, &dv.tri_indices)) || (rc = upload(sc, instances, &dv.instances)) ||
        (rc = upload(sc, meshes, &dv.meshes)) || (rc = upload(sc, materials, &dv.materials)) ||
        (rc = upload(sc, normals, &dv.normals)) || (rc = upload(sc, texcoords, &dv.texcoords)) || (rc = upload(sc, colors, &dv.colors)) ||
        (rc = upload(sc, textures, &dv.textures)) || (rc = upload(sc, texels, &dv.texels)) ||
        (rc = upload(sc, envs, &dv.environments)) || (rc = upload(sc, lights, &dv.lights)) ||
        (rc = upload(sc, alias_ranges, &dv.alias_ranges)) || (rc = upload(sc, env_alias_ranges, &dv.env_alias_ranges)) ||
        (rc = upload(sc, alias_bins, &dv.alias_bins)) || (rc = upload(sc, geo_blob, &dv.geo_blob)) ||
        (rc = upload(sc, light_bounds, &dv.light_bounds)))
    {
        lupin_hip_scene_destroy(sc);
        return rc;
    }
    dv.tlas_root = tlas_root;
    dv.num_lights = s.num_lights;
    dv.num_envs = s.num_environments;
    dv.num_instances = s.num_instances;
    {
        // simple_matte: every instance's material is matte with no texture reference, no mesh carries vertex colours, and
        // there is no environment -- then material type, texture use and environment terms are constants of the scene
        bool simple = s.num_instances > 0 && s.num_environments == 0 && s.num_color_buffers == 0;
        for (uint32_t i = 0; i < s.num_instances && simple; i++)
        {
            const LupinMaterial &mat = s.materials[s.instances[i].mat_idx];
            simple = mat.mat_type == LUPIN_MAT_MATTE && mat.color_tex_idx == LUPIN_SENTINEL_IDX && mat.emission_tex_idx == LUPIN_SENTINEL_IDX &&
                     mat.roughness_tex_idx == LUPIN_SENTINEL_IDX && mat.scattering_tex_idx == LUPIN_SENTINEL_IDX && mat.normal_tex_idx == LUPIN_SENTINEL_IDX;
        }
        sc->simple_matte = simple;
        sc->all_opaque = !any_alpha;
    }
    dv.sort_shade = __builtin_popcount(mat_types_seen) >= 4;   // pays off from about four BSDF families (measured: 3 lose 10 %, 8 win 17 % of k_shade)
    { const char *ss = getenv("LUPIN_SORT_SHADE"); if (ss) dv.sort_shade = strcmp(ss, "0") != 0; }
    dv.geo_blob_words = (uint32_t)geo_blob.size();
    dv.geo_off_blas = off_blas; dv.geo_off_tris = off_tris; dv.geo_off_inst = off_inst;
    static uint64_t next_scene_id = 1;
    sc->id = next_scene_id++;
    hipError_t e = hipStreamSynchronize(ctx->stream);   // host vectors go out of scope
    if (e != hipSuccess) { lupin_hip_scene_destroy(sc); return fail(LUPIN_ERR_HIP, hipGetErrorString(e)); }
    *out_scene = sc;
    return LUPIN_OK;
}

void lupin_hip_scene_destroy(LupinScene *scene)
{
    if (!scene) return;
    hipSetDevice(scene->ctx->device);
    sync_all(scene->ctx);
    for (void *p : scene->allocations) hipFree(p);
    delete scene;
}

// ---- textures / double buffering ----

int lupin_hip_texture_create(LupinContext *ctx, uint32_t width, uint32_t height, LupinTexture **out_tex)
{
    if (!ctx || !out_tex || width == 0 || height == 0) return fail(LUPIN_ERR_INVALID_ARGUMENT, "bad texture size");
    HIP_TRY(hipSetDevice(ctx->device));
    LupinTexture *t = new LupinTexture();
    t->ctx = ctx; t->width = width; t->height = height; t->data = nullptr;
    size_t bytes = (size_t)width * height * 4 * sizeof(__half);
    hipError_t e = hipMalloc((void **)&t->data, bytes);
    if (e != hipSuccess) { delete t; return fail(LUPIN_ERR_OUT_OF_MEMORY, hipGetErrorString(e)); }
    hipMemsetAsync(t->data, 0, bytes, ctx->stream);
    *out_tex = t;
    return LUPIN_OK;
}
void lupin_hip_texture_destroy(LupinTexture *tex)
{
    if (!tex) return;
    hipSetDevice(tex->ctx->device);
    sync_all(tex->ctx);
    hipFree(tex->data);
    delete tex;
}
uint32_t lupin_hip_texture_width(const LupinTexture *tex) { return tex ? tex->width : 0; }
uint32_t lupin_hip_texture_height(const LupinTexture *tex) { return tex ? tex->height : 0; }
void *lupin_hip_texture_device_ptr(const LupinTexture *tex) { return tex ? (void *)tex->data : nullptr; }

int lupin_hip_texture_upload_rgba16f(LupinTexture *tex, const uint16_t *pixels)
{
    if (!tex || !pixels) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(tex->ctx->device));
    join_primary(tex->ctx);
    HIP_TRY(hipMemcpyAsync(tex->data, pixels, (size_t)tex->width * tex->height * 8, hipMemcpyHostToDevice, tex->ctx->stream));
    HIP_TRY(hipStreamSynchronize(tex->ctx->stream));
    return LUPIN_OK;
}
int lupin_hip_texture_download_rgba16f(const LupinTexture *tex, uint16_t *out_pixels)
{
    if (!tex || !out_pixels) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(tex->ctx->device));
    join_primary(tex->ctx);
    HIP_TRY(hipMemcpyAsync(out_pixels, tex->data, (size_t)tex->width * tex->height * 8, hipMemcpyDeviceToHost, tex->ctx->stream));
    HIP_TRY(hipStreamSynchronize(tex->ctx->stream));
    return LUPIN_OK;
}

int lupin_hip_dbuf_create(LupinContext *ctx, uint32_t width, uint32_t height, LupinDoubleBufferedTexture **out)
{
    if (!out) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    LupinDoubleBufferedTexture *d = new LupinDoubleBufferedTexture();
    d->ctx = ctx; d->tex[0] = d->tex[1] = nullptr; d->front_idx = 1; d->back_idx = 0;   // wgpu_utils.rs:293-298
    int rc = lupin_hip_texture_create(ctx, width, height, &d->tex[0]);
    if (rc == LUPIN_OK) rc = lupin_hip_texture_create(ctx, width, height, &d->tex[1]);
    if (rc != LUPIN_OK) { lupin_hip_texture_destroy(d->tex[0]); delete d; return rc; }
    *out = d;
    return LUPIN_OK;
}
void lupin_hip_dbuf_destroy(LupinDoubleBufferedTexture *t)
{
    if (!t) return;
    lupin_hip_texture_destroy(t->tex[0]);
    lupin_hip_texture_destroy(t->tex[1]);
    delete t;
}
LupinTexture *lupin_hip_dbuf_front(LupinDoubleBufferedTexture *t) { return t ? t->tex[t->front_idx] : nullptr; }
LupinTexture *lupin_hip_dbuf_back(LupinDoubleBufferedTexture *t) { return t ? t->tex[t->back_idx] : nullptr; }
int lupin_hip_dbuf_copy_front_to_back(LupinDoubleBufferedTexture *t)
{
    if (!t) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    LupinTexture *f = t->tex[t->front_idx], *b = t->tex[t->back_idx];
    HIP_TRY(hipSetDevice(t->ctx->device));
    join_primary(t->ctx);
    HIP_TRY(hipMemcpyAsync(b->data, f->data, (size_t)f->width * f->height * 8, hipMemcpyDeviceToDevice, t->ctx->stream));
    return LUPIN_OK;
}
void lupin_hip_dbuf_flip(LupinDoubleBufferedTexture *t) { if (t) std::swap(t->front_idx, t->back_idx); }
int lupin_hip_dbuf_resize(LupinDoubleBufferedTexture *t, uint32_t width, uint32_t height)
{
    if (!t) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    if (t->tex[0]->width == width && t->tex[0]->height == height) return LUPIN_OK;   // wgpu_utils.rs:343
    LupinTexture *a = nullptr, *b = nullptr;
    int rc = lupin_hip_texture_create(t->ctx, width, height, &a);
    if (rc == LUPIN_OK) rc = lupin_hip_texture_create(t->ctx, width, height, &b);
    if (rc != LUPIN_OK) { lupin_hip_texture_destroy(a); return rc; }
    lupin_hip_texture_destroy(t->tex[0]);
    lupin_hip_texture_destroy(t->tex[1]);
    t->tex[0] = a; t->tex[1] = b;
    return LUPIN_OK;
}

// ---- the hot path ----

static int pathtrace_impl(LupinContext *ctx, const LupinPathtraceResources *res, const LupinScene *scene,
                          LupinTexture *render_target, uint32_t pathtrace_type, const LupinPathtraceDesc *desc,
                          bool tile_set, uint32_t set_tile_size, uint32_t rank, uint32_t world, int falsecolor_type = -1,
                          const LupinDebugVizDesc *debug = nullptr)
{
    if (!ctx || !res || !scene || !render_target || !desc) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    if (falsecolor_type < 0 && pathtrace_type > LUPIN_PATHTRACE_DIRECT) return fail(LUPIN_ERR_INVALID_ARGUMENT, "unknown pathtrace_type");
    if (falsecolor_type > 11) return fail(LUPIN_ERR_INVALID_ARGUMENT, "unknown falsecolor_type");
    if (debug && debug->viz_type > LUPIN_DEBUG_VIZ_NUM_BOUNCES) return fail(LUPIN_ERR_INVALID_ARGUMENT, "unknown viz_type");
    if (!scene->has_sw_bvh) return fail(LUPIN_ERR_NO_SW_BVH, "no software BVH was built for this scene");   // renderer.rs:774-777
    const uint32_t W = render_target->width, H = render_target->height;
    const LupinTexture *prev = desc->accum_params ? desc->accum_params->prev_frame : nullptr;
    if (prev && prev == render_target) return fail(LUPIN_ERR_SAME_TARGET, "render_target must differ from accum_params.prev_frame");
    if (prev && (prev->width != W || prev->height != H)) return fail(LUPIN_ERR_INVALID_ARGUMENT, "prev_frame size differs from render_target");
    HIP_TRY(hipSetDevice(ctx->device));

    FrameParams fp;
    memset(&fp, 0, sizeof(fp));
    // get_push_constants (renderer.rs:1426-1493)
    LupinPushConstants &pc = fp.pc;
    if (desc->camera_params.is_orthographic) pc.flags |= LUPIN_FLAG_CAMERA_ORTHO;
    const LupinMat3x4 &ct = desc->camera_transform;   // Mat3x4::to_mat4 (base.rs:695-705)
    for (int c = 0; c < 4; c++) { pc.camera_transform.m[c][0] = ct.m[c][0]; pc.camera_transform.m[c][1] = ct.m[c][1]; pc.camera_transform.m[c][2] = ct.m[c][2]; pc.camera_transform.m[c][3] = (c == 3) ? 1.0f : 0.0f; }
    pc.camera_lens = desc->camera_params.lens;
    pc.camera_film = desc->camera_params.film;
    pc.camera_aspect = desc->camera_params.aspect;
    pc.camera_focus = desc->camera_params.focus;
    pc.camera_aperture = desc->camera_params.aperture;
    if (scene->envs_empty) pc.flags |= LUPIN_FLAG_ENVS_EMPTY;
    if (scene->lights_empty) pc.flags |= LUPIN_FLAG_LIGHTS_EMPTY;
    if (scene->instances_empty) pc.flags |= LUPIN_FLAG_INSTANCES_EMPTY;
    if (debug)   // get_push_constants (renderer.rs:1430-1454)
    {
        pc.flags |= debug->viz_type == LUPIN_DEBUG_VIZ_BVH_AABB_CHECKS ? LUPIN_FLAG_DEBUG_AABB_CHECKS
                  : debug->viz_type == LUPIN_DEBUG_VIZ_BVH_TRI_CHECKS ? LUPIN_FLAG_DEBUG_TRI_CHECKS : LUPIN_FLAG_DEBUG_NUM_BOUNCES;
        if (debug->first_hit_only) pc.flags |= LUPIN_FLAG_DEBUG_FIRST_HIT_ONLY;
        pc.heatmap_min = debug->heatmap_min;
        pc.heatmap_max = debug->heatmap_max;
    }
    pc.pathtrace_type = (falsecolor_type < 0 && !debug) ? pathtrace_type : 0u;   // get_push_constants leaves the other selector at 0
    pc.falsecolor_type = falsecolor_type < 0 ? 0u : (uint32_t)falsecolor_type;
    pc.accum_counter = desc->accum_params ? desc->accum_params->accum_counter : 0u;
    pc.max_radiance = desc->advanced.max_radiance;
    pc.rng_seed = desc->advanced.rng_seed;
    pc.ray_epsilon = desc->advanced.ray_epsilon;

    fp.width = W; fp.height = H;
    fp.store_rne = (ctx->store_rounding == 1) ? 1u : 0u;
    fp.max_bounces = res->params.max_bounces;
    fp.spp = res->params.samples_per_pixel;
    uint64_t n64;
    if (tile_set)
    {
        if (set_tile_size == 0 || world == 0 || rank >= world) return fail(LUPIN_ERR_INVALID_ARGUMENT, "bad tile-set arguments");
        const uint32_t tpx = set_tile_size * LUPIN_WORKGROUP_SIZE;
        const uint32_t ntx = (std::max(1u, W) - 1) / tpx + 1, nty = (std::max(1u, H) - 1) / tpx + 1;
        const uint32_t total = ntx * nty;
        const uint32_t owned = (total > rank) ? (total - rank + world - 1) / world : 0;
        fp.tile_px = tpx; fp.tiles_x = ntx; fp.rank = rank; fp.world = world;
        n64 = (uint64_t)owned * tpx * tpx;
    }
    else
    {
        // dispatch extent (renderer.rs:807-838)
        uint32_t groups_x, groups_y;
        if (desc->tile_params)
        {
            uint32_t tile_size = desc->tile_params->tile_size, tile_idx = desc->tile_params->tile_idx;
            if (tile_size == 0) return fail(LUPIN_ERR_INVALID_ARGUMENT, "tile_size must be > 0");
            uint32_t ntx = (std::max(1u, W) - 1) / (tile_size * LUPIN_WORKGROUP_SIZE) + 1;
            uint32_t nty = (std::max(1u, H) - 1) / (tile_size * LUPIN_WORKGROUP_SIZE) + 1;
            if (tile_idx >= ntx * nty) return fail(LUPIN_ERR_TILE_OUT_OF_RANGE, "tile_idx out of range!");
            pc.id_offset[0] = (tile_idx % ntx) * tile_size * LUPIN_WORKGROUP_SIZE;
            pc.id_offset[1] = (tile_idx / ntx) * tile_size * LUPIN_WORKGROUP_SIZE;
            groups_x = std::min(tile_size, (W - pc.id_offset[0]) / LUPIN_WORKGROUP_SIZE);   // floor: edge remainders are skipped
            groups_y = std::min(tile_size, (H - pc.id_offset[1]) / LUPIN_WORKGROUP_SIZE);
        }
        else
        {
            groups_x = (W + LUPIN_WORKGROUP_SIZE - 1) / LUPIN_WORKGROUP_SIZE;
            groups_y = (H + LUPIN_WORKGROUP_SIZE - 1) / LUPIN_WORKGROUP_SIZE;
        }
        fp.off_x = pc.id_offset[0]; fp.off_y = pc.id_offset[1];
        fp.reg_w = std::min(groups_x * LUPIN_WORKGROUP_SIZE, W - fp.off_x);   // texels outside the image are never stored (:287)
        fp.reg_h = std::min(groups_y * LUPIN_WORKGROUP_SIZE, H - fp.off_y);
        n64 = (uint64_t)fp.reg_w * fp.reg_h;
    }
    if (n64 == 0) return LUPIN_OK;
    if (n64 > 0x7FFFFFFFull) return fail(LUPIN_ERR_INVALID_ARGUMENT, "dispatch too large");
    const uint32_t n = (uint32_t)n64;

    if (falsecolor_type >= 0 || debug)
    {
        const uint32_t fblocks = (n + LP_BLOCK - 1) / LP_BLOCK;
        const uint32_t fstack_words = scene->stack_entries * LP_BLOCK;
        const bool flds = scene->dev.geo_blob_words && ctx->lds_geometry;
        const size_t flds_bytes = (size_t)fstack_words * sizeof(uint32_t) + (flds ? (size_t)scene->dev.geo_blob_words * 16 : 0);
        if (flds_bytes > 160 * 1024) return fail(LUPIN_ERR_INVALID_ARGUMENT, "BVH too deep for the LDS traversal stack");
        const __half *pv = prev ? prev->data : (const __half *)nullptr;
        join_primary(ctx);
        if (debug)
        {
            if (flds) hipLaunchKernelGGL(k_debug<true>, dim3(fblocks), dim3(LP_BLOCK), flds_bytes, ctx->stream, scene->dev, fp, n, pv, render_target->data, fstack_words);
            else hipLaunchKernelGGL(k_debug<false>, dim3(fblocks), dim3(LP_BLOCK), flds_bytes, ctx->stream, scene->dev, fp, n, pv, render_target->data, fstack_words);
        }
        else if (flds) hipLaunchKernelGGL(k_falsecolor<true>, dim3(fblocks), dim3(LP_BLOCK), flds_bytes, ctx->stream, scene->dev, fp, n, pv, render_target->data, fstack_words);
        else hipLaunchKernelGGL(k_falsecolor<false>, dim3(fblocks), dim3(LP_BLOCK), flds_bytes, ctx->stream, scene->dev, fp, n, pv, render_target->data, fstack_words);
        HIP_TRY(hipGetLastError());
        return LUPIN_OK;
    }

    // Lane choice: consecutive calls alternate so that their wavefronts overlap; per-kernel timing needs them serial.
    const int w = ctx->timing ? 0 : (int)(ctx->call_index % (uint64_t)ctx->num_lanes);
    ctx->call_index++;
    Lane *ln = &ctx->lanes[w];
    hipStream_t st = ln->stream;

    const uint32_t iterations = fp.spp * (fp.max_bounces + 1);
    int rc = ensure_path_buffers(ctx, ln, n, iterations);
    if (rc != LUPIN_OK) return rc;

    // grid: every shard gets the same number of blocks, block b serves shard b % LP_SHARDS
    const uint32_t blocks_needed = (n + LP_BLOCK - 1) / LP_BLOCK;
    const uint32_t blocks_per_shard = (blocks_needed + LP_SHARDS - 1) / LP_SHARDS;
    const uint32_t blocks = blocks_per_shard * LP_SHARDS;
    ln->pb.shard_cap = blocks_per_shard * LP_BLOCK;
    const uint32_t stack_words = scene->stack_entries * LP_BLOCK;
    const bool lds_geo = scene->dev.geo_blob_words && ctx->lds_geometry;
    const size_t lds = (size_t)stack_words * sizeof(uint32_t) + (lds_geo ? (size_t)scene->dev.geo_blob_words * 16 : 0);
    if (lds > 160 * 1024) return fail(LUPIN_ERR_INVALID_ARGUMENT, "BVH too deep for the LDS traversal stack");

    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (ctx->timing) { t0 = get_event(ctx); t1 = get_event(ctx); hipEventRecord(t0, st); }

    const uint32_t pblocks = persistent_grid(ctx, scene, pathtrace_type, lds_geo, lds);
    hipLaunchKernelGGL(k_set_params, dim3(1), dim3(1), 0, st, fp, ln->d_fp);
    if (ctx->use_graph && !ctx->timing)
    {
        // Everything between k_set_params and the resolve depends on the call only through *d_fp, so it is captured once per
        // (scene, dispatch size, integrator, buffers) and replayed: one graph launch instead of 2-4 launches per iteration.
        Lane::GraphKey key;
        key.scene_id = scene->id; key.pb_generation = ln->pb_generation; key.n = n; key.blocks = blocks; key.type = pathtrace_type;
        key.iterations = iterations; key.stack_words = stack_words; key.lds = (uint32_t)lds; key.pblocks = pblocks;
        key.refill_min = ctx->refill_min; key.node_steps = ctx->node_steps; key.persistent = ctx->persistent_extend;
        key.persistent_shadow = ctx->persistent_shadow ? 1 : 0; key.lds_geometry = lds_geo ? 1 : 0;
        const bool have = ln->graph_exec && key == ln->graph_key;
        if (!have && ln->graph_exec && !(key == ln->seen_key))
        {
            // the lane holds a graph of another shape and this one is new (shapes alternate, e.g. edge tiles): capturing
            // costs about a millisecond, so launch directly and re-capture only if the shape repeats
            ln->seen_key = key;
            HIP_TRY(enqueue_wavefront(ctx, ln, scene, pathtrace_type, lds_geo, n, blocks, pblocks, lds, stack_words, iterations));
        }
        else
        {
            if (!have)
            {
                if (ln->graph_exec) { hipGraphExecDestroy(ln->graph_exec); ln->graph_exec = nullptr; }
                if (ln->graph) { hipGraphDestroy(ln->graph); ln->graph = nullptr; }
                HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                hipError_t ce = enqueue_wavefront(ctx, ln, scene, pathtrace_type, lds_geo, n, blocks, pblocks, lds, stack_words, iterations);
                hipError_t ee = hipStreamEndCapture(st, &ln->graph);
                if (ce != hipSuccess || ee != hipSuccess) return fail(LUPIN_ERR_HIP, std::string("graph capture: ") + hipGetErrorString(ce != hipSuccess ? ce : ee));
                HIP_TRY(hipGraphInstantiate(&ln->graph_exec, ln->graph, nullptr, nullptr, 0));
                ln->graph_key = key;
            }
            else ctx->extend_launches += iterations;
            ln->seen_key = key;
            HIP_TRY(hipGraphLaunch(ln->graph_exec, st));
        }
    }
    else
        HIP_TRY(enqueue_wavefront(ctx, ln, scene, pathtrace_type, lds_geo, n, blocks, pblocks, lds, stack_words, iterations));
    // The frames meet here: the resolve reads prev_frame and overwrites render_target, so it is ordered after everything
    // enqueued so far on the other lane (the previous call's resolve) and, for lane 1, on the primary stream (texture
    // uploads / copies).  The path state itself is private to the lane.
    if (w != 0)
    {
        HIP_TRY(hipEventRecord(ctx->marker, ctx->stream));
        HIP_TRY(hipStreamWaitEvent(st, ctx->marker, 0));
    }
    if (ctx->last_lane >= 0 && ctx->last_lane != w)
        HIP_TRY(hipStreamWaitEvent(st, ctx->lanes[ctx->last_lane].done, 0));
    hipLaunchKernelGGL(k_resolve, dim3(blocks), dim3(LP_BLOCK), 0, st, fp, ln->pb, n,
                       prev ? prev->data : (const __half *)nullptr, render_target->data);
    HIP_TRY(hipEventRecord(ln->done, st));
    ln->used = true;
    ctx->last_lane = w;
    if (ctx->timing) { hipEventRecord(t1, st); ctx->ev_total.push_back({t0, t1}); }
    HIP_TRY(hipGetLastError());
    return LUPIN_OK;
}

int lupin_hip_pathtrace_scene(LupinContext *ctx, const LupinPathtraceResources *res, const LupinScene *scene,
                              LupinTexture *render_target, uint32_t pathtrace_type, const LupinPathtraceDesc *desc)
{
    return pathtrace_impl(ctx, res, scene, render_target, pathtrace_type, desc, false, 0, 0, 1);
}

int lupin_hip_pathtrace_scene_falsecolor(LupinContext *ctx, const LupinPathtraceResources *res, const LupinScene *scene,
                                         LupinTexture *render_target, uint32_t falsecolor_type, const LupinPathtraceDesc *desc)
{
    if (falsecolor_type > 11) return fail(LUPIN_ERR_INVALID_ARGUMENT, "unknown falsecolor_type");
    return pathtrace_impl(ctx, res, scene, render_target, 0, desc, false, 0, 0, 1, (int)falsecolor_type);
}

int lupin_hip_pathtrace_scene_debug(LupinContext *ctx, const LupinPathtraceResources *res, const LupinScene *scene,
                                    LupinTexture *render_target, const LupinDebugVizDesc *debug_desc, const LupinPathtraceDesc *desc)
{
    if (!debug_desc) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    return pathtrace_impl(ctx, res, scene, render_target, 0, desc, false, 0, 0, 1, -1, debug_desc);
}

int lupin_hip_pathtrace_scene_tiles(LupinContext *ctx, const LupinPathtraceResources *res, const LupinScene *scene,
                                    LupinTexture *render_target, uint32_t pathtrace_type, const LupinPathtraceDesc *desc,
                                    uint32_t tile_size, uint32_t rank, uint32_t world)
{
    return pathtrace_impl(ctx, res, scene, render_target, pathtrace_type, desc, true, tile_size, rank, world);
}

// ---- measurement hooks ----

int lupin_hip_stats_reset(LupinContext *ctx, int enable_kernel_timing)
{
    if (!ctx) return fail(LUPIN_ERR_INVALID_ARGUMENT, "ctx is null");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(sync_all(ctx));
    for (int k = 0; k < ctx->num_lanes; k++)
        HIP_TRY(hipMemsetAsync(ctx->lanes[k].stat_counters, 0, 2 * LP_SHARDS * sizeof(unsigned long long), ctx->lanes[k].stream));
    // extend/shade pairs share their middle event: recycle each event once
    for (auto &p : ctx->ev_extend) { ctx->ev_pool.push_back(p.first); ctx->ev_pool.push_back(p.second); }
    for (auto &p : ctx->ev_shade) { ctx->ev_pool.push_back(p.second); }
    ctx->ev_extend.clear();
    ctx->ev_shade.clear();
    for (auto &p : ctx->ev_total) { ctx->ev_pool.push_back(p.first); ctx->ev_pool.push_back(p.second); }
    ctx->ev_total.clear();
    ctx->timing = enable_kernel_timing != 0;
    ctx->extend_launches = 0;
    return LUPIN_OK;
}

int lupin_hip_stats_get(LupinContext *ctx, LupinStats *out)
{
    if (!ctx || !out) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(sync_all(ctx));
    std::vector<unsigned long long> c(2 * LP_SHARDS, 0ull);
    memset(out, 0, sizeof(*out));
    for (int l = 0; l < ctx->num_lanes; l++)
    {
        HIP_TRY(hipMemcpy(c.data(), ctx->lanes[l].stat_counters, c.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (uint32_t k = 0; k < LP_SHARDS; k++) { out->path_bounces += c[2 * k]; out->paths += c[2 * k + 1]; }
    }
    out->extend_launches = ctx->extend_launches;
    auto sum = [](const std::vector<std::pair<hipEvent_t, hipEvent_t>> &v) {
        double ms = 0.0;
        for (auto &p : v) { float f = 0.0f; if (hipEventElapsedTime(&f, p.first, p.second) == hipSuccess) ms += f; }
        return ms;
    };
    out->extend_ms = sum(ctx->ev_extend);
    out->shade_ms = sum(ctx->ev_shade);
    out->total_ms = sum(ctx->ev_total);
    return LUPIN_OK;
}

int lupin_hip_trace_rays(LupinContext *ctx, const LupinScene *scene, uint32_t n, const float *ori_xyz, const float *dir_xyz,
                         float ray_epsilon, uint32_t *out_hit, float *out_dst, float *out_uv, uint32_t *out_instance, uint32_t *out_tri)
{
    if (!ctx || !scene || !ori_xyz || !dir_xyz || !out_hit || !out_dst || !out_uv || !out_instance || !out_tri) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return LUPIN_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    float *d_ori = nullptr, *d_dir = nullptr, *d_dst = nullptr, *d_uv = nullptr;
    uint32_t *d_hit = nullptr, *d_inst = nullptr, *d_tri = nullptr;
    HIP_TRY(hipMalloc((void **)&d_ori, (size_t)n * 12));
    HIP_TRY(hipMalloc((void **)&d_dir, (size_t)n * 12));
    HIP_TRY(hipMalloc((void **)&d_dst, (size_t)n * 4));
    HIP_TRY(hipMalloc((void **)&d_uv, (size_t)n * 8));
    HIP_TRY(hipMalloc((void **)&d_hit, (size_t)n * 4));
    HIP_TRY(hipMalloc((void **)&d_inst, (size_t)n * 4));
    HIP_TRY(hipMalloc((void **)&d_tri, (size_t)n * 4));
    HIP_TRY(hipMemcpyAsync(d_ori, ori_xyz, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_dir, dir_xyz, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
    size_t lds = (size_t)scene->stack_entries * LP_BLOCK * sizeof(uint32_t);
    hipLaunchKernelGGL(k_trace, dim3((n + LP_BLOCK - 1) / LP_BLOCK), dim3(LP_BLOCK), lds, ctx->stream, scene->dev, n, d_ori, d_dir, ray_epsilon,
                       d_hit, d_dst, d_uv, d_inst, d_tri);
    HIP_TRY(hipMemcpyAsync(out_hit, d_hit, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out_dst, d_dst, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out_uv, d_uv, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out_instance, d_inst, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out_tri, d_tri, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    hipFree(d_ori); hipFree(d_dir); hipFree(d_dst); hipFree(d_uv); hipFree(d_hit); hipFree(d_inst); hipFree(d_tri);
    return LUPIN_OK;
}

int lupin_hip_detmath_probe(LupinContext *ctx, int fn, uint32_t n, const float *x, const float *y, float *out)
{
    if (!ctx || !x || !y || !out) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return LUPIN_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    float *dx = nullptr, *dy = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc((void **)&dx, (size_t)n * 4));
    HIP_TRY(hipMalloc((void **)&dy, (size_t)n * 4));
    HIP_TRY(hipMalloc((void **)&dout, (size_t)n * 4));
    HIP_TRY(hipMemcpyAsync(dx, x, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dy, y, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_detmath, dim3((n + LP_BLOCK - 1) / LP_BLOCK), dim3(LP_BLOCK), 0, ctx->stream, fn, n, dx, dy, dout);
    HIP_TRY(hipMemcpyAsync(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    hipFree(dx); hipFree(dy); hipFree(dout);
    return LUPIN_OK;
}

static int pack_common(LupinContext *ctx, const LupinTexture *tex, uint32_t tile_size, uint32_t rank, uint32_t world, void *packed, int unpack)
{
    if (!ctx || !tex || !packed || tile_size == 0 || world == 0 || rank >= world) return fail(LUPIN_ERR_INVALID_ARGUMENT, "bad pack arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    dim3 block(LP_BLOCK, 1, 1), grid((tex->width + LP_BLOCK - 1) / LP_BLOCK, tex->height, 1);
    join_primary(ctx);
    hipLaunchKernelGGL(k_pack_tiles, grid, block, 0, ctx->stream, (const uint2 *)tex->data, (uint2 *)packed, tex->width, tex->height,
                       tile_size * LUPIN_WORKGROUP_SIZE, rank, world, unpack);
    HIP_TRY(hipGetLastError());
    return LUPIN_OK;
}
int lupin_hip_pack_tiles(LupinContext *ctx, const LupinTexture *tex, uint32_t tile_size, uint32_t rank, uint32_t world, void *device_dst, uint64_t *out_pixels)
{
    int rc = pack_common(ctx, tex, tile_size, rank, world, device_dst, 0);
    if (rc == LUPIN_OK && out_pixels) *out_pixels = lupin_hip_packed_tile_pixels(tex->width, tex->height, tile_size, rank, world);
    return rc;
}
int lupin_hip_unpack_tiles(LupinContext *ctx, LupinTexture *tex, uint32_t tile_size, uint32_t rank, uint32_t world, const void *device_src)
{
    return pack_common(ctx, tex, tile_size, rank, world, const_cast<void *>(device_src), 1);
}

int lupin_hip_tonemap_and_fit_aspect(LupinContext *ctx, const LupinTexture *src, uint8_t *dst_rgba8, uint32_t dst_width, uint32_t dst_height,
                                     const LupinTonemapDesc *desc)
{
    if (!ctx || !src || !dst_rgba8 || !desc || dst_width == 0 || dst_height == 0) return fail(LUPIN_ERR_INVALID_ARGUMENT, "bad tonemap arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    TonemapArgs a;
    memset(&a, 0, sizeof(a));
    a.src_w = src->width; a.src_h = src->height; a.dst_w = dst_width; a.dst_h = dst_height;
    if (desc->has_viewport) { a.vp_x = desc->viewport_x; a.vp_y = desc->viewport_y; a.vp_w = desc->viewport_w; a.vp_h = desc->viewport_h; }
    else { a.vp_x = 0.0f; a.vp_y = 0.0f; a.vp_w = (float)dst_width; a.vp_h = (float)dst_height; }
    if (!(a.vp_w > 0.0f) || !(a.vp_h > 0.0f) || !(a.vp_x >= 0.0f) || !(a.vp_y >= 0.0f)) return fail(LUPIN_ERR_INVALID_ARGUMENT, "bad viewport");
    const float src_aspect = (float)src->width / (float)src->height;      // tonemapping.rs:168-174
    const float dst_aspect = a.vp_w / a.vp_h;
    if (src_aspect > dst_aspect) { a.scale_x = 1.0f; a.scale_y = dst_aspect / src_aspect; }
    else { a.scale_x = src_aspect / dst_aspect; a.scale_y = 1.0f; }
    a.exposure = desc->exposure; a.filmic = desc->filmic ? 1u : 0u; a.srgb = desc->srgb ? 1u : 0u;
    // set_scissor_rect(viewport.x as u32, viewport.y as u32, viewport.w as u32, viewport.h as u32) (:217)
    a.sc_x0 = std::min((uint32_t)a.vp_x, dst_width); a.sc_y0 = std::min((uint32_t)a.vp_y, dst_height);
    a.sc_x1 = (uint32_t)std::min<uint64_t>((uint64_t)a.sc_x0 + (uint32_t)a.vp_w, dst_width);
    a.sc_y1 = (uint32_t)std::min<uint64_t>((uint64_t)a.sc_y0 + (uint32_t)a.vp_h, dst_height);

    const size_t bytes = (size_t)dst_width * dst_height * 4;
    uint32_t *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, bytes));
    join_primary(ctx);
    hipError_t e;
    if (desc->clear)   // LoadOp::Clear(0, 0, 0, 1) over the whole attachment
    {
        std::vector<uint32_t> clear_px((size_t)dst_width * dst_height, 0xFF000000u);
        e = hipMemcpyAsync(d, clear_px.data(), bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    else e = hipMemcpyAsync(d, dst_rgba8, bytes, hipMemcpyHostToDevice, ctx->stream);   // LoadOp::Load
    if (e == hipSuccess && a.sc_x1 > a.sc_x0 && a.sc_y1 > a.sc_y0)
    {
        dim3 grid((a.sc_x1 - a.sc_x0 + LP_BLOCK - 1) / LP_BLOCK, a.sc_y1 - a.sc_y0, 1);
        hipLaunchKernelGGL(k_tonemap, grid, dim3(LP_BLOCK), 0, ctx->stream, a, src->data, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(dst_rgba8, d, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(LUPIN_ERR_HIP, hipGetErrorString(e));
    return LUPIN_OK;
}

}  // extern "C"
