// lupin_stages.hpp -- the stage kernels of the wavefront path tracer (gfx950), included once by lupin_hip.hip.
//
// What the reference runs as ONE megakernel invocation per pixel (`pathtrace_main`, pathtracer.wgsl:220-292) is split
// into stages over compacted queues of live paths:
//
//   k_begin               RNG seeding, first camera ray of every pixel of the dispatch      (:224-237, :505-542)
//   k_extend              closest hit with stochastic alpha skipping                       (bvh_custom.wgsl:154-180)
//   k_extend_persistent   the same, phase-scheduled with lane refill (scenes traversed from global memory; also
//                         traces the MIS / Direct shadow rays there)
//   k_shade               the rest of one integrator-loop iteration: medium sampling, material fetch, emission,
//                         BSDF / light sampling + pdfs, volume stack, Russian roulette, and -- when a path ends --
//                         clamp_radiance, next camera sample of the same pixel (the per-pixel RNG stream continues
//                         across samples, :234-239), or retirement
//   k_shadow              MIS / Direct: shadow-ray terms in the reference's order, path finish
//   k_resolve             /spp, progressive blend with prev_frame, Rgba16Float store          (:275-289)
//   k_falsecolor, k_debug, k_tonemap, k_trace, k_detmath, k_pack_tiles    the other entry points and probes
//
// One thread owns one pixel for the whole call, so radiance accumulation needs no atomics; the only atomics are the
// per-wave queue appends (ballot + one atomicAdd per wave).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "lupin_device.hpp"
#include "../../include/lupin_tiles.h"

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

// A per-path field of the path state, `field[slot]`.  The eight fields every bounce touches live in one allocation that is
// laid out either as planes (stride = the element: neighbouring slots coalesce -- scenes whose queues stay in pixel order)
// or as one 128-byte record per path (stride 128: a path's fields share two 64-byte sectors -- scenes whose queues are
// sorted by material, where the slots of a wave are scattered and a plane costs one request per field and lane):
//   sector 0: ori_rng 0, dir_meta 16, hit 32, hit_tri 48      sector 1: weight 64, radiance 80, color 96
template <typename T>
struct PathField
{
    char *base;
    uint32_t stride;
    __device__ __forceinline__ T &operator[](size_t slot) const { return *reinterpret_cast<T *>(base + slot * stride); }
};
constexpr uint32_t LP_PATH_RECORD_BYTES = 128;

struct PathBuffers
{
    PathField<float4> ori_rng;    // ori.xyz | rng state
    PathField<float4> dir_meta;   // dir.xyz | meta
    PathField<float4> weight;     // weight.xyz
    PathField<float4> radiance;   // radiance.xyz
    PathField<float4> color;      // per-pixel sum over samples
    PathField<float4> hit;        // dst u v | instance (HIT_MISS = no hit)
    PathField<uint32_t> hit_tri;  // global triangle
    float4 *vol0;       // medium density.xyz | anisotropy
    float4 *vol1;       // medium scattering.xyz
    // The MIS / Direct fields share a second allocation with the same two layouts; as records:
    //   sector 0: sh_org 0, sh_d0 16, sh_d1 32, next_tri 48      sector 1: sh_f0 64, sh_f1 80, next_hit 96, sh_hit1 112
    PathField<float4> next_hit;   // MIS: hit of the BSDF-sampled shadow ray, reused as next vertex
    PathField<uint32_t> next_tri;
    // MIS / Direct shadow rays, recorded by k_shade and traced by k_shadow (radiance += factor * emission (*|/) scalar)
    PathField<float4> sh_org;     // origin.xyz | flags (bit0: ray 0 valid, bit1: ray 1 valid)
    PathField<float4> sh_d0;      // ray 0 direction | scalar 0
    PathField<float4> sh_f0;      // ray 0 factor = weight * bsdfcos
    PathField<float4> sh_d1;      // ray 1 direction | scalar 1
    PathField<float4> sh_f1;      // ray 1 factor (.w: triangle of the pre-traced hit, see sh_hit1)
    PathField<float4> sh_hit1;    // large scenes trace the shadow rays in the persistent kernel: hit record of ray 1 (ray 0 -> next_hit)
    // Live-path queues, sharded: one global counter per iteration would serialise every wave's append on a
    // single L2 atomic (~88 per microsecond chip-wide -- measured: 186 us per 1M-path iteration, more than the
    // shading itself).  Each of LP_SHARDS shards owns a fixed segment of the queue and its own counter; block b
    // always reads and appends shard b % LP_SHARDS, so a shard never grows beyond its initial size.
    uint32_t *queue[2];   // [parity][shard * shard_cap + i]
    uint32_t *counts;     // counts[k * LP_SHARDS + s] = live paths of shard s entering iteration k
    PathField<uint32_t> skey;     // Standard integrator on the persistent tracer: k_sort_queue's key of the path's hit, written by the tracer (always a plane)
    uint32_t *cursors;    // cursors[(2 k + mode) * LP_SHARDS + s]: how much of shard s the persistent tracer's waves have taken in iteration k
    // Wide tracer: queries it could not certify (lupin_device.hpp "Wide traversal") are appended here, per shard, as job
    // tokens (MODE 0: slot; MODE 1: 2 * slot + ray) and re-traced by the binary tracer in the reference's order.
    uint32_t *retrace;          // [shard * 2 * shard_cap + i]
    uint32_t *retrace_counts;   // [(2 k + mode) * LP_SHARDS + s]
    uint32_t *retrace_cursors;  // hand-out cursors of the re-trace launch, same indexing
    // Light-pdf stage (k_light_pdf, Standard): k_shade does not append; it tags its queue entry with what became of the
    // path (QUEUE_STATE_*), parks numerator and BSDF pdf of a waiting vertex in sh_f0, and k_light_pdf finishes the
    // iteration and appends the survivors in the queue's order.
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
    // Batched calls (DESIGN 5 "Frames per wavefront"): consecutive pathtrace_scene calls that differ only in camera and
    // accum_counter run as ONE wavefront; slot = frame * frame_slots + pixel slot, and the stage kernels read the lane's
    // FrameParams ARRAY: [0] for everything the batch shares, [frame] for the camera and the RNG seed.
    uint32_t frame_slots;        // slots (pixels of the dispatch) per frame
    uint32_t num_frames;         // frames in this wavefront (1 = a single call)
};
constexpr uint32_t LP_MAX_BATCH = 16;

// which frame of the batch a slot belongs to, and its pixel slot inside the frame
__device__ __forceinline__ uint32_t slot_frame(const FrameParams &fp, uint32_t slot, uint32_t &pixel_slot)
{
    const uint32_t frame = fp.num_frames > 1u ? slot / fp.frame_slots : 0u;
    pixel_slot = slot - frame * fp.frame_slots;
    return frame;
}

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
        const uint32_t t = lupin_owned_tile(slot / per_tile, fp.rank, fp.world, fp.tiles_x);
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
    uint32_t gx = 0, gy = 0, frame = 0;
    bool live = slot < n;
    if (live)
    {
        uint32_t pslot;
        frame = slot_frame(fp, slot, pslot);
        slot_to_pixel(fp, pslot, gx, gy);
        live = gx < fp.width && gy < fp.height;   // edge tiles: texels outside the image are never stored (:287)
    }
    const uint32_t shard = blockIdx.x % LP_SHARDS;
    queue_append(live, slot, pb.queue[0] + (size_t)shard * pb.shard_cap, &pb.counts[shard]);
    if (!live) return;
    const FrameParams &ff = fpp[frame];   // this frame's camera and accumulation counter
    uint32_t rng = rng_seed_for(gy * fp.width + gx, ff.pc.accum_counter);
    f3 o, d;
    camera_ray(ff, gx, gy, rng, o, d);
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
#ifndef LP_DEFER_SHADE_WAVES
#define LP_DEFER_SHADE_WAVES 4   // k_shade without the light-pdf march (the light-pdf stage runs it)
#endif
#ifndef LP_LIGHT_PDF_WAVES
#define LP_LIGHT_PDF_WAVES 4
#endif
#ifndef LP_LIGHT_PDF_MIS_WAVES
#define LP_LIGHT_PDF_MIS_WAVES 3
#endif
#ifndef LP_MIS_DEFER_SHADE_WAVES
#define LP_MIS_DEFER_SHADE_WAVES 3
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

// Work counters (bench.py's roofline numerator, lupin_hip_stats_reset(ctx, 2)): the COUNT instantiations of the tracing
// kernels wrap their geometry accessor in GeoTally and add the wave's totals to work[4 * mode + {0 nodes, 1 triangles, 2 instances,
// 3 four-wide nodes}].  All 64 lanes must be active when this is called.
constexpr int LP_TALLY = 4;
constexpr int LP_ROUND_BASE = 12, LP_ROUND_WORDS = 10;   // work[12 .. 21]: round statistics of the persistent tracer (closest-hit mode)
constexpr int LP_CLOCK_BASE = 22, LP_CLOCK_WORDS = 6;    // work[22 .. 27]: shader-clock cycles per wave spent in refill | N | T | I | F rounds, and in the whole loop
__device__ __forceinline__ void tally_flush(const uint32_t (&tally)[LP_TALLY], unsigned long long *work)
{
    #pragma unroll
    for (int k = 0; k < LP_TALLY; k++)
    {
        uint32_t v = tally[k];
        #pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63u) == 0u && v) atomicAdd(&work[k], (unsigned long long)v);
    }
}
template <typename Geo, bool COUNT> struct TallyOf { typedef Geo type; };
template <typename Geo> struct TallyOf<Geo, true> { typedef GeoTally<Geo> type; };
template <bool COUNT, typename Geo>
__device__ __forceinline__ typename TallyOf<Geo, COUNT>::type with_tally(const Geo &geo, uint32_t *tally)
{
    if constexpr (COUNT) { GeoTally<Geo> g; g.base = geo; g.tally = tally; return g; }
    else return geo;
}

template <int TYPE, bool LDSGEO, bool OPAQUE, bool COUNT>
__global__ void __attribute__((amdgpu_waves_per_eu(LP_EXTEND_WAVES, 8))) __launch_bounds__(LP_BLOCK) k_extend(SceneDev sc, const FrameParams *__restrict__ fpp, PathBuffers pb, uint32_t iter,
                                                     unsigned long long *shard_stats, uint32_t stack_words, unsigned long long *work)
{
    const FrameParams fp = *fpp;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    const auto base_geo = make_geo<LDSGEO>(sc, lds_stack, stack_words);
    uint32_t tally[LP_TALLY] = {0u, 0u, 0u, 0u};
    const auto geo = with_tally<COUNT>(base_geo, tally);
    const uint32_t shard = blockIdx.x % LP_SHARDS;
    const uint32_t count = pb.counts[iter * LP_SHARDS + shard];
    const uint32_t i = (blockIdx.x / LP_SHARDS) * LP_BLOCK + threadIdx.x;
    if (i == 0 && count) shard_stats[shard * 2 + 0] += count;   // one writer per shard per launch: no atomic needed
    if (!COUNT && i >= count) return;
    if (i < count)
    {
        const uint32_t slot = pb.queue[iter & 1][(size_t)shard * pb.shard_cap + i];
        float4 orr = pb.ori_rng[slot];
        float4 dm = pb.dir_meta[slot];
        if (TYPE == LUPIN_PATHTRACE_MIS && !(__float_as_uint(dm.w) & META_NEXT_EMISSION))
        {
            pb.hit[slot] = pb.next_hit[slot];
            pb.hit_tri[slot] = pb.next_tri[slot];
        }
        else
        {
            uint32_t rng = __float_as_uint(orr.w);
            const uint32_t rng_in = rng;
            float4 hitrec;
            uint32_t hit_tri;
            trace_alpha<typename TallyOf<typename GeoOf<LDSGEO>::type, COUNT>::type, OPAQUE>(geo, sc, lds_stack, mk3(orr.x, orr.y, orr.z), mk3(dm.x, dm.y, dm.z), rng, fp.pc.ray_epsilon, hitrec, hit_tri);
            pb.hit[slot] = hitrec;
            pb.hit_tri[slot] = hit_tri;
            if (rng != rng_in) pb.ori_rng[slot].w = __uint_as_float(rng);
        }
    }
    if (COUNT) tally_flush(tally, work);
}

// Persistent, phase-scheduled form of k_extend -- the default for scenes traversed from global memory.  Rays of one
// wave need very different numbers of traversal steps and sit in different phases of the traversal, so the
// one-ray-per-lane kernel keeps 11 % of the VALU lanes busy on the bistro-class scene (39 % on the Cornell box; PMC:
// SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64)).  Here each wave owns its lanes for the whole launch, refills empty
// lanes whenever at least `refill_min` of them are free, and executes per round the one phase most lanes wait for.
// The waves of the grid are dealt to the shards round-robin and the waves of a shard hand its queue out among themselves
// (one atomic per refill).  Every ray is still traced by exactly the same sequence of operations as in k_extend, only by
// a different lane.
constexpr uint32_t LP_REFILL_MIN = 16;   // a wave refills when at least this many of its lanes are empty
#ifndef LP_SHORT_WAVES
#define LP_SHORT_WAVES 6   // waves per SIMD the short-stack pass is compiled for (7: 72 registers, 8: 64 -- both measured slower, the spills reach the node loop)
#endif
constexpr uint32_t LP_NODE_STEPS = 4;    // node visits per scheduling round at most

// k_sort_queue's key of a path about to be shaded: what k_shade will execute for it.  Material type of the hit (0..7) | miss (8)
// | inside a medium (9); +16: the material's smooth hint (delta branch); +32 (Standard integrator only): the outcome of the
// path's next random number, which for a surface hit outside a medium is the BSDF-or-light-sampling coin
// (pathtracer.wgsl:640-642) -- the RNG state is read, not advanced.
constexpr uint32_t LP_SORT_KEYS = 64;
template <bool PEEK_COIN>
__device__ __forceinline__ uint32_t shade_sort_key(bool in_medium, bool miss, uint32_t instance_flags, uint32_t rng)
{
    if (in_medium) return 9u;
    if (miss) return 8u;
    uint32_t key = (instance_flags >> 8) & 7u;
    if (instance_flags & (1u << 12)) key |= 16u;
    else if (PEEK_COIN && rnd(rng) < 0.5f) key |= 32u;
    return key;
}

// MODE 0: the integrator's closest-hit queries (one per queue entry, stochastic alpha skipping).
// MODE 1: the shadow rays k_shade recorded for MIS / Direct (two jobs per queue entry, plain closest hit); their hits go
//         to next_hit / next_tri (MIS ray 0, which doubles as the next vertex) or sh_hit1 / sh_f1.w, and
//         k_shadow<.., PRETRACED> folds them into the radiance.
// WIDE:   the four-wide hierarchies with the exactness certificate (lupin_device.hpp "Wide traversal"): a query whose
//         result is not certified writes nothing and leaves its job token in pb.retrace.
// RETRACE: serves those tokens (a second launch of the binary instantiation): the reference's order, by construction.
// SHORT:  the binary traversal on a stack of fewer entries than the scene's depth bound asks for, so that a fifth block fits a
//         CU's LDS: a query whose stack would overflow writes nothing and leaves its token in pb.retrace like an uncertified
//         wide query; every other query has executed exactly the reference's sequence.
// wide_stats (one writer per launch): [0] queries the wide / short tracer took, [1] queries it handed to the re-trace.
template <int TYPE, bool LDSGEO, int MODE, bool COUNT, bool WIDE = false, bool RETRACE = false, bool SHORT = false>
// The short-stack pass is compiled for six waves per SIMD (80 registers; its LDS footprint allows six blocks per CU): the
// tracer is bound by latency x waves in flight, and the few words the compiler spills (kernel-argument pointers,
// reloaded in the triangle and end-of-traversal phases) are cheaper than the missing wave.  (Keeping the world ray in LDS
// instead -- nine registers -- removes no spill and costs an LDS round trip in every scheduling round: measured, rejected.  The
// full-stack instantiation at six waves: no change on the shallow scenes that could use it, materials1 / environments1.)
__global__ void __attribute__((amdgpu_waves_per_eu(SHORT ? LP_SHORT_WAVES : (COUNT ? 2 : 4), 8))) __launch_bounds__(LP_BLOCK) k_extend_persistent(SceneDev sc, const FrameParams *__restrict__ fpp, PathBuffers pb, uint32_t iter,
                                                                unsigned long long *shard_stats, uint32_t refill_min, uint32_t stack_words, uint32_t nsteps,
                                                                unsigned long long *work, unsigned long long *wide_stats)
{
    static_assert(!(WIDE && (LDSGEO || RETRACE)), "the wide hierarchies are traversed from global memory; the re-trace is binary");
    static_assert(!(SHORT && (WIDE || LDSGEO || RETRACE)), "the short stack is a variant of the binary tracer's first pass");
    const FrameParams fp = *fpp;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    const auto base_geo = make_geo<LDSGEO>(sc, lds_stack, stack_words);
    uint32_t tally[LP_TALLY] = {0u, 0u, 0u, 0u};
    const auto geo = with_tally<COUNT>(base_geo, tally);
    static_assert(LP_SHARDS <= LP_BLOCK && 256 % LP_SHARDS == 0, "block 0 books one shard per thread; 64-block grids hold whole waves per shard");
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const size_t mode_row = ((size_t)iter * 2u + (MODE == 1 ? 1u : 0u)) * LP_SHARDS;
    const uint32_t *counts = RETRACE ? pb.retrace_counts + mode_row : pb.counts + (size_t)iter * LP_SHARDS;
    if (!RETRACE && MODE == 0 && blockIdx.x == 0 && tid < LP_SHARDS) { const uint32_t c = counts[tid]; if (c) shard_stats[tid * 2 + 0] += c; }   // one writer per shard per launch
    if ((WIDE || SHORT || RETRACE) && blockIdx.x == 0 && tid < 64u)
    {
        // the first wave books the launch's job count (no LDS: a static allocation here would cost the 40 KB stack its
        // fourth block per CU)
        uint32_t c = 0;
        for (uint32_t k = lane; k < LP_SHARDS; k += 64u) c += counts[k] * ((MODE == 1 && !RETRACE) ? 2u : 1u);
        #pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
        if (lane == 0 && c) wide_stats[RETRACE ? 1 : 0] += c;   // one writer per launch
    }

    // The grid holds `wps` waves per shard; the waves of a shard hand its queue out among themselves, one atomic per
    // refill (a static share per wave leaves the launch waiting for the wave whose few hundred rays happened to be the
    // deep ones: with 1 M rays per launch -- an eighth of the 4K frame -- the tracer ran at half its large-launch rate).
    const uint32_t wave = blockIdx.x * (LP_BLOCK / 64) + tid / 64;         // wave-uniform
    const uint32_t shard = wave % LP_SHARDS;
    const uint32_t cnt = counts[shard] * ((MODE == 1 && !RETRACE) ? 2u : 1u);   // jobs (a re-trace token is one job)
    if (cnt == 0) return;
    uint32_t *cursor = (RETRACE ? pb.retrace_cursors : pb.cursors) + mode_row + shard;
    const size_t shard_base = (size_t)shard * pb.shard_cap;
    const uint32_t *queue = RETRACE ? pb.retrace + 2u * shard_base : pb.queue[iter & 1] + shard_base;
    bool exhausted = false;                                                 // wave-uniform: the shard's queue is handed out

    const float eps = fp.pc.ray_epsilon;
    const float abs_margin = eps;                                           // wide_threshold's absolute part: the scene's own "closer than this is the same place"
    // The instance a lane is inside travels with what its end-of-traversal round needs of that instance's flags (alpha bit,
    // material type, smooth hint: the sort key) in the six bits above the index -- otherwise that round starts with a
    // dependent fetch of the instance record for one word.  (Scene creation refuses more than 2^26 instances.)
    constexpr uint32_t INST_TAG_SHIFT = 26u, INST_INDEX_MASK = (1u << INST_TAG_SHIFT) - 1u;
    constexpr uint32_t REF_DONE = 0xFFFFFFFFu;
    constexpr uint32_t REF_EXIT = 0xFFFFFFFEu;                              // "leave the instance": handled with the instance entries (I phase), see pop()
    constexpr uint32_t REF_OVER = 0xFFFFFFFDu;                              // SHORT: "the stack was too short for this query" (an end of traversal without a result)
    constexpr uint32_t REF_SKIP = 0x3FFFFFFFu;                              // WIDE: "pop again" -- an internal-node reference no scene can hold (index 2^30 - 1)
    const uint32_t stack_entries = stack_words / LP_BLOCK;                  // WIDE: (reference, distance) pairs -> stack_entries / 2 of them; SHORT: references

    // per-lane ray + traversal state
    bool active = false;
    bool flagged = false;     // WIDE: this query goes to the re-trace
    uint32_t slot = 0, rng = 0, rng_in = 0, alpha_k = 0, ray_k = 0;
    bool in_medium = false;   // META_VOLUME of the path (k_sort_queue's key)
    float total_dst = 0.0f;
    f3 o = splat(0.0f), d = splat(0.0f), inv_d = splat(0.0f);
    f3 co = o, cd = d, cinv = inv_d;
    uint32_t sp = 0, blas_base = 0xFFFFFFFFu, cur_inst = 0, cur = REF_DONE;
    Closest best;
    best.t = LP_F32_MAX; best.u = 0.0f; best.v = 0.0f; best.tri = 0u; best.inst = HIT_MISS;

    // a new query from world origin `no` along `nd` (an alpha skip passes the direction it already has: same bits, same reciprocal)
    auto start_traversal = [&](f3 no, f3 nd) {
        const f3 ninv = mk3(1.0f / nd.x, 1.0f / nd.y, 1.0f / nd.z);
        o = no; d = nd; inv_d = ninv;
        co = no; cd = nd; cinv = ninv;
        sp = 0; blas_base = 0xFFFFFFFFu;
        cur = sc.num_instances ? (WIDE ? sc.tlas4_root : sc.tlas_root) : REF_DONE;
        best.t = LP_F32_MAX; best.u = 0.0f; best.v = 0.0f; best.tri = 0u; best.inst = HIT_MISS;
    };
    // The next reference of the lane's stack.  An exhausted BLAS part (sp == blas_base) does NOT restore the world ray here:
    // that is nine registers assigned under a condition inside the hottest loop, which the compiler pays for with copies of
    // the whole ray at every join (a third of the node step's vector instructions were v_mov).  The lane takes REF_EXIT
    // instead, which the top of the next scheduling round serves (restore, then pop on) before the phases are counted.
    auto pop = [&]() {
        if (sp == blas_base) { cur = REF_EXIT; return; }
        if (sp == 0) { cur = REF_DONE; return; }
        sp--;
        if constexpr (WIDE)
        {
            // entries are (reference, entry distance): one whose distance is no longer below the threshold is dropped unfetched.
            // ONE entry per call: a dropped entry leaves REF_SKIP, and the lane pops again in its next node step -- a loop here
            // would run for the whole wave as long as its unluckiest lane, at an LDS round trip per iteration.
            const uint32_t ref = lds_stack[(2u * sp) * LP_BLOCK + tid];
            const float sd = __uint_as_float(lds_stack[(2u * sp + 1u) * LP_BLOCK + tid]);
            cur = sd < wide_threshold(best.t, abs_margin) ? ref : REF_SKIP;
        }
        else cur = lds_stack[sp * LP_BLOCK + tid];
    };

    // COUNT: how the wave's rounds were spent (work[LP_ROUND_BASE ..], MODE 0 only): refill rounds | N rounds, N steps, lanes
    // over the N steps | T rounds, lanes | I rounds, lanes | F rounds, lanes
    uint32_t rs[LP_ROUND_WORDS] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    unsigned long long clk[LP_CLOCK_WORDS] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    unsigned long long t_round = 0, t_loop = 0;
    if (COUNT) t_loop = __builtin_amdgcn_s_memtime();

    // Phase scheduling: a lane is at an internal node (N), a TLAS leaf = instance entry (I), a triangle of a BLAS leaf (T),
    // at the end of a traversal (F) or empty (E).  Every round the wave executes the ONE phase most of its lanes wait for
    // (wave-uniform branch), so each instruction runs with as many lanes as possible; lanes of other phases just wait.
    for (;;)
    {
        // lanes that have left an instance: world ray back, next reference from the TLAS part of the stack (see pop())
        if (__ballot(active && cur == REF_EXIT))
        {
            if (active && cur == REF_EXIT)
            {
                blas_base = 0xFFFFFFFFu;
                co = o; cd = d; cinv = inv_d;
                pop();
            }
        }
        const bool isN = active && !(cur & REF_LEAF);
        const bool isF = active && (SHORT ? cur >= REF_OVER : cur == REF_DONE);   // (REF_EXIT has been served above)
        const bool isLeaf = active && (cur & REF_LEAF) && (SHORT ? cur < REF_OVER : cur != REF_DONE);
        const bool isI = isLeaf && blas_base == 0xFFFFFFFFu;
        const bool isT = isLeaf && !isI;
        const unsigned long long idle = __ballot(!active);
        const uint32_t cE = (uint32_t)__popcll(idle);
        const uint32_t cN = (uint32_t)__popcll(__ballot(isN)), cI = (uint32_t)__popcll(__ballot(isI));
        const uint32_t cT = (uint32_t)__popcll(__ballot(isT)), cF = (uint32_t)__popcll(__ballot(isF));

        if (cE >= refill_min && !exhausted)
        {
            // ---- refill empty lanes ----
            if (COUNT) { rs[0]++; t_round = __builtin_amdgcn_s_memtime(); }
            const uint32_t my_rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            uint32_t first = 0;
            if (lane == 0) first = atomicAdd(cursor, cE);
            first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
            const uint32_t take = first < cnt ? min(cE, cnt - first) : 0u;
            exhausted = take < cE;
            const bool got = !active && my_rank < take;
            const uint32_t job = first + my_rank;
            if (got && MODE == 0)
            {
                slot = queue[job];
                const float4 orr = pb.ori_rng[slot];
                const float4 dm = pb.dir_meta[slot];
                bool trace = true;
                if (TYPE == LUPIN_PATHTRACE_MIS && !RETRACE)
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
                    rng = rng_in = __float_as_uint(orr.w);
                    in_medium = (__float_as_uint(dm.w) & META_VOLUME) != 0;
                    total_dst = 0.0f;
                    alpha_k = 0;
                    flagged = false;
                    start_traversal(mk3(orr.x, orr.y, orr.z), mk3(dm.x, dm.y, dm.z));
                    active = true;
                }
            }
            if (got && MODE == 1)
            {
                if (RETRACE) { const uint32_t token = queue[job]; slot = token >> 1; ray_k = token & 1u; }
                else { slot = queue[job >> 1]; ray_k = job & 1u; }
                const float4 so = pb.sh_org[slot];
                if (__float_as_uint(so.w) & (1u << ray_k))
                {
                    const float4 dd = ray_k ? pb.sh_d1[slot] : pb.sh_d0[slot];
                    flagged = false;
                    start_traversal(mk3(so.x, so.y, so.z), mk3(dd.x, dd.y, dd.z));
                    active = true;
                }
            }
            if (COUNT) clk[0] += __builtin_amdgcn_s_memtime() - t_round;
            continue;
        }
        if (cE == 64u) break;   // nothing in flight and (see above) nothing left to fetch
        if (COUNT) t_round = __builtin_amdgcn_s_memtime();
        int phase_id = 0;

        // The vote: lanes at a triangle or at an instance entry count double.  Those rounds are one step long where a node round
        // is up to four, so serving them early costs little and returns their lanes to the node rounds that dominate
        // (bistro-class 4K: tracer - 2.9 % against a plain majority; end-of-traversal lanes counted double: + 1.3 %).
        const uint32_t vN = cN, vT = 2u * cT, vI = 2u * cI, vF = cF;
        if (vN >= vT && vN >= vI && vN >= vF)
        {
            // ---- N: internal nodes of either level; keeps stepping while at least half of the voters are still at one ----
            if (COUNT) { rs[1]++; phase_id = 1; }
            for (uint32_t r = 0;; r++)
            {
                const bool n = active && !(cur & REF_LEAF);
                if (r > 0 && (r >= nsteps || (uint32_t)__popcll(__ballot(n)) * 2u < cN)) break;
                if (COUNT) { rs[2]++; rs[3] += (uint32_t)__popcll(__ballot(n)); }
                if (n)
                {
                    if constexpr (WIDE)
                    {
                        bool need_pop = cur == REF_SKIP;
                        if (!need_pop)
                        {
                            const auto nd = geo.node4(cur & REF_INDEX_MASK);
                            float dk[4]; uint32_t rk[4];
                            wide_children(nd, co, cinv, wide_threshold(best.t, abs_margin), dk, rk);
                            if (2u * (sp + 3u) > stack_entries)
                            {
                                flagged = true; cur = REF_DONE;   // bounded stack (room for three is required): the binary tracer takes this query
                            }
                            else
                            {
                                #pragma unroll
                                for (int k = 3; k >= 1; k--)
                                    if (dk[k] < __builtin_inff())
                                    {
                                        lds_stack[(2u * sp) * LP_BLOCK + tid] = rk[k];
                                        lds_stack[(2u * sp + 1u) * LP_BLOCK + tid] = __float_as_uint(dk[k]);
                                        sp++;
                                    }
                                need_pop = !(dk[0] < __builtin_inff());
                                cur = rk[0];
                            }
                        }
                        if (need_pop) pop();
                    }
                    else
                    {
                        const NodeRegs nd = geo.node(false, cur);
                        float ld, rd;
                        slab_pair(co, cinv, nd.a, nd.b, nd.c, ld, rd);
                        // bvh_custom.wgsl:63-94 / :252-283: the nearer child (left on a tie) is visited first, the other parked;
                        // each only if its entry distance is below the best hit so far.  near <= far, so "far is entered"
                        // implies "near is entered" (neither is ever NaN): the far child is parked exactly when both are
                        // entered, and a lone entered child is the near one.
                        const bool left_first = ld <= rd;
                        const uint32_t near_ref = left_first ? nd.left : nd.right;
                        const uint32_t far_ref = left_first ? nd.right : nd.left;
                        const float dn = __builtin_fminf(ld, rd), df = __builtin_fmaxf(ld, rd);
                        // SHORT: a push the short stack has no room for hands the query to the full-stack tracer (nothing has been
                        // written for it yet); same straight-line code as the full stack, the verdict is a select at the end
                        const bool over = SHORT && df < best.t && sp >= stack_entries;
                        if (df < best.t && !over) { lds_stack[sp * LP_BLOCK + tid] = far_ref; sp++; }
                        if (dn < best.t) cur = near_ref; else pop();
                        if (SHORT && over) cur = REF_OVER;
                    }
                }
            }
        }
        else if (vT >= vI && vT >= vF)
        {
            // ---- T: one triangle of a BLAS leaf, first-found wins ties (strict <) ----
            if (COUNT) { rs[4]++; rs[5] += cT; phase_id = 2; }
            if (isT)
            {
                const uint32_t ti = cur & (WIDE ? REF_INDEX_MASK : ~REF_LEAF);
                const TriVerts tv = geo.tri(ti);
                bool give_up = false;
                if constexpr (WIDE) give_up = wide_test_triangle(tv, ti, cur_inst, co, cd, eps, abs_margin, best);
                else
                {
                    TriHit h = tri_dst(co, cd, xyz(tv.v0), xyz(tv.v1), xyz(tv.v2), eps);
                    if (h.t < best.t) { best.t = h.t; best.u = h.u; best.v = h.v; best.tri = ti; best.inst = cur_inst; }
                }
                if (give_up) { flagged = true; cur = REF_DONE; }   // no certificate: the rest of this traversal would be discarded anyway
                else if (__float_as_uint(tv.v0.w) & LEAF_END_BITS) pop(); else cur++;
            }
        }
        else if (vI >= vF)
        {
            // ---- I: enter an instance (bvh_custom.wgsl:28-37) ----
            if (COUNT) { rs[6]++; rs[7] += cI; phase_id = 3; }
            if (isI)
            {
                cur_inst = cur & (WIDE ? REF_INDEX_MASK : ~REF_LEAF);
                const InstanceDev in = geo.inst(cur_inst);
                const uint32_t inst_index = cur_inst;
                cur_inst |= ((in.flags & 1u) | (((in.flags >> 8) & 0x1Fu) << 1)) << INST_TAG_SHIFT;
                // a TLAS leaf is reached in world space (blas_base says so): co / cd ARE the world ray, bit for bit
                const f3 wo = co, wd = cd;
                co = mk3(wo.x * in.r0.x + wo.y * in.r0.y + wo.z * in.r0.z + 1.0f * in.r0.w,
                         wo.x * in.r1.x + wo.y * in.r1.y + wo.z * in.r1.z + 1.0f * in.r1.w,
                         wo.x * in.r2.x + wo.y * in.r2.y + wo.z * in.r2.z + 1.0f * in.r2.w);
                cd = mk3(wd.x * in.r0.x + wd.y * in.r0.y + wd.z * in.r0.z + 0.0f * in.r0.w,
                         wd.x * in.r1.x + wd.y * in.r1.y + wd.z * in.r1.z + 0.0f * in.r1.w,
                         wd.x * in.r2.x + wd.y * in.r2.y + wd.z * in.r2.z + 0.0f * in.r2.w);
                uint32_t root = in.blas_root;
                if constexpr (WIDE) root = geo.root4(inst_index);
                if (!(root & REF_LEAF)) cinv = mk3(1.0f / cd.x, 1.0f / cd.y, 1.0f / cd.z);
                blas_base = sp;
                cur = root;
            }
        }
        else
        {
            // ---- F: end of a traversal = one iteration of ray_skip_alpha_stochastically (bvh_custom.wgsl:154-180) ----
            if (COUNT) { rs[8]++; rs[9] += cF; phase_id = 4; }
            if constexpr (WIDE || SHORT)
            {
                // queries without a certificate leave their token for the binary tracer and write nothing: the path state the
                // re-trace starts from is the one this query started from (alpha skips included: the whole chain is redone)
                const bool give_up = isF && (SHORT ? cur == REF_OVER : flagged);
                const unsigned long long gm = __ballot(give_up);
                if (gm)
                {
                    uint32_t base = 0;
                    const int leader = __ffsll((long long)gm) - 1;
                    if ((int)lane == leader) base = atomicAdd(pb.retrace_counts + mode_row + shard, (uint32_t)__popcll(gm));
                    base = __shfl(base, leader);
                    if (give_up)
                    {
                        pb.retrace[2u * shard_base + base + (uint32_t)__popcll(gm & ((1ull << lane) - 1ull))] = MODE == 1 ? (slot * 2u + ray_k) : slot;
                        active = false;
                    }
                }
            }
            if (isF && active && MODE == 1)
            {
                const bool hit = best.t != LP_F32_MAX;
                const float4 rec = make_float4(hit ? best.t : 0.0f, hit ? best.u : 0.0f, hit ? best.v : 0.0f, __uint_as_float(hit ? (best.inst & INST_INDEX_MASK) : HIT_MISS));
                if (TYPE == LUPIN_PATHTRACE_MIS && ray_k == 0) { pb.next_hit[slot] = rec; pb.next_tri[slot] = best.tri; }
                else { pb.sh_hit1[slot] = rec; pb.sh_f1[slot].w = __uint_as_float(best.tri); }
                active = false;
            }
            if (isF && active && MODE == 0)
            {
                const bool hit = best.t != LP_F32_MAX;
                bool again = false;
                uint32_t inst_flags = 0u;
                if (hit)
                {
                    total_dst += best.t;
                    const uint32_t tag = best.inst >> INST_TAG_SHIFT;
                    inst_flags = (tag & 1u) | (((tag >> 1) & 0x1Fu) << 8);
                    best.inst &= INST_INDEX_MASK;
                    if (inst_flags & 1u)
                    {
                        Surface sf = resolve_surface(sc, best.inst, best.tri, best.u, best.v);
                        float opacity = surface_opacity(sc, sf);
                        if (opacity < 1.0f && rnd(rng) >= opacity)
                        {
                            alpha_k++;
                            again = alpha_k < 128u;   // MAX_OPACITY_BOUNCES (pathtracer.wgsl:1263)
                        }
                    }
                }
                if (again)
                {
                    // an end of traversal is reached in world space (pop() leaves an instance before it can run out of stack):
                    // co / cd ARE the world ray here
                    start_traversal(add(co, scale(cd, best.t)), cd);
                }
                else
                {
                    pb.hit[slot] = make_float4(total_dst, best.u, best.v, __uint_as_float(hit ? best.inst : HIT_MISS));
                    pb.hit_tri[slot] = best.tri;
                    if (rng != rng_in) pb.ori_rng[slot].w = __uint_as_float(rng);
                    if (TYPE == LUPIN_PATHTRACE_STANDARD && sc.sort_shade) pb.skey[slot] = shade_sort_key<true>(in_medium, !hit, inst_flags, rng);
                    active = false;
                }
            }
        }
        if (COUNT)
        {
            // the wave must have received what the round loaded before the clock is read: this build measures, it does not race
            __builtin_amdgcn_s_waitcnt(0);
            clk[phase_id] += __builtin_amdgcn_s_memtime() - t_round;
        }
    }
    if (COUNT) tally_flush(tally, work + LP_TALLY * MODE);   // the loop ends wave-uniformly: all lanes are here
    if (COUNT && MODE == 0 && !RETRACE && lane == 0)
    {
        clk[5] = __builtin_amdgcn_s_memtime() - t_loop;
        for (int k = 0; k < LP_ROUND_WORDS; k++) if (rs[k]) atomicAdd(&work[LP_ROUND_BASE + k], (unsigned long long)rs[k]);
        for (int k = 0; k < LP_CLOCK_WORDS; k++) if (clk[k]) atomicAdd(&work[LP_CLOCK_BASE + k], clk[k]);
    }
}

// LUPIN_VERIFY_WIDE=1 (checker, not product): before the tracing stage of an iteration, every queued path's first closest-hit
// query is run twice by one thread -- the reference's binary order and the four-wide traversal -- and compared word for word.
// verify[0] rays checked, [1] rays the wide traversal flagged (the product re-traces those), [2] UNFLAGGED rays whose wide
// result differs (the certificate's claim is that this stays 0), [3] rays that differ, flagged or not, [4..7] flagged rays by
// reason (second hit within the margin | ill-conditioned hit | triangle outside a box above it | stack bound); a ray can
// have several.
template <int MODE>   // 0: the queued paths' next closest-hit query; 1: the shadow rays k_shade recorded (two per entry)
__global__ void __launch_bounds__(LP_BLOCK) k_verify_wide(SceneDev sc, const FrameParams *__restrict__ fpp, PathBuffers pb, uint32_t iter,
                                                          uint32_t wide_pairs, unsigned long long *verify)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    const uint32_t shard = blockIdx.x % LP_SHARDS;
    const uint32_t count = pb.counts[iter * LP_SHARDS + shard];
    const uint32_t i = (blockIdx.x / LP_SHARDS) * LP_BLOCK + threadIdx.x;
    uint32_t checked = 0, flagged = 0, bad = 0, raw = 0, why_tie = 0, why_cond = 0, why_leaky = 0, why_stack = 0;
    const float eps = fpp->pc.ray_epsilon;
    const auto geo = geo_global(sc);
    for (uint32_t k = 0; k < (MODE == 1 ? 2u : 1u); k++)
    {
        bool have = i < count;
        f3 o = splat(0.0f), d = mk3(0.0f, 0.0f, 1.0f);
        if (have)
        {
            const uint32_t slot = pb.queue[iter & 1][(size_t)shard * pb.shard_cap + i];
            if (MODE == 0)
            {
                const float4 orr = pb.ori_rng[slot], dm = pb.dir_meta[slot];
                o = mk3(orr.x, orr.y, orr.z); d = mk3(dm.x, dm.y, dm.z);
            }
            else
            {
                const float4 so = pb.sh_org[slot];
                have = (__float_as_uint(so.w) & (1u << k)) != 0;
                const float4 dd = k ? pb.sh_d1[slot] : pb.sh_d0[slot];
                o = mk3(so.x, so.y, so.z); d = mk3(dd.x, dd.y, dd.z);
            }
        }
        if (have)
        {
            const Closest a = scene_closest(geo, sc, lds_stack, o, d, eps);
            bool flag;
            uint32_t why = 0u;
            const Closest b = scene_closest_wide(geo, sc, lds_stack, wide_pairs, o, d, eps, flag, &why);
            why_tie += (why & WIDE_WHY_TIE) ? 1u : 0u; why_cond += (why & WIDE_WHY_CONDITION) ? 1u : 0u;
            why_leaky += (why & WIDE_WHY_LEAKY) ? 1u : 0u; why_stack += (why & WIDE_WHY_STACK) ? 1u : 0u;
            const bool miss_a = a.t == LP_F32_MAX, miss_b = b.t == LP_F32_MAX;
            const bool same = (miss_a && miss_b) || (!miss_a && !miss_b && __float_as_uint(a.t) == __float_as_uint(b.t) && __float_as_uint(a.u) == __float_as_uint(b.u) &&
                                                     __float_as_uint(a.v) == __float_as_uint(b.v) && a.tri == b.tri && a.inst == b.inst);
            checked += 1u; flagged += flag ? 1u : 0u; raw += same ? 0u : 1u; bad += (!same && !flag) ? 1u : 0u;
            if (!same && !flag)
                printf("[lupin verify] uncertified difference: mode %d o %a %a %a d %a %a %a | binary t %a tri %u inst %u | wide t %a tri %u inst %u\n", MODE,
                       o.x, o.y, o.z, d.x, d.y, d.z, a.t, a.tri, a.inst, b.t, b.tri, b.inst);
        }
    }
    uint32_t v[8] = {checked, flagged, bad, raw, why_tie, why_cond, why_leaky, why_stack};
    #pragma unroll
    for (int k = 0; k < 8; k++)
    {
        uint32_t x = v[k];
        #pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if ((threadIdx.x & 63u) == 0u && x) atomicAdd(&verify[k], (unsigned long long)x);
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

// Shadow rays a vertex wants traced (MIS: BSDF- and light-sampled directions; Direct: the light ray).  A ray goes to the
// path's shadow record (pb.sh_d0 / sh_f0, sh_d1 / sh_f1) the moment it is known, so that nothing of it stays in registers
// across the rest of the vertex (round 2 carried both rays in a 17-float struct to the end of the iteration: with the light-
// pdf marches in between, k_shade<MIS> sat at 256 VGPRs + 448 bytes of scratch).  Only the two valid bits travel on.
struct ShadowRays
{
    PathBuffers *pb;     // nullptr: the caller has no use for shadow rays (debug view, Standard / Naive integrators)
    uint32_t slot;
    uint32_t flags;      // bit 0: ray 0 valid, bit 1: ray 1 valid
    __device__ __forceinline__ void emit(int k, f3 org, f3 dir, f3 factor, float scalar)
    {
        flags |= 1u << k;
        if (!pb) return;
        pb->sh_org[slot] = make_float4(org.x, org.y, org.z, 0.0f);   // both rays leave the vertex; the valid bits follow at the end of the iteration
        if (k == 0) { pb->sh_d0[slot] = make_float4(dir.x, dir.y, dir.z, scalar); pb->sh_f0[slot] = make_float4(factor.x, factor.y, factor.z, 0.0f); }
        else        { pb->sh_d1[slot] = make_float4(dir.x, dir.y, dir.z, scalar); pb->sh_f1[slot] = make_float4(factor.x, factor.y, factor.z, 0.0f); }
    }
};

struct PathRegs
{
    f3 ori, dir, weight, radiance;
    uint32_t rng;
    int bounce;
    bool in_medium;       // volume_stack_len == 1
    bool next_emission;
    Medium medium;
    // deferred weight update (light-pdf stage): weight *= pend_f / (0.5 pend_bp + 0.5 sample_lights_pdf(ori, dir))
    bool pending;
    f3 pend_f;
    float pend_bp;
};

// weight check and Russian roulette (:720-729)
__device__ __forceinline__ bool weight_checks_and_roulette(PathRegs &p)
{
    if (is_zero3(p.weight) || !finite3(p.weight)) return false;
    if (p.bounce > 3)
    {
        float survive = minf(0.99f, maxf(p.weight.x, maxf(p.weight.y, p.weight.z)));
        if (rnd(p.rng) >= survive) return false;
        p.weight = scale(p.weight, 1.0f / survive);
    }
    return true;
}

// One iteration of the integrator loop body after the closest-hit query.  Returns true when the
// path continues with (ori, dir) set for the next bounce, false on `break`.
// TYPE 0: pathtrace_standard (:588-733)   1: pathtrace_mis (:737-933)
//      2: pathtrace_naive (:942-1059)     3: pathtrace_direct (:1062-1245)
// Always inlined: as a real device function (the compiler's choice for the Direct integrator once its callees shrank) the
// call passes SceneDev / PathRegs / ShadowRays through scratch memory, which is slow ("outlined helpers") and,
// on ROCm 7.2, faulted: pointers of the scratch copy of SceneDev read back as material data (address = the bits of
// {roughness, metallic}) in k_shade<Direct> on materials4.
// DEFER (Standard only): the vertex stops before `sample_lights_pdf` -- the numerator and the BSDF pdf go to p.pend_*, and
// k_light_pdf finishes the iteration (weight, checks, Russian roulette: no random number is drawn in between, so the
// sequence of draws is the reference's).
template <int TYPE, typename Geo, bool SIMPLE = false, bool DEFER = false>
__device__ __forceinline__ bool integrate_vertex(const Geo &geo, const SceneDev &sc, uint32_t *stack, const FrameParams &fp, PathRegs &p,
                                 float4 hitrec, uint32_t hit_tri, ShadowRays &sh)
{
    static_assert(!DEFER || TYPE == LUPIN_PATHTRACE_STANDARD || TYPE == LUPIN_PATHTRACE_MIS, "the light-pdf stage serves the Standard and MIS integrators");
    constexpr bool DEFER_WEIGHT = DEFER && TYPE == LUPIN_PATHTRACE_STANDARD;   // MIS defers the two shadow-ray weights instead (below)
    const float eps = fp.pc.ray_epsilon;
    const uint32_t hit_inst = __float_as_uint(hitrec.w);
    p.pending = false;
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
                    sh.emit(1, hit_pos, li, mul(p.weight, bsdfcos), pdf);
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
                if (DEFER_WEIGHT)
                {
                    p.pend_bp = bsdf_pdf(mp, normal, outgoing, incoming);
                    p.pend_f = bsdf_eval(mp, normal, outgoing, incoming);
                    p.pending = true;
                }
                else
                {
                    float prob = 0.5f * bsdf_pdf(mp, normal, outgoing, incoming) + 0.5f * lights_pdf(geo, sc, stack, hit_pos, incoming, eps);
                    p.weight = mul(p.weight, divs(bsdf_eval(mp, normal, outgoing, incoming), prob));
                }
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
                // what the BSDF turn computed for its direction is what the weight update after both turns needs again
                // (:853: weight *= eval_bsdfcos(incoming) / sample_bsdfcos_pdf(incoming), the same pure functions of the same
                // arguments): kept in four registers instead of being evaluated twice
                f3 bsdfcos0 = splat(0.0f);
                float b_pdf0 = 0.0f;
                bool have0 = false;
                // DEFER (the light pdfs are computed by k_light_pdf_mis): two copies of the BSDF code, straight-line -- 138 VGPRs.
                // With the marches inline the loop stays rolled (one copy of the march code) and the barriers below matter.
                #pragma unroll(DEFER ? 2 : 1)
                for (int k = 0; k < 2; k++)
                {
                    const bool light_turn = (k != 0);
                    // The two turns share one copy of the BSDF code (the loop is not unrolled).  Left alone, the compiler hoists
                    // every sub-expression of bsdf_eval / bsdf_pdf that depends only on the normal and the outgoing direction out
                    // of the loop -- for all eight material families at once -- and keeps them alive across both turns and
                    // their light-pdf marches: 256 VGPRs + 424 bytes of scratch.  The turn's own copies of the two vectors pass
                    // through an empty asm, so nothing computed from them is loop-invariant any more.
                    f3 normal_k = normal, outgoing_k = outgoing;
                    MatPoint mp_k = mp;
                    asm volatile("" : "+v"(normal_k.x), "+v"(normal_k.y), "+v"(normal_k.z), "+v"(outgoing_k.x), "+v"(outgoing_k.y), "+v"(outgoing_k.z),
                                      "+v"(mp_k.color.x), "+v"(mp_k.color.y), "+v"(mp_k.color.z), "+v"(mp_k.roughness), "+v"(mp_k.metallic), "+v"(mp_k.ior));
                    f3 mi;
                    if (light_turn) mi = lights_sample(sc, hit_pos, p.rng);
                    else
                    {
                        float rnl = rnd(p.rng);
                        float ra = rnd(p.rng), rb = rnd(p.rng);
                        mi = bsdf_sample(mp_k, normal_k, outgoing_k, rnl, ra, rb);
                    }
                    if (is_zero3(mi)) break;
                    if (!light_turn) incoming = mi;

                    if (DEFER)
                    {
                        // k_light_pdf_mis turns the BSDF pdf parked in the scalar into the MIS weight (or drops the ray)
                        const f3 bsdfcos = bsdf_eval(mp_k, normal_k, outgoing_k, mi);
                        const float bp = bsdf_pdf(mp_k, normal_k, outgoing_k, mi);
                        if (!light_turn) { bsdfcos0 = bsdfcos; b_pdf0 = bp; have0 = true; }
                        if (none_zero3(bsdfcos)) sh.emit(k, hit_pos, mi, mul(p.weight, bsdfcos), bp);
                        continue;
                    }
                    // the march first: nothing of the BSDF terms is alive across it
                    const float light_pdf = lights_pdf(geo, sc, stack, hit_pos, mi, eps);
                    const f3 bsdfcos = bsdf_eval(mp_k, normal_k, outgoing_k, mi);
                    const float b_pdf = bsdf_pdf(mp_k, normal_k, outgoing_k, mi);
                    if (!light_turn) { bsdfcos0 = bsdfcos; b_pdf0 = b_pdf; have0 = true; }
                    float mis_w;
                    if (light_turn) mis_w = (light_pdf * light_pdf) / (light_pdf * light_pdf + b_pdf * b_pdf) / light_pdf;
                    else            mis_w = (b_pdf * b_pdf) / (b_pdf * b_pdf + light_pdf * light_pdf) / b_pdf;

                    // radiance += weight * bsdfcos * emission(mis_ray) * mis_weight   -- traced by k_shadow (:831-849);
                    // the BSDF-sampled ray's hit also becomes `next_intersection`
                    if (none_zero3(bsdfcos) && mis_w != 0.0f) sh.emit(k, hit_pos, mi, mul(p.weight, bsdfcos), mis_w);
                }
                if (have0) p.weight = mul(p.weight, divs(bsdfcos0, b_pdf0));
                else p.weight = mul(p.weight, divs(bsdf_eval(mp, normal, outgoing, incoming), bsdf_pdf(mp, normal, outgoing, incoming)));   // the BSDF turn broke off: incoming == 0
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
            if (DEFER_WEIGHT)
            {
                p.pend_bp = phase_pdf(p.medium, outgoing, incoming);
                p.pend_f = phase_eval(p.medium, outgoing, incoming);
                p.pending = true;
            }
            else
            {
                float prob = 0.5f * phase_pdf(p.medium, outgoing, incoming) + 0.5f * lights_pdf(geo, sc, stack, hit_pos, incoming, eps);
                p.weight = mul(p.weight, divs(phase_eval(p.medium, outgoing, incoming), prob));
            }
        }
    }

    p.ori = hit_pos;
    p.dir = incoming;
    if (DEFER_WEIGHT && p.pending) return true;
    return weight_checks_and_roulette(p);
}

// End of an iteration of the Standard / Naive loop for one path: a continuing path gets its state written back, a finished
// one is folded into the pixel and the pixel's next camera sample started (:234-239).  `r4` is the radiance as stored (only
// emitters change it).  Returns whether the slot still has work.
template <int TYPE>
__device__ __forceinline__ bool path_epilogue(const FrameParams &fp, const FrameParams *fpp, PathBuffers &pb, uint32_t slot, PathRegs &p, uint32_t sample, bool cont,
                                              bool vol_dirty, float4 r4)
{
    bool alive = false;
    if (cont)
    {
        alive = true;
        if (vol_dirty)
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
        float4 c4 = pb.color[slot];
        f3 cr = clamp_radiance(p.radiance, fp.pc.max_radiance);
        pb.color[slot] = make_float4(c4.x + cr.x, c4.y + cr.y, c4.z + cr.z, 0.0f);
        sample++;
        if (sample < fp.spp)
        {
            alive = true;
            uint32_t gx, gy, pslot;
            const uint32_t frame = slot_frame(fp, slot, pslot);
            slot_to_pixel(fp, pslot, gx, gy);
            camera_ray(fpp[frame], gx, gy, p.rng, p.ori, p.dir);
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

enum : int { SLOT_DONE = 0, SLOT_ALIVE = 1, SLOT_WAITS_FOR_LIGHT_PDF = 2 };
// k_shade -> k_light_pdf: the queue entry carries the slot and, above it, what became of the path in k_shade
constexpr uint32_t QUEUE_STATE_SHIFT = 30u, QUEUE_SLOT_MASK = (1u << QUEUE_STATE_SHIFT) - 1u, QUEUE_ENTRY_NONE = 0xFFFFFFFFu;

// Everything of one integrator-loop iteration after the closest-hit query, for one path; writes the path state
// back and returns whether the pixel still has work (the path continues, or its next camera sample was started) or, with
// DEFER, waits for the light-pdf stage.
template <int TYPE, typename Geo, bool SIMPLE = false, bool DEFER = false>
__device__ __forceinline__ int shade_path(const Geo &geo, const SceneDev &sc, uint32_t *stack, const FrameParams &fp, const FrameParams *fpp, PathBuffers &pb,
                                          uint32_t slot, float4 orr, float4 dm, uint32_t rng, float4 hitrec, uint32_t hit_tri)
{
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
    sh.pb = (TYPE == LUPIN_PATHTRACE_MIS || TYPE == LUPIN_PATHTRACE_DIRECT) ? &pb : nullptr;
    sh.slot = slot;
    sh.flags = 0u;
    bool cont = integrate_vertex<TYPE, Geo, SIMPLE, DEFER>(geo, sc, stack, fp, p, hitrec, hit_tri, sh);
    const bool vol_dirty = p.in_medium && !was_in_medium;
    if (DEFER && cont && p.pending)
    {
        // k_light_pdf finishes this iteration: park what it needs (the bounce count is still this iteration's)
        if (vol_dirty)
        {
            pb.vol0[slot] = make_float4(p.medium.density.x, p.medium.density.y, p.medium.density.z, p.medium.anisotropy);
            pb.vol1[slot] = make_float4(p.medium.scattering.x, p.medium.scattering.y, p.medium.scattering.z, 0.0f);
        }
        if (was_in_medium) pb.weight[slot] = make_float4(p.weight.x, p.weight.y, p.weight.z, 0.0f);   // only the transmittance term changed it
        if (p.radiance.x != r4.x || p.radiance.y != r4.y || p.radiance.z != r4.z)
            pb.radiance[slot] = make_float4(p.radiance.x, p.radiance.y, p.radiance.z, 0.0f);
        pb.sh_f0[slot] = make_float4(p.pend_f.x, p.pend_f.y, p.pend_f.z, p.pend_bp);
        const uint32_t nm = ((uint32_t)p.bounce & META_BOUNCE_MASK) | (p.in_medium ? META_VOLUME : 0u) |
                            (p.next_emission ? META_NEXT_EMISSION : 0u) | (sample << META_SAMPLE_SHIFT);
        pb.ori_rng[slot] = make_float4(p.ori.x, p.ori.y, p.ori.z, __uint_as_float(p.rng));
        pb.dir_meta[slot] = make_float4(p.dir.x, p.dir.y, p.dir.z, __uint_as_float(nm));
        return SLOT_WAITS_FOR_LIGHT_PDF;
    }
    if (cont)
    {
        p.bounce++;
        if (p.bounce > (int)fp.max_bounces) cont = false;   // loop condition `bounce <= MAX_BOUNCES` (:596)
    }

    if (TYPE == LUPIN_PATHTRACE_MIS || TYPE == LUPIN_PATHTRACE_DIRECT)
    {
        // hand the vertex to k_shadow: it adds the shadow-ray terms to `radiance` (the order of the additions is the
        // reference's) and only then folds a finished path into the pixel / starts the next sample
        if (vol_dirty)
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
        // the rays themselves went out when they were found (ShadowRays::emit, origin included); what is left are the valid bits
        pb.sh_org[slot].w = __uint_as_float(sh.flags);
        return SLOT_ALIVE;
    }
    return path_epilogue<TYPE>(fp, fpp, pb, slot, p, sample, cont, vol_dirty, r4) ? SLOT_ALIVE : SLOT_DONE;
}

// The light-pdf stage's share of an iteration (Standard): sample_lights_pdf for the direction k_shade chose
// (pathtracer.wgsl:2516-2549 -> bvh_custom.wgsl:112-152), the weight update it feeds (:652-656), the weight checks,
// Russian roulette and the loop condition (:720-729, :596).
template <int TYPE, typename Geo>
__device__ __forceinline__ bool light_pdf_path(const Geo &geo, const SceneDev &sc, uint32_t *stack, const FrameParams &fp, const FrameParams *fpp, PathBuffers &pb, uint32_t slot)
{
    const float4 orr = pb.ori_rng[slot], dm = pb.dir_meta[slot], w4 = pb.weight[slot], pd = pb.sh_f0[slot];
    const uint32_t meta = __float_as_uint(dm.w);
    PathRegs p;
    p.ori = mk3(orr.x, orr.y, orr.z);
    p.dir = mk3(dm.x, dm.y, dm.z);
    p.weight = mk3(w4.x, w4.y, w4.z);
    p.rng = __float_as_uint(orr.w);
    p.bounce = (int)(meta & META_BOUNCE_MASK);
    p.in_medium = (meta & META_VOLUME) != 0;
    p.next_emission = (meta & META_NEXT_EMISSION) != 0;
    p.pending = false;
    const uint32_t sample = meta >> META_SAMPLE_SHIFT;

    const float prob = 0.5f * pd.w + 0.5f * lights_pdf(geo, sc, stack, p.ori, p.dir, fp.pc.ray_epsilon);
    p.weight = mul(p.weight, divs(mk3(pd.x, pd.y, pd.z), prob));
    bool cont = weight_checks_and_roulette(p);
    if (cont)
    {
        p.bounce++;
        if (p.bounce > (int)fp.max_bounces) cont = false;
    }
    float4 r4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (!cont) r4 = pb.radiance[slot];   // a finished path is folded into the pixel
    p.radiance = mk3(r4.x, r4.y, r4.z);
    return path_epilogue<TYPE>(fp, fpp, pb, slot, p, sample, cont, false, r4);
}

// Scenes with several material families: before k_shade, each window of LP_SORT_WINDOW queue entries is counting-sorted in place
// by what the path will execute (material type of the hit | miss | inside a medium), so that a k_shade wave runs one BSDF
// family instead of several (a 256-path window inside k_shade left two or three per wave).  Which queue position holds which
// path does not matter: all path state lives in the path's slot.
#ifndef LP_SORT_WINDOW
#define LP_SORT_WINDOW 4096
#endif
// The key is shade_sort_key (above); FROM_TRACER: the persistent tracer wrote it when it finished the path's query.
template <bool PEEK_COIN, bool FROM_TRACER>
__global__ void __launch_bounds__(LP_BLOCK) k_sort_queue(SceneDev sc, PathBuffers pb, uint32_t iter)
{
    constexpr uint32_t PER_THREAD = LP_SORT_WINDOW / LP_BLOCK;
    constexpr uint32_t NUM_KEYS = LP_SORT_KEYS;
    __shared__ uint32_t sorted[LP_SORT_WINDOW];
    __shared__ uint32_t bins[NUM_KEYS];
    const uint32_t shard = blockIdx.x % LP_SHARDS;
    const uint32_t count = pb.counts[iter * LP_SHARDS + shard];
    const uint32_t base_i = (blockIdx.x / LP_SHARDS) * LP_SORT_WINDOW;
    if (base_i >= count) return;   // block-uniform
    uint32_t *entries = pb.queue[iter & 1] + (size_t)shard * pb.shard_cap + base_i;
    const uint32_t valid = min(LP_SORT_WINDOW, count - base_i);
    if (threadIdx.x < NUM_KEYS) bins[threadIdx.x] = 0u;
    __syncthreads();
    // Rounds of independent loads instead of sixteen dependent chains: the slots, then either the key the persistent tracer
    // left for the path (FROM_TRACER: 4 bytes per path) or the path's hit / meta / RNG words and the hit instances' flags.
    // Entries past the window's end re-read its last entry and are dropped below.
    uint32_t my_slot[PER_THREAD], my_key[PER_THREAD], my_rank[PER_THREAD];
    #pragma unroll
    for (uint32_t r = 0; r < PER_THREAD; r++) my_slot[r] = entries[min(r * LP_BLOCK + threadIdx.x, valid - 1u)];
    if constexpr (FROM_TRACER)
    {
        #pragma unroll
        for (uint32_t r = 0; r < PER_THREAD; r++) my_key[r] = pb.skey[my_slot[r]] & (NUM_KEYS - 1u);
    }
    else
    {
        uint32_t hit_w[PER_THREAD], meta_w[PER_THREAD], rng_w[PER_THREAD], flags_w[PER_THREAD];
        #pragma unroll
        for (uint32_t r = 0; r < PER_THREAD; r++)
        {
            hit_w[r] = __float_as_uint(pb.hit[my_slot[r]].w);
            meta_w[r] = __float_as_uint(pb.dir_meta[my_slot[r]].w);
            rng_w[r] = PEEK_COIN ? __float_as_uint(pb.ori_rng[my_slot[r]].w) : 0u;
        }
        #pragma unroll
        for (uint32_t r = 0; r < PER_THREAD; r++) flags_w[r] = sc.instances[hit_w[r] == HIT_MISS ? 0u : hit_w[r]].flags;   // the launch requires an instance
        #pragma unroll
        for (uint32_t r = 0; r < PER_THREAD; r++) my_key[r] = shade_sort_key<PEEK_COIN>((meta_w[r] & META_VOLUME) != 0, hit_w[r] == HIT_MISS, flags_w[r], rng_w[r]);
    }
    #pragma unroll
    for (uint32_t r = 0; r < PER_THREAD; r++)
    {
        my_rank[r] = 0u;
        if (r * LP_BLOCK + threadIdx.x < valid) my_rank[r] = atomicAdd(&bins[my_key[r]], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64)   // exclusive prefix of the 64 bins by the first wave
    {
        const uint32_t c = bins[threadIdx.x];
        uint32_t incl = c;
        for (int off = 1; off < 64; off <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, off); if ((int)threadIdx.x >= off) incl += v; }
        bins[threadIdx.x] = incl - c;
    }
    __syncthreads();
    #pragma unroll
    for (uint32_t r = 0; r < PER_THREAD; r++)
        if (r * LP_BLOCK + threadIdx.x < valid) sorted[bins[my_key[r]] + my_rank[r]] = my_slot[r];
    __syncthreads();
    #pragma unroll
    for (uint32_t r = 0; r < PER_THREAD; r++)
    {
        const uint32_t j = r * LP_BLOCK + threadIdx.x;
        if (j < valid) entries[j] = sorted[j];
    }
}

// SIMPLE: scenes of untextured matte surfaces without environments (LupinScene::simple_matte, decided at upload) get a
// k_shade in which those facts are compile-time constants: same arithmetic on the paths that exist, none of the code
// for the ones that cannot.
template <int TYPE, bool LDSGEO, bool SIMPLE, bool DEFER = false>
__global__ void __attribute__((amdgpu_waves_per_eu(TYPE == 1 ? (DEFER ? LP_MIS_DEFER_SHADE_WAVES : LP_MIS_SHADE_WAVES) : (SIMPLE ? LP_SIMPLE_SHADE_WAVES : (DEFER ? LP_DEFER_SHADE_WAVES : LP_SHADE_WAVES)), 8))) __launch_bounds__(LP_BLOCK) k_shade(SceneDev sc, const FrameParams *__restrict__ fpp, PathBuffers pb, uint32_t iter,
                                                    unsigned long long *shard_stats, uint32_t stack_words)
{
    const FrameParams fp = *fpp;
    if (SIMPLE) sc.num_envs = 0;   // a fact of a simple_matte scene, constant from here on
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    const auto geo = make_geo<LDSGEO>(sc, lds_stack, stack_words);
    const uint32_t shard = blockIdx.x % LP_SHARDS;
    const uint32_t count = pb.counts[iter * LP_SHARDS + shard];
    const uint32_t i = (blockIdx.x / LP_SHARDS) * LP_BLOCK + threadIdx.x;
    int state = SLOT_DONE;
    bool mine = i < count;
    uint32_t slot = 0;
    if (mine) slot = pb.queue[iter & 1][(size_t)shard * pb.shard_cap + i];
    if (mine)
    {
        const float4 orr = pb.ori_rng[slot];
        state = shade_path<TYPE, typename GeoOf<LDSGEO>::type, SIMPLE, DEFER>(geo, sc, lds_stack, fp, fpp, pb, slot, orr, pb.dir_meta[slot], __float_as_uint(orr.w), pb.hit[slot], pb.hit_tri[slot]);
    }
    if (i == 0 && iter == 0) shard_stats[shard * 2 + 1] += (unsigned long long)count * fp.spp;
    if (TYPE == LUPIN_PATHTRACE_MIS || TYPE == LUPIN_PATHTRACE_DIRECT) return;   // k_shadow appends
    if (DEFER)
    {
        // with the block sort, thread i shaded some other entry of its block: the tagged entries are a permutation of the block's
        if (i < count) pb.queue[iter & 1][(size_t)shard * pb.shard_cap + i] = mine ? (slot | ((uint32_t)state << QUEUE_STATE_SHIFT)) : QUEUE_ENTRY_NONE;
        return;
    }
    queue_append(state == SLOT_ALIVE, slot, pb.queue[(iter + 1) & 1] + (size_t)shard * pb.shard_cap, &pb.counts[(iter + 1) * LP_SHARDS + shard]);
}

// Light-pdf stage: walks the iteration's queue again.  The block's waiting vertices are compacted to its first threads (so
// all lanes of a wave march, except in the block's last active wave), finished there, and the verdicts go back to the
// entries' own threads: the append keeps the queue's order, which the persistent tracer's static partition relies on for
// balance (survivors appended behind k_shade's own appends cost k_extend 4 %).
template <int TYPE, bool LDSGEO>
__global__ void __attribute__((amdgpu_waves_per_eu(LP_LIGHT_PDF_WAVES, 8))) __launch_bounds__(LP_BLOCK) k_light_pdf(SceneDev sc, const FrameParams *__restrict__ fpp, PathBuffers pb, uint32_t iter, uint32_t stack_words)
{
    const FrameParams fp = *fpp;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    __shared__ uint32_t waiting[LP_BLOCK];        // compacted: thread index of the waiting entry
    __shared__ uint32_t entry_slot[LP_BLOCK];
    __shared__ uint32_t verdict[LP_BLOCK];
    __shared__ uint32_t wave_total[LP_BLOCK / 64];
    const auto geo = make_geo<LDSGEO>(sc, lds_stack, stack_words);
    const uint32_t shard = blockIdx.x % LP_SHARDS;
    const uint32_t count = pb.counts[iter * LP_SHARDS + shard];
    const uint32_t i = (blockIdx.x / LP_SHARDS) * LP_BLOCK + threadIdx.x;
    if ((blockIdx.x / LP_SHARDS) * LP_BLOCK >= count) return;   // block-uniform
    uint32_t *entries = pb.queue[iter & 1] + (size_t)shard * pb.shard_cap + (size_t)(blockIdx.x / LP_SHARDS) * LP_BLOCK;
    const uint32_t entry = i < count ? entries[threadIdx.x] : QUEUE_ENTRY_NONE;
    const uint32_t state = entry == QUEUE_ENTRY_NONE ? (uint32_t)SLOT_DONE : entry >> QUEUE_STATE_SHIFT;
    const uint32_t slot = entry & QUEUE_SLOT_MASK;
    const bool waits = state == (uint32_t)SLOT_WAITS_FOR_LIGHT_PDF;
    const unsigned long long wmask = __ballot(waits);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) wave_total[wave] = (uint32_t)__popcll(wmask);
    verdict[threadIdx.x] = 0u;
    entry_slot[threadIdx.x] = slot;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < LP_BLOCK / 64; w++) { const uint32_t c = wave_total[w]; if (w < wave) before += c; total += c; }
    if (waits) waiting[before + (uint32_t)__popcll(wmask & ((1ull << lane) - 1ull))] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x < total)
    {
        const uint32_t w = waiting[threadIdx.x];
        if (light_pdf_path<TYPE, typename GeoOf<LDSGEO>::type>(geo, sc, lds_stack, fp, fpp, pb, entry_slot[w])) verdict[w] = 1u;
    }
    __syncthreads();
    const bool alive = state == (uint32_t)SLOT_ALIVE || (waits && verdict[threadIdx.x] != 0u);
    queue_append(alive, slot, pb.queue[(iter + 1) & 1] + (size_t)shard * pb.shard_cap, &pb.counts[(iter + 1) * LP_SHARDS + shard]);
}

// Light-pdf stage of the MIS integrator: k_shade<MIS, DEFER> records up to two shadow-ray candidates per vertex with the
// BSDF pdf in the scalar; this pass computes sample_lights_pdf for each (pathtracer.wgsl:2516-2549), the power-heuristic
// weight (:822-829), and keeps the ray only if the weight is non-zero -- the reference's condition.  The block's
// candidates (<= 512) are compacted so that whole waves march.
template <bool LDSGEO>
__global__ void __attribute__((amdgpu_waves_per_eu(LP_LIGHT_PDF_MIS_WAVES, 8))) __launch_bounds__(LP_BLOCK) k_light_pdf_mis(SceneDev sc, const FrameParams *__restrict__ fpp, PathBuffers pb, uint32_t iter, uint32_t stack_words)
{
    const FrameParams fp = *fpp;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    __shared__ uint32_t items[2 * LP_BLOCK];      // compacted: 2 * thread index of the entry + ray
    __shared__ uint32_t entry_slot[LP_BLOCK], entry_flags[LP_BLOCK];
    __shared__ uint32_t wave_total[LP_BLOCK / 64];
    const auto geo = make_geo<LDSGEO>(sc, lds_stack, stack_words);
    const uint32_t shard = blockIdx.x % LP_SHARDS;
    const uint32_t count = pb.counts[iter * LP_SHARDS + shard];
    const uint32_t i = (blockIdx.x / LP_SHARDS) * LP_BLOCK + threadIdx.x;
    if ((blockIdx.x / LP_SHARDS) * LP_BLOCK >= count) return;   // block-uniform
    uint32_t slot = 0, flags = 0;
    if (i < count)
    {
        slot = pb.queue[iter & 1][(size_t)shard * pb.shard_cap + i];
        flags = __float_as_uint(pb.sh_org[slot].w) & 3u;
    }
    entry_slot[threadIdx.x] = slot;
    entry_flags[threadIdx.x] = flags;
    const uint32_t mine = (flags & 1u) + (flags >> 1);
    // block-wide exclusive prefix of `mine`
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t incl = mine;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, off); if ((int)lane >= off) incl += v; }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    uint32_t before = incl - mine, total = 0;
    for (uint32_t w = 0; w < LP_BLOCK / 64; w++) { const uint32_t c = wave_total[w]; if (w < wave) before += c; total += c; }
    if (flags & 1u) items[before] = 2u * threadIdx.x;
    if (flags & 2u) items[before + (flags & 1u)] = 2u * threadIdx.x + 1u;
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < total; j += LP_BLOCK)
    {
        const uint32_t it = items[j], t = it >> 1, k = it & 1u;
        const uint32_t s = entry_slot[t];
        const float4 so = pb.sh_org[s];
        float4 *rec = &(k ? pb.sh_d1 : pb.sh_d0)[s];
        const float4 dd = *rec;
        const float light_pdf = lights_pdf(geo, sc, lds_stack, mk3(so.x, so.y, so.z), mk3(dd.x, dd.y, dd.z), fp.pc.ray_epsilon);
        const float b_pdf = dd.w;
        float mis_w;
        if (k) mis_w = (light_pdf * light_pdf) / (light_pdf * light_pdf + b_pdf * b_pdf) / light_pdf;
        else   mis_w = (b_pdf * b_pdf) / (b_pdf * b_pdf + light_pdf * light_pdf) / b_pdf;
        if (mis_w != 0.0f) rec->w = mis_w;
        else atomicAnd(&entry_flags[t], ~(1u << k));
    }
    __syncthreads();
    if (entry_flags[threadIdx.x] != flags)
    {
        const float4 so = pb.sh_org[slot];
        pb.sh_org[slot] = make_float4(so.x, so.y, so.z, __uint_as_float(entry_flags[threadIdx.x]));
    }
}

// emission of a surface point: emission_sample * mat.emission of get_material_point (pathtracer.wgsl:1295-1298,1315)
__device__ __forceinline__ f3 surface_emission(const SceneDev &sc, const Surface &s)
{
    const LupinMaterial *m = &sc.inst_materials[s.inst];
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
                uint32_t gx, gy, pslot;
                const uint32_t frame = slot_frame(fp, slot, pslot);
                slot_to_pixel(fp, pslot, gx, gy);
                f3 o, d;
                camera_ray(fpp[frame], gx, gy, rng, o, d);
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

// pathtrace_main tail (pathtracer.wgsl:275-289): /spp, max(0), progressive blend with prev_frame, Rgba16Float store.
// Accumulation modes (lupin_hip_set_accumulation_mode):
//   F16 running average (reference-faithful): prev_frame is the f16 texel the previous call stored, so the running mean
//       is re-quantised every frame (SURVEY 7 "f16 accumulation semantics");
//   F32: the same recurrence on an f32 shadow of the textures (prev32 / out32, one float4 per pixel); the f16 texel
//       is the rounded view of it.  A prev_frame without a shadow (uploaded, or rendered in the other mode) is read as f16.
// Four channels leave as one 8-byte store.
__device__ __forceinline__ uint32_t f16_bits(float v, bool rne)
{
    return (uint32_t)__half_as_ushort(rne ? __float2half_rn(v) : __float2half_rz(v));
}
__device__ __forceinline__ void store_rgba16f(__half *out, size_t pixel, f3 c, bool rne)
{
    uint2 w;
    w.x = f16_bits(c.x, rne) | (f16_bits(c.y, rne) << 16);
    w.y = f16_bits(c.z, rne) | (0x3C00u << 16);   // alpha = 1.0
    reinterpret_cast<uint2 *>(out)[pixel] = w;
}
__device__ __forceinline__ f3 load_rgb16f(const __half *tex, size_t pixel)
{
    const uint2 w = reinterpret_cast<const uint2 *>(tex)[pixel];
    return mk3(half_bits_to_float(w.x & 0xFFFFu), half_bits_to_float(w.x >> 16), half_bits_to_float(w.y & 0xFFFFu));
}
// The frames of a batch, as the resolve sees them: where each one's result goes, which of those stores survive (a texture
// written again by a later frame of the batch keeps only the later value), and the blend weights' counters.
struct ResolveBatch
{
    __half *target[LP_MAX_BATCH];
    uint32_t accum_counter[LP_MAX_BATCH];
    uint32_t store_mask;          // bit k: frame k's value is the last one written to its texture
    uint32_t count;
};
__global__ void __launch_bounds__(LP_BLOCK) k_resolve(FrameParams fp, PathBuffers pb, uint32_t n, ResolveBatch rb,
                                                      const __half *prev, const float4 *prev32, float4 *out32)
{
    uint32_t slot = blockIdx.x * LP_BLOCK + threadIdx.x;
    if (slot >= n) return;   // n = slots per frame
    uint32_t gx, gy;
    slot_to_pixel(fp, slot, gx, gy);
    if (gx >= fp.width || gy >= fp.height) return;
    const size_t px = (size_t)gy * fp.width + gx;
    const float spp = (float)fp.spp;
    const bool rne = fp.store_rne != 0;
    f3 running = splat(0.0f);   // the value frame k blends with: prev_frame of the first call, then what the previous frame STORED
    bool have_prev = false;
    for (uint32_t k = 0; k < rb.count; k++)
    {
        const float4 c4 = pb.color[(size_t)k * fp.frame_slots + slot];
        f3 c = mk3(maxf(c4.x / spp, 0.0f), maxf(c4.y / spp, 0.0f), maxf(c4.z / spp, 0.0f));
        if (rb.accum_counter[k] != 0)
        {
            const float w = 1.0f / (float)rb.accum_counter[k];
            f3 pc = running;
            if (!have_prev)
            {
                if (prev32) { const float4 p = prev32[px]; pc = mk3(p.x, p.y, p.z); }
                else pc = load_rgb16f(prev, px);
            }
            c = mk3(maxf(pc.x * (1.0f - w) + c.x * w, 0.0f), maxf(pc.y * (1.0f - w) + c.y * w, 0.0f), maxf(pc.z * (1.0f - w) + c.z * w, 0.0f));
        }
        if (out32) out32[px] = make_float4(c.x, c.y, c.z, 1.0f);   // f32 accumulation is never batched: count == 1
        if (rb.store_mask & (1u << k)) store_rgba16f(rb.target[k], px, c, rne);
        // the next frame of the batch reads this texel back as Rgba16Float (pathtracer.wgsl:279-285): round it exactly as the store does
        running = mk3(half_bits_to_float(f16_bits(c.x, rne)), half_bits_to_float(f16_bits(c.y, rne)), half_bits_to_float(f16_bits(c.z, rne)));
        have_prev = true;
    }
}

// device-to-device copy: the measured HBM peak bench.py reports next to the nominal one
__global__ void __launch_bounds__(LP_BLOCK) k_copy_bw(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * LP_BLOCK + threadIdx.x;
    if (i < n) dst[i] = src[i];
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
            sh.pb = nullptr; sh.slot = 0u; sh.flags = 0u;
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

// the same probe through the four-wide traversal; out_flag = the traversal asks for a re-trace (its result is then unspecified)
__global__ void __launch_bounds__(LP_BLOCK) k_trace_wide(SceneDev sc, uint32_t n, const float *ori, const float *dir, float eps, uint32_t wide_pairs,
                                                         uint32_t *out_hit, float *out_dst, float *out_uv, uint32_t *out_inst, uint32_t *out_tri, uint32_t *out_flag)
{
    extern __shared__ uint32_t lds_stack[];
    uint32_t i = blockIdx.x * LP_BLOCK + threadIdx.x;
    if (i >= n) return;
    f3 o = mk3(ori[i * 3 + 0], ori[i * 3 + 1], ori[i * 3 + 2]);
    f3 d = mk3(dir[i * 3 + 0], dir[i * 3 + 1], dir[i * 3 + 2]);
    bool flag;
    Closest c = scene_closest_wide(geo_global(sc), sc, lds_stack, wide_pairs, o, d, eps, flag);
    bool hit = c.t != LP_F32_MAX;
    out_flag[i] = flag ? 1u : 0u;
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

// Tile pack / unpack for the multi-GPU gather.  Payload of a rank = its tiles in ascending order (include/lupin_tiles.h),
// each tile row-major, 8 B per pixel.  One block per tile: the block first sums the pixel counts of the owner's earlier
// tiles (a few hundred terms at most, strided over the threads), then copies the tile's rows.
//   mode 0: pack the tiles `rank` owns (blockIdx.x = j-th owned tile) into packed[0 ..)
//   mode 1: unpack the tiles `rank` owns from packed[0 ..)
//   mode 2: unpack every tile NOT owned by `rank` (blockIdx.x = tile of the frame) from the all-gathered buffer
//           packed[owner * capacity_px + ..) -- the whole readback scatter in one launch
// The unpack modes also write the widened texel into `accum32` when the texture carries a valid f32 accumulator.
__global__ void __launch_bounds__(LP_BLOCK) k_tiles_copy(uint2 *tex, uint2 *packed, float4 *accum32, uint32_t width, uint32_t height, uint32_t tile_px,
                                                        uint32_t rank, uint32_t world, unsigned long long capacity_px, int mode)
{
    const uint32_t ntx = (width - 1) / tile_px + 1;
    uint32_t t, owner, j;
    if (mode == 2)
    {
        t = blockIdx.x;
        owner = lupin_tile_owner(t, ntx, world);
        if (owner == rank) return;
        j = lupin_owned_index(t, world, ntx);
    }
    else { owner = rank; j = blockIdx.x; t = lupin_owned_tile(j, rank, world, ntx); }
    __shared__ unsigned long long part[LP_BLOCK];
    unsigned long long mine = 0;
    for (uint32_t i = threadIdx.x; i < j; i += LP_BLOCK)
    {
        const uint32_t q = lupin_owned_tile(i, owner, world, ntx);
        const uint32_t qx = (q % ntx) * tile_px, qy = (q / ntx) * tile_px;
        mine += (unsigned long long)min(tile_px, width - qx) * min(tile_px, height - qy);
    }
    part[threadIdx.x] = mine;
    __syncthreads();
    for (uint32_t s = LP_BLOCK / 2; s > 0; s >>= 1)
    {
        if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    const unsigned long long before = part[0] + (mode == 2 ? capacity_px * owner : 0ull);
    const uint32_t ox = (t % ntx) * tile_px, oy = (t / ntx) * tile_px;
    const uint32_t w = min(tile_px, width - ox), h = min(tile_px, height - oy);
    for (uint32_t p = threadIdx.x; p < w * h; p += LP_BLOCK)
    {
        const uint32_t x = ox + p % w, y = oy + p / w;
        if (mode == 0) packed[before + p] = tex[(size_t)y * width + x];
        else
        {
            const uint2 w = packed[before + p];
            tex[(size_t)y * width + x] = w;
            if (accum32)   // the texture's f32 accumulator follows (LUPIN_ACCUM_F32): the unpacked texel widened
                accum32[(size_t)y * width + x] = make_float4(half_bits_to_float(w.x & 0xFFFFu), half_bits_to_float(w.x >> 16),
                                                             half_bits_to_float(w.y & 0xFFFFu), half_bits_to_float(w.y >> 16));
        }
    }
}

