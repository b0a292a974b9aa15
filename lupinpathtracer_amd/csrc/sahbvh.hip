// sahbvh.hip -- the reference's BLAS builder on the device ("next" row 8f-1, round-1 review item 10).
//
// lupin_hip_build_bvh_sah_device builds, level by level on the GPU, THE SAME TREE as lp::build_bvh
// (data_structures.rs:196-475, restated for the CPU in builders.cpp): top-down, 5 bins per axis over the centroid bounds
// padded by +-0.001, cost = half surface area x triangle count, a split only if some plane is cheaper than the node and
// both sides are non-empty, partition by centroid[axis] <= pos, depth capped by the reference's 25-entry stack.  Every
// decision is a function of the SET of triangles of a node (min / max reductions, counts) and of the same f32 expressions
// in the same order (-ffp-contract=off, IEEE division), so node boxes, split planes and the triangle set of every node
// equal the CPU builder's bit for bit.  What differs is bookkeeping that cannot change an image except under an exact
// closest-hit tie: nodes are numbered level by level (children adjacent, as the format requires) instead of in the CPU's
// depth-first order, and both sides of a partition keep their relative order (the CPU's in-place swaps permute the right
// side).  tests/test_sah_device.py checks all of that against lupin_build_bvh.
//
// One level = a handful of kernels over all n triangle positions:
//   centroid bounds per node -> bin boxes / counts per node, axis, bin -> one thread per node replays choose_split ->
//   left flags -> prefix sum (hipCUB) -> children allocated by a prefix sum over the splits that happened (deterministic
//   numbering) -> stable scatter.  Waves whose 64 positions belong to one node reduce in registers and issue one atomic
//   per value; only waves straddling nodes (small nodes: little contention) use per-lane atomics.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "lupin_internal.hpp"

namespace {

#define SAH_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { cleanup(); return lupin_internal_fail(LUPIN_ERR_HIP, hipGetErrorString(e__)); } } while (0)

constexpr int kBlock = 256;
constexpr int NUM_BINS = 5;
constexpr uint32_t INACTIVE = 0xFFFFFFFFu;
constexpr uint32_t BIN_WORDS = 7;                           // lo.xyz hi.xyz (order-preserving uints) + count
constexpr uint32_t SLOT_BIN_WORDS = 3 * NUM_BINS * BIN_WORDS;

__device__ __forceinline__ uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

struct Slot { uint32_t node, begin, count, depth; };        // one frontier node of the current level
struct SplitDev
{
    uint32_t performed, axis;
    float pos;
    float lbox[6], rbox[6];                                 // lo.xyz hi.xyz of the two sides (prefix / suffix unions of the bins)
    uint32_t valid, left_count, child_rank;                 // filled by k_finalize: both sides non-empty; rank among the level's valid splits
};

__global__ void __launch_bounds__(kBlock) k_prepare(const float4 *verts, const uint32_t *indices, uint32_t n, float4 *tri_lo, float4 *tri_hi, float4 *cen,
                                                    uint32_t *perm, uint32_t *pslot, uint32_t *root_box)
{
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    const bool live = t < n;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    if (live)
    {
        const float4 a = verts[indices[3 * t + 0]], b = verts[indices[3 * t + 1]], c = verts[indices[3 * t + 2]];
        // compute_tri_centroid (base.rs:1155-1159): (t0 + t1 + t2) / 3.0 ; compute_tri_bounds (:1136-1153)
        cen[t] = make_float4(((a.x + b.x) + c.x) / 3.0f, ((a.y + b.y) + c.y) / 3.0f, ((a.z + b.z) + c.z) / 3.0f, 0.0f);
        lo[0] = fminf(a.x, fminf(b.x, c.x)); lo[1] = fminf(a.y, fminf(b.y, c.y)); lo[2] = fminf(a.z, fminf(b.z, c.z));
        hi[0] = fmaxf(a.x, fmaxf(b.x, c.x)); hi[1] = fmaxf(a.y, fmaxf(b.y, c.y)); hi[2] = fmaxf(a.z, fmaxf(b.z, c.z));
        tri_lo[t] = make_float4(lo[0], lo[1], lo[2], 0.0f);
        tri_hi[t] = make_float4(hi[0], hi[1], hi[2], 0.0f);
        perm[t] = t;
        pslot[t] = 0u;
    }
    // root box: compute_aabb starts from Aabb::default() == zeros (data_structures.rs:529-540), so it contains the origin
    for (int ax = 0; ax < 3; ax++)
    {
        uint32_t l = live ? f2ord(lo[ax]) : 0xFFFFFFFFu, h = live ? f2ord(hi[ax]) : 0u;
        for (int off = 32; off > 0; off >>= 1) { l = min(l, (uint32_t)__shfl_xor((int)l, off)); h = max(h, (uint32_t)__shfl_xor((int)h, off)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&root_box[ax], l); atomicMax(&root_box[3 + ax], h); }
    }
}

__global__ void k_root(const uint32_t *root_box, uint32_t n, LupinBvhNode *nodes, Slot *slots)
{
    LupinBvhNode rn;
    for (int ax = 0; ax < 3; ax++) { rn.aabb_min[ax] = ord2f(root_box[ax]); rn.aabb_max[ax] = ord2f(root_box[3 + ax]); }
    rn.tri_begin_or_first_child = 0; rn.tri_count = n;
    nodes[0] = rn;
    slots[0] = Slot{0u, 0u, n, 1u};
}

__global__ void __launch_bounds__(kBlock) k_clear_acc(uint32_t nslots, uint32_t *cb, uint32_t *bins)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < nslots * 6u) cb[i] = (i % 6u) < 3u ? 0xFFFFFFFFu : 0u;
    if (i < nslots * SLOT_BIN_WORDS) { const uint32_t w = i % BIN_WORDS; bins[i] = w < 3u ? 0xFFFFFFFFu : 0u; }
}

// centroid bounds of every frontier node (choose_split's first loop, data_structures.rs:377-385)
__global__ void __launch_bounds__(kBlock) k_centroid_bounds(uint32_t n, const uint32_t *perm, const uint32_t *pslot, const float4 *cen, uint32_t *cb)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t slot = p < n ? pslot[p] : INACTIVE;
    const bool live = slot != INACTIVE;
    float c[3] = {0, 0, 0};
    if (live) { const float4 v = cen[perm[p]]; c[0] = v.x; c[1] = v.y; c[2] = v.z; }
    const uint32_t first = (uint32_t)__shfl((int)slot, 0);
    const bool uniform = __ballot(slot != first) == 0ull;
    if (uniform)
    {
        if (first == INACTIVE) return;
        for (int ax = 0; ax < 3; ax++)
        {
            uint32_t l = f2ord(c[ax]), h = l;
            for (int off = 32; off > 0; off >>= 1) { l = min(l, (uint32_t)__shfl_xor((int)l, off)); h = max(h, (uint32_t)__shfl_xor((int)h, off)); }
            if ((threadIdx.x & 63) == 0) { atomicMin(&cb[first * 6u + ax], l); atomicMax(&cb[first * 6u + 3u + ax], h); }
        }
    }
    else if (live)
        for (int ax = 0; ax < 3; ax++) { atomicMin(&cb[slot * 6u + ax], f2ord(c[ax])); atomicMax(&cb[slot * 6u + 3u + ax], f2ord(c[ax])); }
}

__device__ __forceinline__ int bin_of(float c, float cmin, float scale)
{
    const float f = floorf((c - cmin) * scale);
    return (f >= (float)(NUM_BINS - 1)) ? NUM_BINS - 1 : ((f > 0.0f) ? (int)f : 0);   // `as usize` saturates, then clamp (data_structures.rs:400-403)
}

// bin boxes and counts (choose_split's second loop, :395-408)
__global__ void __launch_bounds__(kBlock) k_bins(uint32_t n, const uint32_t *perm, const uint32_t *pslot, const float4 *cen, const float4 *tri_lo, const float4 *tri_hi,
                                                 const uint32_t *cb, uint32_t *bins)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t slot = p < n ? pslot[p] : INACTIVE;
    const bool live = slot != INACTIVE;
    const uint32_t first = (uint32_t)__shfl((int)slot, 0);
    const bool uniform = __ballot(slot != first) == 0ull;
    if (uniform && first == INACTIVE) return;
    float c[3] = {0, 0, 0}, lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    if (live)
    {
        const uint32_t t = perm[p];
        const float4 v = cen[t], l = tri_lo[t], h = tri_hi[t];
        c[0] = v.x; c[1] = v.y; c[2] = v.z; lo[0] = l.x; lo[1] = l.y; lo[2] = l.z; hi[0] = h.x; hi[1] = h.y; hi[2] = h.z;
    }
    const uint32_t s = live ? slot : first;
    for (int ax = 0; ax < 3; ax++)
    {
        float cmin = 0.0f, cmax = 0.0f;
        int bi = -1;
        if (live)
        {
            cmin = ord2f(cb[s * 6u + ax]); cmax = ord2f(cb[s * 6u + 3u + ax]);
            if (cmin != cmax)
            {
                const float EPS = 0.001f;
                cmin -= EPS; cmax += EPS;
                const float scale = (float)NUM_BINS / (cmax - cmin);
                bi = bin_of(c[ax], cmin, scale);
            }
        }
        if (uniform)
        {
            for (int b = 0; b < NUM_BINS; b++)
            {
                const bool in = bi == b;
                const unsigned long long who = __ballot(in);
                if (who == 0ull) continue;
                uint32_t v[6];
                for (int k = 0; k < 3; k++) { v[k] = in ? f2ord(lo[k]) : 0xFFFFFFFFu; v[3 + k] = in ? f2ord(hi[k]) : 0u; }
                for (int off = 32; off > 0; off >>= 1)
                    for (int k = 0; k < 3; k++) { v[k] = min(v[k], (uint32_t)__shfl_xor((int)v[k], off)); v[3 + k] = max(v[3 + k], (uint32_t)__shfl_xor((int)v[3 + k], off)); }
                if ((threadIdx.x & 63) == 0)
                {
                    uint32_t *dst = bins + (size_t)first * SLOT_BIN_WORDS + (size_t)(ax * NUM_BINS + b) * BIN_WORDS;
                    for (int k = 0; k < 3; k++) { atomicMin(&dst[k], v[k]); atomicMax(&dst[3 + k], v[3 + k]); }
                    atomicAdd(&dst[6], (uint32_t)__popcll(who));
                }
            }
        }
        else if (bi >= 0)
        {
            uint32_t *dst = bins + (size_t)slot * SLOT_BIN_WORDS + (size_t)(ax * NUM_BINS + bi) * BIN_WORDS;
            for (int k = 0; k < 3; k++) { atomicMin(&dst[k], f2ord(lo[k])); atomicMax(&dst[3 + k], f2ord(hi[k])); }
            atomicAdd(&dst[6], 1u);
        }
    }
}

struct BoxD { float lo[3], hi[3]; };
__device__ __forceinline__ BoxD neutral_box() { BoxD b; for (int k = 0; k < 3; k++) { b.lo[k] = FLT_MAX; b.hi[k] = -FLT_MAX; } return b; }
__device__ __forceinline__ void grow(BoxD &b, const BoxD &o) { for (int k = 0; k < 3; k++) { b.lo[k] = fminf(b.lo[k], o.lo[k]); b.hi[k] = fmaxf(b.hi[k], o.hi[k]); } }
__device__ __forceinline__ float node_cost(const float size[3], uint32_t num_tris)   // data_structures.rs:468-475
{
    const float half_area = size[0] * (size[1] + size[2]) + size[1] * size[2];
    return half_area * (float)num_tris;
}

// one thread per frontier node: the rest of choose_split (:410-466), statement for statement
__global__ void __launch_bounds__(kBlock) k_choose_split(uint32_t nslots, const Slot *slots, const LupinBvhNode *nodes, const uint32_t *cb, const uint32_t *bins, SplitDev *splits)
{
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= nslots) return;
    const Slot sl = slots[s];
    const LupinBvhNode nd = nodes[sl.node];
    const float size[3] = {nd.aabb_max[0] - nd.aabb_min[0], nd.aabb_max[1] - nd.aabb_min[1], nd.aabb_max[2] - nd.aabb_min[2]};
    SplitDev res;
    memset(&res, 0, sizeof(res));
    float best = node_cost(size, sl.count);
    for (int axis = 0; axis < 3; axis++)
    {
        float cmin = ord2f(cb[s * 6u + axis]), cmax = ord2f(cb[s * 6u + 3u + axis]);
        if (cmin == cmax) continue;
        const float EPS = 0.001f;
        cmin -= EPS; cmax += EPS;
        BoxD bin_bounds[NUM_BINS];
        uint32_t bin_count[NUM_BINS];
        for (int b = 0; b < NUM_BINS; b++)
        {
            const uint32_t *src = bins + (size_t)s * SLOT_BIN_WORDS + (size_t)(axis * NUM_BINS + b) * BIN_WORDS;
            bin_count[b] = src[6];
            bin_bounds[b] = neutral_box();
            if (bin_count[b]) for (int k = 0; k < 3; k++) { bin_bounds[b].lo[k] = ord2f(src[k]); bin_bounds[b].hi[k] = ord2f(src[3 + k]); }
        }
        BoxD left_boxes[NUM_BINS - 1], right_boxes[NUM_BINS - 1];
        uint32_t left_count[NUM_BINS - 1], right_count[NUM_BINS - 1];
        BoxD lb = neutral_box(), rb = neutral_box();
        uint32_t lsum = 0, rsum = 0;
        for (int i = 0; i < NUM_BINS - 1; i++)
        {
            lsum += bin_count[i];
            left_count[i] = lsum;
            grow(lb, bin_bounds[i]);
            left_boxes[i] = lb;
            rsum += bin_count[NUM_BINS - 1 - i];
            right_count[NUM_BINS - 2 - i] = rsum;
            grow(rb, bin_bounds[NUM_BINS - 1 - i]);
            right_boxes[NUM_BINS - 2 - i] = rb;
        }
        const float step = (cmax - cmin) / (float)NUM_BINS;
        for (int i = 0; i < NUM_BINS - 1; i++)
        {
            const float ls[3] = {left_boxes[i].hi[0] - left_boxes[i].lo[0], left_boxes[i].hi[1] - left_boxes[i].lo[1], left_boxes[i].hi[2] - left_boxes[i].lo[2]};
            const float rs[3] = {right_boxes[i].hi[0] - right_boxes[i].lo[0], right_boxes[i].hi[1] - right_boxes[i].lo[1], right_boxes[i].hi[2] - right_boxes[i].lo[2]};
            const float plane_cost = node_cost(ls, left_count[i]) + node_cost(rs, right_count[i]);
            if (plane_cost < best)
            {
                res.performed = 1u;
                best = plane_cost;
                res.axis = (uint32_t)axis;
                res.pos = cmin + step * (float)(i + 1);
                for (int k = 0; k < 3; k++) { res.lbox[k] = left_boxes[i].lo[k]; res.lbox[3 + k] = left_boxes[i].hi[k]; res.rbox[k] = right_boxes[i].lo[k]; res.rbox[3 + k] = right_boxes[i].hi[k]; }
            }
        }
    }
    splits[s] = res;
}

// 1 = goes left (bvh_split's partition predicate, :260-268); position n carries a 0 so that the scan has an end value
__global__ void __launch_bounds__(kBlock) k_flags(uint32_t n, const uint32_t *perm, const uint32_t *pslot, const float4 *cen, const SplitDev *splits, uint32_t *flags)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p > n) return;
    uint32_t f = 0u;
    if (p < n)
    {
        const uint32_t slot = pslot[p];
        if (slot != INACTIVE && splits[slot].performed)
        {
            const float4 c = cen[perm[p]];
            const uint32_t ax = splits[slot].axis;
            const float v = ax == 0u ? c.x : (ax == 1u ? c.y : c.z);
            f = v <= splits[slot].pos ? 1u : 0u;
        }
    }
    flags[p] = f;
}

__global__ void __launch_bounds__(kBlock) k_valid(uint32_t nslots, const Slot *slots, const uint32_t *scan, SplitDev *splits, uint32_t *valid)
{
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s > nslots) return;
    uint32_t v = 0u;
    if (s < nslots)
    {
        const Slot sl = slots[s];
        const uint32_t left = scan[sl.begin + sl.count] - scan[sl.begin];
        splits[s].left_count = left;
        v = (splits[s].performed && left != 0u && left != sl.count) ? 1u : 0u;   // both sides non-empty (:270-273)
        splits[s].valid = v;
    }
    valid[s] = v;
}

// children of every split that happened, numbered by the prefix sum over the level's slots (deterministic); the next level's frontier
__global__ void __launch_bounds__(kBlock) k_make_children(uint32_t nslots, const Slot *slots, SplitDev *splits, const uint32_t *valid_scan, uint32_t node_base,
                                                          LupinBvhNode *nodes, Slot *next_slots, uint32_t max_depth_to_push)
{
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= nslots) return;
    const SplitDev sp = splits[s];
    if (!sp.valid) return;
    const Slot sl = slots[s];
    const uint32_t rank = valid_scan[s];
    splits[s].child_rank = rank;
    const uint32_t left = node_base + 2u * rank, right = left + 1u;
    LupinBvhNode l, r;
    for (int k = 0; k < 3; k++) { l.aabb_min[k] = sp.lbox[k]; l.aabb_max[k] = sp.lbox[3 + k]; r.aabb_min[k] = sp.rbox[k]; r.aabb_max[k] = sp.rbox[3 + k]; }
    l.tri_begin_or_first_child = sl.begin; l.tri_count = sp.left_count;
    r.tri_begin_or_first_child = sl.begin + sp.left_count; r.tri_count = sl.count - sp.left_count;
    nodes[left] = l; nodes[right] = r;
    nodes[sl.node].tri_begin_or_first_child = left;
    nodes[sl.node].tri_count = 0u;
    if (sl.depth < max_depth_to_push)   // `if depth < BVH_MAX_DEPTH - 1 { push both }` (:318-322)
    {
        next_slots[2u * rank] = Slot{left, l.tri_begin_or_first_child, l.tri_count, sl.depth + 1u};
        next_slots[2u * rank + 1u] = Slot{right, r.tri_begin_or_first_child, r.tri_count, sl.depth + 1u};
    }
}

// stable partition of every split node's range; positions of nodes that stay leaves keep their triangle and leave the frontier
__global__ void __launch_bounds__(kBlock) k_scatter(uint32_t n, const uint32_t *perm_in, const uint32_t *pslot_in, const uint32_t *flags, const uint32_t *scan,
                                                    const Slot *slots, const SplitDev *splits, uint32_t max_depth_to_push, uint32_t *perm_out, uint32_t *pslot_out)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const uint32_t slot = pslot_in[p];
    const uint32_t t = perm_in[p];
    if (slot == INACTIVE || !splits[slot].valid) { perm_out[p] = t; pslot_out[p] = INACTIVE; return; }
    const Slot sl = slots[slot];
    const SplitDev sp = splits[slot];
    const uint32_t rank = scan[p] - scan[sl.begin];                       // left-going positions before p in this node
    const bool left = flags[p] != 0u;
    const uint32_t q = left ? sl.begin + rank : sl.begin + sp.left_count + ((p - sl.begin) - rank);
    perm_out[q] = t;
    pslot_out[q] = sl.depth < max_depth_to_push ? 2u * sp.child_rank + (left ? 0u : 1u) : INACTIVE;
}

__global__ void __launch_bounds__(kBlock) k_reorder_indices(uint32_t n, const uint32_t *perm, const uint32_t *idx_in, uint32_t *idx_out)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const uint32_t t = perm[p];
    idx_out[3 * p + 0] = idx_in[3 * t + 0]; idx_out[3 * p + 1] = idx_in[3 * t + 1]; idx_out[3 * p + 2] = idx_in[3 * t + 2];
}

}  // namespace

extern "C" {

int64_t lupin_hip_build_bvh_sah_device(LupinContext *ctx, const float *verts_pos4, uint32_t num_verts, uint32_t *indices, uint32_t num_indices,
                                       LupinBvhNode *out_nodes, uint64_t out_capacity)
{
    if (!ctx || !verts_pos4 || !indices || !out_nodes) return lupin_internal_fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    if (num_indices % 3 != 0 || num_indices == 0 || num_verts == 0) return lupin_internal_fail(LUPIN_ERR_INVALID_ARGUMENT, "need at least one triangle");
    const uint32_t n = num_indices / 3;
    for (uint32_t i = 0; i < num_indices; i++)
        if (indices[i] >= num_verts) return lupin_internal_fail(LUPIN_ERR_INVALID_ARGUMENT, "vertex index out of range");
    // min / max run on order-preserving integers here and through fminf / fmaxf on the CPU: they agree on finite values only
    for (size_t i = 0; i < (size_t)num_verts * 4; i++)
        if ((i & 3) != 3 && !std::isfinite(verts_pos4[i])) return lupin_internal_fail(LUPIN_ERR_INVALID_ARGUMENT, "non-finite vertex position (use lupin_build_bvh)");
    if (!lupin_internal_ctx_alive(ctx)) return lupin_internal_fail(LUPIN_ERR_INVALID_ARGUMENT, "the context has been destroyed");
    if (hipSetDevice(lupin_internal_ctx_device(ctx)) != hipSuccess) return lupin_internal_fail(LUPIN_ERR_HIP, "hipSetDevice");
    hipStream_t st = lupin_internal_ctx_stream(ctx);
    const uint64_t max_nodes = 2ull * n - 1ull;

    float4 *d_verts = nullptr, *d_lo = nullptr, *d_hi = nullptr, *d_cen = nullptr;
    uint32_t *d_idx = nullptr, *d_idx_out = nullptr, *d_perm[2] = {nullptr, nullptr}, *d_pslot[2] = {nullptr, nullptr}, *d_flags = nullptr, *d_scan = nullptr;
    uint32_t *d_root = nullptr, *d_cb = nullptr, *d_bins = nullptr, *d_valid = nullptr, *d_valid_scan = nullptr;
    LupinBvhNode *d_nodes = nullptr;
    Slot *d_slots[2] = {nullptr, nullptr};
    SplitDev *d_splits = nullptr;
    void *d_temp = nullptr;
    auto cleanup = [&]() {
        void *ptrs[] = {d_verts, d_lo, d_hi, d_cen, d_idx, d_idx_out, d_perm[0], d_perm[1], d_pslot[0], d_pslot[1], d_flags, d_scan, d_root, d_cb, d_bins, d_valid,
                        d_valid_scan, d_nodes, d_slots[0], d_slots[1], d_splits, d_temp};
        for (void *p : ptrs) if (p) hipFree(p);
    };
    const size_t slot_cap = (size_t)n + 2;   // a level never has more frontier nodes than triangles
    SAH_TRY(hipMalloc((void **)&d_verts, (size_t)num_verts * 16));
    SAH_TRY(hipMalloc((void **)&d_lo, (size_t)n * 16));
    SAH_TRY(hipMalloc((void **)&d_hi, (size_t)n * 16));
    SAH_TRY(hipMalloc((void **)&d_cen, (size_t)n * 16));
    SAH_TRY(hipMalloc((void **)&d_idx, (size_t)num_indices * 4));
    SAH_TRY(hipMalloc((void **)&d_idx_out, (size_t)num_indices * 4));
    for (int k = 0; k < 2; k++)
    {
        SAH_TRY(hipMalloc((void **)&d_perm[k], (size_t)n * 4));
        SAH_TRY(hipMalloc((void **)&d_pslot[k], (size_t)n * 4));
        SAH_TRY(hipMalloc((void **)&d_slots[k], slot_cap * sizeof(Slot)));
    }
    SAH_TRY(hipMalloc((void **)&d_flags, ((size_t)n + 1) * 4));
    SAH_TRY(hipMalloc((void **)&d_scan, ((size_t)n + 1) * 4));
    SAH_TRY(hipMalloc((void **)&d_root, 6 * 4));
    SAH_TRY(hipMalloc((void **)&d_cb, slot_cap * 6 * 4));
    SAH_TRY(hipMalloc((void **)&d_bins, slot_cap * SLOT_BIN_WORDS * 4));
    SAH_TRY(hipMalloc((void **)&d_valid, (slot_cap + 1) * 4));
    SAH_TRY(hipMalloc((void **)&d_valid_scan, (slot_cap + 1) * 4));
    SAH_TRY(hipMalloc((void **)&d_nodes, (size_t)max_nodes * sizeof(LupinBvhNode)));
    SAH_TRY(hipMalloc((void **)&d_splits, slot_cap * sizeof(SplitDev)));
    size_t temp_bytes = 0, temp2 = 0;
    SAH_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, temp_bytes, d_flags, d_scan, (int)(n + 1), st));
    SAH_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, temp2, d_valid, d_valid_scan, (int)(slot_cap + 1), st));
    temp_bytes = std::max(std::max(temp_bytes, temp2), (size_t)16);
    SAH_TRY(hipMalloc(&d_temp, temp_bytes));

    SAH_TRY(hipMemcpyAsync(d_verts, verts_pos4, (size_t)num_verts * 16, hipMemcpyHostToDevice, st));
    SAH_TRY(hipMemcpyAsync(d_idx, indices, (size_t)num_indices * 4, hipMemcpyHostToDevice, st));
    const uint32_t zero_box[6] = {0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u};   // f2ord(0.0f) on both sides
    SAH_TRY(hipMemcpyAsync(d_root, zero_box, sizeof(zero_box), hipMemcpyHostToDevice, st));

    const uint32_t tri_blocks = (n + kBlock - 1) / kBlock, tri_blocks1 = (n + 1 + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_prepare, dim3(tri_blocks), dim3(kBlock), 0, st, d_verts, d_idx, n, d_lo, d_hi, d_cen, d_perm[0], d_pslot[0], d_root);
    hipLaunchKernelGGL(k_root, dim3(1), dim3(1), 0, st, d_root, n, d_nodes, d_slots[0]);

    uint32_t nslots = 1, num_nodes = 1;
    int cur = 0;
    const uint32_t max_depth_to_push = LUPIN_BVH_MAX_DEPTH - 1;
    while (nslots > 0)
    {
        const uint32_t slot_blocks = (nslots + kBlock - 1) / kBlock, slot_blocks1 = (nslots + 1 + kBlock - 1) / kBlock;
        const uint32_t acc_blocks = (uint32_t)(((size_t)nslots * SLOT_BIN_WORDS + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_clear_acc, dim3(acc_blocks), dim3(kBlock), 0, st, nslots, d_cb, d_bins);
        hipLaunchKernelGGL(k_centroid_bounds, dim3(tri_blocks), dim3(kBlock), 0, st, n, d_perm[cur], d_pslot[cur], d_cen, d_cb);
        hipLaunchKernelGGL(k_bins, dim3(tri_blocks), dim3(kBlock), 0, st, n, d_perm[cur], d_pslot[cur], d_cen, d_lo, d_hi, d_cb, d_bins);
        hipLaunchKernelGGL(k_choose_split, dim3(slot_blocks), dim3(kBlock), 0, st, nslots, d_slots[cur], d_nodes, d_cb, d_bins, d_splits);
        hipLaunchKernelGGL(k_flags, dim3(tri_blocks1), dim3(kBlock), 0, st, n, d_perm[cur], d_pslot[cur], d_cen, d_splits, d_flags);
        SAH_TRY(hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, d_flags, d_scan, (int)(n + 1), st));
        hipLaunchKernelGGL(k_valid, dim3(slot_blocks1), dim3(kBlock), 0, st, nslots, d_slots[cur], d_scan, d_splits, d_valid);
        SAH_TRY(hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, d_valid, d_valid_scan, (int)(nslots + 1), st));
        uint32_t nvalid = 0;
        SAH_TRY(hipMemcpyAsync(&nvalid, d_valid_scan + nslots, 4, hipMemcpyDeviceToHost, st));
        SAH_TRY(hipStreamSynchronize(st));
        if (nvalid == 0) break;
        if ((uint64_t)num_nodes + 2ull * nvalid > max_nodes) { cleanup(); return lupin_internal_fail(LUPIN_ERR_HIP, "SAH builder produced more nodes than a binary tree can hold"); }
        // the depth of a level is uniform: every slot of this level has the same depth, so either all valid splits push their children or none does
        hipLaunchKernelGGL(k_make_children, dim3(slot_blocks), dim3(kBlock), 0, st, nslots, d_slots[cur], d_splits, d_valid_scan, num_nodes, d_nodes, d_slots[1 - cur], max_depth_to_push);
        hipLaunchKernelGGL(k_scatter, dim3(tri_blocks), dim3(kBlock), 0, st, n, d_perm[cur], d_pslot[cur], d_flags, d_scan, d_slots[cur], d_splits, max_depth_to_push, d_perm[1 - cur], d_pslot[1 - cur]);
        num_nodes += 2u * nvalid;
        // depth of this level = 1 + number of levels done; children join the frontier while depth < BVH_MAX_DEPTH - 1
        Slot first_slot;
        SAH_TRY(hipMemcpyAsync(&first_slot, d_slots[cur], sizeof(Slot), hipMemcpyDeviceToHost, st));
        SAH_TRY(hipStreamSynchronize(st));
        nslots = first_slot.depth < max_depth_to_push ? 2u * nvalid : 0u;
        cur = 1 - cur;
    }
    if (num_nodes > out_capacity) { cleanup(); return lupin_internal_fail(LUPIN_ERR_INVALID_ARGUMENT, "node buffer too small (2 * triangles - 1 always suffices)"); }
    hipLaunchKernelGGL(k_reorder_indices, dim3(tri_blocks), dim3(kBlock), 0, st, n, d_perm[cur], d_idx, d_idx_out);
    SAH_TRY(hipGetLastError());
    SAH_TRY(hipMemcpyAsync(out_nodes, d_nodes, (size_t)num_nodes * sizeof(LupinBvhNode), hipMemcpyDeviceToHost, st));
    SAH_TRY(hipMemcpyAsync(indices, d_idx_out, (size_t)num_indices * 4, hipMemcpyDeviceToHost, st));
    SAH_TRY(hipStreamSynchronize(st));
    cleanup();
    return (int64_t)num_nodes;
}

}  // extern "C"
