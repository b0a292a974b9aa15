// Device BLAS builder ("next" row 8f-1): a linear BVH over Morton-sorted triangles, emitted in the reference's own
// node format (BvhNode 32 B, children adjacent, leaf = tri_begin + tri_count, renderer.rs:228-238) together with the
// reordered index buffer, so that it drops into lupin_hip_scene_create -- and into the oracle -- exactly where the
// reference's CPU SAH builder output (data_structures.rs:196-475 / lupin_build_bvh) goes.
//
// Shape: triangles sorted by the 30-bit Morton code of their AABB centre (stable radix sort, rocPRIM via hipCUB);
// a COMPLETE binary tree over the sorted sequence with 2^D leaves of 1-2 triangles each: node k of level d covers
// [floor(k n / 2^d), floor((k+1) n / 2^d)) and sits at index 2^d - 1 + k, children at 2 i + 1, 2 i + 2.  Depth
// D = ceil(log2(n / 2)) <= 22 for 2^23 triangles, inside the reference's BVH_MAX_DEPTH = 25 stack.  Boxes are exact
// unions (min / max only), refitted level by level.  Everything is deterministic: same input, same bytes.
//
// Image results do not depend on the builder except where two triangles tie for the closest hit, and through the
// reference's alias-table / reordered-index quirk for emissive meshes with unequal triangle areas.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cstdint>
#include <vector>

#include "lupin_internal.hpp"

namespace {

#define LBVH_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { cleanup(); return lupin_internal_fail(LUPIN_ERR_HIP, hipGetErrorString(e__)); } } while (0)

constexpr int kBlock = 256;

// order-preserving float <-> uint mapping for atomicMin / atomicMax
__device__ __forceinline__ uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

// bounds[0..2] = min of triangle-box centres, bounds[3..5] = max (order-preserving uints)
__global__ void __launch_bounds__(kBlock) k_centre_bounds(const float4 *verts, const uint32_t *indices, uint32_t num_tris, float4 *centres, uint32_t *bounds)
{
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    float c[3] = {0.0f, 0.0f, 0.0f};
    const bool live = t < num_tris;
    if (live)
    {
        const float4 a = verts[indices[3 * t + 0]], b = verts[indices[3 * t + 1]], d = verts[indices[3 * t + 2]];
        c[0] = (fminf(a.x, fminf(b.x, d.x)) + fmaxf(a.x, fmaxf(b.x, d.x))) * 0.5f;
        c[1] = (fminf(a.y, fminf(b.y, d.y)) + fmaxf(a.y, fmaxf(b.y, d.y))) * 0.5f;
        c[2] = (fminf(a.z, fminf(b.z, d.z)) + fmaxf(a.z, fmaxf(b.z, d.z))) * 0.5f;
        centres[t] = make_float4(c[0], c[1], c[2], 0.0f);
    }
    for (int ax = 0; ax < 3; ax++)
    {
        uint32_t lo = live ? f2ord(c[ax]) : 0xFFFFFFFFu, hi = live ? f2ord(c[ax]) : 0u;
        for (int off = 32; off > 0; off >>= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, off)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, off)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&bounds[ax], lo); atomicMax(&bounds[3 + ax], hi); }
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v)   // 10 bits -> every third bit
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void __launch_bounds__(kBlock) k_morton(const float4 *centres, uint32_t num_tris, const uint32_t *bounds, uint32_t *keys, uint32_t *vals)
{
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= num_tris) return;
    const float4 c = centres[t];
    const float cc[3] = {c.x, c.y, c.z};
    uint32_t q[3];
    for (int ax = 0; ax < 3; ax++)
    {
        const float lo = ord2f(bounds[ax]), hi = ord2f(bounds[3 + ax]);
        const float ext = hi - lo;
        float f = ext > 0.0f ? (cc[ax] - lo) / ext * 1024.0f : 0.0f;   // IEEE division (part of the library's contract)
        f = fminf(fmaxf(f, 0.0f), 1023.0f);
        q[ax] = (uint32_t)f;   // truncation; non-finite centres land in cell 0 through the clamp above (NaN -> fmaxf -> 0)
    }
    keys[t] = (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
    vals[t] = t;
}

// leaves: level D of the complete tree.  Node (D, k) covers sorted triangles [k n / 2^D, (k+1) n / 2^D).
__global__ void __launch_bounds__(kBlock) k_leaves(const float4 *verts, const uint32_t *indices, const uint32_t *order, uint32_t num_tris,
                                                   uint32_t depth, LupinBvhNode *nodes, uint32_t *indices_out)
{
    const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t leaves = 1ull << depth;
    if (k >= leaves) return;
    const uint32_t lo = (uint32_t)((k * num_tris) >> depth), hi = (uint32_t)(((k + 1) * num_tris) >> depth);
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = lo; i < hi; i++)
    {
        const uint32_t t = order[i];
        for (int v = 0; v < 3; v++)
        {
            const uint32_t vi = indices[3 * t + v];
            indices_out[3 * i + v] = vi;
            const float4 p = verts[vi];
            mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
            mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
        }
    }
    LupinBvhNode nd;
    nd.aabb_min[0] = mn[0]; nd.aabb_min[1] = mn[1]; nd.aabb_min[2] = mn[2];
    nd.aabb_max[0] = mx[0]; nd.aabb_max[1] = mx[1]; nd.aabb_max[2] = mx[2];
    nd.tri_begin_or_first_child = lo;
    nd.tri_count = hi - lo;   // >= 1 because num_tris >= 2^depth
    nodes[(leaves - 1) + k] = nd;
}

// one internal level: node (d, k) at 2^d - 1 + k = union of its children at 2 i + 1, 2 i + 2
__global__ void __launch_bounds__(kBlock) k_refit_level(uint32_t level, LupinBvhNode *nodes)
{
    const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= (1ull << level)) return;
    const uint64_t i = ((1ull << level) - 1) + k;
    const LupinBvhNode a = nodes[2 * i + 1], b = nodes[2 * i + 2];
    LupinBvhNode nd;
    for (int ax = 0; ax < 3; ax++) { nd.aabb_min[ax] = fminf(a.aabb_min[ax], b.aabb_min[ax]); nd.aabb_max[ax] = fmaxf(a.aabb_max[ax], b.aabb_max[ax]); }
    nd.tri_begin_or_first_child = (uint32_t)(2 * i + 1);
    nd.tri_count = 0;
    nodes[i] = nd;
}

}  // namespace

extern "C" {

uint32_t lupin_hip_lbvh_depth(uint32_t num_tris)
{
    uint32_t d = 0;
    while (d < 31 && (2ull << d) < (uint64_t)num_tris) d++;   // smallest D with 2^D >= n / 2  => leaves hold 1-2 triangles
    return d;
}

uint64_t lupin_hip_lbvh_node_count(uint32_t num_tris) { return num_tris ? (2ull << lupin_hip_lbvh_depth(num_tris)) - 1 : 0; }

int64_t lupin_hip_build_bvh_device(LupinContext *ctx, const float *verts_pos4, uint32_t num_verts, uint32_t *indices, uint32_t num_indices,
                                   LupinBvhNode *out_nodes, uint64_t out_capacity)
{
    if (!ctx || !verts_pos4 || !indices || !out_nodes) return lupin_internal_fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    if (num_indices % 3 != 0 || num_indices == 0 || num_verts == 0) return lupin_internal_fail(LUPIN_ERR_INVALID_ARGUMENT, "need at least one triangle");
    const uint32_t n = num_indices / 3;
    for (uint32_t i = 0; i < num_indices; i++)
        if (indices[i] >= num_verts) return lupin_internal_fail(LUPIN_ERR_INVALID_ARGUMENT, "vertex index out of range");
    const uint32_t depth = lupin_hip_lbvh_depth(n);
    const uint64_t num_nodes = (2ull << depth) - 1;
    if (num_nodes > out_capacity) return lupin_internal_fail(LUPIN_ERR_INVALID_ARGUMENT, "node buffer too small (see lupin_hip_lbvh_node_count)");
    if (!lupin_internal_ctx_alive(ctx)) return lupin_internal_fail(LUPIN_ERR_INVALID_ARGUMENT, "the context has been destroyed");
    if (hipSetDevice(lupin_internal_ctx_device(ctx)) != hipSuccess) return lupin_internal_fail(LUPIN_ERR_HIP, "hipSetDevice");
    hipStream_t st = lupin_internal_ctx_stream(ctx);

    float4 *d_verts = nullptr, *d_centres = nullptr;
    uint32_t *d_idx = nullptr, *d_idx_out = nullptr, *d_bounds = nullptr, *d_keys = nullptr, *d_vals = nullptr, *d_keys2 = nullptr, *d_vals2 = nullptr;
    LupinBvhNode *d_nodes = nullptr;
    void *d_temp = nullptr;
    auto cleanup = [&]() {
        void *ptrs[] = {d_verts, d_centres, d_idx, d_idx_out, d_bounds, d_keys, d_vals, d_keys2, d_vals2, d_nodes, d_temp};
        for (void *p : ptrs) if (p) hipFree(p);
    };
    LBVH_TRY(hipMalloc((void **)&d_verts, (size_t)num_verts * 16));
    LBVH_TRY(hipMalloc((void **)&d_centres, (size_t)n * 16));
    LBVH_TRY(hipMalloc((void **)&d_idx, (size_t)num_indices * 4));
    LBVH_TRY(hipMalloc((void **)&d_idx_out, (size_t)num_indices * 4));
    LBVH_TRY(hipMalloc((void **)&d_bounds, 6 * 4));
    LBVH_TRY(hipMalloc((void **)&d_keys, (size_t)n * 4));
    LBVH_TRY(hipMalloc((void **)&d_vals, (size_t)n * 4));
    LBVH_TRY(hipMalloc((void **)&d_keys2, (size_t)n * 4));
    LBVH_TRY(hipMalloc((void **)&d_vals2, (size_t)n * 4));
    LBVH_TRY(hipMalloc((void **)&d_nodes, (size_t)num_nodes * sizeof(LupinBvhNode)));
    LBVH_TRY(hipMemcpyAsync(d_verts, verts_pos4, (size_t)num_verts * 16, hipMemcpyHostToDevice, st));
    LBVH_TRY(hipMemcpyAsync(d_idx, indices, (size_t)num_indices * 4, hipMemcpyHostToDevice, st));
    const uint32_t init_bounds[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    LBVH_TRY(hipMemcpyAsync(d_bounds, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, st));

    const uint32_t tri_blocks = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_centre_bounds, dim3(tri_blocks), dim3(kBlock), 0, st, d_verts, d_idx, n, d_centres, d_bounds);
    hipLaunchKernelGGL(k_morton, dim3(tri_blocks), dim3(kBlock), 0, st, d_centres, n, d_bounds, d_keys, d_vals);
    size_t temp_bytes = 0;
    LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_keys, d_keys2, d_vals, d_vals2, (int)n, 0, 30, st));
    LBVH_TRY(hipMalloc(&d_temp, std::max<size_t>(temp_bytes, 16)));
    LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_keys, d_keys2, d_vals, d_vals2, (int)n, 0, 30, st));
    const uint64_t leaves = 1ull << depth;
    hipLaunchKernelGGL(k_leaves, dim3((uint32_t)((leaves + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, d_verts, d_idx, d_vals2, n, depth, d_nodes, d_idx_out);
    for (int level = (int)depth - 1; level >= 0; level--)
        hipLaunchKernelGGL(k_refit_level, dim3((uint32_t)(((1ull << level) + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, (uint32_t)level, d_nodes);
    LBVH_TRY(hipGetLastError());
    LBVH_TRY(hipMemcpyAsync(out_nodes, d_nodes, (size_t)num_nodes * sizeof(LupinBvhNode), hipMemcpyDeviceToHost, st));
    LBVH_TRY(hipMemcpyAsync(indices, d_idx_out, (size_t)num_indices * 4, hipMemcpyDeviceToHost, st));
    LBVH_TRY(hipStreamSynchronize(st));
    cleanup();
    return (int64_t)num_nodes;
}

}  // extern "C"
