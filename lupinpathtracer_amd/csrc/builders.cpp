// builders.cpp -- host-side preprocessing that produces the hot path's inputs.
//
// Restates the CPU builders of the reference (lupin/src/data_structures.rs:20-641): binned-SAH
// BLAS, agglomerative TLAS, alias tables and light weights.  Same split rule, same node order,
// same triangle reordering, so the traversal visits the same nodes in the same order as a scene
// built by `lp::build_accel_structures_and_upload`.  Plain f32 arithmetic, single thread per
// call (callers parallelise across meshes).

#include <cstdint>
#include <cstring>
#include <cmath>
#include <cfloat>
#include <vector>
#include <algorithm>

#include "../../include/lupin_hip.h"
#include "../../include/lupin_tiles.h"

namespace {

struct V3 { float x, y, z; float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); } };
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 vmin(V3 a, V3 b) { return {fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }

struct Box { V3 lo, hi; };
// Aabb::neutral() (base.rs:237-244)
inline Box neutral_box() { return {{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}}; }
// grow_aabb_to_include_aabb (base.rs:1091-1099)
inline void grow(Box &b, const Box &o) { b.lo = vmin(b.lo, o.lo); b.hi = vmax(b.hi, o.hi); }

// node_cost (data_structures.rs:468-475): half surface area x triangle count
inline float node_cost(V3 size, uint32_t num_tris)
{
    float half_area = size.x * (size.y + size.z) + size.y * size.z;
    return half_area * (float)num_tris;
}

struct Split
{
    bool performed = false;
    int axis = 0;
    float pos = 0.0f;
    float cost = 0.0f;
    Box left, right;
};

struct BlasBuild
{
    const float *verts;  // stride 4
    uint32_t *indices;
    std::vector<V3> centroids;
    std::vector<Box> tri_bounds;
    std::vector<LupinBvhNode> nodes;

    V3 vert(uint32_t i) const { return {verts[(size_t)i * 4], verts[(size_t)i * 4 + 1], verts[(size_t)i * 4 + 2]}; }

    // choose_split (data_structures.rs:366-466): 5 bins per axis over padded centroid bounds
    Split choose_split(uint32_t node) const
    {
        const int NUM_BINS = 5;
        const LupinBvhNode &nd = nodes[node];
        V3 size = {nd.aabb_max[0] - nd.aabb_min[0], nd.aabb_max[1] - nd.aabb_min[1], nd.aabb_max[2] - nd.aabb_min[2]};
        size_t tri_begin = nd.tri_begin_or_first_child, tri_count = nd.tri_count;

        Split res;
        res.cost = node_cost(size, (uint32_t)tri_count);
        for (int axis = 0; axis < 3; axis++)
        {
            float cmin = FLT_MAX, cmax = -FLT_MAX;
            for (size_t t = tri_begin; t < tri_begin + tri_count; t++)
            {
                float c = centroids[t][axis];
                cmin = fminf(cmin, c);
                cmax = fmaxf(cmax, c);
            }
            if (cmin == cmax) continue;
            const float EPS = 0.001f;
            cmin -= EPS;
            cmax += EPS;

            Box bin_bounds[NUM_BINS];
            uint32_t bin_count[NUM_BINS];
            for (int b = 0; b < NUM_BINS; b++) { bin_bounds[b] = neutral_box(); bin_count[b] = 0; }
            float scale = (float)NUM_BINS / (cmax - cmin);
            for (size_t t = tri_begin; t < tri_begin + tri_count; t++)
            {
                float f = floorf((centroids[t][axis] - cmin) * scale);
                int bi = (f >= (float)(NUM_BINS - 1)) ? NUM_BINS - 1 : ((f > 0.0f) ? (int)f : 0);  // `as usize` saturates, then clamp
                grow(bin_bounds[bi], tri_bounds[t]);
                bin_count[bi] += 1;
            }

            Box left_boxes[NUM_BINS - 1], right_boxes[NUM_BINS - 1];
            uint32_t left_count[NUM_BINS - 1], right_count[NUM_BINS - 1];
            Box lb = neutral_box(), rb = neutral_box();
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

            float step = (cmax - cmin) / (float)NUM_BINS;
            for (int i = 0; i < NUM_BINS - 1; i++)
            {
                V3 ls = left_boxes[i].hi - left_boxes[i].lo;
                V3 rs = right_boxes[i].hi - right_boxes[i].lo;
                float plane_cost = node_cost(ls, left_count[i]) + node_cost(rs, right_count[i]);
                if (plane_cost < res.cost)
                {
                    res.performed = true;
                    res.cost = plane_cost;
                    res.axis = axis;
                    res.pos = cmin + step * (float)(i + 1);
                    res.left = left_boxes[i];
                    res.right = right_boxes[i];
                }
            }
        }
        return res;
    }

    void swap_tris(uint32_t a, uint32_t b)  // data_structures.rs:502-527
    {
        for (int k = 0; k < 3; k++) std::swap(indices[(size_t)a * 3 + k], indices[(size_t)b * 3 + k]);
        std::swap(centroids[a], centroids[b]);
        std::swap(tri_bounds[a], tri_bounds[b]);
    }

    // bvh_split (data_structures.rs:237-325): explicit stack, depth cap BVH_MAX_DEPTH - 1
    void split_all()
    {
        struct Item { uint32_t node, depth; };
        Item stack[LUPIN_BVH_MAX_DEPTH + 1];
        int sp = 1;
        stack[0] = {0u, 1u};
        while (sp > 0)
        {
            sp--;
            uint32_t node = stack[sp].node, depth = stack[sp].depth;
            Split split = choose_split(node);
            if (!split.performed) continue;

            uint32_t begin = nodes[node].tri_begin_or_first_child;
            uint32_t count = nodes[node].tri_count;
            uint32_t end = begin + count;
            uint32_t left_idx = begin;
            for (uint32_t t = begin; t < end; t++)
            {
                if (centroids[t][split.axis] <= split.pos)
                {
                    if (t != left_idx) swap_tris(left_idx, t);
                    left_idx++;
                }
            }
            uint32_t left_count = left_idx - begin;
            uint32_t right_count = count - left_count;
            if (left_count == 0 || right_count == 0) continue;

            uint32_t left = (uint32_t)nodes.size();
            uint32_t right = left + 1;
            LupinBvhNode l, r;
            memset(&l, 0, sizeof(l)); memset(&r, 0, sizeof(r));
            l.tri_begin_or_first_child = begin; l.tri_count = left_count;
            r.tri_begin_or_first_child = left_idx; r.tri_count = right_count;
            l.aabb_min[0] = split.left.lo.x; l.aabb_min[1] = split.left.lo.y; l.aabb_min[2] = split.left.lo.z;
            l.aabb_max[0] = split.left.hi.x; l.aabb_max[1] = split.left.hi.y; l.aabb_max[2] = split.left.hi.z;
            r.aabb_min[0] = split.right.lo.x; r.aabb_min[1] = split.right.lo.y; r.aabb_min[2] = split.right.lo.z;
            r.aabb_max[0] = split.right.hi.x; r.aabb_max[1] = split.right.hi.y; r.aabb_max[2] = split.right.hi.z;
            nodes.push_back(l);
            nodes.push_back(r);
            nodes[node].tri_begin_or_first_child = left;
            nodes[node].tri_count = 0;

            if (depth < (uint32_t)(LUPIN_BVH_MAX_DEPTH - 1))
            {
                stack[sp + 0] = {left, depth + 1};
                stack[sp + 1] = {right, depth + 1};
                sp += 2;
            }
        }
    }
};

// Mat4::inverse (base.rs:542-578) on column-major m[col][row]
void mat4_inverse(const float m[4][4], float b[4][4])
{
    float s0 = m[0][0] * m[1][1] - m[1][0] * m[0][1];
    float s1 = m[0][0] * m[1][2] - m[1][0] * m[0][2];
    float s2 = m[0][0] * m[1][3] - m[1][0] * m[0][3];
    float s3 = m[0][1] * m[1][2] - m[1][1] * m[0][2];
    float s4 = m[0][1] * m[1][3] - m[1][1] * m[0][3];
    float s5 = m[0][2] * m[1][3] - m[1][2] * m[0][3];
    float c5 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float c4 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float c3 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float c2 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float c1 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float c0 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float invdet = 1.0f / (s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0);
    b[0][0] = ( m[1][1] * c5 - m[1][2] * c4 + m[1][3] * c3) * invdet;
    b[0][1] = (-m[0][1] * c5 + m[0][2] * c4 - m[0][3] * c3) * invdet;
    b[0][2] = ( m[3][1] * s5 - m[3][2] * s4 + m[3][3] * s3) * invdet;
    b[0][3] = (-m[2][1] * s5 + m[2][2] * s4 - m[2][3] * s3) * invdet;
    b[1][0] = (-m[1][0] * c5 + m[1][2] * c2 - m[1][3] * c1) * invdet;
    b[1][1] = ( m[0][0] * c5 - m[0][2] * c2 + m[0][3] * c1) * invdet;
    b[1][2] = (-m[3][0] * s5 + m[3][2] * s2 - m[3][3] * s1) * invdet;
    b[1][3] = ( m[2][0] * s5 - m[2][2] * s2 + m[2][3] * s1) * invdet;
    b[2][0] = ( m[1][0] * c4 - m[1][1] * c2 + m[1][3] * c0) * invdet;
    b[2][1] = (-m[0][0] * c4 + m[0][1] * c2 - m[0][3] * c0) * invdet;
    b[2][2] = ( m[3][0] * s4 - m[3][1] * s2 + m[3][3] * s0) * invdet;
    b[2][3] = (-m[2][0] * s4 + m[2][1] * s2 - m[2][3] * s0) * invdet;
    b[3][0] = (-m[1][0] * c3 + m[1][1] * c1 - m[1][2] * c0) * invdet;
    b[3][1] = ( m[0][0] * c3 - m[0][1] * c1 + m[0][2] * c0) * invdet;
    b[3][2] = (-m[3][0] * s3 + m[3][1] * s1 - m[3][2] * s0) * invdet;
    b[3][3] = ( m[2][0] * s3 - m[2][1] * s1 + m[2][2] * s0) * invdet;
}

}  // namespace

extern "C" {

void lupin_mat3x4_inverse(const LupinMat3x4 *in, LupinMat3x4 *out)
{
    // Mat3x4::to_mat4 + Mat4::inverse + truncation (base.rs:695-722)
    float m[4][4], b[4][4];
    for (int c = 0; c < 4; c++) { m[c][0] = in->m[c][0]; m[c][1] = in->m[c][1]; m[c][2] = in->m[c][2]; m[c][3] = (c == 3) ? 1.0f : 0.0f; }
    mat4_inverse(m, b);
    for (int c = 0; c < 4; c++) for (int r = 0; r < 3; r++) out->m[c][r] = b[c][r];
}

int64_t lupin_build_bvh(const float *verts_pos4, uint32_t num_verts, uint32_t *indices,
                        uint32_t num_indices, LupinBvhNode *out_nodes, uint64_t out_capacity)
{
    if (!verts_pos4 || !indices) return LUPIN_ERR_INVALID_ARGUMENT;
    uint32_t num_tris = num_indices / 3;
    for (uint32_t i = 0; i < num_tris * 3; i++) if (indices[i] >= num_verts) return LUPIN_ERR_INVALID_ARGUMENT;

    std::vector<uint32_t> work;
    BlasBuild bb;
    bb.verts = verts_pos4;
    if (out_nodes) { bb.indices = indices; }
    else { work.assign(indices, indices + (size_t)num_tris * 3); bb.indices = work.data(); }  // count query: leave caller's order alone
    bb.centroids.reserve(num_tris);
    bb.tri_bounds.reserve(num_tris);
    for (uint32_t t = 0; t < num_tris; t++)
    {
        V3 t0 = bb.vert(bb.indices[(size_t)t * 3 + 0]), t1 = bb.vert(bb.indices[(size_t)t * 3 + 1]), t2 = bb.vert(bb.indices[(size_t)t * 3 + 2]);
        // compute_tri_centroid (base.rs:1155-1159): (t0 + t1 + t2) / 3.0
        V3 s = (t0 + t1) + t2;
        bb.centroids.push_back({s.x / 3.0f, s.y / 3.0f, s.z / 3.0f});
        // compute_tri_bounds (base.rs:1136-1153)
        Box b;
        b.lo = {fminf(t0.x, fminf(t1.x, t2.x)), fminf(t0.y, fminf(t1.y, t2.y)), fminf(t0.z, fminf(t1.z, t2.z))};
        b.hi = {fmaxf(t0.x, fmaxf(t1.x, t2.x)), fmaxf(t0.y, fmaxf(t1.y, t2.y)), fmaxf(t0.z, fmaxf(t1.z, t2.z))};
        bb.tri_bounds.push_back(b);
    }
    // compute_aabb (data_structures.rs:529-540) starts from Aabb::default() == all zeros, so the
    // root box always contains the origin.  Kept: it feeds the root's SAH cost.
    Box root = {{0, 0, 0}, {0, 0, 0}};
    for (uint32_t t = 0; t < num_tris; t++) grow(root, bb.tri_bounds[t]);
    LupinBvhNode rn;
    memset(&rn, 0, sizeof(rn));
    rn.aabb_min[0] = root.lo.x; rn.aabb_min[1] = root.lo.y; rn.aabb_min[2] = root.lo.z;
    rn.aabb_max[0] = root.hi.x; rn.aabb_max[1] = root.hi.y; rn.aabb_max[2] = root.hi.z;
    rn.tri_begin_or_first_child = 0;
    rn.tri_count = num_tris;
    bb.nodes.push_back(rn);
    bb.split_all();

    if (out_nodes)
    {
        if (out_capacity < bb.nodes.size()) return LUPIN_ERR_INVALID_ARGUMENT;
        memcpy(out_nodes, bb.nodes.data(), bb.nodes.size() * sizeof(LupinBvhNode));
    }
    return (int64_t)bb.nodes.size();
}

int64_t lupin_build_tlas(const LupinInstance *instances, uint32_t num_instances,
                         const float *model_aabbs, uint32_t num_meshes, LupinTlasNode *out_nodes)
{
    if (num_instances == 0 || num_meshes == 0) return 0;  // data_structures.rs:547
    if (!instances || !model_aabbs || !out_nodes) return LUPIN_ERR_INVALID_ARGUMENT;

    std::vector<uint32_t> node_indices;
    std::vector<LupinTlasNode> tlas;
    node_indices.reserve(num_instances);
    tlas.reserve((size_t)num_instances * 2);

    for (uint32_t i = 0; i < num_instances; i++)
    {
        const LupinInstance &inst = instances[i];
        if (inst.mesh_idx >= num_meshes) return LUPIN_ERR_INVALID_ARGUMENT;
        const float *ab = model_aabbs + (size_t)inst.mesh_idx * 6;
        // transform = transpose_inverse_transform.transpose().inverse()  (local -> world)
        LupinMat3x4 w2l, l2w;
        for (int c = 0; c < 4; c++) for (int r = 0; r < 3; r++) w2l.m[c][r] = inst.transpose_inverse_transform.m[r][c];
        lupin_mat3x4_inverse(&w2l, &l2w);
        // transform_aabb (base.rs:1113-1134): 8 corners, z fastest
        Box res = neutral_box();
        for (int k = 0; k < 8; k++)
        {
            float x = (k & 4) ? ab[3] : ab[0], y = (k & 2) ? ab[4] : ab[1], z = (k & 1) ? ab[5] : ab[2];
            V3 p = {l2w.m[0][0] * x + l2w.m[1][0] * y + l2w.m[2][0] * z + l2w.m[3][0] * 1.0f,
                    l2w.m[0][1] * x + l2w.m[1][1] * y + l2w.m[2][1] * z + l2w.m[3][1] * 1.0f,
                    l2w.m[0][2] * x + l2w.m[1][2] * y + l2w.m[2][2] * z + l2w.m[3][2] * 1.0f};
            res.lo = vmin(res.lo, p);
            res.hi = vmax(res.hi, p);
        }
        LupinTlasNode nd;
        memset(&nd, 0, sizeof(nd));
        nd.aabb_min[0] = res.lo.x; nd.aabb_min[1] = res.lo.y; nd.aabb_min[2] = res.lo.z;
        nd.aabb_max[0] = res.hi.x; nd.aabb_max[1] = res.hi.y; nd.aabb_max[2] = res.hi.z;
        nd.instance_idx = i;
        tlas.push_back(nd);
        node_indices.push_back((uint32_t)tlas.size() - 1);
    }

    // tlas_find_best_match (data_structures.rs:670-692)
    auto best_match = [&](uint32_t node_a) -> uint32_t {
        const LupinTlasNode &a = tlas[node_indices[node_a]];
        float smallest = FLT_MAX;
        uint32_t best_b = 0xFFFFFFFFu;
        for (uint32_t i = 0; i < (uint32_t)node_indices.size(); i++)
        {
            if (node_a == i) continue;
            const LupinTlasNode &b = tlas[node_indices[i]];
            float ex = fmaxf(a.aabb_max[0], b.aabb_max[0]) - fminf(a.aabb_min[0], b.aabb_min[0]);
            float ey = fmaxf(a.aabb_max[1], b.aabb_max[1]) - fminf(a.aabb_min[1], b.aabb_min[1]);
            float ez = fmaxf(a.aabb_max[2], b.aabb_max[2]) - fminf(a.aabb_min[2], b.aabb_min[2]);
            float area = ex * ey + ey * ez + ez * ex;
            if (area < smallest) { smallest = area; best_b = i; }
        }
        return best_b;
    };

    // agglomerative clustering (data_structures.rs:572-610)
    uint32_t a = 0;
    uint32_t b = best_match(a);
    while (node_indices.size() > 1)
    {
        uint32_t c = best_match(b);
        if (a == c)
        {
            uint32_t ia = node_indices[a], ib = node_indices[b];
            LupinTlasNode na = tlas[ia], nb = tlas[ib];
            LupinTlasNode nn;
            memset(&nn, 0, sizeof(nn));
            nn.left = ia;
            nn.right = ib;
            // Deviation from the reference, on purpose: `left == 0` is the leaf marker, but before
            // the final reversal leaf 0 sits at index 0, so a merge whose left operand is
            // instance 0 would later be traversed as a leaf (data_structures.rs:624-628 never
            // remaps it).  Swapping the operands keeps the tree intact; only the tie order of
            // that one node differs.
            if (nn.left == 0) { nn.left = ib; nn.right = ia; }
            for (int k = 0; k < 3; k++)
            {
                nn.aabb_min[k] = fminf(na.aabb_min[k], nb.aabb_min[k]);
                nn.aabb_max[k] = fmaxf(na.aabb_max[k], nb.aabb_max[k]);
            }
            tlas.push_back(nn);
            node_indices[a] = (uint32_t)tlas.size() - 1;
            node_indices[b] = node_indices.back();
            node_indices.pop_back();
            if (a >= (uint32_t)node_indices.size()) a = (uint32_t)node_indices.size() - 1;
            b = best_match(a);
        }
        else
        {
            a = b;
            b = c;
        }
    }

    // push a copy of the root, then reverse so the root is node 0 (data_structures.rs:612-635)
    tlas.push_back(tlas[node_indices[a]]);
    uint32_t len = (uint32_t)tlas.size();
    std::reverse(tlas.begin(), tlas.end());
    for (uint32_t i = 0; i < len; i++)
    {
        // internal nodes: remap children; `right == 0` now legitimately means old index 0
        bool internal = (tlas[i].left != 0) || (tlas[i].right != 0);
        if (internal)
        {
            tlas[i].left = len - 1 - tlas[i].left;
            tlas[i].right = len - 1 - tlas[i].right;
        }
    }
    memcpy(out_nodes, tlas.data(), (size_t)len * sizeof(LupinTlasNode));
    return (int64_t)len;
}

int64_t lupin_build_alias_table(const float *weights, uint64_t n, LupinAliasBin *out_bins)
{
    // data_structures.rs:116-193 (PBRT-4 alias method)
    if (n == 0) return 0;
    if (!weights || !out_bins) return LUPIN_ERR_INVALID_ARGUMENT;
    double sum = 0.0;
    for (uint64_t i = 0; i < n; i++) sum += (double)weights[i];
    if (sum == 0.0) return 0;
    double normalize_factor = 1.0 / sum;
    for (uint64_t i = 0; i < n; i++)
    {
        out_bins[i].prob = (float)((double)weights[i] * normalize_factor);
        out_bins[i].alias_threshold = 0.0f;
        out_bins[i].alias = 0;
    }
    struct Outcome { float prob_estimate; uint32_t idx; };
    std::vector<Outcome> under, over;
    for (uint64_t i = 0; i < n; i++)
    {
        float pe = out_bins[i].prob * (float)n;
        if (pe < 1.0f) under.push_back({pe, (uint32_t)i}); else over.push_back({pe, (uint32_t)i});
    }
    while (!under.empty() && !over.empty())
    {
        Outcome u = under.back(); under.pop_back();
        Outcome o = over.back(); over.pop_back();
        out_bins[u.idx].alias_threshold = u.prob_estimate;
        out_bins[u.idx].alias = o.idx;
        float excess = u.prob_estimate + o.prob_estimate - 1.0f;
        if (excess < 1.0f) under.push_back({excess, o.idx}); else over.push_back({excess, o.idx});
    }
    while (!over.empty()) { Outcome o = over.back(); over.pop_back(); out_bins[o.idx].alias_threshold = 1.0f; out_bins[o.idx].alias = 0; }
    while (!under.empty()) { Outcome u = under.back(); under.pop_back(); out_bins[u.idx].alias_threshold = 1.0f; out_bins[u.idx].alias = 0; }
    return (int64_t)n;
}

float lupin_mesh_light_weights(const float *verts_pos4, const uint32_t *indices, uint32_t num_indices, float *out_weights)
{
    // data_structures.rs:40-51 with tri_area (:106-112)
    float total = 0.0f;
    for (uint32_t i = 0; i + 2 < num_indices; i += 3)
    {
        const float *p0 = verts_pos4 + (size_t)indices[i] * 4, *p1 = verts_pos4 + (size_t)indices[i + 1] * 4, *p2 = verts_pos4 + (size_t)indices[i + 2] * 4;
        V3 a = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
        V3 b = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
        // Vec3::cross (base.rs:172-179)
        V3 c = {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
        float area = sqrtf(c.x * c.x + c.y * c.y + c.z * c.z) / 2.0f;
        if (out_weights) out_weights[i / 3] = area;
        total += area;
    }
    return total;
}

void lupin_env_light_weights(const float *texels, uint32_t width, uint32_t height, const float scale[3], float *out_weights)
{
    // data_structures.rs:65-93
    const float PI_F = 3.14159265358979323846f;
    bool uniform = scale[0] <= 0.0f && scale[1] <= 0.0f && scale[2] <= 0.0f;
    for (uint32_t y = 0; y < height; y++)
    {
        float angle = ((float)y + 0.5f) * PI_F / (float)height;
        float s = sinf(angle);
        for (uint32_t x = 0; x < width; x++)
        {
            size_t i = (size_t)y * width + x;
            const float *p = texels + i * 4;
            float e = fmaxf(fmaxf(p[0] * scale[0], p[1] * scale[1]), p[2] * scale[2]);
            out_weights[i] = uniform ? 1.0f : e * s;
        }
    }
}

uint32_t lupin_hip_get_num_tiles(uint32_t tile_size, uint32_t width, uint32_t height)
{
    // renderer.rs:675-681
    if (tile_size == 0) return 0;
    uint32_t ntx = ((width > 1 ? width : 1) - 1) / (tile_size * LUPIN_WORKGROUP_SIZE) + 1;
    uint32_t nty = ((height > 1 ? height : 1) - 1) / (tile_size * LUPIN_WORKGROUP_SIZE) + 1;
    return ntx * nty;
}

uint64_t lupin_hip_packed_tile_pixels(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t rank, uint32_t world)
{
    if (tile_size == 0 || world == 0) return 0;
    uint32_t tpx = tile_size * LUPIN_WORKGROUP_SIZE;
    uint32_t ntx = ((width > 1 ? width : 1) - 1) / tpx + 1;
    uint32_t nty = ((height > 1 ? height : 1) - 1) / tpx + 1;
    uint64_t total = 0;
    const uint32_t owned = lupin_owned_tile_count(ntx * nty, rank, world);
    for (uint32_t j = 0; j < owned; j++)
    {
        const uint32_t t = lupin_owned_tile(j, rank, world, ntx);
        uint32_t ox = (t % ntx) * tpx, oy = (t / ntx) * tpx;
        uint32_t w = std::min(tpx, width - ox), h = std::min(tpx, height - oy);
        total += (uint64_t)w * h;
    }
    return total;
}

}  // extern "C"
