// lupin_hip.hip -- host side of liblupin_hip.so: the C ABI of include/lupin_hip.h (contexts and their lanes, scene upload,
// textures, the pathtrace_scene family, measurement hooks, probes).  The stage kernels live in lupin_stages.hpp, the
// traversal / material / light device functions in lupin_device.hpp, the CPU builders in builders.cpp, the device BLAS
// builder in lbvh.hip.
//
// THERE IS NO CPU FALLBACK: without a HIP device every entry point that needs one fails with LUPIN_ERR_NO_DEVICE.

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <set>
#include <mutex>
#include <algorithm>
#include <atomic>

#include "lupin_stages.hpp"
#include "lupin_internal.hpp"

// ------------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------------

static inline float host_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static thread_local std::string g_last_error;
static int fail(int code, const std::string &msg) { g_last_error = msg; return code; }
#define HIP_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return fail(LUPIN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); } while (0)

#define LP_MAX_LANES 8
#define LP_WIDE_WORDS 10   // [0] wide queries, [1] re-traced, [2..9] LUPIN_VERIFY_WIDE: checked, flagged, mismatches among unflagged, raw mismatches, 4 reasons
#define LP_COUNTER_ROWS 7   // per iteration and shard: 1 queue counter + 2 hand-out cursors + 2 re-trace counters + 2 re-trace cursors
#define LP_WORK_WORDS 28   // 3 tracing modes (closest hit | shadow rays | light-pdf marching) x {nodes, triangles, instances, four-wide nodes}, then 10 round statistics
// "Lanes" (stream + path buffers + counters) let consecutive pathtrace_scene calls overlap: the wavefront of
// frame k+1 starts while the thin tail of frame k is still draining.  Frames only meet at k_resolve (frame k+1 blends
// with frame k's output), which waits on the previous call's completion event.
struct Lane
{
    hipStream_t stream = nullptr;
    PathBuffers pb{};
    char *hot = nullptr;            // the eight per-bounce fields of the path state: capacity x 128 bytes, planes or records (set_path_layout)
    char *shadow = nullptr;         // the eight MIS / Direct fields, likewise
    char *skey = nullptr;           // k_sort_queue's keys, 4 bytes per slot
    uint64_t capacity = 0;          // slots the path buffers hold
    uint32_t counts_capacity = 0;
    unsigned long long *stat_counters = nullptr;   // per shard: [2s] path bounces, [2s+1] paths
    unsigned long long *work_counters = nullptr;   // [4 * mode + {nodes, triangles, instances, wide nodes}] of the COUNT kernels (stats mode 2)
    unsigned long long *wide_counters = nullptr;   // [0] queries the wide tracer took, [1] queries re-traced by the binary tracer
    hipEvent_t done = nullptr;      // recorded after the last kernel of the lane's latest call
    bool used = false;
    FrameParams *d_fp = nullptr;    // this lane's per-call parameters (k_set_params writes, the stage kernels read)
    uint64_t pb_generation = 0;     // bumped when the path buffers are reallocated
    // the lane's wavefront (memset + k_begin + all iterations) as a replayable graph
    struct GraphKey
    {
        uint64_t scene_id = 0, pb_generation = 0;
        uint32_t n = 0, blocks = 0, type = 0, iterations = 0, stack_words = 0, lds = 0, pblocks = 0;
        uint32_t wide_blocks = 0, wide_stack_words = 0;
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
    int num_lanes = LP_MAX_LANES;   // LUPIN_LANES=1..8; see the lane choice in flush_pending
    bool lanes_from_env = false;
    uint64_t call_index = 0;
    int last_lane = -1;
    hipEvent_t marker = nullptr;
    bool timing = false;
    int path_records = -1;          // LUPIN_PATH_RECORDS=0/1: path state as planes / 128-byte records (default: records where the queues are sorted)
    uint32_t short_stack = 26;      // LUPIN_SHORT_STACK=n: first pass of the binary tracer on n stack entries per lane when the scene's depth bound
                                    // asks for more (26 KB per block: six blocks per CU instead of four); 0 = always the full stack, one pass
    int light_stage = -1;           // LUPIN_LIGHT_STAGE=0/1: sample_lights_pdf inline in k_shade / in its own stage (k_light_pdf, k_light_pdf_mis); default: stage for MIS only
    bool debug_sync = false;        // LUPIN_DEBUG_SYNC=1: synchronise and report after every stage launch (fault localisation)
    bool counting = false;          // lupin_hip_stats_reset(ctx, 2): the tracing kernels run their work-counting instantiation
    int accum_mode = 0;             // LUPIN_ACCUM_F16_RUNNING_AVERAGE | LUPIN_ACCUM_F32
    int runtime_version = 0;        // hipRuntimeGetVersion of the libamdhip64 this process bound
    int store_rounding = 0;        // LUPIN_STORE_ROUND_TOWARD_ZERO
    bool lds_geometry = true;       // LUPIN_LDS_GEOMETRY=0 keeps small scenes in global memory (A/B runs)
    int persistent_extend = 2;      // LUPIN_EXTEND: "simple" 0 | "persistent" 1 (always) | default 2: persistent for scenes traversed from global memory
    uint32_t num_cus = 256;
    bool specialize_simple = true;          // LUPIN_SIMPLE_SHADE=0: always the general k_shade
    bool use_graph = false;                 // LUPIN_GRAPH=1: replay the lane-private wavefront as a HIP graph (opt-in)
    bool persistent_shadow = true;          // LUPIN_SHADOW=simple: MIS / Direct shadow rays stay in k_shadow even on large scenes
    bool wide_traversal = false;           // LUPIN_TRAVERSAL=wide / lupin_hip_set_traversal: four-wide hierarchy + certificate + re-trace (same images, not faster: DESIGN 5)
    static constexpr uint32_t wide_stack_pairs = 20;   // (reference, distance) stack entries per lane of the wide tracer: 40 KB per block, four blocks per CU
    bool verify_wide = false;              // LUPIN_VERIFY_WIDE=1: every closest-hit query is also checked wide-vs-binary on the device (stats)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_extend, ev_shade, ev_total;
    std::vector<hipEvent_t> ev_pool;
    uint64_t extend_launches = 0;
    // Frames per wavefront: pathtrace_scene calls that differ only in camera / accum_counter and chain their textures
    // (each call's prev_frame is the previous call's render_target) are recorded here and run as ONE wavefront when the batch is
    // full or anything needs their result (flush_pending: sync, texture reads and writes, another kind of call, teardown).
    struct PendingFrame { FrameParams fp; LupinTexture *target; const LupinTexture *prev; };
    std::vector<PendingFrame> pending;
    const LupinScene *pending_scene = nullptr;
    uint32_t pending_type = 0;
    uint32_t batch_frames = 0;             // LUPIN_BATCH=1..16: calls per wavefront (1 = every call is its own wavefront); 0 = by dispatch size,
                                           // see frames_per_wavefront()
    bool in_flush = false;
    int last_lanes = 0;                    // frames in flight the latest pathtrace call could use (reported by lupin_hip_stats_get)
    bool last_wide = false;                // ... and whether it ran the four-wide tracer
    uint32_t last_batch = 0, last_short = 0;   // frames the latest wavefront carried; its first-pass stack entries (0 = one pass)
};

struct LupinPathtraceResources
{
    LupinContext *ctx;
    LupinBakedPathtraceParams params;
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
    int device = 0;                                 // the context's device ordinal (the scene may outlive the context)
    SceneDev dev{};
    std::vector<void *> allocations;
    uint32_t stack_entries = 1;
    uint32_t persistent_blocks[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};   // grid of k_extend_persistent per integrator (lazy); [1]: sharing the chip with other lanes' stages
    uint32_t wide_blocks[4] = {0, 0, 0, 0};         // grid of its four-wide instantiation
    uint32_t short_blocks[4] = {0, 0, 0, 0};        // grid of its short-stack instantiation
    bool has_wide = false;                          // the four-wide hierarchies were built (scenes traversed from global memory)
    uint64_t leaky_triangles = 0;                   // triangles outside some box above them (reference builder: bins vs partition)
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

// Points the lane's per-bounce fields into its `hot` allocation: planes (neighbouring slots coalesce) or 128-byte records
// (scattered slots cost two sectors per path, not one per field).  See PathField.
static void set_path_layout(Lane *ln, bool records)
{
    PathBuffers &pb = ln->pb;
    char *b = ln->hot;
    const size_t cap = (size_t)ln->capacity;
    pb.skey = {ln->skey, 4};   // always a plane: k_sort_queue reads nothing else of a path
    if (records)
    {
        const uint32_t st = LP_PATH_RECORD_BYTES;
        pb.ori_rng = {b + 0, st}; pb.dir_meta = {b + 16, st}; pb.hit = {b + 32, st}; pb.hit_tri = {b + 48, st};
        pb.weight = {b + 64, st}; pb.radiance = {b + 80, st}; pb.color = {b + 96, st};
        char *m = ln->shadow;
        pb.sh_org = {m + 0, st}; pb.sh_d0 = {m + 16, st}; pb.sh_d1 = {m + 32, st}; pb.next_tri = {m + 48, st};
        pb.sh_f0 = {m + 64, st}; pb.sh_f1 = {m + 80, st}; pb.next_hit = {m + 96, st}; pb.sh_hit1 = {m + 112, st};
    }
    else
    {
        pb.ori_rng = {b, 16}; pb.dir_meta = {b + 16 * cap, 16}; pb.hit = {b + 32 * cap, 16}; pb.weight = {b + 48 * cap, 16};
        pb.radiance = {b + 64 * cap, 16}; pb.color = {b + 80 * cap, 16}; pb.hit_tri = {b + 96 * cap, 4};
        char *m = ln->shadow;
        pb.sh_org = {m, 16}; pb.sh_d0 = {m + 16 * cap, 16}; pb.sh_d1 = {m + 32 * cap, 16}; pb.sh_f0 = {m + 48 * cap, 16};
        pb.sh_f1 = {m + 64 * cap, 16}; pb.next_hit = {m + 80 * cap, 16}; pb.sh_hit1 = {m + 96 * cap, 16}; pb.next_tri = {m + 112 * cap, 4};
    }
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
        void **ptrs[] = {(void **)&ctx->hot, (void **)&ctx->shadow, (void **)&ctx->skey, (void **)&pb.vol0, (void **)&pb.vol1, (void **)&pb.queue[0], (void **)&pb.queue[1],
                         (void **)&pb.retrace};
        size_t elem[] = {LP_PATH_RECORD_BYTES, LP_PATH_RECORD_BYTES, 4, 16, 16, 4, 4, 8};   // re-trace tokens: up to two jobs per slot (shadow rays)
        constexpr int NPTRS = 8;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        // the lane holds nothing until every allocation has succeeded: a failure part-way (hipMalloc returns through
        // HIP_TRY) must not leave a capacity behind that a later, smaller dispatch would trust
        ctx->capacity = 0;
        ctx->pb_generation++;
        for (int k = 0; k < NPTRS; k++)
            if (*ptrs[k]) { hipFree(*ptrs[k]); *ptrs[k] = nullptr; }
        for (int k = 0; k < NPTRS; k++)
            HIP_TRY(hipMalloc(ptrs[k], (size_t)slots * elem[k]));
        // touch every page now (a fill runs at several TB/s): otherwise the first wavefront that reaches a slot pays the
        // mapping of its pages inside its kernels -- 1 % of the first full wavefront of eight 4K frames
        for (int k = 0; k < NPTRS; k++)
            HIP_TRY(hipMemsetAsync(*ptrs[k], 0, (size_t)slots * elem[k], ctx->stream));
        ctx->capacity = slots;
    }
    if (iterations + 2 > ctx->counts_capacity)
    {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->counts_capacity = 0;
        if (ctx->pb.counts) hipFree(ctx->pb.counts);
        ctx->pb.counts = nullptr;
        ctx->pb.cursors = ctx->pb.retrace_counts = ctx->pb.retrace_cursors = nullptr;
        // queue counters, then the persistent tracer's hand-out cursors, the re-trace counters and the re-trace cursors (two
        // per iteration each): one allocation, one clear per call
        HIP_TRY(hipMalloc((void **)&ctx->pb.counts, LP_COUNTER_ROWS * (size_t)(iterations + 2) * LP_SHARDS * sizeof(uint32_t)));
        ctx->counts_capacity = iterations + 2;
        ctx->pb.cursors = ctx->pb.counts + (size_t)ctx->counts_capacity * LP_SHARDS;
        ctx->pb.retrace_counts = ctx->pb.cursors + 2 * (size_t)ctx->counts_capacity * LP_SHARDS;
        ctx->pb.retrace_cursors = ctx->pb.retrace_counts + 2 * (size_t)ctx->counts_capacity * LP_SHARDS;
        ctx->pb_generation++;
    }
    return LUPIN_OK;
}

// Four-wide collapse of a binary hierarchy given as child-pair nodes (`bin`, references as in WideNode.d): starting from
// a node's two children, the internal child with the largest box surface is replaced by its own two children until the
// node has four children or only leaves.  Every child box is a box of the reference's tree, every leaf keeps its
// reference, so the set of triangles under a reference is unchanged.
// A child is opened only if BOTH its children's boxes lie inside its own box (exact comparison): then passing a
// grandchild's slab test implies passing the dropped child's (the slab test is monotone under box inclusion: same
// roundings of monotone operations), so the wide traversal skips nothing the reference's would not skip.  The
// reference's builder does produce boxes that are not nested (its child boxes come from centroid BINS, its partition from
// a comparison with the split plane; a triangle on the wrong side of that rounding is outside its node's stored box --
// first seen as the one ray in 8 x 10^5 whose closest hit the binary order misses): such a child stays a child.
// WideNode.d.z carries, per child of a pair (bit 0 left, bit 1 right), "some triangle below is not inside this child's
// box" (blas_child_pairs); it becomes REF_LEAKY in the wide node's child reference: those children are never pruned by distance.  Nodes are appended to `out` in depth-first order
// (a 128-byte node is a cache line of its own: order among nodes does not matter to the caches).  Returns the root
// reference (a leaf root passes through).
static uint32_t collapse_to_wide4(const std::vector<WideNode> &bin, uint32_t root_ref, std::vector<Wide4> &out, uint32_t *out_depth = nullptr)
{
    if (out_depth) *out_depth = 0;
    if (root_ref & REF_LEAF) return root_ref;
    struct Child { float lo[3], hi[3]; uint32_t ref; bool leaky, closed; };
    auto children_of = [&](uint32_t b, Child &l, Child &r) {
        const WideNode &w = bin[b];
        l.lo[0] = w.a.x; l.lo[1] = w.a.z; l.lo[2] = w.b.x; l.hi[0] = w.b.z; l.hi[1] = w.c.x; l.hi[2] = w.c.z; l.ref = w.d.x;
        r.lo[0] = w.a.y; r.lo[1] = w.a.w; r.lo[2] = w.b.y; r.hi[0] = w.b.w; r.hi[1] = w.c.y; r.hi[2] = w.c.w; r.ref = w.d.y;
        l.leaky = (w.d.z & 1u) != 0; r.leaky = (w.d.z & 2u) != 0;
        l.closed = r.closed = false;
    };
    auto inside = [](const Child &in, const Child &out) {   // exact; a NaN bound is "not inside"
        for (int ax = 0; ax < 3; ax++)
            if (!(in.lo[ax] >= out.lo[ax] && in.hi[ax] <= out.hi[ax])) return false;
        return true;
    };
    auto area = [](const Child &c) {
        const double ex = (double)c.hi[0] - c.lo[0], ey = (double)c.hi[1] - c.lo[1], ez = (double)c.hi[2] - c.lo[2];
        const double a = ex * ey + ey * ez + ez * ex;
        return a == a ? a : 0.0;
    };
    struct Item { uint32_t bin_node, wide_node, depth; };
    std::vector<Item> todo;
    const uint32_t root = (uint32_t)out.size();
    out.emplace_back();
    todo.push_back({root_ref, root, 1u});
    const float nanf_ = std::nanf("");
    while (!todo.empty())
    {
        const Item it = todo.back();
        todo.pop_back();
        if (out_depth) *out_depth = std::max(*out_depth, it.depth);
        Child c[4];
        int n = 2;
        children_of(it.bin_node, c[0], c[1]);
        while (n < 4)
        {
            int pick = -1;
            double best = -1.0;
            for (int k = 0; k < n; k++)
                if (!(c[k].ref & REF_LEAF) && !c[k].closed) { const double a = area(c[k]); if (a > best) { best = a; pick = k; } }
            if (pick < 0) break;
            Child l, r;
            children_of(c[pick].ref, l, r);
            if (!inside(l, c[pick]) || !inside(r, c[pick])) { c[pick].closed = true; continue; }   // not nested: this box must be tested
            c[pick] = l;
            c[n++] = r;
        }
        Wide4 w;
        float *lo[3] = {&w.lox.x, &w.loy.x, &w.loz.x}, *hi[3] = {&w.hix.x, &w.hiy.x, &w.hiz.x};
        uint32_t refs[4];
        for (int k = 0; k < 4; k++)
        {
            for (int ax = 0; ax < 3; ax++) { lo[ax][k] = k < n ? c[k].lo[ax] : nanf_; hi[ax][k] = k < n ? c[k].hi[ax] : nanf_; }
            refs[k] = REF_NONE;
            if (k < n)
            {
                if (c[k].ref & REF_LEAF) refs[k] = c[k].ref;
                else
                {
                    refs[k] = (uint32_t)out.size();
                    out.emplace_back();
                    todo.push_back({c[k].ref, refs[k], it.depth + 1});
                }
            }
        }
        for (int k = 0; k < n; k++) if (c[k].leaky) refs[k] |= REF_LEAKY;
        w.ref = make_uint4(refs[0], refs[1], refs[2], refs[3]);
        w.pad = make_uint4(0u, 0u, 0u, 0u);
        out[it.wide_node] = w;
    }
    return root;
}

// One mesh's BLAS (the reference's BvhNode array, data_structures.rs:196-325) as child-pair nodes appended to `blas`:
// node index -> child reference (leaf: REF_LEAF | global index of its first triangle; internal: index into `blas`).
// `leaf_ends` receives the global index of every leaf's last triangle.  Returns the root reference; *why != nullptr on a
// malformed array (the caller has checked that the array is a tree).
// With vertex data (verts_pos4 / indices, mesh-local) the boxes are also checked against what they should bound -- the wide
// tracer's certificate needs to know where the reference's boxes do not:
//   WideNode.d.z bit 0 / 1  the left / right child's stored box does not contain every triangle below it
//   leaky_tris              (global indices) triangles that are not inside every box above them, leaf box included
static uint32_t blas_child_pairs(const LupinBvhNode *nodes, uint32_t num_nodes, uint32_t ntris, uint32_t tri_offset, std::vector<WideNode> &blas,
                                 std::vector<uint32_t> &leaf_ends, const char **why,
                                 const float *verts_pos4 = nullptr, const uint32_t *indices = nullptr, std::vector<uint32_t> *leaky_tris = nullptr)
{
    *why = nullptr;
    std::vector<uint32_t> ref(num_nodes);
    uint32_t wide_base = (uint32_t)blas.size(), wide_count = 0;
    for (uint32_t n = 0; n < num_nodes; n++)
    {
        const LupinBvhNode &nd = nodes[n];
        if (nd.tri_count > 0)
        {
            if ((uint64_t)nd.tri_begin_or_first_child + nd.tri_count > ntris) { *why = "BLAS leaf range out of bounds"; return 0; }
            ref[n] = REF_LEAF | (tri_offset + nd.tri_begin_or_first_child);
            leaf_ends.push_back(tri_offset + nd.tri_begin_or_first_child + nd.tri_count - 1);
        }
        else
        {
            if ((uint64_t)nd.tri_begin_or_first_child + 1 >= num_nodes) { *why = "BLAS child index out of bounds"; return 0; }
            ref[n] = wide_base + wide_count++;
        }
    }
    // what the boxes should bound: true bounds per node (bottom-up) and, top-down, the intersection of the boxes above a leaf
    std::vector<uint8_t> leaky_node(num_nodes, 0);
    if (verts_pos4 && indices)
    {
        struct B { float lo[3], hi[3]; };
        auto tri_bounds = [&](uint32_t t) {
            B b;
            for (int ax = 0; ax < 3; ax++) { b.lo[ax] = INFINITY; b.hi[ax] = -INFINITY; }
            for (int k = 0; k < 3; k++)
                for (int ax = 0; ax < 3; ax++)
                {
                    const float x = verts_pos4[(size_t)indices[(size_t)t * 3 + k] * 4 + ax];
                    b.lo[ax] = std::min(b.lo[ax], x); b.hi[ax] = std::max(b.hi[ax], x);
                }
            return b;
        };
        std::vector<B> truth(num_nodes);
        // iterative post-order over the tree rooted at node 0 (children may have any index)
        std::vector<std::pair<uint32_t, int>> st;
        st.push_back({0u, 0});
        while (!st.empty())
        {
            auto [n, phase] = st.back();
            const LupinBvhNode &nd = nodes[n];
            if (nd.tri_count > 0)
            {
                B b;
                for (int ax = 0; ax < 3; ax++) { b.lo[ax] = INFINITY; b.hi[ax] = -INFINITY; }
                for (uint32_t t = nd.tri_begin_or_first_child; t < nd.tri_begin_or_first_child + nd.tri_count; t++)
                {
                    const B tb = tri_bounds(t);
                    for (int ax = 0; ax < 3; ax++) { b.lo[ax] = std::min(b.lo[ax], tb.lo[ax]); b.hi[ax] = std::max(b.hi[ax], tb.hi[ax]); }
                }
                truth[n] = b;
                st.pop_back();
            }
            else if (phase == 0)
            {
                st.back().second = 1;
                st.push_back({nd.tri_begin_or_first_child, 0});
                st.push_back({nd.tri_begin_or_first_child + 1, 0});
                continue;
            }
            else
            {
                const B &l = truth[nd.tri_begin_or_first_child], &r = truth[nd.tri_begin_or_first_child + 1];
                for (int ax = 0; ax < 3; ax++) { truth[n].lo[ax] = std::min(l.lo[ax], r.lo[ax]); truth[n].hi[ax] = std::max(l.hi[ax], r.hi[ax]); }
                st.pop_back();
            }
            bool in = true;
            for (int ax = 0; ax < 3; ax++) in = in && truth[n].lo[ax] >= nd.aabb_min[ax] && truth[n].hi[ax] <= nd.aabb_max[ax];
            leaky_node[n] = in ? 0 : 1;
        }
        if (leaky_tris)
        {
            // top-down: clip = intersection of the stored boxes from the root down to the node
            std::vector<std::pair<uint32_t, B>> down;
            B all;
            for (int ax = 0; ax < 3; ax++) { all.lo[ax] = -INFINITY; all.hi[ax] = INFINITY; }
            down.push_back({0u, all});
            while (!down.empty())
            {
                auto [n, clip] = down.back();
                down.pop_back();
                const LupinBvhNode &nd = nodes[n];
                for (int ax = 0; ax < 3; ax++) { clip.lo[ax] = std::max(clip.lo[ax], nd.aabb_min[ax]); clip.hi[ax] = std::min(clip.hi[ax], nd.aabb_max[ax]); }
                if (nd.tri_count > 0)
                {
                    for (uint32_t t = nd.tri_begin_or_first_child; t < nd.tri_begin_or_first_child + nd.tri_count; t++)
                    {
                        const B tb = tri_bounds(t);
                        bool in = true;
                        for (int ax = 0; ax < 3; ax++) in = in && tb.lo[ax] >= clip.lo[ax] && tb.hi[ax] <= clip.hi[ax];
                        if (!in) leaky_tris->push_back(tri_offset + t);
                    }
                }
                else
                {
                    down.push_back({nd.tri_begin_or_first_child, clip});
                    down.push_back({nd.tri_begin_or_first_child + 1, clip});
                }
            }
        }
    }
    blas.resize(wide_base + wide_count);
    for (uint32_t n = 0; n < num_nodes; n++)
    {
        const LupinBvhNode &nd = nodes[n];
        if (nd.tri_count > 0) continue;
        const uint32_t lc = nd.tri_begin_or_first_child, rc = lc + 1;
        const LupinBvhNode &l = nodes[lc];
        const LupinBvhNode &r = nodes[rc];
        WideNode w;
        w.a = make_float4(l.aabb_min[0], r.aabb_min[0], l.aabb_min[1], r.aabb_min[1]);
        w.b = make_float4(l.aabb_min[2], r.aabb_min[2], l.aabb_max[0], r.aabb_max[0]);
        w.c = make_float4(l.aabb_max[1], r.aabb_max[1], l.aabb_max[2], r.aabb_max[2]);
        w.d = make_uint4(ref[lc], ref[rc], (leaky_node[lc] ? 1u : 0u) | (leaky_node[rc] ? 2u : 0u), 0u);
        blas[ref[n]] = w;
    }
    return ref[0];
}

// depth of a hierarchy in internal levels = worst-case number of parked far children
// depth of a BLAS in internal levels; UINT32_MAX for a node array that is not a tree (bounded walk, like the TLAS check)
static uint32_t blas_depth(const LupinBvhNode *nodes, uint32_t count)
{
    if (count == 0) return 0;
    uint32_t best = 0;
    uint64_t visited = 0;
    std::vector<std::pair<uint32_t, uint32_t>> st;
    st.push_back({0u, 0u});
    while (!st.empty())
    {
        auto [n, d] = st.back();
        st.pop_back();
        if (++visited > (uint64_t)count + 1) return 0xFFFFFFFFu;
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
template <int TYPE>
static uint32_t persistent_grid_t(LupinContext *ctx, const LupinScene *scene, size_t lds, bool shares_chip)
{
    uint32_t &cached = const_cast<LupinScene *>(scene)->persistent_blocks[shares_chip ? 1 : 0][TYPE];
    if (cached == 0)
    {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_extend_persistent<TYPE, false, 0, false>, LP_BLOCK, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        // With frames in flight the persistent tracer of one frame shares the chip with the shading of another: when five
        // or more of its blocks fit per CU, three leave that room and the pair finishes sooner (materials1 / environments1
        // +5 %); deeper scenes fit four at most and are latency-bound, they keep them all (bistro-class -9 % with two).
        if (shares_chip && per_cu > 4) per_cu = 3;
        cached = std::max(64u, ctx->num_cus * (uint32_t)per_cu / 64u * 64u);
    }
    return cached;
}

// grid of the four-wide instantiation (its own LDS footprint and register count)
template <int TYPE>
static uint32_t wide_grid_t(LupinContext *ctx, const LupinScene *scene, size_t lds)
{
    uint32_t &cached = const_cast<LupinScene *>(scene)->wide_blocks[TYPE];
    if (cached == 0)
    {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_extend_persistent<TYPE, false, 0, false, true, false>, LP_BLOCK, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        cached = std::max(64u, ctx->num_cus * (uint32_t)per_cu / 64u * 64u);
    }
    return cached;
}
// grid of the short-stack instantiation of the binary tracer
template <int TYPE>
static uint32_t short_grid_t(LupinContext *ctx, const LupinScene *scene, size_t lds)
{
    uint32_t &cached = const_cast<LupinScene *>(scene)->short_blocks[TYPE];
    if (cached == 0)
    {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_extend_persistent<TYPE, false, 0, false, false, false, true>, LP_BLOCK, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        cached = std::max(64u, ctx->num_cus * (uint32_t)per_cu / 64u * 64u);
    }
    return cached;
}
static uint32_t short_grid(LupinContext *ctx, const LupinScene *scene, uint32_t type, size_t lds)
{
    switch (type)
    {
    case LUPIN_PATHTRACE_STANDARD: return short_grid_t<LUPIN_PATHTRACE_STANDARD>(ctx, scene, lds);
    case LUPIN_PATHTRACE_MIS: return short_grid_t<LUPIN_PATHTRACE_MIS>(ctx, scene, lds);
    case LUPIN_PATHTRACE_NAIVE: return short_grid_t<LUPIN_PATHTRACE_NAIVE>(ctx, scene, lds);
    default: return short_grid_t<LUPIN_PATHTRACE_DIRECT>(ctx, scene, lds);
    }
}
static uint32_t wide_grid(LupinContext *ctx, const LupinScene *scene, uint32_t type, size_t lds)
{
    switch (type)
    {
    case LUPIN_PATHTRACE_STANDARD: return wide_grid_t<LUPIN_PATHTRACE_STANDARD>(ctx, scene, lds);
    case LUPIN_PATHTRACE_MIS: return wide_grid_t<LUPIN_PATHTRACE_MIS>(ctx, scene, lds);
    case LUPIN_PATHTRACE_NAIVE: return wide_grid_t<LUPIN_PATHTRACE_NAIVE>(ctx, scene, lds);
    default: return wide_grid_t<LUPIN_PATHTRACE_DIRECT>(ctx, scene, lds);
    }
}

// The phase-scheduled persistent tracer serves scenes traversed from global memory; scenes staged in LDS (a few dozen node
// visits per ray) are faster with one ray per lane (k_extend; Cornell box 6.4 vs 5.4 Gsamples/s in round 1).
// LUPIN_EXTEND=simple forces k_extend everywhere.
static bool use_persistent(const LupinContext *ctx, bool lds_geo) { return ctx->persistent_extend != 0 && !lds_geo; }
static uint32_t persistent_grid(LupinContext *ctx, const LupinScene *scene, uint32_t type, bool lds_geo, size_t lds, bool shares_chip)
{
    if (!use_persistent(ctx, lds_geo)) return 0;
    switch (type)
    {
    case LUPIN_PATHTRACE_STANDARD: return persistent_grid_t<LUPIN_PATHTRACE_STANDARD>(ctx, scene, lds, shares_chip);
    case LUPIN_PATHTRACE_MIS: return persistent_grid_t<LUPIN_PATHTRACE_MIS>(ctx, scene, lds, shares_chip);
    case LUPIN_PATHTRACE_NAIVE: return persistent_grid_t<LUPIN_PATHTRACE_NAIVE>(ctx, scene, lds, shares_chip);
    default: return persistent_grid_t<LUPIN_PATHTRACE_DIRECT>(ctx, scene, lds, shares_chip);
    }
}

// LUPIN_LIGHT_STAGE=1: sample_lights_pdf of the Standard / MIS integrators runs in its own stage (k_light_pdf / k_light_pdf_mis) instead of inline
// in k_shade.  Same results; off by default (faster kernel for kernel, slower with frames in flight).
static bool use_light_stage(const LupinContext *ctx, const LupinScene *scene, uint32_t type)
{
    if (scene->simple_matte && ctx->specialize_simple) return false;
    if (ctx->light_stage >= 0) return ctx->light_stage > 0;
    // default: MIS runs its two sample_lights_pdf evaluations per vertex in k_light_pdf_mis (k_shade<MIS, DEFER> is then
    // straight-line BSDF code at 138 VGPRs instead of 256 + scratch with the marches inline); the Standard integrator keeps
    // its single evaluation inline (faster with frames in flight)
    return type == LUPIN_PATHTRACE_MIS;
}

// launch shape of one call's stage kernels
struct Shape
{
    uint32_t blocks = 0;         // one thread per slot, LP_SHARDS-aligned
    uint32_t pblocks = 0;        // persistent tracer (binary hierarchy); 0 = the one-ray-per-lane k_extend
    size_t lds = 0;              // traversal stacks (+ staged geometry)
    uint32_t stack_words = 0;
    uint32_t wblocks = 0;        // four-wide tracer; 0 = off
    size_t wlds = 0;
    uint32_t wstack_words = 0;
    uint32_t sblocks = 0;        // short-stack first pass of the binary tracer; 0 = off
    size_t slds = 0;
    uint32_t sstack_words = 0;
};

// the tracing stage of MODE 0 (closest hits of the integrator loop) or 1 (recorded shadow rays) on the persistent tracer
template <int TYPE, int MODE>
static void launch_persistent_tracer(LupinContext *ctx, Lane *ln, const LupinScene *scene, const Shape &sh, uint32_t iter)
{
    hipStream_t st = ln->stream;
    const FrameParams *fp = ln->d_fp;
    unsigned long long *work = ln->work_counters, *wide = ln->wide_counters;
    {
        if (sh.wblocks)
        {
            // four-wide traversal with the exactness certificate, then the queries it did not certify in the reference's order
            if (ctx->counting)
            {
                hipLaunchKernelGGL((k_extend_persistent<TYPE, false, MODE, true, true, false>), dim3(sh.wblocks), dim3(LP_BLOCK), sh.wlds, st,
                                   scene->dev, fp, ln->pb, iter, ln->stat_counters, LP_REFILL_MIN, sh.wstack_words, LP_NODE_STEPS, work, wide);
                hipLaunchKernelGGL((k_extend_persistent<TYPE, false, MODE, true, false, true>), dim3(sh.pblocks), dim3(LP_BLOCK), sh.lds, st,
                                   scene->dev, fp, ln->pb, iter, ln->stat_counters, LP_REFILL_MIN, sh.stack_words, LP_NODE_STEPS, work, wide);
            }
            else
            {
                hipLaunchKernelGGL((k_extend_persistent<TYPE, false, MODE, false, true, false>), dim3(sh.wblocks), dim3(LP_BLOCK), sh.wlds, st,
                                   scene->dev, fp, ln->pb, iter, ln->stat_counters, LP_REFILL_MIN, sh.wstack_words, LP_NODE_STEPS, work, wide);
                hipLaunchKernelGGL((k_extend_persistent<TYPE, false, MODE, false, false, true>), dim3(sh.pblocks), dim3(LP_BLOCK), sh.lds, st,
                                   scene->dev, fp, ln->pb, iter, ln->stat_counters, LP_REFILL_MIN, sh.stack_words, LP_NODE_STEPS, work, wide);
            }
            return;
        }
    }
    if (sh.sblocks && !ctx->counting)
    {
        // the binary traversal on a short stack (one more block per CU), then the few queries that needed the full one
        hipLaunchKernelGGL((k_extend_persistent<TYPE, false, MODE, false, false, false, true>), dim3(sh.sblocks), dim3(LP_BLOCK), sh.slds, st,
                           scene->dev, fp, ln->pb, iter, ln->stat_counters, LP_REFILL_MIN, sh.sstack_words, LP_NODE_STEPS, work, wide);
        hipLaunchKernelGGL((k_extend_persistent<TYPE, false, MODE, false, false, true>), dim3(sh.pblocks), dim3(LP_BLOCK), sh.lds, st,
                           scene->dev, fp, ln->pb, iter, ln->stat_counters, LP_REFILL_MIN, sh.stack_words, LP_NODE_STEPS, work, wide);
        return;
    }
    if (ctx->counting)
        hipLaunchKernelGGL((k_extend_persistent<TYPE, false, MODE, true>), dim3(sh.pblocks), dim3(LP_BLOCK), sh.lds, st,
                           scene->dev, fp, ln->pb, iter, ln->stat_counters, LP_REFILL_MIN, sh.stack_words, LP_NODE_STEPS, work, wide);
    else
        hipLaunchKernelGGL((k_extend_persistent<TYPE, false, MODE, false>), dim3(sh.pblocks), dim3(LP_BLOCK), sh.lds, st,
                           scene->dev, fp, ln->pb, iter, ln->stat_counters, LP_REFILL_MIN, sh.stack_words, LP_NODE_STEPS, work, wide);
}

template <int TYPE, bool LDSGEO>
static void launch_iteration_t(LupinContext *ctx, Lane *ln, const LupinScene *scene, const Shape &sh, uint32_t iter)
{
    const uint32_t blocks = sh.blocks, pblocks = sh.pblocks, stack_words = sh.stack_words;
    const size_t lds = sh.lds;
    hipStream_t st = ln->stream;
    const FrameParams *fp = ln->d_fp;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    if (ctx->timing) { e0 = get_event(ctx); e1 = get_event(ctx); e2 = get_event(ctx); hipEventRecord(e0, st); }
    const bool persistent = pblocks != 0;
    unsigned long long *work = ln->work_counters;
    if constexpr (!LDSGEO)
    {
        if (ctx->verify_wide && scene->has_wide)   // checker only: every query of this iteration, binary vs wide, one ray per lane
            hipLaunchKernelGGL(k_verify_wide<0>, dim3(blocks), dim3(LP_BLOCK), std::max(lds, (size_t)ctx->wide_stack_pairs * 2u * LP_BLOCK * sizeof(uint32_t)), st,
                               scene->dev, fp, ln->pb, iter, ctx->wide_stack_pairs, ln->wide_counters + 2);
    }
    bool traced = false;
    if constexpr (!LDSGEO) { if (persistent) { launch_persistent_tracer<TYPE, 0>(ctx, ln, scene, sh, iter); traced = true; } }
    if (!traced)
    {
        const bool opaque = scene->all_opaque && ctx->specialize_simple;
        if (ctx->counting)
        {
            if (opaque) hipLaunchKernelGGL((k_extend<TYPE, LDSGEO, true, true>), dim3(blocks), dim3(LP_BLOCK), lds, st, scene->dev, fp, ln->pb, iter, ln->stat_counters, stack_words, work);
            else hipLaunchKernelGGL((k_extend<TYPE, LDSGEO, false, true>), dim3(blocks), dim3(LP_BLOCK), lds, st, scene->dev, fp, ln->pb, iter, ln->stat_counters, stack_words, work);
        }
        else if (opaque)
            hipLaunchKernelGGL((k_extend<TYPE, LDSGEO, true, false>), dim3(blocks), dim3(LP_BLOCK), lds, st, scene->dev, fp, ln->pb, iter, ln->stat_counters, stack_words, work);
        else
            hipLaunchKernelGGL((k_extend<TYPE, LDSGEO, false, false>), dim3(blocks), dim3(LP_BLOCK), lds, st, scene->dev, fp, ln->pb, iter, ln->stat_counters, stack_words, work);
    }
    if (ctx->timing) hipEventRecord(e1, st);
    if (ctx->debug_sync)
    {
        hipError_t de = hipStreamSynchronize(st);
        fprintf(stderr, "[lupin] iteration %u type %d: extend done: %s\n", iter, TYPE, hipGetErrorString(de)); fflush(stderr);
    }
    bool light_stage = false;
    if constexpr (TYPE == LUPIN_PATHTRACE_STANDARD || TYPE == LUPIN_PATHTRACE_MIS) light_stage = use_light_stage(ctx, scene, TYPE);
    // several material families: sort the queue in windows first, k_shade then finds its 256 paths (nearly) uniform
    SceneDev shade_dev = scene->dev;
    if (scene->dev.sort_shade && scene->dev.num_instances)
    {
        const uint32_t windows = ((blocks / LP_SHARDS) * LP_BLOCK + LP_SORT_WINDOW - 1) / LP_SORT_WINDOW;
        // the Standard integrator's persistent tracer leaves the key with the hit; otherwise the pass derives it
        if (TYPE == LUPIN_PATHTRACE_STANDARD && persistent)
            hipLaunchKernelGGL((k_sort_queue<true, true>), dim3(windows * LP_SHARDS), dim3(LP_BLOCK), 0, st, scene->dev, ln->pb, iter);
        else
            hipLaunchKernelGGL((k_sort_queue<TYPE == LUPIN_PATHTRACE_STANDARD, false>), dim3(windows * LP_SHARDS), dim3(LP_BLOCK), 0, st, scene->dev, ln->pb, iter);
        shade_dev.sort_shade = 0;
    }
    if (scene->simple_matte && ctx->specialize_simple)
        hipLaunchKernelGGL((k_shade<TYPE, LDSGEO, true>), dim3(blocks), dim3(LP_BLOCK), lds, st, shade_dev, fp, ln->pb, iter, ln->stat_counters, stack_words);
    else if (!light_stage)
        hipLaunchKernelGGL((k_shade<TYPE, LDSGEO, false>), dim3(blocks), dim3(LP_BLOCK), lds, st, shade_dev, fp, ln->pb, iter, ln->stat_counters, stack_words);
    if constexpr (TYPE == LUPIN_PATHTRACE_STANDARD)
    {
        if (light_stage)
        {
            // sample_lights_pdf has its own stage: k_shade tags the vertices that need it, k_light_pdf finishes them and appends
            // (this k_shade never traverses: without LDS-staged geometry it needs the block sort's 256 words only)
            const size_t shade_lds = LDSGEO ? lds : std::min(lds, (size_t)LP_BLOCK * sizeof(uint32_t));
            hipLaunchKernelGGL((k_shade<TYPE, LDSGEO, false, true>), dim3(blocks), dim3(LP_BLOCK), shade_lds, st, shade_dev, fp, ln->pb, iter, ln->stat_counters, stack_words);
            hipLaunchKernelGGL((k_light_pdf<TYPE, LDSGEO>), dim3(blocks), dim3(LP_BLOCK), lds, st, scene->dev, fp, ln->pb, iter, stack_words);
        }
    }
    if constexpr (TYPE == LUPIN_PATHTRACE_MIS)
    {
        if (light_stage)
        {
            // the two MIS weights per vertex need sample_lights_pdf: k_shade parks the candidates, this pass weighs them
            hipLaunchKernelGGL((k_shade<TYPE, LDSGEO, false, true>), dim3(blocks), dim3(LP_BLOCK), lds, st, shade_dev, fp, ln->pb, iter, ln->stat_counters, stack_words);
            hipLaunchKernelGGL((k_light_pdf_mis<LDSGEO>), dim3(blocks), dim3(LP_BLOCK), lds, st, scene->dev, fp, ln->pb, iter, stack_words);
        }
    }
    if constexpr (TYPE == LUPIN_PATHTRACE_MIS || TYPE == LUPIN_PATHTRACE_DIRECT)   // shadow rays + path finish (booked with "shade" in the timing)
    {
        bool pretraced = false;
        if constexpr (!LDSGEO)
        {
            if (persistent && ctx->persistent_shadow)
            {
                // large scenes: the shadow rays go through the phase-scheduled persistent tracer as well, then a light finish pass
                if (ctx->verify_wide && scene->has_wide)
                    hipLaunchKernelGGL(k_verify_wide<1>, dim3(blocks), dim3(LP_BLOCK), std::max(lds, (size_t)ctx->wide_stack_pairs * 2u * LP_BLOCK * sizeof(uint32_t)), st,
                                       scene->dev, fp, ln->pb, iter, ctx->wide_stack_pairs, ln->wide_counters + 2);
                launch_persistent_tracer<TYPE, 1>(ctx, ln, scene, sh, iter);
                hipLaunchKernelGGL((k_shadow<TYPE, false, true>), dim3(blocks), dim3(LP_BLOCK), lds, st, scene->dev, fp, ln->pb, iter, stack_words);
                pretraced = true;
            }
        }
        if (pretraced) {}   // (k_shadow<.., PRETRACED> has run)
        else
            hipLaunchKernelGGL((k_shadow<TYPE, LDSGEO, false>), dim3(blocks), dim3(LP_BLOCK), lds, st, scene->dev, fp, ln->pb, iter, stack_words);
    }
    if (ctx->debug_sync)
    {
        fprintf(stderr, "[lupin] iteration %u type %d: stages launched, syncing\n", iter, TYPE); fflush(stderr);
        hipError_t de = hipStreamSynchronize(st);
        fprintf(stderr, "[lupin] iteration %u: %s\n", iter, hipGetErrorString(de)); fflush(stderr);
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
static void launch_iteration(LupinContext *ctx, Lane *ln, const LupinScene *scene, bool lds_geo, const Shape &sh, uint32_t iter)
{
    if (lds_geo) launch_iteration_t<TYPE, true>(ctx, ln, scene, sh, iter);
    else launch_iteration_t<TYPE, false>(ctx, ln, scene, sh, iter);
}

// the lane-private part of one call: clear the queue counters, first rays, every iteration of the wavefront
static hipError_t enqueue_wavefront(LupinContext *ctx, Lane *ln, const LupinScene *scene, uint32_t pathtrace_type, bool lds_geo, uint32_t n,
                                    const Shape &sh, uint32_t iterations)
{
    const uint32_t blocks = sh.blocks;
    hipStream_t st = ln->stream;
    hipError_t e = hipMemsetAsync(ln->pb.counts, 0, LP_COUNTER_ROWS * (size_t)ln->counts_capacity * LP_SHARDS * sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_begin, dim3(blocks), dim3(LP_BLOCK), 0, st, (const FrameParams *)ln->d_fp, ln->pb, n);
    for (uint32_t it = 0; it < iterations; it++)
    {
        switch (pathtrace_type)
        {
        case LUPIN_PATHTRACE_STANDARD: launch_iteration<LUPIN_PATHTRACE_STANDARD>(ctx, ln, scene, lds_geo, sh, it); break;
        case LUPIN_PATHTRACE_MIS: launch_iteration<LUPIN_PATHTRACE_MIS>(ctx, ln, scene, lds_geo, sh, it); break;
        case LUPIN_PATHTRACE_NAIVE: launch_iteration<LUPIN_PATHTRACE_NAIVE>(ctx, ln, scene, lds_geo, sh, it); break;
        default: launch_iteration<LUPIN_PATHTRACE_DIRECT>(ctx, ln, scene, lds_geo, sh, it); break;
        }
    }
    return hipSuccess;
}

static int flush_pending(LupinContext *ctx);

// The resolves form a chain (each waits for the previous call's), so the latest call's event covers all lanes' texture writes.
// Every reader / writer of a texture on the primary stream comes through here (or through sync_all): recorded calls run first.
static void join_primary(LupinContext *ctx)
{
    flush_pending(ctx);
    if (ctx->last_lane > 0) hipStreamWaitEvent(ctx->stream, ctx->lanes[ctx->last_lane].done, 0);
}
static hipError_t sync_all(LupinContext *ctx)
{
    if (flush_pending(ctx) != LUPIN_OK) return hipErrorUnknown;
    hipError_t e = hipSuccess;
    for (int k = 0; k < LP_MAX_LANES && e == hipSuccess; k++)
        if (ctx->lanes[k].stream) e = hipStreamSynchronize(ctx->lanes[k].stream);
    return e;
}

// Handles outlive their context in host code that tears down in the wrong order (garbage-collected hosts do): every entry
// point that reaches a context through a texture / scene / communicator checks this registry instead of dereferencing a
// freed pointer.  (An address reused by a later context counts as alive again; its objects are then merely foreign.)
static std::mutex g_live_mu;
static std::set<const LupinContext *> g_live_contexts;
static bool ctx_alive(const LupinContext *ctx)
{
    std::lock_guard<std::mutex> lock(g_live_mu);
    return ctx && g_live_contexts.count(ctx) != 0;
}
#define CTX_ALIVE_TRY(c) do { if (!ctx_alive(c)) return fail(LUPIN_ERR_INVALID_ARGUMENT, "the context of this object has been destroyed"); } while (0)

bool lupin_internal_ctx_alive(const LupinContext *ctx) { return ctx_alive(ctx); }
int lupin_internal_fail(int code, const char *msg) { return fail(code, msg); }
void lupin_internal_join_primary(LupinContext *ctx) { join_primary(ctx); }
int lupin_internal_sync_all(LupinContext *ctx) { HIP_TRY(hipSetDevice(ctx->device)); HIP_TRY(sync_all(ctx)); return LUPIN_OK; }
// pack / unpack launches on the primary stream, ordered after every frame enqueued so far (see k_tiles_copy for `mode`)
int lupin_internal_tiles_copy(LupinContext *ctx, const LupinTexture *tex, void *packed, uint32_t tile_size, uint32_t rank, uint32_t world,
                              uint64_t capacity_px, int mode)
{
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t tpx = tile_size * LUPIN_WORKGROUP_SIZE;
    const uint32_t ntx = (tex->width - 1) / tpx + 1, nty = (tex->height - 1) / tpx + 1;
    const uint32_t blocks = mode == 2 ? ntx * nty : lupin_owned_tile_count(ntx * nty, rank, world);
    join_primary(ctx);
    // unpacked texels also refresh a valid f32 accumulator (widened: another rank's tile arrives as f16), so that
    // lupin_hip_texture_download_rgba32f and the next frame's blend see the gathered frame, not stale or zero words
    float4 *shadow = (mode != 0 && tex->accum32 && tex->accum32_valid) ? tex->accum32 : nullptr;
    if (blocks)
        hipLaunchKernelGGL(k_tiles_copy, dim3(blocks), dim3(LP_BLOCK), 0, ctx->stream, (uint2 *)tex->data, (uint2 *)packed, shadow, tex->width, tex->height,
                           tpx, rank, world, (unsigned long long)capacity_px, mode);
    HIP_TRY(hipGetLastError());
    return LUPIN_OK;
}
int lupin_internal_ctx_device(const LupinContext *ctx) { return ctx->device; }
hipStream_t lupin_internal_ctx_stream(const LupinContext *ctx) { return ctx->stream; }

// Frames in flight run on one stream each; the HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues
// (default 4).  Ask for 8 unless the host already chose -- effective when this library loads before the process's first
// HIP call; otherwise the lanes beyond the queue count share queues (correct, less overlap).
__attribute__((constructor)) static void lupin_hw_queues_default() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

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
    LupinRuntimeInfo ri;
    lupin_hip_runtime_info(&ri);
    if (ri.num_hip_runtimes_mapped > 1)
        return fail(LUPIN_ERR_HIP, std::string("two HIP runtimes are mapped into this process (") + ri.hip_runtime_paths +
                                   "): load liblupin_hip.so in a process that has not imported another copy (e.g. a PyTorch wheel's)");
    LupinContext *ctx = new LupinContext();
    { std::lock_guard<std::mutex> lock(g_live_mu); g_live_contexts.insert(ctx); }
    ctx->device = device_ordinal;
    ctx->runtime_version = ri.runtime_hip_version;
    hipError_t e = hipSuccess;
    const char *nl = getenv("LUPIN_LANES");
    if (nl) { ctx->num_lanes = std::min(LP_MAX_LANES, std::max(1, atoi(nl))); ctx->lanes_from_env = true; }
    for (int k = 0; k < ctx->num_lanes && e == hipSuccess; k++)
    {
        Lane &ln = ctx->lanes[k];
        e = hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipMalloc((void **)&ln.d_fp, LP_MAX_BATCH * sizeof(FrameParams));
        if (e == hipSuccess) e = hipMalloc((void **)&ln.stat_counters, 2 * LP_SHARDS * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemsetAsync(ln.stat_counters, 0, 2 * LP_SHARDS * sizeof(unsigned long long), ln.stream);
        if (e == hipSuccess) e = hipMalloc((void **)&ln.work_counters, LP_WORK_WORDS * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemsetAsync(ln.work_counters, 0, LP_WORK_WORDS * sizeof(unsigned long long), ln.stream);
        if (e == hipSuccess) e = hipMalloc((void **)&ln.wide_counters, LP_WIDE_WORDS * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemsetAsync(ln.wide_counters, 0, LP_WIDE_WORDS * sizeof(unsigned long long), ln.stream);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->marker, hipEventDisableTiming);
    if (e != hipSuccess)
    {
        lupin_hip_destroy_context(ctx);   // one cleanup path: registry entry, streams, events and buffers of the lanes made so far
        return fail(LUPIN_ERR_HIP, std::string("context setup: ") + hipGetErrorString(e));
    }
    ctx->stream = ctx->lanes[0].stream;
    const char *ext = getenv("LUPIN_EXTEND");
    if (ext && strcmp(ext, "persistent") == 0) ctx->persistent_extend = 1;
    else if (ext && strcmp(ext, "simple") == 0) ctx->persistent_extend = 0;
    const char *lg = getenv("LUPIN_LDS_GEOMETRY");
    ctx->lds_geometry = !(lg && strcmp(lg, "0") == 0);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0)
    {
        ctx->num_cus = (uint32_t)prop.multiProcessorCount;
    }
    const char *dbs = getenv("LUPIN_DEBUG_SYNC");
    ctx->debug_sync = dbs && strcmp(dbs, "0") != 0;
    if (const char *lsg = getenv("LUPIN_LIGHT_STAGE")) ctx->light_stage = atoi(lsg) != 0 ? 1 : 0;
    if (const char *pr = getenv("LUPIN_PATH_RECORDS")) ctx->path_records = atoi(pr) != 0 ? 1 : 0;
    const char *ssh = getenv("LUPIN_SIMPLE_SHADE");
    if (ssh && strcmp(ssh, "0") == 0) ctx->specialize_simple = false;
    const char *gr = getenv("LUPIN_GRAPH");
    if (gr) ctx->use_graph = strcmp(gr, "0") != 0;
    if (ctx->use_graph && ctx->runtime_version / 100000 != HIP_VERSION / 100000)
    {
        // Graph replay is validated on the runtime this library was built against (HIP_VERSION major.minor).  Round 1 saw a
        // memory fault on replay when a PyTorch wheel's older libamdhip64 served the process: refuse rather than risk it.
        lupin_hip_destroy_context(ctx);
        return fail(LUPIN_ERR_HIP, "LUPIN_GRAPH=1 needs the HIP runtime this library was built against (built " + std::to_string(HIP_VERSION) +
                                   ", running on " + std::to_string(ri.runtime_hip_version) + ")");
    }
    const char *shd = getenv("LUPIN_SHADOW");
    if (shd && strcmp(shd, "simple") == 0) ctx->persistent_shadow = false;
    if (const char *bf = getenv("LUPIN_BATCH")) ctx->batch_frames = (uint32_t)std::min((int)LP_MAX_BATCH, std::max(0, atoi(bf)));
    if (const char *tv = getenv("LUPIN_TRAVERSAL")) ctx->wide_traversal = strcmp(tv, "wide") == 0;
    if (const char *vw = getenv("LUPIN_VERIFY_WIDE")) ctx->verify_wide = atoi(vw) != 0;
    if (const char *ss = getenv("LUPIN_SHORT_STACK")) ctx->short_stack = (uint32_t)std::max(0, atoi(ss));
    *out_ctx = ctx;
    return LUPIN_OK;
}

void lupin_hip_destroy_context(LupinContext *ctx)
{
    if (!ctx) return;
    {
        std::lock_guard<std::mutex> lock(g_live_mu);
        if (!g_live_contexts.erase(ctx)) return;   // not ours, or destroyed already
    }
    hipSetDevice(ctx->device);
    sync_all(ctx);
    for (int k = 0; k < LP_MAX_LANES; k++)
    {
        PathBuffers &pb = ctx->lanes[k].pb;
        void *ptrs[] = {ctx->lanes[k].hot, ctx->lanes[k].shadow, ctx->lanes[k].skey, pb.vol0, pb.vol1, pb.queue[0], pb.queue[1], pb.counts, ctx->lanes[k].stat_counters,
                        ctx->lanes[k].work_counters, ctx->lanes[k].wide_counters, pb.retrace};
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
    CTX_ALIVE_TRY(ctx);
    if (!ctx) return fail(LUPIN_ERR_INVALID_ARGUMENT, "ctx is null");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(sync_all(ctx));
    return LUPIN_OK;
}

int lupin_hip_set_f16_store_rounding(LupinContext *ctx, int mode)
{
    CTX_ALIVE_TRY(ctx);
    if (ctx) { int frc = flush_pending(ctx); if (frc != LUPIN_OK) return frc; }   // calls recorded so far ran under the old setting
    if (!ctx || (mode != 0 && mode != 1)) return fail(LUPIN_ERR_INVALID_ARGUMENT, "mode must be 0 (toward zero) or 1 (nearest even)");
    ctx->store_rounding = mode;
    return LUPIN_OK;
}

// How many recorded calls one wavefront of `pixels`-pixel dispatches may carry.  Left to the library (batch_frames == 0): sixteen up
// to 4 M pixels -- 1080p launches are the small ones: materials1 / environments1 + 5 % / + 7 % over eight -- and eight above
// (3840 x 2160: + 0.9 % for twice the path state).  Always within the queue entries' slot bits.
static uint32_t frames_per_wavefront(const LupinContext *ctx, uint64_t pixels)
{
    uint64_t k = ctx->batch_frames ? ctx->batch_frames : (pixels <= (4ull << 20) ? 16u : 8u);
    if (pixels) k = std::min<uint64_t>(k, (uint64_t)QUEUE_SLOT_MASK / pixels);
    return (uint32_t)std::max<uint64_t>(k, 1);
}

int lupin_hip_set_batch_frames(LupinContext *ctx, uint32_t frames)
{
    CTX_ALIVE_TRY(ctx);
    if (!ctx || frames > LP_MAX_BATCH) return fail(LUPIN_ERR_INVALID_ARGUMENT, "frames per wavefront must be in [1, 16] (0: chosen by dispatch size)");
    int frc = flush_pending(ctx);
    if (frc != LUPIN_OK) return frc;
    ctx->batch_frames = frames;
    return LUPIN_OK;
}

int lupin_hip_set_traversal(LupinContext *ctx, int mode)
{
    CTX_ALIVE_TRY(ctx);
    if (ctx) { int frc = flush_pending(ctx); if (frc != LUPIN_OK) return frc; }   // calls recorded so far ran under the old setting
    if (!ctx || (mode != LUPIN_TRAVERSAL_WIDE && mode != LUPIN_TRAVERSAL_BINARY)) return fail(LUPIN_ERR_INVALID_ARGUMENT, "unknown traversal mode");
    ctx->wide_traversal = mode == LUPIN_TRAVERSAL_WIDE;
    return LUPIN_OK;
}

int lupin_hip_set_accumulation_mode(LupinContext *ctx, int mode)
{
    CTX_ALIVE_TRY(ctx);
    if (ctx) { int frc = flush_pending(ctx); if (frc != LUPIN_OK) return frc; }   // calls recorded so far ran under the old setting
    if (!ctx || (mode != LUPIN_ACCUM_F16_RUNNING_AVERAGE && mode != LUPIN_ACCUM_F32)) return fail(LUPIN_ERR_INVALID_ARGUMENT, "unknown accumulation mode");
    ctx->accum_mode = mode;
    return LUPIN_OK;
}

// Allocates, for every frame in flight, the path state of dispatches up to `pixels` pixels with the given baked parameters --
// what the first `num_lanes` pathtrace calls would otherwise do one after the other inside the caller's frame loop.
int lupin_hip_reserve_path_state(LupinContext *ctx, uint64_t pixels, uint32_t max_bounces, uint32_t samples_per_pixel)
{
    CTX_ALIVE_TRY(ctx);
    if (pixels == 0 || pixels > (uint64_t)QUEUE_SLOT_MASK || samples_per_pixel == 0) return fail(LUPIN_ERR_INVALID_ARGUMENT, "bad path-state reservation");
    const uint32_t batch = frames_per_wavefront(ctx, pixels);
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t iterations = samples_per_pixel * (max_bounces + 1);
    // the lanes the dispatches will rotate over (flush_pending's choice for a scene traced from global memory)
    const int lanes = ctx->lanes_from_env ? ctx->num_lanes : std::min(batch > 1 ? 1 : LP_MAX_LANES, ctx->num_lanes);
    for (int k = 0; k < lanes; k++)
    {
        int rc = ensure_path_buffers(ctx, &ctx->lanes[k], pixels * batch, iterations);
        if (rc != LUPIN_OK) return rc;
    }
    return LUPIN_OK;
}

int lupin_hip_build_pathtrace_resources(LupinContext *ctx, const LupinBakedPathtraceParams *params, LupinPathtraceResources **out_res)
{
    CTX_ALIVE_TRY(ctx);
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
    CTX_ALIVE_TRY(ctx);
    if (!ctx || !desc || !out_scene) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const LupinSceneDesc &s = *desc;

    // ---- validation (validate_scene, data_structures.rs:876-928, plus what the kernels index) ----
    if (s.num_instances > (1u << 26)) return fail(LUPIN_ERR_INVALID_ARGUMENT, "more than 2^26 instances (the tracer keeps an instance's flags above its 26-bit index)");
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
    sc->device = ctx->device;
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
        for (uint32_t n = 0; n < m.num_bvh_nodes; n++)
            if (m.bvh_nodes[n].tri_count == 0 && (uint64_t)m.bvh_nodes[n].tri_begin_or_first_child + 1 >= m.num_bvh_nodes)
            { lupin_hip_scene_destroy(sc); return fail(LUPIN_ERR_INVALID_ARGUMENT, "BLAS child index out of bounds"); }
        const uint32_t bd = blas_depth(m.bvh_nodes, m.num_bvh_nodes);
        if (bd == 0xFFFFFFFFu) { lupin_hip_scene_destroy(sc); return fail(LUPIN_ERR_INVALID_ARGUMENT, "BLAS is not a tree"); }
        max_blas_depth = std::max(max_blas_depth, bd);
        std::vector<uint32_t> leaf_ends, leaky_tris;
        const char *why = nullptr;
        const uint32_t root_ref = blas_child_pairs(m.bvh_nodes, m.num_bvh_nodes, ntris, md.tri_offset, blas, leaf_ends, &why, m.verts_pos, m.indices, &leaky_tris);
        if (why) { lupin_hip_scene_destroy(sc); return fail(LUPIN_ERR_INVALID_ARGUMENT, why); }
        // v0.w of a triangle: bit 0 = last of its leaf, bit 1 = not inside every box above it (the wide tracer re-traces hits on it)
        for (uint32_t t : leaky_tris) tris[t].v0.w = host_u2f(TRI_LEAKY_BITS);
        for (uint32_t last : leaf_ends) tris[last].v0.w = host_u2f(__builtin_bit_cast(uint32_t, tris[last].v0.w) | LEAF_END_BITS);
        sc->leaky_triangles += leaky_tris.size();
        mesh_root[mi] = root_ref;
    }

    // ---- TLAS ----
    std::vector<WideNode> tlas;
    uint32_t tlas_root = REF_LEAF;
    uint32_t tlas_depth = 0;
    if (s.num_tlas_nodes > 0)
    {
        // TLAS nodes go behind the BLAS nodes in ONE array (no per-lane base select in the traversal step): global references
        std::vector<uint32_t> ref(s.num_tlas_nodes);
        const uint32_t nblas = (uint32_t)blas.size();
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
                ref[n] = nblas + wide_count++;
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
            w.a = make_float4(l.aabb_min[0], r.aabb_min[0], l.aabb_min[1], r.aabb_min[1]);
            w.b = make_float4(l.aabb_min[2], r.aabb_min[2], l.aabb_max[0], r.aabb_max[0]);
            w.c = make_float4(l.aabb_max[1], r.aabb_max[1], l.aabb_max[2], r.aabb_max[2]);
            w.d = make_uint4(ref[nd.left], ref[nd.right], 0u, 0u);
            tlas[ref[n] - nblas] = w;
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
    blas.insert(blas.end(), tlas.begin(), tlas.end());   // the one node array: [BLAS | TLAS]

    // ---- the same hierarchies four-wide (wide tracer; scenes small enough for LDS staging are traced by k_extend) ----
    std::vector<Wide4> wide4;   // TLAS nodes first, then every mesh's BLAS nodes: one array, global indices
    std::vector<uint32_t> mesh_root4(s.num_meshes, REF_LEAF);
    uint32_t tlas4_root = tlas_root;
    const size_t lds_bytes_if_staged = blas.size() * 80 + tris.size() * 48 + (size_t)s.num_instances * 80;
    const bool build_wide = s.num_instances > 0 && !(lds_bytes_if_staged <= LP_GEO_LDS_LIMIT && ctx->lds_geometry);
    if (build_wide)
    {
        tlas4_root = collapse_to_wide4(blas, tlas_root, wide4);
        for (uint32_t mi = 0; mi < s.num_meshes; mi++) mesh_root4[mi] = collapse_to_wide4(blas, mesh_root[mi], wide4);
        if (wide4.size() >= (size_t)REF_INDEX_MASK || tris.size() >= (size_t)REF_INDEX_MASK) { lupin_hip_scene_destroy(sc); return fail(LUPIN_ERR_INVALID_ARGUMENT, "scene too large for 30-bit references"); }
        sc->has_wide = true;
    }

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
        // bits 8..11: material type, bit 12: the material's own roughness is zero (a hint: smooth reflective / refractive /
        // transparent surfaces take the delta branch unless a roughness texture says otherwise) = k_sort_queue's key
        const bool smooth = mat.roughness == 0.0f && (mat.mat_type == LUPIN_MAT_REFLECTIVE || mat.mat_type == LUPIN_MAT_REFRACTIVE || mat.mat_type == LUPIN_MAT_TRANSPARENT);
        d.flags = (maybe_alpha ? 1u : 0u) | ((mat.mat_type & 0xFu) << 8) | (smooth ? 1u << 12 : 0u);
        mat_types_seen |= 1u << (mat.mat_type & 0xFu);
        instances[i] = d;
    }

    // per-instance copies of the mesh record and the material (shading fetches them beside the instance record)
    std::vector<MeshDev> inst_meshes(s.num_instances);
    std::vector<LupinMaterial> inst_materials(s.num_instances);
    for (uint32_t i = 0; i < s.num_instances; i++)
    {
        inst_meshes[i] = meshes[s.instances[i].mesh_idx];
        inst_materials[i] = s.materials[s.instances[i].mat_idx];
    }
    std::vector<uint32_t> inst_root4(s.num_instances);
    for (uint32_t i = 0; i < s.num_instances; i++) inst_root4[i] = mesh_root4[s.instances[i].mesh_idx];

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
    // (padded to whole groups of four: lights_pdf fetches the bounds four at a time, 64 aligned bytes per scalar load)
    std::vector<float4> light_bounds(((size_t)s.num_lights + 3) / 4 * 4, make_float4(0.0f, 0.0f, 0.0f, -1.0f));
    for (uint32_t i = 0; i < s.num_lights; i++)
    {
        const LupinInstance &in = s.instances[s.lights[i].instance_idx];
        const LupinMeshDesc &m = s.meshes[in.mesh_idx];
        const float (*r)[4] = in.transpose_inverse_transform.m;   // 3 rows x 4: world -> local
        const double a = r[0][0], b = r[0][1], c = r[0][2], d = r[1][0], e = r[1][1], f = r[1][2], g = r[2][0], h = r[2][1], k = r[2][2];
        const double det = a * (e * k - f * h) - b * (d * k - f * g) + c * (d * h - e * g);
        float4 bound = make_float4(0.0f, 0.0f, 0.0f, INFINITY);   // singular / non-finite transform: never culled (.w = radius squared)
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
                const double padded = rad * 1.02 + 1e-4 * (cmag + rad) + 1e-6;
                bound = make_float4((float)cx, (float)cy, (float)cz, (float)(padded * padded * (1.0 + 1e-6)));   // .w = padded radius squared
            }
        }
        light_bounds[i] = bound;
    }

    // small scenes: one blob [tlas | blas | tris | instances] in 16-byte words for LDS staging.  Nodes and instances are
    // laid out at a stride of FIVE words (80 bytes) instead of four: a wave's lanes read records at divergent indices with
    // ds_read_b128, and at a 64-byte stride records i and i + 4 start in the same LDS bank (round 1's PMC: 62 M
    // SQ_LDS_BANK_CONFLICT cycles against 184 M SQ_ACTIVE_INST_LDS on the Cornell box); at 80 bytes only i and i + 16
    // collide.  Triangles keep 48 bytes (3 words: i and i + 16 collide as well).
    std::vector<float4> geo_blob;
    uint32_t off_blas = 0, off_tris = 0, off_inst = 0;
    {
        const size_t bytes = blas.size() * 80 + tris.size() * 48 + instances.size() * 80;
        if (bytes > 0 && bytes <= LP_GEO_LDS_LIMIT)
        {
            auto append = [&](const void *p, size_t count, size_t words) {   // records of `words` 16-byte words, padded to LP_GEO_LDS_STRIDE
                const float4 *f = reinterpret_cast<const float4 *>(p);
                for (size_t r = 0; r < count; r++)
                {
                    geo_blob.insert(geo_blob.end(), f + r * words, f + (r + 1) * words);
                    if (words == 4) geo_blob.push_back(make_float4(0.0f, 0.0f, 0.0f, 0.0f));
                }
            };
            append(blas.data(), blas.size(), 4);   // [BLAS | TLAS] nodes, indexed by the global references
            off_blas = 0;
            off_tris = (uint32_t)geo_blob.size();
            append(tris.data(), tris.size(), 3);
            off_inst = (uint32_t)geo_blob.size();
            append(instances.data(), instances.size(), 4);
        }
    }

    SceneDev &dv = sc->dev;
    int rc = LUPIN_OK;
    std::vector<LupinMaterial> materials(s.materials, s.materials + s.num_materials);
    std::vector<LupinEnvironment> envs(s.environments, s.environments + s.num_environments);
    std::vector<LupinLight> lights(s.lights, s.lights + s.num_lights);
    if ((rc = upload(sc, blas, &dv.blas)) || (rc = upload(sc, tris, &dv.tris)) ||
        (rc = upload(sc, tri_indices, &dv.tri_indices)) || (rc = upload(sc, instances, &dv.instances)) ||
        (rc = upload(sc, meshes, &dv.meshes)) || (rc = upload(sc, materials, &dv.materials)) ||
        (rc = upload(sc, inst_meshes, &dv.inst_meshes)) || (rc = upload(sc, inst_materials, &dv.inst_materials)) ||
        (rc = upload(sc, normals, &dv.normals)) || (rc = upload(sc, texcoords, &dv.texcoords)) || (rc = upload(sc, colors, &dv.colors)) ||
        (rc = upload(sc, textures, &dv.textures)) || (rc = upload(sc, texels, &dv.texels)) ||
        (rc = upload(sc, envs, &dv.environments)) || (rc = upload(sc, lights, &dv.lights)) ||
        (rc = upload(sc, alias_ranges, &dv.alias_ranges)) || (rc = upload(sc, env_alias_ranges, &dv.env_alias_ranges)) ||
        (rc = upload(sc, alias_bins, &dv.alias_bins)) || (rc = upload(sc, geo_blob, &dv.geo_blob)) ||
        (rc = upload(sc, light_bounds, &dv.light_bounds)) ||
        (rc = upload(sc, wide4, &dv.wide4)) || (rc = upload(sc, inst_root4, &dv.inst_root4)))
    {
        lupin_hip_scene_destroy(sc);
        return rc;
    }
    dv.tlas = dv.blas;
    dv.tlas_root = tlas_root;
    dv.tlas4_root = tlas4_root;
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
    static std::atomic<uint64_t> next_scene_id{1};   // contexts may live on different host threads
    sc->id = next_scene_id.fetch_add(1);
    hipError_t e = hipStreamSynchronize(ctx->stream);   // host vectors go out of scope
    if (e != hipSuccess) { lupin_hip_scene_destroy(sc); return fail(LUPIN_ERR_HIP, hipGetErrorString(e)); }
    *out_scene = sc;
    return LUPIN_OK;
}

void lupin_hip_scene_destroy(LupinScene *scene)
{
    if (!scene) return;
    hipSetDevice(scene->device);
    if (ctx_alive(scene->ctx)) sync_all(scene->ctx);   // a destroyed context has drained its streams already
    for (void *p : scene->allocations) hipFree(p);
    delete scene;
}

// ---- textures / double buffering ----

int lupin_hip_texture_create(LupinContext *ctx, uint32_t width, uint32_t height, LupinTexture **out_tex)
{
    CTX_ALIVE_TRY(ctx);
    if (!ctx || !out_tex || width == 0 || height == 0) return fail(LUPIN_ERR_INVALID_ARGUMENT, "bad texture size");
    HIP_TRY(hipSetDevice(ctx->device));
    LupinTexture *t = new LupinTexture();
    t->ctx = ctx; t->device = ctx->device; t->width = width; t->height = height; t->data = nullptr; t->accum32 = nullptr; t->accum32_valid = false;
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
    hipSetDevice(tex->device);
    if (ctx_alive(tex->ctx)) sync_all(tex->ctx);
    hipFree(tex->data);
    if (tex->accum32) hipFree(tex->accum32);
    delete tex;
}
uint32_t lupin_hip_texture_width(const LupinTexture *tex) { return tex ? tex->width : 0; }
uint32_t lupin_hip_texture_height(const LupinTexture *tex) { return tex ? tex->height : 0; }
void *lupin_hip_texture_device_ptr(const LupinTexture *tex) { return tex ? (void *)tex->data : nullptr; }

int lupin_hip_texture_upload_rgba16f(LupinTexture *tex, const uint16_t *pixels)
{
    if (!tex || !pixels) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    CTX_ALIVE_TRY(tex->ctx);
    HIP_TRY(hipSetDevice(tex->ctx->device));
    join_primary(tex->ctx);
    HIP_TRY(hipMemcpyAsync(tex->data, pixels, (size_t)tex->width * tex->height * 8, hipMemcpyHostToDevice, tex->ctx->stream));
    HIP_TRY(hipStreamSynchronize(tex->ctx->stream));
    tex->accum32_valid = false;   // the f16 texels are now the only truth
    return LUPIN_OK;
}
int lupin_hip_texture_download_rgba32f(const LupinTexture *tex, float *out_pixels)
{
    if (!tex || !out_pixels) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    if (!tex->accum32 || !tex->accum32_valid) return fail(LUPIN_ERR_INVALID_ARGUMENT, "texture has no f32 accumulator (render into it with LUPIN_ACCUM_F32 first)");
    CTX_ALIVE_TRY(tex->ctx);
    HIP_TRY(hipSetDevice(tex->ctx->device));
    join_primary(tex->ctx);
    HIP_TRY(hipMemcpyAsync(out_pixels, tex->accum32, (size_t)tex->width * tex->height * 16, hipMemcpyDeviceToHost, tex->ctx->stream));
    HIP_TRY(hipStreamSynchronize(tex->ctx->stream));
    return LUPIN_OK;
}
int lupin_hip_texture_download_rgba16f(const LupinTexture *tex, uint16_t *out_pixels)
{
    if (!tex || !out_pixels) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    CTX_ALIVE_TRY(tex->ctx);
    HIP_TRY(hipSetDevice(tex->ctx->device));
    join_primary(tex->ctx);
    HIP_TRY(hipMemcpyAsync(out_pixels, tex->data, (size_t)tex->width * tex->height * 8, hipMemcpyDeviceToHost, tex->ctx->stream));
    HIP_TRY(hipStreamSynchronize(tex->ctx->stream));
    return LUPIN_OK;
}

int lupin_hip_dbuf_create(LupinContext *ctx, uint32_t width, uint32_t height, LupinDoubleBufferedTexture **out)
{
    CTX_ALIVE_TRY(ctx);
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
    CTX_ALIVE_TRY(t->ctx);
    HIP_TRY(hipSetDevice(t->ctx->device));
    join_primary(t->ctx);
    HIP_TRY(hipMemcpyAsync(b->data, f->data, (size_t)f->width * f->height * 8, hipMemcpyDeviceToDevice, t->ctx->stream));
    b->accum32_valid = false;
    if (f->accum32 && f->accum32_valid)
    {
        if (!b->accum32) HIP_TRY(hipMalloc((void **)&b->accum32, (size_t)f->width * f->height * 16));
        HIP_TRY(hipMemcpyAsync(b->accum32, f->accum32, (size_t)f->width * f->height * 16, hipMemcpyDeviceToDevice, t->ctx->stream));
        b->accum32_valid = true;
    }
    return LUPIN_OK;
}
void lupin_hip_dbuf_flip(LupinDoubleBufferedTexture *t) { if (t) std::swap(t->front_idx, t->back_idx); }
int lupin_hip_dbuf_resize(LupinDoubleBufferedTexture *t, uint32_t width, uint32_t height)
{
    if (!t) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    if (t->tex[0]->width == width && t->tex[0]->height == height) return LUPIN_OK;   // wgpu_utils.rs:343
    LupinTexture *a = nullptr, *b = nullptr;
    CTX_ALIVE_TRY(t->ctx);
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
    CTX_ALIVE_TRY(ctx);
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
        const uint32_t owned = lupin_owned_tile_count(total, rank, world);
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
    if (n64 * 8u > (uint64_t)QUEUE_SLOT_MASK) return fail(LUPIN_ERR_INVALID_ARGUMENT, "dispatch too large");   // queue entries keep two bits for the light-pdf stage; a wavefront of dispatches this size holds up to eight frames (frames_per_wavefront)
    const uint32_t n = (uint32_t)n64;

    if (falsecolor_type >= 0 || debug)
    {
        const uint32_t fblocks = (n + LP_BLOCK - 1) / LP_BLOCK;
        const uint32_t fstack_words = scene->stack_entries * LP_BLOCK;
        const bool flds = scene->dev.geo_blob_words && ctx->lds_geometry;
        const size_t flds_bytes = (size_t)fstack_words * sizeof(uint32_t) + (flds ? (size_t)scene->dev.geo_blob_words * 16 : 0);
        if (flds_bytes > 160 * 1024) return fail(LUPIN_ERR_INVALID_ARGUMENT, "BVH too deep for the LDS traversal stack");
        const __half *pv = prev ? prev->data : (const __half *)nullptr;
        render_target->accum32_valid = false;
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

    // ---- record the call; run the batch when it is full or cannot grow (DESIGN 5 "Frames per wavefront") ----
    fp.frame_slots = n;
    fp.num_frames = 1;
    const uint32_t max_frames = frames_per_wavefront(ctx, n64);
    const bool batchable = max_frames > 1 && !ctx->counting && !ctx->verify_wide && !ctx->debug_sync && ctx->accum_mode != LUPIN_ACCUM_F32;
    if (!ctx->pending.empty())
    {
        // a call joins the batch if it differs from the batch's first call only in camera and accum_counter, and blends with
        // what the previous call of the batch stores (or with nothing: accum_counter 0)
        FrameParams a = ctx->pending.front().fp, b = fp;
        for (FrameParams *q : {&a, &b})
        {
            memset(&q->pc.camera_transform, 0, sizeof(q->pc.camera_transform));
            q->pc.camera_lens = q->pc.camera_film = q->pc.camera_aspect = q->pc.camera_focus = q->pc.camera_aperture = 0.0f;
            q->pc.accum_counter = 0;
            q->pc.flags &= ~(uint32_t)LUPIN_FLAG_CAMERA_ORTHO;
            q->num_frames = 1;
        }
        const bool joins = batchable && ctx->pending_scene == scene && ctx->pending_type == pathtrace_type && memcmp(&a, &b, sizeof(a)) == 0 &&
                           (fp.pc.accum_counter == 0 || prev == ctx->pending.back().target) && ctx->pending.size() < max_frames;
        if (!joins)
        {
            int rc = flush_pending(ctx);
            if (rc != LUPIN_OK) return rc;
        }
    }
    ctx->pending.push_back({fp, render_target, prev});
    ctx->pending_scene = scene;
    ctx->pending_type = pathtrace_type;
    if (!batchable || ctx->pending.size() >= max_frames) return flush_pending(ctx);
    return LUPIN_OK;
}

// Runs the recorded calls as one wavefront: slot = frame * frame_slots + pixel slot, one resolve that applies the frames'
// blends per pixel in call order.  Errors of a deferred call surface here, i.e. at the call that needed its result.
static int flush_pending(LupinContext *ctx)
{
    if (ctx->pending.empty() || ctx->in_flush) return LUPIN_OK;
    ctx->in_flush = true;
    struct Done { LupinContext *c; ~Done() { c->pending.clear(); c->pending_scene = nullptr; c->in_flush = false; } } done{ctx};
    HIP_TRY(hipSetDevice(ctx->device));
    const LupinScene *scene = ctx->pending_scene;
    const uint32_t pathtrace_type = ctx->pending_type;
    const uint32_t K = (uint32_t)ctx->pending.size();
    FrameParams fp = ctx->pending.front().fp;
    fp.num_frames = K;
    const uint32_t frame_slots = fp.frame_slots;
    const uint32_t n = K * frame_slots;                         // slots of the wavefront
    LupinTexture *render_target = ctx->pending.back().target;    // (f32 accumulation is never batched: K == 1 there)
    const LupinTexture *prev = ctx->pending.front().prev;
    const uint32_t W = fp.width, H = fp.height;

    // Lane choice: consecutive wavefronts alternate over `lanes` streams so that they overlap; per-kernel timing needs them serial.
    // * A wavefront of ONE frame of a scene traced by the persistent kernel does not fill the chip in its middle iterations
    //   (bounded by the latency of one traversal): every lane is used (an eighth of the 4K frame: 34.1 -> 30.1 ms with 8 lanes).
    // * A wavefront of SEVERAL frames (the default: up to eight calls) fills it on its own, and then stages of different
    //   wavefronts only take LDS, registers and cache from each other: one lane, every stage with the chip to itself
    //   (bistro-class 4K, eight frames: 159 ms per step on one lane, 175 - 183 on two, 177 on three; profiles/r03_batch_matrix_short.txt).
    // * Launch-bound LDS-resident scenes are best with four lanes (Cornell box 8.0 against 7.2 Gsamples/s on one).
    const bool lds_scene = scene->dev.geo_blob_words && ctx->lds_geometry;
    const int lanes = ctx->lanes_from_env ? ctx->num_lanes : std::min(lds_scene ? 4 : (K > 1 ? 1 : LP_MAX_LANES), ctx->num_lanes);
    const bool chip_to_itself = lanes == 1 || ctx->timing;
    const int w = ctx->timing ? 0 : (int)(ctx->call_index % (uint64_t)lanes);
    ctx->call_index++;
    Lane *ln = &ctx->lanes[w];
    hipStream_t st = ln->stream;

    const uint32_t iterations = fp.spp * (fp.max_bounces + 1);
    int rc = ensure_path_buffers(ctx, ln, n, iterations);
    if (rc != LUPIN_OK) return rc;
    // scenes whose queues get sorted by material read the path state at scattered slots: records; otherwise planes
    set_path_layout(ln, ctx->path_records < 0 ? (scene->dev.sort_shade != 0) : ctx->path_records != 0);

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

    const uint32_t pblocks = persistent_grid(ctx, scene, pathtrace_type, lds_geo, lds, !chip_to_itself);
    Shape sh;
    sh.blocks = blocks; sh.pblocks = pblocks; sh.lds = lds; sh.stack_words = stack_words;
    if (pblocks && !lds_geo && scene->has_wide && ctx->wide_traversal)
    {
        // the wide tracer keeps (reference, distance) pairs on a bounded stack; a query that would overflow it is re-traced
        sh.wstack_words = 2u * ctx->wide_stack_pairs * LP_BLOCK;
        sh.wlds = (size_t)sh.wstack_words * sizeof(uint32_t);
        sh.wblocks = wide_grid(ctx, scene, pathtrace_type, sh.wlds);
    }
    if (pblocks && !lds_geo && !sh.wblocks && chip_to_itself && ctx->short_stack && scene->stack_entries > ctx->short_stack)
    {
        // A fifth block per CU for scenes whose depth bound asks for more than 32 KB of stack per block (bistro-class: 40):
        // the tracer is bound by latency x waves (four blocks instead of three: -18 %, five instead of four: -10 %), and the
        // stack a query actually uses is far below the bound (bistro-class 4K: 33 of 3.5 G queries need more than 20 entries).
        // Only with the chip to itself: sharing it, the larger tracer loses more to the other lanes' stages than it gains.
        sh.sstack_words = ctx->short_stack * LP_BLOCK;
        sh.slds = (size_t)sh.sstack_words * sizeof(uint32_t);
        sh.sblocks = short_grid(ctx, scene, pathtrace_type, sh.slds);
    }
    ctx->last_lanes = lanes;
    ctx->last_wide = sh.wblocks != 0;
    ctx->last_batch = K;
    ctx->last_short = sh.sblocks ? ctx->short_stack : 0u;
    for (uint32_t k = 0; k < K; k++)
    {
        FrameParams fk = ctx->pending[k].fp;
        fk.num_frames = K;
        hipLaunchKernelGGL(k_set_params, dim3(1), dim3(1), 0, st, fk, ln->d_fp + k);
    }
    if (ctx->use_graph && !ctx->timing && !ctx->counting && !ctx->verify_wide)
    {
        // Everything between k_set_params and the resolve depends on the call only through *d_fp, so it is captured once per
        // (scene, dispatch size, integrator, buffers) and replayed: one graph launch instead of 2-4 launches per iteration.
        Lane::GraphKey key;
        key.scene_id = scene->id; key.pb_generation = ln->pb_generation; key.n = n; key.blocks = blocks; key.type = pathtrace_type;
        key.iterations = iterations; key.stack_words = stack_words; key.lds = (uint32_t)lds; key.pblocks = pblocks;
        key.persistent = ctx->persistent_extend;
        key.persistent_shadow = ctx->persistent_shadow ? 1 : 0; key.lds_geometry = lds_geo ? 1 : 0;
        key.wide_blocks = sh.wblocks ? sh.wblocks : sh.sblocks; key.wide_stack_words = sh.wblocks ? sh.wstack_words : sh.sstack_words;
        const bool have = ln->graph_exec && key == ln->graph_key;
        if (!have && ln->graph_exec && !(key == ln->seen_key))
        {
            // the lane holds a graph of another shape and this one is new (shapes alternate, e.g. edge tiles): capturing
            // costs about a millisecond, so launch directly and re-capture only if the shape repeats
            ln->seen_key = key;
            HIP_TRY(enqueue_wavefront(ctx, ln, scene, pathtrace_type, lds_geo, n, sh, iterations));
        }
        else
        {
            if (!have)
            {
                if (ln->graph_exec) { hipGraphExecDestroy(ln->graph_exec); ln->graph_exec = nullptr; }
                if (ln->graph) { hipGraphDestroy(ln->graph); ln->graph = nullptr; }
                HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                hipError_t ce = enqueue_wavefront(ctx, ln, scene, pathtrace_type, lds_geo, n, sh, iterations);
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
        HIP_TRY(enqueue_wavefront(ctx, ln, scene, pathtrace_type, lds_geo, n, sh, iterations));
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
    const float4 *prev32 = nullptr;
    float4 *out32 = nullptr;
    if (ctx->accum_mode == LUPIN_ACCUM_F32)
    {
        if (!render_target->accum32)
        {
            HIP_TRY(hipMalloc((void **)&render_target->accum32, (size_t)W * H * 16));
            HIP_TRY(hipMemsetAsync(render_target->accum32, 0, (size_t)W * H * 16, st));
        }
        out32 = render_target->accum32;
        if (prev && prev->accum32 && prev->accum32_valid) prev32 = prev->accum32;
    }
    render_target->accum32_valid = out32 != nullptr;
    ResolveBatch rb;
    memset(&rb, 0, sizeof(rb));
    rb.count = K;
    for (uint32_t k = 0; k < K; k++)
    {
        rb.target[k] = ctx->pending[k].target->data;
        rb.accum_counter[k] = ctx->pending[k].fp.pc.accum_counter;
        bool last_write = true;
        for (uint32_t q = k + 1; q < K; q++) last_write = last_write && ctx->pending[q].target != ctx->pending[k].target;
        if (last_write) rb.store_mask |= 1u << k;
        if (k + 1 < K) ctx->pending[k].target->accum32_valid = false;
    }
    hipLaunchKernelGGL(k_resolve, dim3((frame_slots + LP_BLOCK - 1) / LP_BLOCK), dim3(LP_BLOCK), 0, st, fp, ln->pb, frame_slots, rb,
                       prev ? prev->data : (const __half *)nullptr, prev32, out32);
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
    CTX_ALIVE_TRY(ctx);
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
    for (int k = 0; k < ctx->num_lanes; k++)
    {
        HIP_TRY(hipMemsetAsync(ctx->lanes[k].work_counters, 0, LP_WORK_WORDS * sizeof(unsigned long long), ctx->lanes[k].stream));
        HIP_TRY(hipMemsetAsync(ctx->lanes[k].wide_counters, 0, LP_WIDE_WORDS * sizeof(unsigned long long), ctx->lanes[k].stream));
    }
    ctx->timing = enable_kernel_timing == LUPIN_STATS_KERNEL_TIMING;
    ctx->counting = enable_kernel_timing == LUPIN_STATS_WORK_COUNTERS;
    ctx->extend_launches = 0;
    return LUPIN_OK;
}

int lupin_hip_stats_get(LupinContext *ctx, LupinStats *out)
{
    CTX_ALIVE_TRY(ctx);
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
    for (int l = 0; l < ctx->num_lanes; l++)
    {
        unsigned long long w[LP_WORK_WORDS];
        HIP_TRY(hipMemcpy(w, ctx->lanes[l].work_counters, sizeof(w), hipMemcpyDeviceToHost));
        for (int m = 0; m < 3; m++)
        {
            out->node_visits[m] += w[4 * m + 0]; out->tri_tests[m] += w[4 * m + 1]; out->instance_entries[m] += w[4 * m + 2];
            out->wide_node_visits[m] += w[4 * m + 3];
        }
        for (int k = 0; k < 10; k++) out->tracer_rounds[k] += w[12 + k];
        for (int k = 0; k < 6; k++) out->tracer_cycles[k] += w[22 + k];
        unsigned long long wd[LP_WIDE_WORDS];
        HIP_TRY(hipMemcpy(wd, ctx->lanes[l].wide_counters, sizeof(wd), hipMemcpyDeviceToHost));
        out->wide_queries += wd[0]; out->wide_retraced += wd[1];
        out->verify_checked += wd[2]; out->verify_flagged += wd[3]; out->verify_mismatches += wd[4]; out->verify_raw_mismatches += wd[5];
        for (int k = 0; k < 4; k++) out->verify_reasons[k] += wd[6 + k];
    }
    out->frames_in_flight = (uint32_t)std::max(0, ctx->last_lanes);
    out->wide_traversal = ctx->last_wide ? 1u : 0u;
    out->frames_per_wavefront = ctx->last_batch;
    out->short_stack_entries = ctx->last_short;
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

int lupin_hip_measure_copy_bandwidth(LupinContext *ctx, uint64_t bytes, uint32_t reps, double *out_gb_per_s)
{
    CTX_ALIVE_TRY(ctx);
    if (!ctx || !out_gb_per_s || bytes < 16 || reps == 0) return fail(LUPIN_ERR_INVALID_ARGUMENT, "bad copy-bandwidth arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(sync_all(ctx));
    const size_t n = bytes / 16;
    float4 *a = nullptr, *b = nullptr;
    hipError_t e = hipMalloc((void **)&a, n * 16);
    if (e == hipSuccess) e = hipMalloc((void **)&b, n * 16);
    if (e != hipSuccess) { if (a) hipFree(a); return fail(LUPIN_ERR_OUT_OF_MEMORY, hipGetErrorString(e)); }
    hipEvent_t e0 = get_event(ctx), e1 = get_event(ctx);
    HIP_TRY(hipMemsetAsync(a, 0x3C, n * 16, ctx->stream));
    const uint32_t blocks = (uint32_t)((n + LP_BLOCK - 1) / LP_BLOCK);   // one 16-byte element per thread: the shape that reaches the guide's 6.29 TB/s (tools/calib/copy_probe.hip)
    hipLaunchKernelGGL(k_copy_bw, dim3(blocks), dim3(LP_BLOCK), 0, ctx->stream, (const float4 *)a, b, n);   // warm-up: pages touched
    HIP_TRY(hipEventRecord(e0, ctx->stream));
    for (uint32_t r = 0; r < reps; r++)
        hipLaunchKernelGGL(k_copy_bw, dim3(blocks), dim3(LP_BLOCK), 0, ctx->stream, (const float4 *)((r & 1) ? b : a), (r & 1) ? a : b, n);
    HIP_TRY(hipEventRecord(e1, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    ctx->ev_pool.push_back(e0); ctx->ev_pool.push_back(e1);
    hipFree(a); hipFree(b);
    *out_gb_per_s = ms > 0.0f ? 2.0 * (double)(n * 16) * reps / (ms * 1e-3) / 1e9 : 0.0;   // bytes read + bytes written
    return LUPIN_OK;
}

int lupin_hip_runtime_info(LupinRuntimeInfo *out)
{
    if (!out) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    memset(out, 0, sizeof(*out));
    out->build_hip_version = HIP_VERSION;
    int v = 0;
    if (hipRuntimeGetVersion(&v) == hipSuccess) out->runtime_hip_version = v;
    // every distinct libamdhip64 mapped into this process (a PyTorch wheel bundles its own copy)
    std::vector<std::string> libs;
    if (FILE *f = fopen("/proc/self/maps", "r"))
    {
        char line[4096];
        while (fgets(line, sizeof(line), f))
        {
            const char *p = strstr(line, "libamdhip64");
            if (!p) continue;
            const char *path = strchr(line, '/');
            if (!path) continue;
            std::string sp(path);
            while (!sp.empty() && (sp.back() == '\n' || sp.back() == ' ')) sp.pop_back();
            if (std::find(libs.begin(), libs.end(), sp) == libs.end()) libs.push_back(sp);
        }
        fclose(f);
    }
    out->num_hip_runtimes_mapped = (uint32_t)libs.size();
    std::string joined;
    for (auto &l : libs) { if (!joined.empty()) joined += ";"; joined += l; }
    strncpy(out->hip_runtime_paths, joined.c_str(), sizeof(out->hip_runtime_paths) - 1);
    return LUPIN_OK;
}

int64_t lupin_hip_collapse_bvh4(const LupinBvhNode *nodes, uint32_t num_nodes, const float *verts_pos4, uint32_t num_verts, const uint32_t *indices,
                                uint32_t num_indices, void *out_nodes, uint64_t capacity, uint32_t *out_root, uint8_t *out_tri_flags)
{
    if (!nodes || num_nodes == 0 || !out_root) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    const uint32_t num_tris = num_indices / 3;
    for (uint32_t n = 0; n < num_nodes; n++)
        if (nodes[n].tri_count == 0 && (uint64_t)nodes[n].tri_begin_or_first_child + 1 >= num_nodes) return fail(LUPIN_ERR_INVALID_ARGUMENT, "BLAS child index out of bounds");
    if (blas_depth(nodes, num_nodes) == 0xFFFFFFFFu) return fail(LUPIN_ERR_INVALID_ARGUMENT, "BLAS is not a tree");
    if (verts_pos4 && indices)
        for (uint32_t i = 0; i < num_tris * 3; i++) if (indices[i] >= num_verts) return fail(LUPIN_ERR_INVALID_ARGUMENT, "vertex index out of range");
    std::vector<WideNode> pairs;
    std::vector<uint32_t> leaf_ends, leaky;
    const char *why = nullptr;
    const uint32_t root = blas_child_pairs(nodes, num_nodes, num_tris, 0u, pairs, leaf_ends, &why, verts_pos4, indices, &leaky);
    if (why) return fail(LUPIN_ERR_INVALID_ARGUMENT, why);
    std::vector<Wide4> wide;
    *out_root = collapse_to_wide4(pairs, root, wide);
    if (out_tri_flags)
    {
        memset(out_tri_flags, 0, num_tris);
        for (uint32_t t : leaky) out_tri_flags[t] |= 2u;
        for (uint32_t t : leaf_ends) out_tri_flags[t] |= 1u;
    }
    if (out_nodes)
    {
        if (wide.size() > capacity) return fail(LUPIN_ERR_INVALID_ARGUMENT, "out_nodes too small");
        if (!wide.empty()) memcpy(out_nodes, wide.data(), wide.size() * sizeof(Wide4));
    }
    return (int64_t)wide.size();
}

static int trace_rays_impl(LupinContext *ctx, const LupinScene *scene, uint32_t n, const float *ori_xyz, const float *dir_xyz,
                           float ray_epsilon, uint32_t *out_hit, float *out_dst, float *out_uv, uint32_t *out_instance, uint32_t *out_tri, uint32_t *out_flag)
{
    CTX_ALIVE_TRY(ctx);
    if (!ctx || !scene || !ori_xyz || !dir_xyz || !out_hit || !out_dst || !out_uv || !out_instance || !out_tri) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    if (out_flag && !scene->has_wide) return fail(LUPIN_ERR_INVALID_ARGUMENT, "this scene has no four-wide hierarchy (it is small enough to be staged in LDS)");
    if (n == 0) return LUPIN_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    float *d_ori = nullptr, *d_dir = nullptr, *d_dst = nullptr, *d_uv = nullptr;
    uint32_t *d_hit = nullptr, *d_inst = nullptr, *d_tri = nullptr, *d_flag = nullptr;
    HIP_TRY(hipMalloc((void **)&d_ori, (size_t)n * 12));
    HIP_TRY(hipMalloc((void **)&d_dir, (size_t)n * 12));
    HIP_TRY(hipMalloc((void **)&d_dst, (size_t)n * 4));
    HIP_TRY(hipMalloc((void **)&d_uv, (size_t)n * 8));
    HIP_TRY(hipMalloc((void **)&d_hit, (size_t)n * 4));
    HIP_TRY(hipMalloc((void **)&d_inst, (size_t)n * 4));
    HIP_TRY(hipMalloc((void **)&d_tri, (size_t)n * 4));
    if (out_flag) HIP_TRY(hipMalloc((void **)&d_flag, (size_t)n * 4));
    HIP_TRY(hipMemcpyAsync(d_ori, ori_xyz, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_dir, dir_xyz, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
    if (out_flag)
    {
        const size_t lds = (size_t)ctx->wide_stack_pairs * 2u * LP_BLOCK * sizeof(uint32_t);
        hipLaunchKernelGGL(k_trace_wide, dim3((n + LP_BLOCK - 1) / LP_BLOCK), dim3(LP_BLOCK), lds, ctx->stream, scene->dev, n, d_ori, d_dir, ray_epsilon,
                           ctx->wide_stack_pairs, d_hit, d_dst, d_uv, d_inst, d_tri, d_flag);
        HIP_TRY(hipMemcpyAsync(out_flag, d_flag, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    else
    {
        const size_t lds = (size_t)scene->stack_entries * LP_BLOCK * sizeof(uint32_t);
        hipLaunchKernelGGL(k_trace, dim3((n + LP_BLOCK - 1) / LP_BLOCK), dim3(LP_BLOCK), lds, ctx->stream, scene->dev, n, d_ori, d_dir, ray_epsilon,
                           d_hit, d_dst, d_uv, d_inst, d_tri);
    }
    HIP_TRY(hipMemcpyAsync(out_hit, d_hit, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out_dst, d_dst, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out_uv, d_uv, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out_instance, d_inst, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out_tri, d_tri, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    hipFree(d_ori); hipFree(d_dir); hipFree(d_dst); hipFree(d_uv); hipFree(d_hit); hipFree(d_inst); hipFree(d_tri);
    if (d_flag) hipFree(d_flag);
    return LUPIN_OK;
}

int lupin_hip_trace_rays(LupinContext *ctx, const LupinScene *scene, uint32_t n, const float *ori_xyz, const float *dir_xyz,
                         float ray_epsilon, uint32_t *out_hit, float *out_dst, float *out_uv, uint32_t *out_instance, uint32_t *out_tri)
{
    return trace_rays_impl(ctx, scene, n, ori_xyz, dir_xyz, ray_epsilon, out_hit, out_dst, out_uv, out_instance, out_tri, nullptr);
}

int lupin_hip_trace_rays_wide(LupinContext *ctx, const LupinScene *scene, uint32_t n, const float *ori_xyz, const float *dir_xyz,
                              float ray_epsilon, uint32_t *out_hit, float *out_dst, float *out_uv, uint32_t *out_instance, uint32_t *out_tri,
                              uint32_t *out_needs_retrace)
{
    if (!out_needs_retrace) return fail(LUPIN_ERR_INVALID_ARGUMENT, "null argument");
    return trace_rays_impl(ctx, scene, n, ori_xyz, dir_xyz, ray_epsilon, out_hit, out_dst, out_uv, out_instance, out_tri, out_needs_retrace);
}

int lupin_hip_detmath_probe(LupinContext *ctx, int fn, uint32_t n, const float *x, const float *y, float *out)
{
    CTX_ALIVE_TRY(ctx);
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
    return lupin_internal_tiles_copy(ctx, tex, packed, tile_size, rank, world, 0, unpack ? 1 : 0);
}
int lupin_hip_pack_tiles(LupinContext *ctx, const LupinTexture *tex, uint32_t tile_size, uint32_t rank, uint32_t world, void *device_dst, uint64_t *out_pixels)
{
    CTX_ALIVE_TRY(ctx);
    int rc = pack_common(ctx, tex, tile_size, rank, world, device_dst, 0);
    if (rc == LUPIN_OK && out_pixels) *out_pixels = lupin_hip_packed_tile_pixels(tex->width, tex->height, tile_size, rank, world);
    return rc;
}
int lupin_hip_unpack_tiles(LupinContext *ctx, LupinTexture *tex, uint32_t tile_size, uint32_t rank, uint32_t world, const void *device_src)
{
    CTX_ALIVE_TRY(ctx);
    return pack_common(ctx, tex, tile_size, rank, world, const_cast<void *>(device_src), 1);
}

int lupin_hip_unpack_gathered_tiles(LupinContext *ctx, LupinTexture *tex, uint32_t tile_size, uint32_t rank, uint32_t world,
                                    const void *device_gathered, uint64_t capacity_pixels)
{
    CTX_ALIVE_TRY(ctx);
    if (!ctx || !tex || !device_gathered || tile_size == 0 || world == 0 || rank >= world) return fail(LUPIN_ERR_INVALID_ARGUMENT, "bad unpack arguments");
    return lupin_internal_tiles_copy(ctx, tex, const_cast<void *>(device_gathered), tile_size, rank, world, capacity_pixels, 2);
}

int lupin_hip_tonemap_and_fit_aspect(LupinContext *ctx, const LupinTexture *src, uint8_t *dst_rgba8, uint32_t dst_width, uint32_t dst_height,
                                     const LupinTonemapDesc *desc)
{
    CTX_ALIVE_TRY(ctx);
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
