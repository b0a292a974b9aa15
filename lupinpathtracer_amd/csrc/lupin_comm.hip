// lupin_comm.hip -- the one exchange step of tile-sharded rendering behind the C ABI: RCCL communicators and the
// framebuffer gather (pack -> ncclAllGather over xGMI -> unpack) on the context's primary stream.
//
// The reference is single-device ("Distributed communication backend: absent"); its tile mathematics
// (renderer.rs:807-829) is what the shards are made of.  librccl (573 MB of code objects) is loaded lazily with dlopen
// the first time a communicator is asked for, so single-GPU hosts never map it and liblupin_hip.so has no link-time
// dependency on it.

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "lupin_internal.hpp"
#include "../../include/lupin_tiles.h"

namespace {

struct Rccl
{
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl *rccl()
{
    static Rccl r;
    if (r.handle || !r.error.empty()) return &r;
    // LUPIN_RCCL_LIB names THE library to use (no fallback: a host that points at a build wants that build or an error);
    // without it the usual sonames are tried in order.
    const char *forced = getenv("LUPIN_RCCL_LIB");
    const char *defaults[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string why;
    auto try_load = [&](const char *n) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.handle)
        {
            const char *e = dlerror();   // one call: dlerror() clears the message it returns
            if (!why.empty()) why += "; ";
            why += e ? e : (std::string(n) + ": unknown dlopen error");
        }
        return r.handle != nullptr;
    };
    if (forced && *forced) try_load(forced);
    else for (const char *n : defaults) if (try_load(n)) break;
    if (!r.handle) { r.error = "librccl could not be loaded: " + why; return &r; }
    bool ok = true;
    auto sym = [&](const char *n) { void *p = dlsym(r.handle, n); if (!p) { ok = false; r.error = std::string("librccl lacks ") + n; } return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(r.handle); r.handle = nullptr; }
    return &r;
}

}  // namespace

struct LupinComm
{
    int device = 0;            // the context's device ordinal (the communicator may outlive the context)
    LupinContext *ctx = nullptr;
    ncclComm_t comm = nullptr;
    bool owns_comm = true;
    uint32_t rank = 0, world = 1;
    // staging for the gather: [send: capacity px][recv: world * capacity px], 8 B per pixel; grown on demand
    uint2 *send = nullptr, *recv = nullptr;
    uint64_t capacity_px = 0;
    double *scratch = nullptr;      // device words for barrier / all-reduce
    uint32_t scratch_words = 0;
};

#define COMM_FAIL(code, msg) lupin_internal_fail((code), (msg))
#define HIP_TRY_C(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return lupin_internal_fail(LUPIN_ERR_HIP, (std::string(#expr) + ": " + hipGetErrorString(e__)).c_str()); } while (0)
#define NCCL_TRY(expr) do { ncclResult_t r__ = (expr); if (r__ != ncclSuccess) return lupin_internal_fail(LUPIN_ERR_RCCL, (std::string(#expr) + ": " + (R->GetErrorString ? R->GetErrorString(r__) : "rccl error")).c_str()); } while (0)

static int need_rccl(Rccl **out)
{
    Rccl *R = rccl();
    if (!R->handle) return COMM_FAIL(LUPIN_ERR_RCCL, R->error.c_str());
    *out = R;
    return LUPIN_OK;
}

static int ensure_scratch(LupinComm *c, uint32_t words)
{
    if (words <= c->scratch_words) return LUPIN_OK;
    HIP_TRY_C(hipSetDevice(lupin_internal_ctx_device(c->ctx)));
    if (c->scratch) { HIP_TRY_C(hipStreamSynchronize(lupin_internal_ctx_stream(c->ctx))); hipFree(c->scratch); c->scratch = nullptr; }
    const uint32_t n = words < 16 ? 16 : words;
    HIP_TRY_C(hipMalloc((void **)&c->scratch, (size_t)n * sizeof(double)));
    c->scratch_words = n;
    return LUPIN_OK;
}

// capacity = the largest payload any rank packs (all-gather needs equal counts)
static uint64_t gather_capacity(uint32_t w, uint32_t h, uint32_t tile_size, uint32_t world)
{
    uint64_t cap = 0;
    for (uint32_t r = 0; r < world; r++)
    {
        const uint64_t p = lupin_hip_packed_tile_pixels(w, h, tile_size, r, world);
        cap = p > cap ? p : cap;
    }
    return cap;
}

static int ensure_staging(LupinComm *c, uint64_t capacity_px)
{
    if (capacity_px <= c->capacity_px) return LUPIN_OK;
    HIP_TRY_C(hipSetDevice(lupin_internal_ctx_device(c->ctx)));
    HIP_TRY_C(hipStreamSynchronize(lupin_internal_ctx_stream(c->ctx)));
    if (c->send) hipFree(c->send);
    if (c->recv) hipFree(c->recv);
    c->send = c->recv = nullptr; c->capacity_px = 0;
    HIP_TRY_C(hipMalloc((void **)&c->send, capacity_px * 8));
    HIP_TRY_C(hipMalloc((void **)&c->recv, capacity_px * 8 * c->world));
    HIP_TRY_C(hipMemsetAsync(c->send, 0, capacity_px * 8, lupin_internal_ctx_stream(c->ctx)));   // the padding behind a short payload is gathered too
    c->capacity_px = capacity_px;
    return LUPIN_OK;
}

extern "C" {

int lupin_hip_comm_get_unique_id(uint8_t *out_id)
{
    if (!out_id) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "out_id is null");
    Rccl *R; int rc = need_rccl(&R); if (rc) return rc;
    ncclUniqueId id;
    NCCL_TRY(R->GetUniqueId(&id));
    static_assert(sizeof(id) == LUPIN_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(out_id, &id, sizeof(id));
    return LUPIN_OK;
}

int lupin_hip_comm_init_rank(LupinContext *ctx, const uint8_t *id_bytes, uint32_t rank, uint32_t world, LupinComm **out_comm)
{
    if (!ctx || !id_bytes || !out_comm || world == 0 || rank >= world) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "bad communicator arguments");
    if (!lupin_internal_ctx_alive(ctx)) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "the context of this object has been destroyed");
    Rccl *R; int rc = need_rccl(&R); if (rc) return rc;
    HIP_TRY_C(hipSetDevice(lupin_internal_ctx_device(ctx)));
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t comm = nullptr;
    NCCL_TRY(R->CommInitRank(&comm, (int)world, id, (int)rank));
    LupinComm *c = new LupinComm();
    c->ctx = ctx; c->device = lupin_internal_ctx_device(ctx); c->comm = comm; c->rank = rank; c->world = world;
    *out_comm = c;
    return LUPIN_OK;
}

int lupin_hip_comm_from_nccl(LupinContext *ctx, void *nccl_comm, uint32_t rank, uint32_t world, LupinComm **out_comm)
{
    if (!ctx || !nccl_comm || !out_comm || world == 0 || rank >= world) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "bad communicator arguments");
    if (!lupin_internal_ctx_alive(ctx)) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "the context of this object has been destroyed");
    Rccl *R; int rc = need_rccl(&R); if (rc) return rc;
    LupinComm *c = new LupinComm();
    c->ctx = ctx; c->device = lupin_internal_ctx_device(ctx); c->comm = (ncclComm_t)nccl_comm; c->owns_comm = false; c->rank = rank; c->world = world;
    *out_comm = c;
    return LUPIN_OK;
}

int lupin_hip_comm_init_all(LupinContext *const *ctxs, uint32_t n, LupinComm **out_comms)
{
    if (!ctxs || !out_comms || n == 0) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "bad communicator arguments");
    Rccl *R; int rc = need_rccl(&R); if (rc) return rc;
    std::vector<int> devs(n);
    for (uint32_t i = 0; i < n; i++)
    {
        if (!ctxs[i]) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "null context");
        if (!lupin_internal_ctx_alive(ctxs[i])) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "the context of this object has been destroyed");
        devs[i] = lupin_internal_ctx_device(ctxs[i]);
        for (uint32_t k = 0; k < i; k++) if (devs[k] == devs[i]) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "one context per device");
    }
    std::vector<ncclComm_t> comms(n, nullptr);
    NCCL_TRY(R->CommInitAll(comms.data(), (int)n, devs.data()));
    for (uint32_t i = 0; i < n; i++)
    {
        LupinComm *c = new LupinComm();
        c->ctx = ctxs[i]; c->device = devs[i]; c->comm = comms[i]; c->rank = i; c->world = n;
        out_comms[i] = c;
    }
    return LUPIN_OK;
}

void lupin_hip_comm_destroy(LupinComm *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (lupin_internal_ctx_alive(c->ctx)) lupin_internal_sync_all(c->ctx);   // a destroyed context has drained its streams already
    Rccl *R = rccl();
    if (c->owns_comm && c->comm && R->handle) R->CommDestroy(c->comm);
    if (c->send) hipFree(c->send);
    if (c->recv) hipFree(c->recv);
    if (c->scratch) hipFree(c->scratch);
    delete c;
}

uint32_t lupin_hip_comm_rank(const LupinComm *c) { return c ? c->rank : 0; }
uint32_t lupin_hip_comm_world(const LupinComm *c) { return c ? c->world : 0; }

// pack (own tiles) -> all-gather -> unpack (everyone else's tiles), all enqueued on the context's primary stream
static int gather_enqueue(Rccl *R, LupinComm *c, LupinTexture *tex, uint32_t tile_size, int stage)
{
    LupinContext *ctx = c->ctx;
    hipStream_t st = lupin_internal_ctx_stream(ctx);
    if (stage == 0)
    {
        if (tex->ctx != ctx) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "texture belongs to another context");
        int rc = ensure_staging(c, gather_capacity(tex->width, tex->height, tile_size, c->world));
        if (rc) return rc;
        return lupin_internal_tiles_copy(ctx, tex, c->send, tile_size, c->rank, c->world, 0, 0);
    }
    if (stage == 1)
    {
        HIP_TRY_C(hipSetDevice(lupin_internal_ctx_device(ctx)));
        NCCL_TRY(R->AllGather(c->send, c->recv, (size_t)c->capacity_px * 8, ncclUint8, c->comm, st));
        return LUPIN_OK;
    }
    return lupin_internal_tiles_copy(ctx, tex, c->recv, tile_size, c->rank, c->world, c->capacity_px, 2);
}

int lupin_hip_gather_framebuffer(LupinComm *c, LupinTexture *tex, uint32_t tile_size)
{
    if (!c || !tex || tile_size == 0) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "bad gather arguments");
    if (!lupin_internal_ctx_alive(c->ctx)) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "the context of this object has been destroyed");
    Rccl *R; int rc = need_rccl(&R); if (rc) return rc;
    for (int stage = 0; stage < 3; stage++)
        if ((rc = gather_enqueue(R, c, tex, tile_size, stage))) return rc;
    return LUPIN_OK;
}

// Readback gather: only `root` receives.  Every other rank packs its tiles and sends the exact payload; the root posts one
// receive per peer into that peer's slice of the staging buffer (the all-gather's layout) and scatters them with the same
// one-launch unpack.  One RCCL group per call, so the sends and receives of a rank cannot order-deadlock.  Ranks other than
// the root keep a framebuffer that holds their own tiles only.
int lupin_hip_gather_framebuffer_to(LupinComm *c, LupinTexture *tex, uint32_t tile_size, uint32_t root)
{
    if (!c || !tex || tile_size == 0 || root >= c->world) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "bad gather arguments");
    if (!lupin_internal_ctx_alive(c->ctx)) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "the context of this object has been destroyed");
    Rccl *R; int rc = need_rccl(&R); if (rc) return rc;
    if ((rc = gather_enqueue(R, c, tex, tile_size, 0))) return rc;   // staging + pack of the own tiles
    if (c->world == 1) return LUPIN_OK;
    LupinContext *ctx = c->ctx;
    hipStream_t st = lupin_internal_ctx_stream(ctx);
    HIP_TRY_C(hipSetDevice(lupin_internal_ctx_device(ctx)));
    NCCL_TRY(R->GroupStart());
    if (c->rank == root)
    {
        for (uint32_t r = 0; r < c->world; r++)
        {
            if (r == root) continue;
            const uint64_t px = lupin_hip_packed_tile_pixels(tex->width, tex->height, tile_size, r, c->world);
            if (px == 0) continue;
            ncclResult_t e = R->Recv(c->recv + (size_t)r * c->capacity_px, (size_t)px * 8, ncclUint8, (int)r, c->comm, st);
            if (e != ncclSuccess) { R->GroupEnd(); return COMM_FAIL(LUPIN_ERR_RCCL, R->GetErrorString ? R->GetErrorString(e) : "ncclRecv"); }
        }
    }
    else
    {
        const uint64_t px = lupin_hip_packed_tile_pixels(tex->width, tex->height, tile_size, c->rank, c->world);
        if (px)
        {
            ncclResult_t e = R->Send(c->send, (size_t)px * 8, ncclUint8, (int)root, c->comm, st);
            if (e != ncclSuccess) { R->GroupEnd(); return COMM_FAIL(LUPIN_ERR_RCCL, R->GetErrorString ? R->GetErrorString(e) : "ncclSend"); }
        }
    }
    NCCL_TRY(R->GroupEnd());
    if (c->rank == root) return gather_enqueue(R, c, tex, tile_size, 2);
    return LUPIN_OK;
}

// one process driving n contexts (lupin_hip_comm_init_all): the n all-gathers form one RCCL group
int lupin_hip_gather_framebuffer_all(LupinComm *const *comms, LupinTexture *const *texs, uint32_t n, uint32_t tile_size)
{
    if (!comms || !texs || n == 0 || tile_size == 0) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "bad gather arguments");
    Rccl *R; int rc = need_rccl(&R); if (rc) return rc;
    for (uint32_t i = 0; i < n; i++)
    {
        if (!comms[i] || !texs[i]) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "null communicator / texture");
        if (!lupin_internal_ctx_alive(comms[i]->ctx)) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "the context of this object has been destroyed");
        if (texs[i]->width != texs[0]->width || texs[i]->height != texs[0]->height) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "framebuffers differ in size");
        if ((rc = gather_enqueue(R, comms[i], texs[i], tile_size, 0))) return rc;
    }
    NCCL_TRY(R->GroupStart());
    for (uint32_t i = 0; i < n; i++)
        if ((rc = gather_enqueue(R, comms[i], texs[i], tile_size, 1))) { R->GroupEnd(); return rc; }
    NCCL_TRY(R->GroupEnd());
    for (uint32_t i = 0; i < n; i++)
        if ((rc = gather_enqueue(R, comms[i], texs[i], tile_size, 2))) return rc;
    return LUPIN_OK;
}

// sum (op 0) / max (op 1) of n doubles over the ranks; synchronous: every frame this rank enqueued has finished when it returns
int lupin_hip_comm_allreduce_f64(LupinComm *c, double *inout, uint32_t n, uint32_t op)
{
    if (!c || !inout || n == 0 || op > 1) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "bad all-reduce arguments");
    if (!lupin_internal_ctx_alive(c->ctx)) return COMM_FAIL(LUPIN_ERR_INVALID_ARGUMENT, "the context of this object has been destroyed");
    Rccl *R; int rc = need_rccl(&R); if (rc) return rc;
    if ((rc = ensure_scratch(c, n))) return rc;
    if ((rc = lupin_internal_sync_all(c->ctx))) return rc;
    hipStream_t st = lupin_internal_ctx_stream(c->ctx);
    HIP_TRY_C(hipMemcpyAsync(c->scratch, inout, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    NCCL_TRY(R->AllReduce(c->scratch, c->scratch, n, ncclFloat64, op == 0 ? ncclSum : ncclMax, c->comm, st));
    HIP_TRY_C(hipMemcpyAsync(inout, c->scratch, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY_C(hipStreamSynchronize(st));
    return LUPIN_OK;
}

int lupin_hip_comm_barrier(LupinComm *c)
{
    double one = 1.0;
    return lupin_hip_comm_allreduce_f64(c, &one, 1, 0);
}

}  // extern "C"
