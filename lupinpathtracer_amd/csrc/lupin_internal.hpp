// Shared between the translation units of liblupin_hip.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "../../include/lupin_hip.h"

struct LupinTexture
{
    LupinContext *ctx;
    int device;            // the context's device ordinal (valid after the context is gone: a texture may outlive it)
    uint32_t width, height;
    __half *data;          // Rgba16Float, row-major, row 0 = top
    float4 *accum32;       // f32 shadow (LUPIN_ACCUM_F32), allocated by the first frame rendered into it in that mode
    bool accum32_valid;    // the shadow holds the value `data` is the rounded view of
};

int lupin_internal_fail(int code, const char *msg);          // records the message lupin_hip_last_error() returns
bool lupin_internal_ctx_alive(const LupinContext *ctx);       // created by lupin_hip_create_context and not destroyed since
int lupin_internal_ctx_device(const LupinContext *ctx);
hipStream_t lupin_internal_ctx_stream(const LupinContext *ctx);   // the primary stream
void lupin_internal_join_primary(LupinContext *ctx);         // primary stream waits for the frames enqueued so far
int lupin_internal_sync_all(LupinContext *ctx);              // host waits for every lane
int lupin_internal_tiles_copy(LupinContext *ctx, const LupinTexture *tex, void *packed, uint32_t tile_size, uint32_t rank, uint32_t world,
                              uint64_t capacity_px, int mode);   // k_tiles_copy on the primary stream
