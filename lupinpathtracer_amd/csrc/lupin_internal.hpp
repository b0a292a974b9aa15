// Shared between the translation units of liblupin_hip.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/lupin_hip.h"

int lupin_internal_fail(int code, const char *msg);          // records the message lupin_hip_last_error() returns
int lupin_internal_ctx_device(const LupinContext *ctx);
hipStream_t lupin_internal_ctx_stream(const LupinContext *ctx);   // the primary stream
