// gather_probe.hip -- what limits random 64-byte record fetches (the tracer's BVH node visits) on MI355X?
//   gather_probe <table MiB> <record bytes: 64|128|256> <stream: none|plain|nt> [stream MiB]
// Random records from a table of the given size; optionally a second stream (own HIP stream) copies a large buffer at the
// same time with plain or non-temporal loads/stores -- the path-state traffic of the wavefront stages -- to see whether it
// evicts the table from the Infinity Cache.  Prints the gather's GB/s.
//   hipcc --offload-arch=gfx950 -O3 tools/calib/gather_probe.hip -o tools/calib/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>

__device__ __forceinline__ uint32_t hash_u32(uint32_t x)
{
    x ^= x >> 17; x *= 0xed5ad4bbu; x ^= x >> 11; x *= 0xac4c1b51u; x ^= x >> 15; x *= 0x31848babu; x ^= x >> 14;
    return x;
}

template <int WORDS>   // 16-byte words per record
__global__ void k_gather(const float4 *t, size_t records, uint32_t per_thread, float *out)
{
    float acc = 0.0f;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    #pragma unroll 2
    for (uint32_t k = 0; k < per_thread; k++)
    {
        const size_t r = (size_t)(((uint64_t)hash_u32(gid * 7919u + k * 104729u + 1u) * records) >> 32);
        const float4 *p = t + r * WORDS;
        #pragma unroll
        for (int w = 0; w < WORDS; w++) { const float4 v = p[w]; acc += v.x + v.w; }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <bool NT>
__global__ void k_stream_copy(const float4 *src, float4 *dst, size_t n, int reps)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int r = 0; r < reps; r++)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        {
            if (NT) { float4 v; v.x = __builtin_nontemporal_load(&src[i].x); v.y = __builtin_nontemporal_load(&src[i].y); v.z = __builtin_nontemporal_load(&src[i].z); v.w = __builtin_nontemporal_load(&src[i].w);
                      __builtin_nontemporal_store(v.x, &dst[i].x); __builtin_nontemporal_store(v.y, &dst[i].y); __builtin_nontemporal_store(v.z, &dst[i].z); __builtin_nontemporal_store(v.w, &dst[i].w); }
            else dst[i] = src[i];
        }
}

int main(int argc, char **argv)
{
    const size_t table_mib = argc > 1 ? atol(argv[1]) : 320;
    const int rec = argc > 2 ? atoi(argv[2]) : 64;
    const char *stream = argc > 3 ? argv[3] : "none";
    const size_t stream_mib = argc > 4 ? atol(argv[4]) : 2048;
    const size_t bytes = table_mib << 20;
    float4 *t = nullptr, *sa = nullptr, *sb = nullptr; float *out = nullptr;
    hipMalloc((void **)&t, bytes); hipMalloc((void **)&out, 64);
    hipMemset(t, 0x3C, bytes);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const bool streaming = strcmp(stream, "none") != 0;
    if (streaming) { hipMalloc((void **)&sa, stream_mib << 20); hipMalloc((void **)&sb, stream_mib << 20); hipMemset(sa, 1, stream_mib << 20); }
    hipDeviceSynchronize();
    const uint32_t blocks = 256 * 16, threads = 256, per_thread = 64;
    const double req = (double)blocks * threads * per_thread * rec;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
        if (rec == 64) hipLaunchKernelGGL(k_gather<4>, dim3(blocks), dim3(threads), 0, s1, (const float4 *)t, bytes / 64, per_thread, out);
        else if (rec == 128) hipLaunchKernelGGL(k_gather<8>, dim3(blocks), dim3(threads), 0, s1, (const float4 *)t, bytes / 128, per_thread, out);
        else hipLaunchKernelGGL(k_gather<16>, dim3(blocks), dim3(threads), 0, s1, (const float4 *)t, bytes / 256, per_thread, out);
    };
    launch();   // warm: table into the caches
    hipStreamSynchronize(s1);
    if (streaming)
    {
        const size_t n = (stream_mib << 20) / 16;
        if (!strcmp(stream, "nt")) hipLaunchKernelGGL(k_stream_copy<true>, dim3(256 * 4), dim3(256), 0, s2, (const float4 *)sa, sb, n, 64);
        else hipLaunchKernelGGL(k_stream_copy<false>, dim3(256 * 4), dim3(256), 0, s2, (const float4 *)sa, sb, n, 64);
    }
    float best = 1e30f, total = 0;
    for (int it = 0; it < 5; it++)
    {
        hipEventRecord(e0, s1); launch(); hipEventRecord(e1, s1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; total += ms;
    }
    hipDeviceSynchronize();
    printf("{\"table_MiB\": %zu, \"record_bytes\": %d, \"stream\": \"%s\", \"requested_bytes\": %.0f, \"best_ms\": %.3f, \"mean_ms\": %.3f, \"GBps_best\": %.1f, \"Grecords_per_s\": %.2f}\n",
           table_mib, rec, stream, req, best, total / 5, req / best / 1e6, req / rec / best / 1e6);
    return 0;
}
