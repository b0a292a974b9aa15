// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE on gfx950 for THIS build's access patterns (MI355X_MICROARCH.md, HBM:
// "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Three kernels over a 2 GiB table (far beyond the L2s):
//   k_stream     16 B per lane, coalesced            (the guide's case: expect FETCH_SIZE x 1024 = bytes / 2)
//   k_gather64   one random 64-byte record per lane  (a WideNode / InstanceDev fetch of the tracer)
//   k_gather48   one random 48-byte record per lane  (a TriVerts fetch; records straddle 64-byte lines)
// Run each under `rocprofv3 --pmc FETCH_SIZE` and compare with the requested bytes this program prints.
//   hipcc --offload-arch=gfx950 -O3 tools/calib/fetch_calib.hip -o tools/calib/fetch_calib
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

__global__ void k_stream(const float4 *t, size_t n, float *out)
{
    float acc = 0.0f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const float4 v = t[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ void k_gather64(const float4 *t, size_t records, uint32_t per_thread, float *out)
{
    float acc = 0.0f;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = 0; k < per_thread; k++)
    {
        const size_t r = (size_t)(((uint64_t)hash_u32(gid * 7919u + k * 104729u + 1u) * records) >> 32);
        const float4 *p = t + r * 4;
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ void k_gather48(const float4 *t, size_t records, uint32_t per_thread, float *out)
{
    float acc = 0.0f;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = 0; k < per_thread; k++)
    {
        const size_t r = (size_t)(((uint64_t)hash_u32(gid * 7919u + k * 104729u + 1u) * records) >> 32);
        const float4 *p = t + r * 3;
        const float4 a = p[0], b = p[1], c = p[2];
        acc += a.x + b.y + c.z;
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char **argv)
{
    const char *which = argc > 1 ? argv[1] : "all";
    const size_t bytes = (size_t)2 << 30;
    float4 *t = nullptr; float *out = nullptr;
    if (hipMalloc((void **)&t, bytes) != hipSuccess || hipMalloc((void **)&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(t, 0x3C, bytes);
    hipDeviceSynchronize();
    const uint32_t blocks = 256 * 16, threads = 256, per_thread = 32;
    const size_t lanes = (size_t)blocks * threads;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch, double req_bytes) {
        if (strcmp(which, "all") && strcmp(which, name)) return;
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("{\"kernel\": \"%s\", \"requested_bytes\": %.0f, \"ms\": %.3f, \"GBps\": %.1f}\n", name, req_bytes, ms, req_bytes / ms / 1e6);
    };
    run("k_stream", [&] { hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(threads), 0, 0, (const float4 *)t, bytes / 16, out); }, (double)bytes);
    run("k_gather64", [&] { hipLaunchKernelGGL(k_gather64, dim3(blocks), dim3(threads), 0, 0, (const float4 *)t, bytes / 64, per_thread, out); }, (double)lanes * per_thread * 64.0);
    run("k_gather48", [&] { hipLaunchKernelGGL(k_gather48, dim3(blocks), dim3(threads), 0, 0, (const float4 *)t, bytes / 48, per_thread, out); }, (double)lanes * per_thread * 48.0);
    hipFree(t); hipFree(out);
    return 0;
}
