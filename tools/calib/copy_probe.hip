// copy_probe.hip -- which device-to-device copy kernel shape reaches the HBM rate the guide quotes (6.29 TB/s float4 copy)?
//   hipcc --offload-arch=gfx950 -O3 tools/calib/copy_probe.hip -o tools/calib/copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride)
    {
        float4 v[U];
        #pragma unroll
        for (int k = 0; k < U; k++)
        {
            if (NT) { const float *p = (const float *)(src + i + k * stride); v[k] = make_float4(__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1), __builtin_nontemporal_load(p + 2), __builtin_nontemporal_load(p + 3)); }
            else v[k] = src[i + k * stride];
        }
        #pragma unroll
        for (int k = 0; k < U; k++)
        {
            if (NT) { float *p = (float *)(dst + i + k * stride); __builtin_nontemporal_store(v[k].x, p); __builtin_nontemporal_store(v[k].y, p + 1); __builtin_nontemporal_store(v[k].z, p + 2); __builtin_nontemporal_store(v[k].w, p + 3); }
            else dst[i + k * stride] = v[k];
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
int main()
{
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    float4 *a, *b; hipMalloc((void **)&a, bytes); hipMalloc((void **)&b, bytes); hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < 6; r++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("{\"copy\": \"%s\", \"GBps\": %.1f}\n", name, 2.0 * bytes * 6 / ms / 1e6);
    };
    for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 64})
    {
        char nm[64];
        snprintf(nm, 64, "u1 b%d", blocks); run(nm, [&] { hipLaunchKernelGGL((k_copy<1, false>), dim3(blocks), dim3(256), 0, 0, a, b, n); });
        snprintf(nm, 64, "u4 b%d", blocks); run(nm, [&] { hipLaunchKernelGGL((k_copy<4, false>), dim3(blocks), dim3(256), 0, 0, a, b, n); });
        snprintf(nm, 64, "u8 b%d", blocks); run(nm, [&] { hipLaunchKernelGGL((k_copy<8, false>), dim3(blocks), dim3(256), 0, 0, a, b, n); });
        snprintf(nm, 64, "u4nt b%d", blocks); run(nm, [&] { hipLaunchKernelGGL((k_copy<4, true>), dim3(blocks), dim3(256), 0, 0, a, b, n); });
    }
    run("one block per 256 float4 (grid = n/256)", [&] { hipLaunchKernelGGL((k_copy<1, false>), dim3((unsigned)(n / 256)), dim3(256), 0, 0, a, b, n); });
    run("hipMemcpyDtoD", [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
