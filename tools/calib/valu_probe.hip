// tools/calib/valu_probe.hip -- issue cost of the vector instructions the tracer is made of, per wave-instruction, with
// 1 / 2 / 4 waves per SIMD (gfx950).  Independent chains (8 registers in rotation), no memory traffic: what is measured is
// the issue rate of one SIMD's vector ALU as the waves on it share it.
//   hipcc --offload-arch=gfx950 -O2 -o valu_probe tools/calib/valu_probe.hip && ./valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void __launch_bounds__(256) k_probe(float *out, int iters, float seed)
{
    float r[8]; v2f p[8];
    for (int k = 0; k < 8; k++) { r[k] = seed + (float)(threadIdx.x + k); p[k] = (v2f){r[k], r[k] + 1.0f}; }
    const float c = seed * 0.5f; const v2f pc = {c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++)
    {
        #pragma unroll
        for (int u = 0; u < 8; u++)
        {
            if (OP == 0) {
#define X(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[k]) : "v"(c));
                REP8(X)
#undef X
            }
            if (OP == 1) {
#define X(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc));
                REP8(X)
#undef X
            }
            if (OP == 2) {
#define X(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc));
                REP8(X)
#undef X
            }
            if (OP == 3) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[k]) : "v"(c));
                REP8(X)
#undef X
            }
            if (OP == 4) {
#define X(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[k]));
                REP8(X)
#undef X
            }
            if (OP == 5) {
#define X(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[k]) : "v"(c) : "vcc");
                REP8(X)
#undef X
            }
            if (OP == 6) {
#define X(k) asm volatile("v_min_f32 %0, %0, %1" : "+v"(r[k]) : "v"(c));
                REP8(X)
#undef X
            }
            if (OP == 7) {
#define X(k) asm volatile("v_mov_b32 %0, %1" : "+v"(r[k]) : "v"(c));
                REP8(X)
#undef X
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int k = 0; k < 8; k++) s += r[k] + p[k].x + p[k].y;
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
}

template <int OP>
static double run(int waves_per_simd, int iters, float *d_out, int cus)
{
    // a 256-thread block = 4 waves = one wave per SIMD of its CU; waves_per_simd blocks per CU
    const int blocks = cus * waves_per_simd;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_probe<OP><<<blocks, 256>>>(d_out, 16, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_probe<OP><<<blocks, 256>>>(d_out, iters, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return (double)ms * 1e-3;
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double hz = (double)prop.clockRate * 1e3;
    float *d_out; hipMalloc(&d_out, sizeof(float) * 256 * cus * 8);
    const int iters = 20000;   // x 64 instructions (x 2 for the compare + select pair)
    const char *names[8] = {"v_mul_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_fma_f32", "v_rcp_f32", "v_cmp+v_cndmask", "v_min_f32", "v_mov_b32"};
    for (int op = 0; op < 8; op++)
        for (int w : {1, 2, 4})
        {
            double s = 0;
            switch (op) {
            case 0: s = run<0>(w, iters, d_out, cus); break; case 1: s = run<1>(w, iters, d_out, cus); break;
            case 2: s = run<2>(w, iters, d_out, cus); break; case 3: s = run<3>(w, iters, d_out, cus); break;
            case 4: s = run<4>(w, iters, d_out, cus); break; case 5: s = run<5>(w, iters, d_out, cus); break;
            case 6: s = run<6>(w, iters, d_out, cus); break; default: s = run<7>(w, iters, d_out, cus); break; }
            const double insts = (double)iters * 64.0 * (op == 5 ? 2.0 : 1.0);
            // cycles of one SIMD per wave-instruction: elapsed cycles / (instructions per wave * waves on the SIMD)
            printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"seconds\": %.6f, \"simd_cycles_per_wave_instruction\": %.3f}\n",
                   names[op], w, s, s * hz / (insts * w));
        }
    hipFree(d_out);
    return 0;
}
