// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950 (wave64).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_t __attribute__((ext_vector_type(2)));
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters, float s)
{
    float a[8]; float2_t p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = float2_t{a[i], a[i] + 1.f}; }
    float2_t s2{s, s};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = __builtin_fmaf(a[i], s, 0.5f);
                else p[i] = __builtin_elementwise_fma(p[i], s2, float2_t{0.5f, 0.5f});
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += (MODE == 0) ? a[i] : (p[i].x + p[i].y);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
int main()
{
    float *out; hipMalloc(&out, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int blocks_per_cu : {1, 2, 4, 8}) {
        int grid = 256 * blocks_per_cu;
        for (int mode = 0; mode < 2; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters, 0.999f);
                else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters, 0.999f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double insts = (double)grid * 4 * iters * 32;           // wave-instructions
            double per_simd_cycles = ms * 1e-3 * 2.4e9 / (insts / 1024.0);
            double flops = insts * 64 * 2 * (mode ? 2 : 1);
            printf("waves/SIMD %d %s: %.3f ms  %.2f cycles per wave-instr per SIMD (at 2.4GHz)  %.1f TFLOP/s\n", blocks_per_cu,
                   mode ? "v_pk_fma_f32" : "v_fma_f32   ", ms, per_simd_cycles, flops / ms / 1e9);
        }
    }
    return 0;
}
