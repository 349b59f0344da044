// dp_rates.hip -- issue cost of the fp64 instructions of the integral-image scans on gfx950: cycles per wave-instruction for
// v_cvt_f64_f32, v_cvt_f32_f64, v_add_f64 (independent streams and one dependent chain), at 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o dp_rates tools/micro/dp_rates.hip && ./dp_rates
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(float *out, long long *cyc, float seed)
{
    float x[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { x[i] = seed + threadIdx.x * 0.001f + i; d[i] = (double)x[i]; }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(x[i])); }
            if (MODE == 1) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(x[i]) : "v"(d[i])); }
            if (MODE == 2) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7])); }          // 8 independent chains
            if (MODE == 3) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[0]) : "v"(d[1])); }                    // one dependent chain
            if (MODE == 4) { asm volatile("v_cvt_f64_f32 %0, %2\n\tv_add_f64 %1, %1, %0\n\tv_cvt_f32_f64 %2, %1" : "=&v"(d[1]), "+v"(d[0]), "+v"(x[i])); } // the scan step, dependent
            if (MODE == 5) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 7])); }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char *name, int n_per)
{
    float *out; long long *cyc;
    hipMalloc(&out, 4 * 2048 * 256); hipMalloc(&cyc, 8 * 256);
    for (int waves = 1; waves <= 4; waves *= 2) { // waves per SIMD (4 SIMDs per CU)
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * 4 * waves), 0, 0, out, cyc, 1.0f);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * 4 * waves), 0, 0, out, cyc, 1.0f);
        long long h[256];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i];
        avg /= 256;
        printf("%-28s %d wave(s)/SIMD: %.1f cycles per wave-instruction (%.1f per SIMD-instruction)\n", name, waves, avg / (256.0 * 8 * n_per), avg / (256.0 * 8 * n_per) / waves);
    }
}
int main()
{
    run<0>("v_cvt_f64_f32", 1); run<1>("v_cvt_f32_f64", 1); run<2>("v_add_f64 independent", 1); run<3>("v_add_f64 dependent", 1);
    run<4>("cvt+add+cvt (scan step)", 3); run<5>("v_add_f32 independent", 1);
    return 0;
}
