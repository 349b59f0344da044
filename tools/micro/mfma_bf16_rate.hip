// Sustained v_mfma_f32_32x32x16_bf16 rate under chip-wide load (random operands), and the clock the chip holds.
// hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_bf16_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(512) void k(const uint4 *in, float *out, int iters, long long *clk)
{
    const int tid = threadIdx.x;
    bf16x8 a[4], b[8];
    for (int i = 0; i < 4; ++i) { uint4 x = in[(tid + 64 * i) & 1023]; a[i] = *reinterpret_cast<bf16x8 *>(&x); }
    for (int i = 0; i < 8; ++i) { uint4 x = in[(tid * 3 + 64 * i) & 1023]; b[i] = *reinterpret_cast<bf16x8 *>(&x); }
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 48; ++s) acc[s % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s & 3], b[s & 7], acc[s % CHAINS], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float r = 0.f;
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) r += acc[c][i];
    out[blockIdx.x * 512 + tid] = r;
    if (tid == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int CHAINS> void run(int waves_per_simd, const uint4 *in, float *out, long long *clk)
{
    const int iters = 2000, threads = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAINS>, dim3(256), dim3(threads), 0, 0, in, out, 10, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(256), dim3(threads), 0, 0, in, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double mfma = 256.0 * threads / 64 * iters * 48, flops = mfma * 32 * 32 * 16 * 2;
    printf("chains %d, %d waves/SIMD: %.3f ms, %.0f TFLOP/s bf16, shader clock %.2f GHz (clock64 %lld / wall %lld @100MHz), %.1f cycles per MFMA per SIMD\n",
           CHAINS, waves_per_simd, ms, flops / ms / 1e9, (double)h[0] / ((double)h[1] * 10.0) , h[0], h[1],
           (double)h[0] / (iters * 48.0 * waves_per_simd));
}

int main()
{
    std::vector<unsigned short> h(8192);
    for (auto &x : h) x = (unsigned short)(0x3f80 + (rand() & 0x7f) + ((rand() & 1) << 15)); // random bf16 in +-[1,2)
    uint4 *in; float *out; long long *clk;
    hipMalloc(&in, 16384); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 16);
    hipMemcpy(in, h.data(), 16384, hipMemcpyHostToDevice);
    run<1>(1, in, out, clk); run<1>(2, in, out, clk); run<2>(1, in, out, clk); run<2>(2, in, out, clk);
    return 0;
}
