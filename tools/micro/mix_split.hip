// mix_split.hip -- v_fma_mixlo/hi_f16 as the second piece of the fp16 operand split (vfa_split.h: split_f16x4): lo = RN_f16(x - float(hi)) in ONE
// instruction per value, compared bit for bit with the three-instruction form on 1.7e7 values.  hipcc --offload-arch=gfx950 -O3 -o mix_split tools/micro/mix_split.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_ref(float x0, float x1, float x2, float x3, uint2 &hi, uint2 &lo)
{
    const f32x4v x = {x0, x1, x2, x3};
    const f16x4 h = __builtin_convertvector(x, f16x4);
    const f32x4v r = x - __builtin_convertvector(h, f32x4v);
    const f16x4 l = __builtin_convertvector(r, f16x4);
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}
__device__ __forceinline__ void split_mix(float x0, float x1, float x2, float x3, uint2 &hi, uint2 &lo)
{
    const f32x4v x = {x0, x1, x2, x3};
    const f16x4 h = __builtin_convertvector(x, f16x4);
    hi = __builtin_bit_cast(uint2, h);
    unsigned l0 = 0, l1 = 0;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(l0) : "v"(hi.x), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l0) : "v"(hi.x), "v"(x1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(l1) : "v"(hi.y), "v"(x2));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l1) : "v"(hi.y), "v"(x3));
    lo = make_uint2(l0, l1);
}
template <int SAT> __global__ void k(const float *in, uint4 *ref, uint4 *mix, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (SAT) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
    uint2 h, l;
    split_ref(in[4 * i], in[4 * i + 1], in[4 * i + 2], in[4 * i + 3], h, l);
    ref[i] = make_uint4(h.x, h.y, l.x, l.y);
    split_mix(in[4 * i], in[4 * i + 1], in[4 * i + 2], in[4 * i + 3], h, l);
    mix[i] = make_uint4(h.x, h.y, l.x, l.y);
}
int main()
{
    const int n = 1 << 22;
    float *hin = (float *)malloc((size_t)n * 16);
    unsigned *u = (unsigned *)hin;
    srand(1);
    for (size_t i = 0; i < (size_t)n * 4; ++i) {
        const int kind = rand() % 10;
        if (kind < 6) hin[i] = ((float)rand() / RAND_MAX - 0.5f) * 40000.0f * (float)(rand() % 3 == 0 ? 1e-3 : 1.0);
        else if (kind < 8) u[i] = ((unsigned)rand() << 16) ^ (unsigned)rand();           // any bit pattern (NaN, Inf, denormals, huge)
        else if (kind == 8) hin[i] = ldexpf((float)rand() / RAND_MAX, -(rand() % 40));  // tiny
        else hin[i] = (rand() & 1 ? 65504.0f : -65520.0f) * (1.0f + (float)(rand() % 100) * 1e-3f); // around the fp16 limit
    }
    float *din; uint4 *dr, *dm;
    hipMalloc(&din, (size_t)n * 16); hipMalloc(&dr, (size_t)n * 16); hipMalloc(&dm, (size_t)n * 16);
    hipMemcpy(din, hin, (size_t)n * 16, hipMemcpyHostToDevice);
    unsigned *hr = (unsigned *)malloc((size_t)n * 16), *hm = (unsigned *)malloc((size_t)n * 16);
    for (int sat = 0; sat < 2; ++sat) {
        if (sat) hipLaunchKernelGGL(k<1>, dim3(n / 256), dim3(256), 0, 0, din, dr, dm, n);
        else hipLaunchKernelGGL(k<0>, dim3(n / 256), dim3(256), 0, 0, din, dr, dm, n);
        hipMemcpy(hr, dr, (size_t)n * 16, hipMemcpyDeviceToHost); hipMemcpy(hm, dm, (size_t)n * 16, hipMemcpyDeviceToHost);
        size_t bad = 0, nanok = 0;
        for (size_t i = 0; i < (size_t)n * 4; ++i) if (hr[i] != hm[i]) { if (bad < 5) printf("  mismatch word %zu: ref %08x mix %08x (in %08x %08x)\n", i, hr[i], hm[i], u[(i / 4) * 4 + 2 * (i % 2)], u[(i / 4) * 4 + 2 * (i % 2) + 1]); ++bad; }
        printf("MIX FP16_OVFL=%d: %zu mismatching words of %zu\n", sat, bad, (size_t)n * 4);
    }
    return 0;
}
