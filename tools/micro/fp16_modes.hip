// fp16_modes.hip -- what MODE.FP16_OVFL does to fp32 -> fp16 conversions, and whether the fp16 MFMA keeps NaN / Inf (gfx950).
//   hipcc --offload-arch=gfx950 -O2 -o fp16_modes tools/micro/fp16_modes.hip && ./fp16_modes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void cvt(const float *in, unsigned *out, int ovfl)
{
    if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
    const f32x2 x = {in[2 * threadIdx.x], in[2 * threadIdx.x + 1]};
    const f16x2 h = __builtin_convertvector(x, f16x2);
    out[threadIdx.x] = __builtin_bit_cast(unsigned, h);
}
__global__ void mm(const _Float16 *a, const _Float16 *b, float *c, int ovfl)
{
    if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
    f16x8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = a[threadIdx.x * 8 + j]; bv[j] = b[threadIdx.x * 8 + j]; }
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) c[threadIdx.x * 16 + i] = acc[i];
}
int main()
{
    float h_in[8] = {1.0f, 70000.0f, -1e6f, NAN, INFINITY, -INFINITY, 65520.0f, 3e-8f};
    float *d_in; unsigned *d_out;
    hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, 16);
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    for (int ov = 0; ov < 2; ++ov) {
        unsigned h_out[4];
        hipLaunchKernelGGL(cvt, dim3(1), dim3(4), 0, 0, d_in, d_out, ov);
        hipMemcpy(h_out, d_out, 16, hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d:", ov);
        for (int i = 0; i < 8; ++i) printf("  %g -> 0x%04x", h_in[i], (h_out[i / 2] >> (16 * (i & 1))) & 0xffff);
        printf("\n");
    }
    // MFMA: A row 0 has a NaN at k = 0, row 1 an Inf, others 1; B = 1
    _Float16 ha[64 * 8], hb[64 * 8];
    for (int i = 0; i < 64 * 8; ++i) { ha[i] = (_Float16)1.0f; hb[i] = (_Float16)1.0f; }
    // lane l holds row (l & 31), k = 8 (l >> 5) + j
    unsigned short nanb = 0x7e00, infb = 0x7c00;
    memcpy(&ha[0 * 8 + 0], &nanb, 2);
    memcpy(&ha[1 * 8 + 0], &infb, 2);
    _Float16 *da, *db; float *dc;
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dc, 64 * 16 * 4);
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    for (int ov = 0; ov < 2; ++ov) {
        float hc[64 * 16];
        hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, da, db, dc, ov);
        hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
        // element (row i, col j): lane j + 32 * ((i / 4) % 2), register (i % 4) + 4 * (i / 8)
        auto at = [&](int i, int j) { return hc[(j + 32 * ((i / 4) % 2)) * 16 + (i % 4) + 4 * (i / 8)]; };
        printf("MFMA f16, FP16_OVFL=%d: row 0 (NaN operand) -> %g, row 1 (Inf operand) -> %g, row 2 -> %g\n", ov, at(0, 5), at(1, 5), at(2, 5));
    }
    return 0;
}
