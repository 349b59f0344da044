// xy_roles.hip -- role split of the single-layer frame kernel at TWO waves per SIMD and 256 registers each (round 5 study):
//   X (waves 0-3, one per SIMD): the matrix role.  64 output columns each; the fp16 HI plane of its weight slice resident (128
//     registers), the LO plane streamed from L2 every item (128 KB per item and CU), A fragments from the LDS planes one k-step ahead;
//     96 MFMAs per item (3 072 cycles of pipe).
//   Y (waves 4-7, one per SIMD): the pooling role of the NEXT item: 32 boxes x 64 channels per wave = 8 units of 16 taps, the taps of
//     unit u + 1 requested before unit u is consumed (32 taps = 128 registers in flight), reference FMA chains, exact quotient, fp16
//     split, planes double-buffered.
// One barrier per item.  The shipped serial kernel (both roles in every wave, one after the other): ~11 100 cycles per item.
//   MODE 0: both roles   1: X only   2: Y only
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o xy_roles tools/micro/xy_roles.hip && ./xy_roles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWinBytes = 88 * 1024;            // tap window(s): 88 slots of 1 KiB (256 channels fp32)
constexpr int kChunkStride = 32 * 16 + 32;      // A tile: chunk (8 k) x 32 rows x 16 B, padded
constexpr int kPlane = 32 * kChunkStride;       // 32 chunks = 256 k
constexpr int kTile = 2 * kPlane;               // hi + lo

struct Box { float fx[4], fy[4]; float rs, as; unsigned rowb[4], colb[4]; };

__device__ __forceinline__ f32x4 fma4(f32x4 a, float w, f32x4 c) { return f32x4{fmaf(a[0], w, c[0]), fmaf(a[1], w, c[1]), fmaf(a[2], w, c[2]), fmaf(a[3], w, c[3])}; }
__device__ __forceinline__ f32x4 mul4(f32x4 a, float w) { return f32x4{a[0] * w, a[1] * w, a[2] * w, a[3] * w}; }
__device__ __forceinline__ f32x4 sample4(f32x4 a, f32x4 b, f32x4 c, f32x4 d, float w0, float w1, float w2, float w3)
{
    return fma4(d, w3, fma4(c, w2, fma4(b, w1, mul4(a, w0))));
}
__device__ __forceinline__ float quot(float v, float as, float rs)
{
    const float q0 = v * rs;
    const float q1 = fmaf(fmaf(-as, q0, v), rs, q0);
    return fmaf(fmaf(-as, q1, v), rs, q1);
}

template <int MODE, int DMA>
__global__ __launch_bounds__(512) void k(const uint4 *__restrict__ w_hi, const uint4 *__restrict__ w_lo, const unsigned *__restrict__ boxes,
                                         const float *__restrict__ image, float *out, int items, unsigned long long *cyc)
{
    __shared__ __align__(16) unsigned char s_win[kWinBytes];
    __shared__ __align__(16) unsigned char s_planes[2][kTile];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    for (int i = tid; i < kWinBytes / 4; i += 512) reinterpret_cast<float *>(s_win)[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
    for (int i = tid; i < 2 * kTile / 4; i += 512) reinterpret_cast<unsigned *>(s_planes)[i] = 0x3c003c00u;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        // ------------------------------------------------------------------ X: the matrix role
        if (MODE == 2) { for (int item = 0; item < items; ++item) __syncthreads(); return; }
        const int r = lane & 31, h = lane >> 5;
        f16x8 wh[16][2];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const uint4 u = w_hi[((wave * 16 + ks) * 2 + cb) * 64 + lane];
                wh[ks][cb] = __builtin_bit_cast(f16x8, u);
                asm volatile("" : "+v"(wh[ks][cb]));
            }
        f32x16 acc[2], sum[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = sum[0][i] = sum[1][i] = 0.0f;
        const uint4 *wl_base = w_lo + (size_t)wave * 16 * 2 * 64 + lane;
        constexpr int PF = 4; // k-steps of W lo in flight
        for (int item = 0; item < items; ++item) {
            const unsigned char *pr = s_planes[item & 1];
            const uint4 *wl = wl_base + (size_t)(item & 3) * 4 * 16 * 2 * 64; // (four copies of the plane: a new address every item)
            uint4 ring[PF][2];
#pragma unroll
            for (int p = 0; p < PF; ++p) { ring[p][0] = wl[(p * 2 + 0) * 64]; ring[p][1] = wl[(p * 2 + 1) * 64]; }
            f16x8 ah = *reinterpret_cast<const f16x8 *>(pr + (0 + h) * kChunkStride + r * 16);
            f16x8 al = *reinterpret_cast<const f16x8 *>(pr + kPlane + (0 + h) * kChunkStride + r * 16);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                f16x8 ah_n = ah, al_n = al;
                if (ks + 1 < 16) {
                    ah_n = *reinterpret_cast<const f16x8 *>(pr + (2 * (ks + 1) + h) * kChunkStride + r * 16);
                    al_n = *reinterpret_cast<const f16x8 *>(pr + kPlane + (2 * (ks + 1) + h) * kChunkStride + r * 16);
                }
                const f16x8 wl0 = __builtin_bit_cast(f16x8, ring[ks % PF][0]), wl1 = __builtin_bit_cast(f16x8, ring[ks % PF][1]);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl1, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[ks][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[ks][1], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[ks][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[ks][1], acc[1], 0, 0, 0);
                if (ks + PF < 16) { ring[ks % PF][0] = wl[((ks + PF) * 2 + 0) * 64]; ring[ks % PF][1] = wl[((ks + PF) * 2 + 1) * 64]; }
                ah = ah_n; al = al_n;
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float a = acc[cb][i];
                    sum[cb][i] = fmaf(a > 0.0f ? a : 0.0f, 0x1p-20f, sum[cb][i]);
                    acc[cb][i] = 1.0f;
                }
            __syncthreads();
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) out[((size_t)blockIdx.x * 512 + tid) * 32 + cb * 16 + i] = sum[cb][i];
    } else {
        // ------------------------------------------------------------------ Y: the pooling role (next item)
        if (MODE == 1) { for (int item = 0; item < items; ++item) __syncthreads(); return; }
        const int yw = wave - 4, pb = lane >> 2, pi = lane & 3;
        Box bx[2];
        auto load_boxes = [&](int item) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const unsigned *b = boxes + (((size_t)blockIdx.x * 8 + (item & 7)) * 32 + hf * 16 + pb) * 16;
#pragma unroll
                for (int i = 0; i < 4; ++i) { bx[hf].fx[i] = __uint_as_float(0x3e800000u + (b[i] & 0xffffu)); bx[hf].fy[i] = __uint_as_float(0x3e800000u + (b[8 + i] & 0xffffu)); }
                bx[hf].rs = 0.25f; bx[hf].as = 4.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    bx[hf].rowb[i] = (b[i] % 8u) * 10u * 1024u;
                    bx[hf].colb[i] = (b[4 + i] % 10u) * 1024u + (unsigned)(yw * 256 + pi * 16);
                }
            }
        };
        auto issue = [&](int it, f32x4 (&t)[16]) {
            const Box &b = bx[it >> 2];
            const unsigned rot = (unsigned)((pb + it) & 3) << 6;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    t[i * 4 + j] = *reinterpret_cast<const f32x4 *>(s_win + (b.rowb[i] + b.colb[j] + rot));
        };
        auto consume = [&](int it, f32x4 (&t)[16], unsigned char *planes) {
            const Box &b = bx[it >> 2];
            // weights from the fractions (the shipped kernels read 16 rounded weights from the record: same count of multiplies here)
            const f32x4 lt = sample4(t[0], t[1], t[4], t[5], b.fy[0] * b.fx[0], b.fy[0] * b.fx[1], b.fy[1] * b.fx[0], b.fy[1] * b.fx[1]);
            const f32x4 rb = sample4(t[10], t[11], t[14], t[15], b.fy[2] * b.fx[2], b.fy[2] * b.fx[3], b.fy[3] * b.fx[2], b.fy[3] * b.fx[3]);
            const f32x4 rt = sample4(t[2], t[3], t[6], t[7], b.fy[0] * b.fx[2], b.fy[0] * b.fx[3], b.fy[1] * b.fx[2], b.fy[1] * b.fx[3]);
            const f32x4 lb = sample4(t[8], t[9], t[12], t[13], b.fy[2] * b.fx[0], b.fy[2] * b.fx[1], b.fy[3] * b.fx[0], b.fy[3] * b.fx[1]);
            f32x4 v = ((lt + rb) - rt) - lb;
            v = f32x4{quot(v[0], b.as, b.rs), quot(v[1], b.as, b.rs), quot(v[2], b.as, b.rs), quot(v[3], b.as, b.rs)};
            const f16x4 hi = __builtin_convertvector(v, f16x4);
            const f32x4 back = __builtin_convertvector(hi, f32x4);
            const f16x4 lo = __builtin_convertvector(v - back, f16x4);
            const unsigned piece = (unsigned)((pb + it) & 3);
            const int row = (it >> 2) * 16 + pb;
            const int off = (int)(yw * 8 + 2 * piece + (pi >> 1)) * kChunkStride + row * 16 + (pi & 1) * 8;
            *reinterpret_cast<f16x4 *>(planes + off) = hi;
            *reinterpret_cast<f16x4 *>(planes + kPlane + off) = lo;
        };
        f32x4 ta[16], tb[16];
        load_boxes(0);
        issue(0, ta);
        for (int item = 0; item < items; ++item) {
            unsigned char *pw = s_planes[(item + 1) & 1];
            if (DMA && yw < 2) { // the window of the item after the next: ~40 slots of 1 KiB by LDS-DMA, 20 per wave
                const char *src = reinterpret_cast<const char *>(image) + ((size_t)(blockIdx.x * 64 + (item & 63)) * 64 + yw * 20) * 1024 + lane * 16;
#pragma unroll
                for (int s = 0; s < 20; ++s)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + s * 1024),
                                                     (__attribute__((address_space(3))) void *)(s_win + (44 + yw * 20 + s) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int it = 0; it < 8; it += 2) {
                issue(it + 1, tb);
                consume(it, ta, pw);
                if (it + 2 < 8) issue(it + 2, ta);
                else { load_boxes(item + 1); issue(0, ta); }
                consume(it + 1, tb, pw);
            }
            if (DMA) __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0)
            __syncthreads();
        }
        if (tid == 511) out[0] += ta[0][0];
    }
    if (tid == 0) cyc[blockIdx.x] = (unsigned long long)(__builtin_amdgcn_s_memtime() - t0);
}

template <int MODE, int DMA> void run(const char *name, const uint4 *wh, const uint4 *wl, const unsigned *boxes, const float *image, float *out, unsigned long long *cyc)
{
    const int items = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, DMA>), dim3(256), dim3(512), 0, 0, wh, wl, boxes, image, out, items, cyc);
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<MODE, DMA>), dim3(256), dim3(512), 0, 0, wh, wl, boxes, image, out, items, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc[256];
    hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < 256; ++i) mean += (double)hc[i] / 256;
    const hipError_t err = hipGetLastError();
    printf("%-44s %8.1f us per launch, %6.3f us per item, s_memtime %7.0f ticks per item (100 MHz: x clock/100MHz = cycles)  (%s)\n", name, ms * 1e3 / 5,
           ms * 1e3 / 5 / items, mean / items, hipGetErrorString(err));
}

int main()
{
    uint4 *wh, *wl; unsigned *boxes; float *out, *image; unsigned long long *cyc;
    const size_t wbytes = (size_t)4 * 16 * 2 * 64 * 16;
    hipMalloc(&wh, wbytes); hipMalloc(&wl, 4 * wbytes);
    hipMalloc(&boxes, (size_t)256 * 8 * 32 * 16 * 4);
    hipMalloc(&out, (size_t)256 * 512 * 32 * 4);
    hipMalloc(&image, (size_t)256 * 64 * 64 * 1024);
    hipMalloc(&cyc, 256 * 8);
    hipMemset(image, 0, (size_t)256 * 64 * 64 * 1024);
    const size_t nb = (size_t)256 * 8 * 32 * 16;
    unsigned *hb = (unsigned *)malloc(nb * 4);
    for (size_t i = 0; i < nb; ++i) hb[i] = (unsigned)rand();
    hipMemcpy(boxes, hb, nb * 4, hipMemcpyHostToDevice);
    unsigned *hw = (unsigned *)malloc(4 * wbytes);
    for (size_t i = 0; i < 4 * wbytes / 4; ++i) hw[i] = 0x2c002c00u + (rand() & 0x03ff03ff);
    hipMemcpy(wh, hw, wbytes, hipMemcpyHostToDevice);
    hipMemcpy(wl, hw, 4 * wbytes, hipMemcpyHostToDevice);
    run<1, 0>("X only (96 MFMAs, W lo streamed)", wh, wl, boxes, image, out, cyc);
    run<2, 0>("Y only (8 units per wave, 32 taps in flight)", wh, wl, boxes, image, out, cyc);
    run<0, 0>("X and Y side by side", wh, wl, boxes, image, out, cyc);
    run<0, 1>("X and Y side by side + 40 KB window DMA", wh, wl, boxes, image, out, cyc);
    return 0;
}
