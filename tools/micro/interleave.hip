// interleave.hip -- can ONE wave per SIMD (512 registers: the whole fp16 hi + lo weight of its 64 output columns resident, 256 of
// them) run the box pooling of the NEXT 32-row item inside the MFMA shadows of the CURRENT one?  The serial fused kernel
// (vfa_fused.hip) runs the two phases back to back on two waves per SIMD (W takes half of a 256-register wave: nothing left to
// prefetch taps with); per SIMD and item the pooling issues ~920 vector instructions (3 700 cycles), the product 96 MFMAs
// (3 072 cycles of pipe, 768 of issue): interleaved perfectly ~4 450 cycles against ~8 750 today.
//   MODE 0: interleaved    1: MFMAs only    2: pooling only    3: pooling, then the MFMAs (one wave per SIMD, serial)
//   hipcc --offload-arch=gfx950 -O3 -o interleave tools/micro/interleave.hip && ./interleave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWinBytes = 92 * 1024;            // tap window: 92 slots of 1 KiB (256 channels fp32)
constexpr int kChunkStride = 32 * 16 + 32;      // A tile: chunk (8 k) x 32 rows x 16 B, padded
constexpr int kPlane = 32 * kChunkStride;       // 32 chunks = 256 k
constexpr int kTile = 2 * kPlane;               // hi + lo

struct Box { float fx[4], fy[4]; float scl; unsigned rowb[4], colb[4]; }; // (the sixteen tap weights are formed where they are used: 8 registers instead of 16)

__device__ __forceinline__ f32x4 fma4(f32x4 a, float w, f32x4 c) { return f32x4{fmaf(a[0], w, c[0]), fmaf(a[1], w, c[1]), fmaf(a[2], w, c[2]), fmaf(a[3], w, c[3])}; }
__device__ __forceinline__ f32x4 mul4(f32x4 a, float w) { return f32x4{a[0] * w, a[1] * w, a[2] * w, a[3] * w}; }
__device__ __forceinline__ f32x4 sample4(f32x4 a, f32x4 b, f32x4 c, f32x4 d, float y0, float y1, float x0, float x1)
{
    return fma4(d, y1 * x1, fma4(c, y1 * x0, fma4(b, y0 * x1, mul4(a, y0 * x0))));
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4 *__restrict__ wsrc, const unsigned *__restrict__ boxes, float *out, int items)
{
    __shared__ __align__(16) unsigned char s_win[kWinBytes];
    __shared__ __align__(16) unsigned char s_planes[2][kTile];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < kWinBytes / 4; i += 256) reinterpret_cast<float *>(s_win)[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
    for (int i = tid; i < 2 * kTile / 4; i += 256) reinterpret_cast<unsigned *>(s_planes)[i] = 0x3c003c00u;
    // the wave's weight: 16 k-steps x 2 column blocks x {hi, lo}
    f16x8 w[16][2][2];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const uint4 u = wsrc[(((wave * 16 + ks) * 2 + cb) * 2 + p) * 64 + lane];
                w[ks][cb][p] = __builtin_bit_cast(f16x8, u);
                asm volatile("" : "+v"(w[ks][cb][p])); // (resident: not to be loaded again inside the loop)
            }
    // two boxes per lane (the wave's 32 boxes x 64 channels: lane = (box 0..15, piece 0..3), halves 0 / 1)
    Box bx[2];
    const int pb = lane >> 2, pi = lane & 3;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const unsigned *b = boxes + ((size_t)blockIdx.x * 32 + hf * 16 + pb) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) { bx[hf].fx[i] = __uint_as_float(0x3e800000u + (b[i] & 0xffffu)); bx[hf].fy[i] = __uint_as_float(0x3e800000u + (b[8 + i] & 0xffffu)); }
        bx[hf].scl = 0.25f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bx[hf].rowb[i] = (b[i] % 9u) * 10u * 1024u;                                    // row part: slot row x 10 columns
            bx[hf].colb[i] = (b[4 + i] % 10u) * 1024u + (unsigned)(wave * 256 + pi * 16);  // column part + the wave's channel quarter
        }
    }
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.0f;
    float *s_sum = reinterpret_cast<float *>(s_win); // (the tile's running sums would live in LDS: 32 registers the wave does not have)
    __syncthreads();
    const int r = lane & 31, h = lane >> 5;
    const unsigned win = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)s_win;

    // corners lt, rb (taps A) and rt, lb (taps B) of iteration `it`: eight taps each
    auto issue_a = [&](int it, f32x4 (&t)[8]) {
        const Box &b = bx[it >> 2];
        const unsigned rot = (unsigned)((pb + it) & 3) << 6;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                t[i * 2 + j] = *reinterpret_cast<const f32x4 *>(s_win + (b.rowb[i] + b.colb[j] + rot));
                t[4 + i * 2 + j] = *reinterpret_cast<const f32x4 *>(s_win + (b.rowb[2 + i] + b.colb[2 + j] + rot));
            }
    };
    auto issue_b = [&](int it, f32x4 (&t)[8]) {
        const Box &b = bx[it >> 2];
        const unsigned rot = (unsigned)((pb + it) & 3) << 6;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                t[i * 2 + j] = *reinterpret_cast<const f32x4 *>(s_win + (b.rowb[i] + b.colb[2 + j] + rot));
                t[4 + i * 2 + j] = *reinterpret_cast<const f32x4 *>(s_win + (b.rowb[2 + i] + b.colb[j] + rot));
            }
    };
    auto half_a = [&](int it, f32x4 (&t)[8], f32x4 &v) {
        const Box &b = bx[it >> 2];
        const f32x4 lt = sample4(t[0], t[1], t[2], t[3], b.fy[0], b.fy[1], b.fx[0], b.fx[1]);
        const f32x4 rb = sample4(t[4], t[5], t[6], t[7], b.fy[2], b.fy[3], b.fx[2], b.fx[3]);
        v = lt + rb;
    };
    auto half_b = [&](int it, f32x4 (&t)[8], f32x4 v, unsigned char *planes) {
        const Box &b = bx[it >> 2];
        const f32x4 rt = sample4(t[0], t[1], t[2], t[3], b.fy[0], b.fy[1], b.fx[2], b.fx[3]);
        const f32x4 lb = sample4(t[4], t[5], t[6], t[7], b.fy[2], b.fy[3], b.fx[0], b.fx[1]);
        v = (v - rt) - lb;
        v = v * b.scl;
        const f16x4 hi = __builtin_convertvector(v, f16x4);
        const f32x4 back = __builtin_convertvector(hi, f32x4);
        const f16x4 lo = __builtin_convertvector(v - back, f16x4);
        const unsigned piece = (unsigned)((pb + it) & 3);
        const int row = (it >> 2) * 16 + pb;
        const int off = (int)(wave * 8 + 2 * piece + (pi >> 1)) * kChunkStride + row * 16 + (pi & 1) * 8;
        *reinterpret_cast<f16x4 *>(planes + off) = hi;
        *reinterpret_cast<f16x4 *>(planes + kPlane + off) = lo;
    };
    auto kstep = [&](int ks, const unsigned char *planes) {
        const f16x8 ah = *reinterpret_cast<const f16x8 *>(planes + (2 * ks + h) * kChunkStride + r * 16);
        const f16x8 al = *reinterpret_cast<const f16x8 *>(planes + kPlane + (2 * ks + h) * kChunkStride + r * 16);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w[ks][cb][1], acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w[ks][cb][0], acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, w[ks][cb][0], acc[cb], 0, 0, 0);
        }
    };

    f32x4 ta[8], tb[8], v;
    if (MODE != 1) issue_a(0, ta);
    for (int item = 0; item < items; ++item) {
        // (the boxes change with every item: nothing derived from them may be kept across the loop)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(bx[hf].rowb[i]), "+v"(bx[hf].colb[i]), "+v"(bx[hf].fx[i]), "+v"(bx[hf].fy[i]));
        unsigned char *pw = s_planes[(item + 1) & 1];       // the next item's A tile (written)
        const unsigned char *pr = s_planes[item & 1];       // this item's (read)
        if (MODE == 0) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                issue_b(it, tb);
                kstep(2 * it, pr);
                half_a(it, ta, v);
                issue_a((it + 1) & 7, ta);
                kstep(2 * it + 1, pr);
                half_b(it, tb, v, pw);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) kstep(ks, pr);
        } else {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                issue_b(it, tb);
                half_a(it, ta, v);
                issue_a((it + 1) & 7, ta);
                half_b(it, tb, v, pw);
            }
            if (MODE == 3) {
                __syncthreads();
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) kstep(ks, pw);
            }
        }
        // relu and the tile's running sum (in LDS)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float *p = s_sum + ((wave * 2 + cb) * 16 + i) * 64 + lane;
                *p += acc[cb][i] > 0.0f ? acc[cb][i] : 0.0f;
                acc[cb][i] = 1.0f;
            }
        __syncthreads();
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) out[((size_t)blockIdx.x * 256 + tid) * 32 + cb * 16 + i] = s_sum[((wave * 2 + cb) * 16 + i) * 64 + lane] + ta[i & 7][0];
}

template <int MODE> void run(const char *name, const uint4 *w, const unsigned *boxes, float *out)
{
    const int items = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, w, boxes, out, items);
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, w, boxes, out, items);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const hipError_t err = hipGetLastError();
    printf("%-34s %8.1f us per launch, %7.2f us per item = %6.0f cycles @2.4 GHz  (%s)\n", name, ms * 1e3 / 5, ms * 1e3 / 5 / items,
           ms * 1e-3 / 5 / items * 2.4e9, hipGetErrorString(err));
}

int main()
{
    uint4 *w; unsigned *boxes; float *out;
    hipMalloc(&w, 4 * 16 * 2 * 2 * 64 * 16);
    hipMalloc(&boxes, 256 * 32 * 16 * 4);
    hipMalloc(&out, (size_t)256 * 256 * 32 * 4);
    unsigned *hb = (unsigned *)malloc(256 * 32 * 16 * 4);
    for (int i = 0; i < 256 * 32 * 16; ++i) hb[i] = (unsigned)rand();
    hipMemcpy(boxes, hb, 256 * 32 * 16 * 4, hipMemcpyHostToDevice);
    unsigned *hw = (unsigned *)malloc(4 * 16 * 2 * 2 * 64 * 16);
    for (int i = 0; i < 4 * 16 * 2 * 2 * 64 * 4; ++i) hw[i] = 0x2c002c00u + (rand() & 0x03ff03ff);
    hipMemcpy(w, hw, 4 * 16 * 2 * 2 * 64 * 16, hipMemcpyHostToDevice);
    run<1>("MFMAs only (96 per wave and item)", w, boxes, out);
    run<2>("pooling only", w, boxes, out);
    run<3>("pooling, then MFMAs", w, boxes, out);
    run<0>("interleaved in one wave", w, boxes, out);
    return 0;
}
