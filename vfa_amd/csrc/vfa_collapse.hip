// vfa_collapse.hip -- `collapse` (Linear + ReLU, reference vfa/model/vfa_op.py:121-124) fused with the cross-view sum
// (vfa/model/vfanet.py:82) for single-layer grids: K = C = 256 inputs, N = 256 outputs.
//
//   out[m, :] = (accumulate ? out[m, :] : 0) + sum_v relu(vox[v, m, :] . W^T + bias)
//
// The fp32 MFMA of gfx950 (v_mfma_f32_32x32x2_f32) is clock/power bound at ~110 TFLOP/s and made this product 58 % of
// the frame.  Here every fp32 operand is split EXACTLY into bf16 pieces (x = hi + lo + r, |r| <= 2^-18 |x|) and the
// product is formed from three (or four) bf16 MFMAs with fp32 accumulation:
//   a.w ~= a_lo.w_hi + a_hi.w_lo + a_hi.w_hi (+ a_lo.w_lo)
// (the "3xBF16" scheme; dropped terms <= 3 * 2^-18 |a||w| per product, measured 4-7e-6 of max|out| against the 1e-5
// the path's tolerance allows, the fp32 library GEMM itself sits at 1e-6).  16x the MFMA rate for 3x the MFMAs.
//
// One persistent 512-thread workgroup per CU (all 160 KiB of LDS).  Wave w owns output columns 32 w .. 32 w + 31 and
// keeps ITS slice of W (256 k x 32 columns, hi and lo planes) in 128 VGPRs for the whole launch, split on the fly from
// the fp32 weight: no host-side weight preparation, W is read once per workgroup.  An item is (tile of 32 cells, view):
// its 32 x 256 fp32 rows are one contiguous 32 KiB read, brought in by LDS-DMA (global_load_lds_dwordx4, one row per
// wave instruction) into a ring of three raw slots -- two items stay in flight across the (raw) barriers, which is
// what hides the ~3 us loaded HBM latency at one workgroup per CU; register staging can only keep one.  Each thread
// then splits its share of the next item into two XOR-swizzled bf16 planes (conflict-free ds_read_b128 fragments),
// double-buffered against the MFMAs of the current item; relu(acc + bias) of every view is added in registers, so
// neither `lin` (n_views x M x N) nor a separate epilogue pass touches HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vfa_hip.h"


namespace {

constexpr int kRows = 32;         // cells per tile = one 32x32 MFMA row block
constexpr int kK = 256, kN = 256; // the only shape this kernel is built for
constexpr int kThreads = 512;     // 8 waves: wave w owns columns 32 w ..
constexpr int kRowBytes = 2 * kK; // one bf16 plane row; 16-byte chunks are XOR-swizzled with (row & 15)
constexpr int kPlane = kRows * kRowBytes;
constexpr int kSteps = kK / 16;   // k-steps of v_mfma_f32_32x32x16_bf16
constexpr int kRing = 3;          // raw fp32 slots
constexpr int kRaw = kRows * kK * 4;
constexpr int kLdsBytes = kRing * kRaw + 2 * 2 * kPlane; // 163840 = all of the CU's LDS
constexpr int kDmaPerItem = kRows / (kThreads / 64);      // LDS-DMA instructions per wave and item (4)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float relu_t(float x) { return (x < 0.0f) ? 0.0f : x; } // NaN stays NaN

// x = hi + lo + r exactly in fp32 arithmetic: hi = RNE bf16(x), lo = RNE bf16(x - hi)
__device__ __forceinline__ void split_bf16(float x, __bf16 &hi, __bf16 &lo)
{
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

struct Frag { bf16x8 hi, lo; };

template <int TERMS>
__global__ __launch_bounds__(kThreads) void collapse_relu_sum_kernel(const float *__restrict__ vox, const float *__restrict__ weight,
                                                                   const float *__restrict__ bias, float *__restrict__ out,
                                                                   int n_views, long long M, int accumulate)
{

    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *raw = smem;                       // [kRing][32 rows][1 KiB]
    unsigned char *planes = smem + kRing * kRaw;     // [buffer][hi / lo][32 rows][512 B]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;
    const long long n_tiles = (M + kRows - 1) / kRows;
    long long tile = blockIdx.x;
    if (tile >= n_tiles) return;

    // W slice of this wave as MFMA B fragments: lane (r, h) holds W[n = 32 wave + r][k = 16 s + 8 h + j], j = 0..7
    Frag w[kSteps];
    {
        const float *wrow = weight + (size_t)(wave * 32 + r) * kK + 8 * h;
#pragma unroll
        for (int s = 0; s < kSteps; ++s) {
            const float4 x0 = *reinterpret_cast<const float4 *>(wrow + 16 * s);
            const float4 x1 = *reinterpret_cast<const float4 *>(wrow + 16 * s + 4);
            const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __bf16 a, b;
                split_bf16(x[j], a, b);
                w[s].hi[j] = a;
                w[s].lo[j] = b;
            }
        }
    }
    const float bcol = bias ? bias[wave * 32 + r] : 0.0f;

    // LDS-DMA of one item: wave w brings rows 4 w .. 4 w + 3 (a row = 64 lanes x 16 B = the instruction's 1 KiB)
    auto fetch = [&](long long t, int v, int slot) {
#pragma unroll
        for (int j = 0; j < kDmaPerItem; ++j) {
            const int row = wave * kDmaPerItem + j;
            long long m = t * kRows + row;
            m = m < M ? m : M - 1; // rows past the end: any valid row, their outputs are never stored
            const float *src = vox + ((size_t)v * M + (size_t)m) * kK + lane * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(raw + slot * kRaw + row * 1024), 16, 0, 0);
        }
    };
    // split this thread's share of a landed raw slot into the bf16 planes of `buf`
    constexpr int kPieces = kRows * kK / 4 / kThreads; // float4 per thread and item (4)
    static_assert(kPieces == 4, "the raw reads below are written for 4 pieces");
    auto stage = [&](int slot, int buf) {
        // The raw reads are inline asm: for a plain LDS load hipcc first drains every LDS-DMA in flight (vmcnt(0), it
        // cannot tell the slots apart), which would cut the prefetch depth to nothing.  wait_oldest() + the barrier
        // of the previous item already guarantee that this slot has landed.
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
        const unsigned src = lds0 + slot * kRaw + tid * 16;
        const unsigned dst = lds0 + kRing * kRaw + buf * 2 * kPlane;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4 x4[2];
            if (half == 0)
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:8192\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(x4[0]), "=&v"(x4[1]) : "v"(src) : "memory");
            else
                asm volatile("ds_read_b128 %0, %2 offset:16384\n\tds_read_b128 %1, %2 offset:24576\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(x4[0]), "=&v"(x4[1]) : "v"(src) : "memory");
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = tid + kThreads * (2 * half + k), row = idx >> 6, c4 = idx & 63;
                const float x[4] = {x4[k].x, x4[k].y, x4[k].z, x4[k].w};
                union { __bf16 b[4]; unsigned long long u; } hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) split_bf16(x[j], hi.b[j], lo.b[j]);
                const unsigned off = dst + row * kRowBytes + ((((c4 >> 1) ^ (row & 15)) << 4) | ((c4 & 1) << 3));
                // hi plane at off, lo plane at off + kPlane (= 32 x 64 x 8 bytes)
                asm volatile("ds_write2st64_b64 %0, %1, %2 offset1:32" : : "v"(off), "v"(hi.u), "v"(lo.u) : "memory");
            }
        }
    };
    auto next_item = [&](long long &t, int &v) {
        if (++v == n_views) { v = 0; t += gridDim.x; }
    };
    auto barrier = [&]() { // raw: a __syncthreads() would drain the LDS-DMA in flight (vmcnt(0))
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // Items are (tile, view) pairs in the order this workgroup meets them; item i lives in raw slot i % 3 and, split,
    // in plane buffer i & 1.  All waves run the same phase at the same time: the two waves of a SIMD hide each other's
    // fragment-read latency in the MFMA phase.  Measured alternatives, none faster: shifting waves 4-7 half an item
    // (their split under the others' MFMAs), reading the next chunk's fragments ahead, a quarter of the split behind
    // every second chunk (102-118 us each against 109-115 us on the bench shape).
    long long tf = tile; // next item to fetch
    int vf = 0, slot_f = 0;
    int in_flight = 0;   // items whose LDS-DMA this wave has issued and not yet waited for
    for (int k = 0; k < kRing; ++k) {
        if (tf < n_tiles) {
            fetch(tf, vf, slot_f);
            ++in_flight;
        }
        next_item(tf, vf);
        slot_f = slot_f == kRing - 1 ? 0 : slot_f + 1;
    }
    auto wait_oldest = [&]() { // wave-uniform: leave the younger items in flight
        if (in_flight >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (in_flight == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (in_flight > 0) --in_flight;
    };
    wait_oldest();
    barrier();
    stage(0, 0);
    wait_oldest();
    barrier();

    int buf = 0, slot = 0; // raw slot of the CURRENT item (already split): free for the fetch below
    const int key = r & 15;
    const int frag_base = r * kRowBytes + ((h ^ (key & 1)) << 4);
    long long t1 = tile; // item after the current one
    int v1 = 0;
    next_item(t1, v1);
    for (; tile < n_tiles; tile += gridDim.x) {
        f32x16 sum;
#pragma unroll
        for (int i = 0; i < 16; ++i) sum[i] = 0.0f;
        for (int v = 0; v < n_views; ++v) {
            const bool has_next = t1 < n_tiles;
            const int slot1 = slot == kRing - 1 ? 0 : slot + 1;
            if (tf < n_tiles) { // item + 3 into the slot the current item was split from
                fetch(tf, vf, slot);
                ++in_flight;
            }
            next_item(tf, vf);

            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
            const unsigned char *pa = planes + buf * 2 * kPlane + frag_base;
            // Two k-steps at a time; a chunk whose hi fragments are all zero in this wave's 32 rows (a view that does
            // not see these cells: 15-67 % of the rows are masked) costs two LDS reads and no MFMA.
#pragma unroll
            for (int c = 0; c < kSteps / 2; ++c) {
                const int off0 = ((2 * c) ^ (key >> 1)) << 5, off1 = ((2 * c + 1) ^ (key >> 1)) << 5;
                const bf16x8 h0 = *reinterpret_cast<const bf16x8 *>(pa + off0);
                const bf16x8 h1 = *reinterpret_cast<const bf16x8 *>(pa + off1);
                const uint4 u0 = *reinterpret_cast<const uint4 *>(&h0), u1 = *reinterpret_cast<const uint4 *>(&h1);
                const unsigned any = (u0.x | u0.y) | (u0.z | u0.w) | (u1.x | u1.y) | (u1.z | u1.w);
                if (__ballot((any & 0x7fff7fffu) != 0u) == 0ull) continue; // +-0 only
                const bf16x8 l0 = *reinterpret_cast<const bf16x8 *>(pa + kPlane + off0);
                const bf16x8 l1 = *reinterpret_cast<const bf16x8 *>(pa + kPlane + off1);
                // the four hi products first: they cover the latency of the lo reads
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h0, w[2 * c].lo, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1, w[2 * c + 1].lo, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h0, w[2 * c].hi, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1, w[2 * c + 1].hi, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l0, w[2 * c].hi, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l1, w[2 * c + 1].hi, acc, 0, 0, 0);
                if (TERMS >= 4) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l0, w[2 * c].lo, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l1, w[2 * c + 1].lo, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) sum[i] = sum[i] + relu_t(acc[i] + bcol); // vfa_op.py:124, vfanet.py:82
            if (has_next) stage(slot1, buf ^ 1);
            wait_oldest(); // item + 2 has landed (this wave's rows); the barrier makes all rows visible
            if (v + 1 == n_views) {
                // C/D map of the 32x32 MFMA: register i of lane (r, h) is row (i & 3) + 8 (i >> 2) + 4 h, column r
                float *orow = out + (size_t)tile * kRows * kN + wave * 32 + r;
                const long long rows_left = M - tile * kRows;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (row < rows_left) {
                        // this workgroup owns the rows: a fire-and-forget atomic is load + add + store without the
                        // round trip (a load's wait would also drain the LDS-DMA in flight)
                        if (accumulate) unsafeAtomicAdd(orow + (size_t)row * kN, sum[i]);
                        else orow[(size_t)row * kN] = sum[i];
                    }
                }
            }
            barrier();
            buf ^= 1;
            slot = slot1;
            next_item(t1, v1);
        }
    }
}

} // namespace

extern "C" int vfa_collapse_relu_sum_f32(const float *vox, const float *weight, const float *bias, float *out, int n_views,
                                         size_t M, int K, int N, int accumulate, int flags, void *stream)
{
    const int terms = flags & VFA_FLAG_TERMS_MASK, reserved_cus = (flags >> 8) & 0xff;
    if (flags & ~(VFA_FLAG_TERMS_MASK | 0xff00)) return VFA_ERR_BAD_ARGUMENT;
    if (n_views < 0 || K <= 0 || N <= 0 || (terms != 0 && terms != 2 && terms != 3 && terms != 4 && terms != 6)) return VFA_ERR_BAD_ARGUMENT;
    if (K != kK || N != kN) return VFA_ERR_UNSUPPORTED;
    if (M == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (n_views == 0) {
        if (!accumulate) return (int)hipMemsetAsync(out, 0, M * (size_t)N * sizeof(float), s);
        return 0;
    }
    constexpr size_t lds_bytes = kLdsBytes;
    static bool attr_set = false; // idempotent: a race only repeats the call
    if (!attr_set) {
        hipError_t e = hipSuccess;
        const void *fns[2] = {(const void *)collapse_relu_sum_kernel<3>, (const void *)collapse_relu_sum_kernel<4>};
        for (int i = 0; i < 2 && e == hipSuccess; ++i)
            e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    int n_cu = 256;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            cus > 0)
            n_cu = cus;
    }
    const long long n_tiles = ((long long)M + kRows - 1) / kRows;
    // Persistent workgroups, one per CU: every workgroup walks `rounds` tiles; launch only as many as that takes (an even
    // load, and the CUs a ragged last round would idle stay free).  VFA_FLAG_RESERVED_CUS(n) lowers the count further when
    // that does not add a round.
    long long rounds = (n_tiles + n_cu - 1) / n_cu;
    if (reserved_cus > 0 && n_cu - reserved_cus >= 8 && (n_tiles + (n_cu - reserved_cus) - 1) / (n_cu - reserved_cus) == rounds)
        n_cu -= reserved_cus;
    long long wgs = (n_tiles + rounds - 1) / rounds;
    if (wgs > n_cu) wgs = n_cu;
    const unsigned blocks = (unsigned)wgs;
    if (terms == 4)
        hipLaunchKernelGGL((collapse_relu_sum_kernel<4>), dim3(blocks), dim3(kThreads), lds_bytes, s, vox, weight, bias, out, n_views,
                           (long long)M, accumulate);
    else
        hipLaunchKernelGGL((collapse_relu_sum_kernel<3>), dim3(blocks), dim3(kThreads), lds_bytes, s, vox, weight, bias, out, n_views,
                           (long long)M, accumulate);
    return (int)hipGetLastError();
}
