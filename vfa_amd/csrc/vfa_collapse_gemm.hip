// vfa_collapse_gemm.hip -- the `collapse` product (reference vfa/model/vfa_op.py:121-123) for multi-layer grids:
//
//   lin[m, :] = vox[m, :] . W^T          m < M = n_views * cells,  K = nl * 256 inputs,  N = 256 outputs
//
// (no bias, no ReLU: the epilogue kernels of vfa_kernels.hip add them while summing views, exactly as behind the library
// GEMM this replaces).  Same arithmetic as vfa_collapse.hip: every fp32 product is three bf16 MFMA products of an exact
// hi/lo split of both operands, accumulated in fp32 (error ~3e-6 of max|out|, tolerance of the path 1e-5).
//
// K does not fit the register-resident W of the single-layer kernel, so this one is a K-looped tile GEMM:
//   * a prep kernel splits W once per call into bf16 hi / lo planes stored in MFMA B-fragment order in a caller-owned
//     workspace (K * 256 * 4 bytes): a wave fetches the fragments of a k-step with one coalesced 1 KiB load per plane;
//   * one persistent 512-thread workgroup per CU owns tiles of 128 rows; wave w owns output columns 32 w .. 32 w + 31
//     and keeps 4 x (32 x 32) fp32 accumulators (64 VGPRs);
//   * K advances in chunks of 128: the chunk of the NEXT work item is loaded from HBM into registers (8 float4 per
//     thread) while the MFMAs of the current one run, then split into two XOR-swizzled bf16 LDS planes (double-buffered,
//     128 KiB); the W fragments of a chunk arrive in two halves, the second in flight under the MFMAs of the first;
//   * 96 MFMAs per wave between barriers (4x the single-layer kernel's);
//   * row blocks of 32 rows whose chunk is all zero -- masked voxels -- skip their MFMAs (flags set by the split pass).
// Measured (MI355X): 260-310 TFLOP/s fp32-equivalent (K = 1280 / 2048) against 110-135 for the fp32 library GEMM, error
// 5e-6 of max|out|.  In-kernel stamps: the MFMA phase of an item is 6 200-8 200 cycles (the two waves of a SIMD need
// 6 144), the HBM side delivers a CU's 64 KiB in ~7 500 (2.1-2.5 TB/s chip-wide while the MFMAs run, 4.8 TB/s alone),
// the output stores ~3 400 per item.  Variants measured within +-5 %: 4 dedicated loader waves beside 8 MFMA waves
// (768 threads), two items in flight per loader, LDS fragment reads pipelined one half-step ahead.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vfa_hip.h"
#include "vfa_split.h"


namespace {

constexpr int kN = 256;          // output channels (the only width this kernel is built for)
constexpr int kTileRows = 128;   // rows per tile = 4 MFMA row blocks
constexpr int kRowBlocks = kTileRows / 32;
constexpr int kChunk = 128;      // k per chunk
constexpr int kStepsPerChunk = kChunk / 16;
constexpr int kThreads = 512;
constexpr int kRowBytes = kChunk * 2;            // one bf16 plane row of a chunk (256 B); 16-byte pieces XOR (row & 15)
constexpr int kPlane = kTileRows * kRowBytes;    // 32 KiB
constexpr int kLdsBytes = 2 * 2 * kPlane + 64;   // [buffer][hi / lo] = 128 KiB, + live flags [3 items][4 row blocks]
constexpr int kLoads = kTileRows * kChunk / 4 / kThreads; // float4 per thread and chunk (8)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
using vfa_dev::f16x8;
using vfa_dev::kExpA;
using vfa_dev::kExpW;
using vfa_dev::pow2f;
using vfa_dev::split_exponent;
using vfa_dev::split_f16x4;
using vfa_dev::wave_max_u32;

__device__ __forceinline__ void split_bf16(float x, __bf16 &hi, __bf16 &lo)
{
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

// Workspace layout: fragment (chunk c, wave w, step s, plane p) = 64 lanes x 16 B at
//   ((((c * 8 + w) * kStepsPerChunk + s) * 2 + p) * 64 + lane) * 16
// lane (r, h) holds W[n = 32 w + r][k = 128 c + 16 s + 8 h + j], j = 0..7  (MFMA B operand of v_mfma_f32_32x32x16_bf16)
__global__ __launch_bounds__(256) void split_weight_kernel(const float *__restrict__ weight, uint4 *__restrict__ ws, int K)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; // one (c, w, s, lane) per thread, both planes
    const size_t total = (size_t)(K / kChunk) * 8 * kStepsPerChunk * 64;
    if (i >= total) return;
    const int lane = (int)(i & 63);
    const int s = (int)((i >> 6) % kStepsPerChunk);
    const int w = (int)((i >> 6) / kStepsPerChunk % 8);
    const int c = (int)((i >> 6) / kStepsPerChunk / 8);
    const int r = lane & 31, h = lane >> 5;
    const float *src = weight + (size_t)(32 * w + r) * K + (size_t)kChunk * c + 16 * s + 8 * h;
    union { __bf16 b[8]; uint4 u; } hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) split_bf16(src[j], hi.b[j], lo.b[j]);
    const size_t base = ((((size_t)c * 8 + w) * kStepsPerChunk + s) * 2) * 64 + lane;
    ws[base] = hi.u;
    ws[base + 64] = lo.u;
}

// fp16 form (VFA_FLAG_TERMS 2: the arithmetic of the fused frame kernels, vfa_split.h): max |W| in kWmaxParts partial maxima, then the
// fragments of W 2^ew split into two fp16 pieces, ew = the exponent that brings max|W| into [2^14, 2^15) -- what
// split_block (vfa_fused.hip) / pipe_split_weight_kernel of the frame kernels compute, so that the recomputed product of the training
// backward sees the forward's operands bit for bit.  `tail` (behind the fragments): kWmaxParts partial maxima, then ew.
constexpr int kWmaxParts = 32;
__global__ __launch_bounds__(256) void gemm_weight_absmax_kernel(const float *__restrict__ weight, unsigned *tail, size_t count)
{
    __shared__ unsigned part[4];
    unsigned m = 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)kWmaxParts * 256) m = max(m, __float_as_uint(weight[i]) & 0x7fffffffu);
    m = wave_max_u32(m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) tail[blockIdx.x] = max(max(part[0], part[1]), max(part[2], part[3]));
}
__global__ __launch_bounds__(256) void split_weight_f16_kernel(const float *__restrict__ weight, uint4 *__restrict__ ws, int K, unsigned *tail)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)(K / kChunk) * 8 * kStepsPerChunk * 64;
    if (i >= total) return;
    vfa_dev::fp16_saturate_mode(true);
    unsigned m = 0u;
    for (int k = 0; k < kWmaxParts; ++k) m = max(m, tail[k]);
    const int ew = split_exponent(m, kExpW);
    if (i == 0) reinterpret_cast<int *>(tail)[kWmaxParts] = ew;
    const float sc = pow2f(ew);
    const int lane = (int)(i & 63);
    const int s = (int)((i >> 6) % kStepsPerChunk);
    const int w = (int)((i >> 6) / kStepsPerChunk % 8);
    const int c = (int)((i >> 6) / kStepsPerChunk / 8);
    const int r = lane & 31, h = lane >> 5;
    const float *src = weight + (size_t)(32 * w + r) * K + (size_t)kChunk * c + 16 * s + 8 * h;
    uint2 h0, l0, h1, l1;
    split_f16x4(src[0] * sc, src[1] * sc, src[2] * sc, src[3] * sc, h0, l0);
    split_f16x4(src[4] * sc, src[5] * sc, src[6] * sc, src[7] * sc, h1, l1);
    const size_t base = ((((size_t)c * 8 + w) * kStepsPerChunk + s) * 2) * 64 + lane;
    ws[base] = make_uint4(h0.x, h0.y, h1.x, h1.y);
    ws[base + 64] = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

// fp16 form: what the kernel needs beside the operands -- the feature statistics of the scale (the frame kernels' 2^ea), the weight
// exponent (tail of the workspace) and the sliver shift of every row's item (vfa_sliver_shifts_u8; NULL: none)
struct F16Args {
    const unsigned *amax; int amax_n;
    const int *wexp;                // -> ew
    const unsigned char *shift;     // (M) per row, or NULL
    const unsigned char *tile_any;  // (ceil(M / 128)) 1 = a row of the tile has a shift (NULL: look at every row)
};

// MASK: the epilogue of the training backward (vfa_collapse_gemm_relu_backward_f32).  The product is the recomputed pre-activation;
// instead of storing it the kernel stores  d lin = (lin + bias > 0) ? d out[cell] : 0  (row m = view * cells + cell: the gradient
// of the view sum reaches every view alike) and adds the column sums of d lin to d bias -- `lin` never exists in memory.
struct MaskArgs {
    const float *bias;   // (256) or NULL
    const float *grad;   // (cells, 256): d out
    float *gbias;        // (256), accumulated with atomics (like vfa_relu_mask_backward_f32), or NULL
    long long cells;
};

template <int TERMS, bool MASK>
__global__ __launch_bounds__(kThreads) void collapse_gemm_kernel(const float *__restrict__ vox, const uint4 *__restrict__ ws,
                                                               float *__restrict__ lin, long long M, int K, MaskArgs ma, F16Args fa)
{
    constexpr bool F16 = TERMS == 2; // two fp16 pieces per operand under a power-of-two scale (vfa_split.h) instead of two bf16 pieces
    extern __shared__ __align__(16) unsigned char planes[]; // [buffer][hi / lo][128 rows][256 B]
    int *live = reinterpret_cast<int *>(planes + 2 * 2 * kPlane); // [item % 3][row block]: the chunk has a non-zero element there
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;
    const long long n_tiles = (M + kTileRows - 1) / kTileRows;
    const int n_chunks = K / kChunk;
    long long tile = blockIdx.x;
    if (tile >= n_tiles) return;
    // fp16 form: 2^ea of the voxel features (largest |feature| the integral-image kernels saw: the frame kernels' own reduction) and ew
    int ea = 0, ew = 0;
    if constexpr (F16) {
        __shared__ unsigned s_amax;
        if (tid == 0) s_amax = 0u;
        __syncthreads();
        unsigned mx = 0u;
        for (int i = tid; i < fa.amax_n; i += kThreads) mx = max(mx, fa.amax[i]);
        mx = wave_max_u32(mx);
        if (lane == 0) atomicMax(&s_amax, mx);
        __syncthreads();
        ea = split_exponent((unsigned)__builtin_amdgcn_readfirstlane((int)s_amax), kExpA);
        ew = __builtin_amdgcn_readfirstlane(*fa.wexp);
    }
    // sliver shift of row m (its item's: vfa_geom.h).  Tiles without a shifted row -- nearly all -- are told by one byte per tile.
    auto tile_shifted = [&](long long t) -> bool {
        if (!F16 || !fa.shift) return false;
        return !fa.tile_any || __builtin_amdgcn_readfirstlane((int)fa.tile_any[t]) != 0;
    };
    // this thread's share of a chunk: float4 number tid + 512 i of the 128 x 32 float4 (row = idx / 32).
    // Every load of the main loop is UNCONDITIONAL (addresses are clamped instead): with straight-line loads the in-order
    // vmcnt counter lets the compiler wait for exactly the W fragments it needs and leave the younger HBM loads in
    // flight; with predicated loads it fell back to vmcnt(0) between them and before the first MFMA (1.5x slower).
    float4 pre[kLoads];
    float row_scale[kLoads]; // fp16 form: 2^(ea - shift) of the row of pre[i]
    auto fetch = [&](long long t, int c) {
        const long long row0 = t * kTileRows;
        const bool shifted = tile_shifted(t);
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {
            const int idx = tid + kThreads * i, row = idx >> 5, c4 = idx & 31;
            long long m = row0 + row;
            m = m < M ? m : M - 1; // rows past the end read a valid row; their outputs are never stored
            pre[i] = *reinterpret_cast<const float4 *>(vox + (size_t)m * K + (size_t)c * kChunk + 4 * c4);
            if constexpr (F16) row_scale[i] = pow2f(ea - (shifted ? (int)fa.shift[m] : 0));
        }
    };
    auto stage = [&](int buf, int slot) {
        if constexpr (F16) vfa_dev::fp16_saturate_mode(true); // (conversions saturate; the MFMAs need the default mode: vfa_split.h)
        unsigned nz = 0; // bit rb: this thread saw a non-zero (or NaN) element in row block rb
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {
            const int idx = tid + kThreads * i, row = idx >> 5, c4 = idx & 31;
            const float x[4] = {pre[i].x, pre[i].y, pre[i].z, pre[i].w};
            union { __bf16 b[4]; uint2 u; } hi, lo;
            bool any = false;
            if constexpr (F16) {
                const float sc = row_scale[i]; // 2^(ea - shift) of this row (set by `fetch`)
                split_f16x4(x[0] * sc, x[1] * sc, x[2] * sc, x[3] * sc, hi.u, lo.u);
#pragma unroll
                for (int j = 0; j < 4; ++j) any |= !(x[j] == 0.0f);
            } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                split_bf16(x[j], hi.b[j], lo.b[j]);
                any |= !(x[j] == 0.0f);
            }
            }
            nz |= any ? 1u << (row >> 5) : 0u;
            const int off = row * kRowBytes + ((((c4 >> 1) ^ (row & 15)) << 4) | ((c4 & 1) << 3));
            *reinterpret_cast<uint2 *>(planes + (buf * 2 + 0) * kPlane + off) = hi.u;
            *reinterpret_cast<uint2 *>(planes + (buf * 2 + 1) * kPlane + off) = lo.u;
        }
        // a thread's rows are (tid + 512 i) / 32 = tid / 32 + 16 i: one row of each 16-row half of every row block
#pragma unroll
        for (int rb = 0; rb < kRowBlocks; ++rb)
            if (__ballot((nz >> rb) & 1u) != 0ull && lane == 0) live[slot * kRowBlocks + rb] = 1; // every writer stores 1
        if constexpr (F16) vfa_dev::fp16_saturate_mode(false);
    };
    // W fragments of half a chunk (4 k-steps x 2 planes = 32 VGPRs); two halves are alive at a time
    struct WHalf { bf16x8 hi[kStepsPerChunk / 2], lo[kStepsPerChunk / 2]; };
    auto load_w = [&](WHalf &wq, int c, int half) {
        const uint4 *p = ws + ((((size_t)c * 8 + wave) * kStepsPerChunk + half * (kStepsPerChunk / 2)) * 2) * 64 + lane;
#pragma unroll
        for (int s = 0; s < kStepsPerChunk / 2; ++s) {
            const uint4 a = p[(size_t)s * 128], b = p[(size_t)s * 128 + 64];
            wq.hi[s] = *reinterpret_cast<const bf16x8 *>(&a);
            wq.lo[s] = *reinterpret_cast<const bf16x8 *>(&b);
        }
    };

    const int key = r & 15;
    const int frag_base = r * kRowBytes + ((h ^ (key & 1)) << 4);
    f32x16 acc[kRowBlocks];
    auto mfma_half = [&](const WHalf &wq, int buf, int half, int slot) {
        const unsigned char *pa = planes + buf * 2 * kPlane + frag_base;
#pragma unroll
        for (int rb = 0; rb < kRowBlocks; ++rb) {
            if (!__builtin_amdgcn_readfirstlane(live[slot * kRowBlocks + rb])) continue; // 32 masked voxels: all +-0
#pragma unroll
            for (int s = 0; s < kStepsPerChunk / 2; ++s) {
                const int off = ((half * (kStepsPerChunk / 2) + s) ^ (key >> 1)) << 5;
                const unsigned char *pr = pa + rb * 32 * kRowBytes + off;
                const bf16x8 a_hi = *reinterpret_cast<const bf16x8 *>(pr);
                const bf16x8 a_lo = *reinterpret_cast<const bf16x8 *>(pr + kPlane);
                if constexpr (F16) { // (the order of the frame kernels: hi.lo, hi.hi, lo.hi)
                    const f16x8 ah = __builtin_bit_cast(f16x8, a_hi), al = __builtin_bit_cast(f16x8, a_lo);
                    const f16x8 wh = __builtin_bit_cast(f16x8, wq.hi[s]), wl = __builtin_bit_cast(f16x8, wq.lo[s]);
                    acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl, acc[rb], 0, 0, 0);
                    acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, acc[rb], 0, 0, 0);
                    acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh, acc[rb], 0, 0, 0);
                } else {
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, wq.lo[s], acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, wq.hi[s], acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, wq.hi[s], acc[rb], 0, 0, 0);
                if (TERMS >= 4) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, wq.lo[s], acc[rb], 0, 0, 0);
                }
            }
        }
    };

    // work items = (tile, chunk) in the order this workgroup meets them; item i: plane buffer i & 1, flags slot i % 3
    if (tid < 3 * kRowBlocks) live[tid] = 0;
    __syncthreads();
    fetch(tile, 0);
    stage(0, 0);
    WHalf w0, w1;
    load_w(w0, 0, 0);
    load_w(w1, 0, 1);
    __syncthreads();
    int buf = 0, slot = 0;
    float bias_c = 0.0f, gb = 0.0f; // MASK: this lane's column (32 wave + r): its bias, its running column sum of d lin
    if constexpr (MASK) bias_c = ma.bias ? ma.bias[wave * 32 + r] : 0.0f;
    for (; tile < n_tiles; tile += gridDim.x) {
        const bool shifted_tile = tile_shifted(tile);
#pragma unroll
        for (int rb = 0; rb < kRowBlocks; ++rb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[rb][i] = 0.0f;
                if constexpr (F16 && MASK) { // the bias rides in the accumulator, in the row's units: exactly how the frame kernels start it
                    const int row = rb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    long long m = tile * kTileRows + row;
                    m = m < M ? m : M - 1;
                    acc[rb][i] = (bias_c * pow2f(ea + ew)) * pow2f(shifted_tile ? -(int)fa.shift[m] : 0);
                }
            }
        for (int c = 0; c < n_chunks; ++c) {
            const bool last_chunk = c + 1 == n_chunks;
            const long long nt = last_chunk ? tile + gridDim.x : tile;
            const bool has_next = nt < n_tiles;
            const int cn = last_chunk ? 0 : c + 1; // the next item's chunk (its W fragments do not depend on the tile)
            const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
            if (tid < kRowBlocks) live[slot2 * kRowBlocks + tid] = 0; // read last in the previous item, set in the next
            fetch(has_next ? nt : tile, cn); // lands under the MFMAs below (after the last item: a harmless re-read)
            mfma_half(w0, buf, 0, slot);
            load_w(w0, cn, 0); // in flight under the second half's MFMAs
            mfma_half(w1, buf, 1, slot);
            load_w(w1, cn, 1); // in flight under the split below and the next first half
            stage(buf ^ 1, slot1);
            __syncthreads();
            buf ^= 1;
            slot = slot1;
        }
        // C/D map of the 32x32 MFMA: register i of lane (r, h) is row (i & 3) + 8 (i >> 2) + 4 h, column r
#pragma unroll
        for (int rb = 0; rb < kRowBlocks; ++rb) {
            const long long row0 = tile * kTileRows + rb * 32;
            float *orow = lin + (size_t)row0 * kN + wave * 32 + r;
            // MASK: the cell of the block's first row, once per block and in scalar registers (row0 is the same in every lane); a
            // row then adds its offset and wraps at most once (cells >= 32: the host checks) -- a 64-bit modulo per element cost
            // more than the product at K = 256
            long long cell0 = 0;
            if constexpr (MASK) cell0 = (long long)__builtin_amdgcn_readfirstlane((int)(row0 % ma.cells));
            if constexpr (MASK) {
                // The sixteen gradients of the lane FIRST, all in flight together, then the sixteen stores.  Written element by
                // element -- load, compare, store under the `row < M` branch -- every load waited with vmcnt(0) for itself AND for
                // the store in front of it: 64 dependent round trips per tile, ~95 000 cycles, 2.4 x the whole product (round 6:
                // 436 -> see DESIGN.md section 4.5 us per bench scale).  The loads need no branch (a cell index is always in range).
                // Offsets are 32-bit from uniform bases (the host checks the sizes): one register per address, none spilled.
                float gq[16];
                const unsigned lane_byte = (unsigned)(wave * 32 + r) * 4u;
                const int cell0_i = (int)cell0, cells_i = (int)ma.cells; // (cells < 2^22: the host checks)
                const char *gbase = reinterpret_cast<const char *>(ma.grad);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                    int cell = cell0_i + row;
                    cell = cell >= cells_i ? cell - cells_i : cell;
                    gq[i] = *reinterpret_cast<const float *>(gbase + ((unsigned)cell * (unsigned)(kN * 4) + lane_byte));
                }
                char *otile = reinterpret_cast<char *>(lin + (size_t)row0 * kN); // (uniform)
                const int rows_left = (int)(M - row0 < 32 ? M - row0 : 32);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                    const float pre_act = F16 ? acc[rb][i] : acc[rb][i] + bias_c; // (fp16 form: the bias is in the accumulator)
                    const float o = (pre_act > 0.0f) ? gq[i] : 0.0f;
                    if (row < rows_left) {
                        *reinterpret_cast<float *>(otile + ((unsigned)row * (unsigned)(kN * 4) + lane_byte)) = o;
                        gb += o;
                    }
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                if (row0 + row < M) {
                    if constexpr (MASK) {
                    } else {
                        // (fp16 form: back from the row's units 2^(ea + ew - shift): exact)
                        orow[(size_t)row * kN] = F16 ? acc[rb][i] * pow2f((shifted_tile ? (int)fa.shift[row0 + row] : 0) - (ea + ew)) : acc[rb][i];
                    }
                }
            }
        }
    }
    if constexpr (MASK) {
        if (ma.gbias) {
            gb += __shfl_xor(gb, 32); // the two row halves of the column
            if (h == 0) unsafeAtomicAdd(ma.gbias + wave * 32 + r, gb);
        }
    }
}

} // namespace

extern "C" size_t vfa_collapse_gemm_workspace_bytes(int K, int N)
{
    if (K <= 0 || N <= 0) return 0;
    return (size_t)K * (size_t)N * 4 + 256; // (+ the tail of the fp16 form: partial maxima of |W| and its exponent)
}

// f16: the fp16 form (statistics + shifts given): only through vfa_collapse_gemm_relu_backward_f16_f32
static int collapse_gemm_launch(const float *vox, const float *weight, float *out, void *workspace, size_t workspace_bytes, size_t M,
                                int K, int N, int flags, const MaskArgs *mask, void *stream, const F16Args *f16 = nullptr)
{
    const int terms = flags & VFA_FLAG_TERMS_MASK, reserved_cus = (flags >> 8) & 0xff;
    if (flags & ~(VFA_FLAG_TERMS_MASK | 0xff00)) return VFA_ERR_BAD_ARGUMENT;
    if (K <= 0 || N <= 0 || (terms != 0 && terms != 2 && terms != 3 && terms != 4 && terms != 6)) return VFA_ERR_BAD_ARGUMENT;
    if (N != kN || K % kChunk != 0) return VFA_ERR_UNSUPPORTED;
    if (M == 0) return 0;
    if (!workspace || workspace_bytes < vfa_collapse_gemm_workspace_bytes(K, N)) return VFA_ERR_BAD_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    static bool attr_set = false; // idempotent: a race only repeats the call
    if (!attr_set) {
        const void *fns[5] = {(const void *)collapse_gemm_kernel<3, false>, (const void *)collapse_gemm_kernel<4, false>,
                              (const void *)collapse_gemm_kernel<3, true>, (const void *)collapse_gemm_kernel<4, true>,
                              (const void *)collapse_gemm_kernel<2, true>};
        for (const void *fn : fns) {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
            if (e != hipSuccess) return (int)e;
        }
        attr_set = true;
    }
    int n_cu = 256;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            cus > 0)
            n_cu = cus;
    }
    const size_t frags = (size_t)(K / kChunk) * 8 * kStepsPerChunk * 64;
    F16Args fa = {};
    if (f16) {
        unsigned *tail = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(workspace) + (size_t)K * N * 4);
        hipLaunchKernelGGL(gemm_weight_absmax_kernel, dim3(kWmaxParts), dim3(256), 0, s, weight, tail, (size_t)K * N);
        hipLaunchKernelGGL(split_weight_f16_kernel, dim3((unsigned)((frags + 255) / 256)), dim3(256), 0, s, weight, (uint4 *)workspace, K, tail);
        fa = *f16;
        fa.wexp = reinterpret_cast<const int *>(tail) + kWmaxParts;
    } else {
        hipLaunchKernelGGL(split_weight_kernel, dim3((unsigned)((frags + 255) / 256)), dim3(256), 0, s, weight, (uint4 *)workspace, K);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const long long n_tiles = ((long long)M + kTileRows - 1) / kTileRows;
    // Persistent workgroups, one per CU: every workgroup walks `rounds` tiles; launch only as many as that takes (an even
    // load, and the CUs a ragged last round would idle stay free).  VFA_FLAG_RESERVED_CUS(n) lowers the count further when
    // that does not add a round.
    long long rounds = (n_tiles + n_cu - 1) / n_cu;
    if (reserved_cus > 0 && n_cu - reserved_cus >= 8 && (n_tiles + (n_cu - reserved_cus) - 1) / (n_cu - reserved_cus) == rounds)
        n_cu -= reserved_cus;
    long long wgs = (n_tiles + rounds - 1) / rounds;
    if (wgs > n_cu) wgs = n_cu;
    const unsigned blocks = (unsigned)wgs;
    const MaskArgs none = {nullptr, nullptr, nullptr, 1};
    if (f16) {
        if (!mask) return VFA_ERR_BAD_ARGUMENT;
        hipLaunchKernelGGL((collapse_gemm_kernel<2, true>), dim3(blocks), dim3(kThreads), kLdsBytes, s, vox, (const uint4 *)workspace, out,
                           (long long)M, K, *mask, fa);
    } else if (mask) {
        if (terms == 4)
            hipLaunchKernelGGL((collapse_gemm_kernel<4, true>), dim3(blocks), dim3(kThreads), kLdsBytes, s, vox, (const uint4 *)workspace, out,
                               (long long)M, K, *mask, fa);
        else
            hipLaunchKernelGGL((collapse_gemm_kernel<3, true>), dim3(blocks), dim3(kThreads), kLdsBytes, s, vox, (const uint4 *)workspace, out,
                               (long long)M, K, *mask, fa);
    } else {
        if (terms == 4)
            hipLaunchKernelGGL((collapse_gemm_kernel<4, false>), dim3(blocks), dim3(kThreads), kLdsBytes, s, vox, (const uint4 *)workspace, out,
                               (long long)M, K, none, fa);
        else
            hipLaunchKernelGGL((collapse_gemm_kernel<3, false>), dim3(blocks), dim3(kThreads), kLdsBytes, s, vox, (const uint4 *)workspace, out,
                               (long long)M, K, none, fa);
    }
    return (int)hipGetLastError();
}

extern "C" int vfa_collapse_gemm_f32(const float *vox, const float *weight, float *lin, void *workspace, size_t workspace_bytes,
                                     size_t M, int K, int N, int flags, void *stream)
{
    return collapse_gemm_launch(vox, weight, lin, workspace, workspace_bytes, M, K, N, flags, nullptr, stream);
}

extern "C" int vfa_collapse_gemm_relu_backward_f32(const float *vox, const float *weight, const float *bias, const float *grad_out,
                                                   float *grad_lin, float *grad_bias, void *workspace, size_t workspace_bytes,
                                                   int n_views, size_t cells, int K, int N, int flags, void *stream)
{
    if (n_views < 0 || !grad_out || !grad_lin) return VFA_ERR_BAD_ARGUMENT;
    if (n_views == 0 || cells == 0) return 0;
    if (cells < 32 || cells >= (1ull << 22)) return VFA_ERR_UNSUPPORTED; // (a 32-row block wraps over the cells at most once; 32-bit byte offsets into d out)
    const MaskArgs ma = {bias, grad_out, grad_bias, (long long)cells};
    return collapse_gemm_launch(vox, weight, grad_lin, workspace, workspace_bytes, (size_t)n_views * cells, K, N, flags, &ma, stream);
}

// The same with the product of the FUSED FRAME KERNELS (two fp16 pieces per operand under the frame's power-of-two scales, vfa_split.h):
// feat_absmax / absmax_count = the feature statistics of this scale's integral images (what the forward reduced to 2^ea), row_shift =
// the sliver shift of every row (n_views * cells bytes, from vfa_sliver_shifts_u8; NULL: none), tile_any = one byte per 128 rows: a
// row of that tile has a shift (NULL: every row is looked up).  Operands, scales,
// accumulator start and the order of the MFMA products are the forward's, so the recomputed pre-activation -- and with it the ReLU
// mask of the backward -- is the forward's bit for bit.
extern "C" int vfa_collapse_gemm_relu_backward_f16_f32(const float *vox, const float *weight, const float *bias, const float *grad_out,
                                                       float *grad_lin, float *grad_bias, void *workspace, size_t workspace_bytes,
                                                       int n_views, size_t cells, int K, int N, const unsigned *feat_absmax,
                                                       int absmax_count, const unsigned char *row_shift, const unsigned char *tile_any,
                                                       int flags, void *stream)
{
    if (n_views < 0 || !grad_out || !grad_lin || !feat_absmax || absmax_count <= 0 || (tile_any && !row_shift)) return VFA_ERR_BAD_ARGUMENT;
    if (flags & VFA_FLAG_TERMS_MASK & ~2) return VFA_ERR_BAD_ARGUMENT; // (terms 0 / 2: this entry point IS the fp16 form)
    if (n_views == 0 || cells == 0) return 0;
    if (cells < 32 || cells >= (1ull << 22)) return VFA_ERR_UNSUPPORTED;
    const MaskArgs ma = {bias, grad_out, grad_bias, (long long)cells};
    const F16Args fa = {feat_absmax, absmax_count, nullptr, row_shift, tile_any};
    return collapse_gemm_launch(vox, weight, grad_lin, workspace, workspace_bytes, (size_t)n_views * cells, K, N, flags & ~VFA_FLAG_TERMS_MASK, &ma, stream, &fa);
}
