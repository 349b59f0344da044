// vfa_lateral.hip -- the producer of the path's input: the lateral branch of one feature scale,
//     relu(GroupNorm16(conv1x1(feat)))            reference vfa/model/vfanet.py:37-42 (lat8/16/32, bn8/16/32), :72-74
// as far as the integral image needs it: the 1 x 1 convolution written CHANNELS-LAST with the GroupNorm statistics gathered in
// its epilogue, and the per-(view, channel) affine (gamma * rstd, beta - mean * gamma * rstd) that the row scan of the integral
// image applies together with the ReLU (vfa_integral.hip, rows_hwc_kernel).  The normalised lateral map itself is never stored,
// and no NCHW copy of the convolution output exists.
//
// The convolution is a (pixels x K) . (K x 256) product at the width of an sgemm: SIX bf16 MFMA products of a three-piece split of
// both operands with fp32 accumulation (lateral_body; 6 / 16 of the matrix-pipe time of v_mfma_f32_32x32x2_f32, which the first form
// of this kernel used).
//
// Tiling: a workgroup of four waves takes 32 PXW consecutive pixels of one view (PXW = 2 on the large map, 1 on the small ones:
// blocks_per_workgroup); wave w owns pixel block w % PXW and 8 PXW / 4 of the eight 32-channel blocks.  The CHANNELS are the rows
// of the MFMA, so that a lane ends up with four consecutive channels of one pixel per register quad: channels-last stores, through a
// wave-private LDS tile.  The pixel operand comes straight from HBM two 16-k chunks ahead (the NCHW input is k-major: lane (pixel p,
// k half) loads feat[k][p], 32 consecutive floats per half wave) and is split in registers; the weight comes pre-split in MFMA
// fragment order (lateral_split_weight_kernel) and is streamed by every wave from L2 / L1: no LDS staging, no workgroup barrier.
// vfa_lateral_convs_f32 runs the scales of a frame as ONE launch of each of the three kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vfa_hip.h"
#include "vfa_geom.h"

namespace {
using namespace vfa_dev;

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kCo = 256;        // output channels (the path's channel count)
constexpr int kGroups = 16;     // GroupNorm(16, 256): 16 channels per group
constexpr int kTilePx = 128;    // pixels per workgroup
constexpr int kKc = 16;         // k per chunk (one v_mfma_f32_32x32x16_bf16 deep)
constexpr int kMaxK = 1024;     // input channels the workspace has room for (ResNet laterals: 128, 256, 512)
constexpr int kThreads = 256;

struct LateralArgs {
    const float *feat;    // (n_views, K, H * W)
    const float *weight;  // (256, K)
    const float *bias;    // (256)
    float *out;           // (n_views, H * W, 256)
    double *partial;      // (n_views, 16 groups, parts, 2): sum, sum of squares of 32 pixels x 16 channels; parts = blocks * PXW
    const uint4 *wfrag;   // the weight as three bf16 planes in MFMA fragment order (lateral_split_weight_kernel)
    int K, HW, blocks;
};

// The weight as three bf16 planes (x = p0 + p1 + p2, exact to 2^-25 |x|) in MFMA fragment order, once per call:
//   frag[((chunk * 8 + cb) * 3 + plane) * 64 + lane] (16 B) = W[co = 32 cb + (lane & 31)][k = 16 chunk + 8 (lane >> 5) + j], j = 0..7
__device__ __forceinline__ void lateral_split_weight(const float *__restrict__ w, uint4 *__restrict__ frag, int K)
{
    const int idx = blockIdx.x * 256 + threadIdx.x; // (chunk, cb, lane)
    if (idx >= (K / 16) * 8 * 64) return;
    const int lane = idx & 63, cb = (idx >> 6) & 7, chunk = idx >> 9;
    const float *src = w + (size_t)(32 * cb + (lane & 31)) * K + 16 * chunk + 8 * (lane >> 5);
    union { __bf16 b[8]; uint4 u; } p0, p1, p2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = src[j];
        p0.b[j] = (__bf16)x;
        const float r1 = x - (float)p0.b[j];
        p1.b[j] = (__bf16)r1;
        p2.b[j] = (__bf16)(r1 - (float)p1.b[j]);
    }
    uint4 *o = frag + (size_t)(chunk * 8 + cb) * 3 * 64 + lane;
    o[0] = p0.u; o[64] = p1.u; o[128] = p2.u;
}
__global__ __launch_bounds__(256) void lateral_split_weight_kernel(const float *__restrict__ w, uint4 *__restrict__ frag, int K)
{
    lateral_split_weight(w, frag, K);
}
constexpr int kMaxMaps = 3; // feature scales of a frame in one launch (vfa_lateral_convs_f32)
struct SplitBatch { const float *w[kMaxMaps]; uint4 *frag[kMaxMaps]; int K[kMaxMaps]; };
__global__ __launch_bounds__(256) void lateral_split_weight_batched_kernel(SplitBatch b) // blockIdx.y = map
{
    const int m = blockIdx.y;
    lateral_split_weight(m == 0 ? b.w[0] : (m == 1 ? b.w[1] : b.w[2]), m == 0 ? b.frag[0] : (m == 1 ? b.frag[1] : b.frag[2]),
                         m == 0 ? b.K[0] : (m == 1 ? b.K[1] : b.K[2]));
}

// PXW = 32-pixel blocks per workgroup (4, 2 or 1): wave w owns pixel block w % PXW and 8 PXW / 4 of the eight 32-channel
// blocks.  PXW = 4: a wave holds all 256 channels of its pixels (the pixel operand is loaded once); the small maps of strides 16
// and 32 take 2 or 1 so that the launch still has a few hundred workgroups.
//
// Arithmetic (round 4): the fp32 product as SIX bf16 MFMA products of a three-piece split of both operands -- p0 p0, p0 p1, p1 p0,
// p0 p2, p2 p0, p1 p1: everything down to 2^-16 of the largest, what is dropped is <= 2^-23 of a product (vfa_pipe.hip,
// VFA_FLAG_TERMS 6; tests/test_split_arithmetic.py: <= 3e-8 normwise from the split) -- with fp32 accumulation: the class of an
// sgemm, no scale needed (bf16 has fp32's exponent range), at 6 / 16 of the matrix-pipe time of v_mfma_f32_32x32x2_f32 (whose
// 64 cycles per instruction left the three convolutions of a bench frame at 221 us against 75 us of pipe time and 28 us of HBM
// time).  The pixel operand is split in registers as it arrives; the weight planes come pre-split (lateral_split_weight_kernel)
// through LDS, one 16-k chunk (24 KB) ahead.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kChunkK = 16;                    // k per MFMA (v_mfma_f32_32x32x16_bf16)
constexpr int kFragChunk = 8 * 3 * 64;         // uint4 per 16-k chunk of the split weight: 8 channel blocks x 3 planes x 64 lanes
template <int PXW>
__device__ __forceinline__ void lateral_body(const LateralArgs &a, const int v, const int blk, float4 (*s_out)[32 * 9])
{
    constexpr int CBW = 8 * PXW / 4; // channel blocks per wave: 8, 4, 2
    constexpr int D = CBW >= 4 ? 4 : 2; // weight fragments are requested D channel blocks ahead of the products that take them
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int col = lane & 31, kg = lane >> 5;
    const int pbw = wave % PXW, cb0 = (wave / PXW) * CBW;
    const int p0 = (blk * PXW + pbw) * 32;
    const int px = min(p0 + col, a.HW - 1); // (tail: a clamped pixel is computed and thrown away)
    const float *src = a.feat + (size_t)v * a.K * a.HW + (size_t)(8 * kg) * a.HW + px;

    f32x16 acc[CBW];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[cb][i] = 0.0f;

    // No LDS and no workgroup barrier: every wave streams the pre-split weight fragments it needs straight from L2 (fragment
    // order: one coalesced 1 KiB load per (channel block, plane); the four waves of a workgroup ask for the same lines within a few
    // hundred cycles, so most of it is a vector-L1 hit) and the pixel operand from HBM, two chunks ahead.  (Staged through LDS with
    // a barrier per 16-k chunk the small maps -- 12 or 24 MFMAs per wave and chunk -- spent their time at the barrier: 79 and 91 us
    // for the stride-32 / stride-16 maps of the bench frame.)
    const int chunks = a.K / kChunkK;
    const bf16x8 *wbase = reinterpret_cast<const bf16x8 *>(a.wfrag) + (size_t)cb0 * 3 * 64 + lane;
    bf16x8 fr[D][3];
    auto load_frag = [&](int slot, int c, int cb) { // fragments of channel block cb0 + cb of chunk c (clamped: the last ones are re-read, unused)
        const bf16x8 *p = wbase + (size_t)min(c, chunks - 1) * kFragChunk + cb * 192;
        fr[slot][0] = p[0]; fr[slot][1] = p[64]; fr[slot][2] = p[128];
    };
    constexpr int AD = 2;              // the pixel operand is requested AD chunks ahead (4: 256 registers and spills at PXW = 4, no faster)
    float an[AD][8];                   // the pixel operand of chunks c + 1 .. c + AD, on their way from HBM (ring: chunk c + 1 + d in slot (c + d) % AD)
    bf16x8 a0, a1, a2;                 // ... and of chunk c, split: lane (pixel col, k group kg), k = 8 kg + j
    auto load_a = [&](float (&dst)[8], int c) {
        const int cc = min(c, chunks - 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[j] = src[(size_t)(cc * kChunkK + j) * a.HW];
    };
    auto split_a = [&](const float (&x)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const __bf16 q0 = (__bf16)x[j];
            const float r1 = x[j] - (float)q0;
            const __bf16 q1 = (__bf16)r1;
            a0[j] = q0; a1[j] = q1; a2[j] = (__bf16)(r1 - (float)q1);
        }
    };
    load_a(an[AD - 1], 0);
#pragma unroll
    for (int d = 0; d < AD - 1; ++d) load_a(an[d], 1 + d);
#pragma unroll
    for (int d = 0; d < D; ++d) load_frag(d, 0, d);
    split_a(an[AD - 1]);
    load_a(an[AD - 1], AD);
    // (chunks = K / 16 is a multiple of AD for K = 128, 256, 512: the ring position is a compile-time fact of the unrolled body)
    for (int c4 = 0; c4 < chunks; c4 += AD) {
#pragma unroll
      for (int ci = 0; ci < AD; ++ci) {
        const int c = c4 + ci;
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) {
            const bf16x8 w0 = fr[cb % D][0], w1 = fr[cb % D][1], w2 = fr[cb % D][2];
            // the fragments D blocks ahead (of this chunk or of the next one) into the registers just read
            if (cb + D < CBW) load_frag(cb % D, c, cb + D);
            else load_frag(cb % D, c + 1, cb + D - CBW);
            // D[channel][pixel]: the weight is the row operand; small terms first
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, a2, acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, a0, acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, a1, acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, a1, acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, a0, acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, a0, acc[cb], 0, 0, 0);
        }
        split_a(an[ci]);          // chunk c + 1 (requested AD chunks ago)
        load_a(an[ci], c + 1 + AD); // ... and its slot takes chunk c + 1 + AD
      }
    }

    // epilogue: + bias, channels-last store, GroupNorm partial sums (double: the variance is a difference of two of them).
    // The weight is the MFMA's row operand, so lane (col, kg) holds PIXEL p0 + col and register i of block cb is channel
    // 32 cb + (i & 3) + 8 (i >> 2) + 4 kg: four consecutive channels per i >> 2 -- one 16-byte store each (with the pixels as rows a
    // lane held one channel of 16 pixels and stored 4 bytes at a time: 70 of the 255 us of the stride-8 map).
    // Stores go through a wave-private LDS tile (32 pixels x 32 channels): straight from the accumulators a store instruction
    // wrote 64 scattered 16-byte pieces (one per lane, 1 KiB apart) -- with the MFMAs switched off the stride-8 convolution of the
    // bench frame still took 69 of its 94 us --; from the tile it writes the 128 contiguous bytes of eight pixels.
    const int kh = kg;
    const int p = p0 + col;
    const bool on = p < a.HW;
    float4 *tile = s_out[wave];
    const int parts = a.blocks * PXW;
#pragma unroll
    for (int cbi = 0; cbi < CBW; ++cbi) {
        const int cb = cb0 + cbi;
        double s1[2] = {0.0, 0.0}, s2[2] = {0.0, 0.0}; // the block's two groups: channels 0..15 (i < 8) and 16..31
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = cb * 32 + 8 * g + 4 * kh;
            const float4 bc = *reinterpret_cast<const float4 *>(a.bias + c);
            const float4 y = make_float4(acc[cbi][4 * g + 0] + bc.x, acc[cbi][4 * g + 1] + bc.y, acc[cbi][4 * g + 2] + bc.z,
                                         acc[cbi][4 * g + 3] + bc.w);
            tile[col * 9 + 2 * g + kh] = y;
            if (on) {
                const double y0 = (double)y.x, y1 = (double)y.y, y2 = (double)y.z, y3 = (double)y.w;
                s1[g >> 1] += (y0 + y1) + (y2 + y3);
                s2[g >> 1] += __builtin_fma(y0, y0, y1 * y1) + __builtin_fma(y2, y2, y3 * y3);
            }
        }
        // (one wave: LDS operations complete in order; the wait makes the tile visible to the reads below)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2) {
            const int pl = 8 * r2 + (lane >> 3), q = lane & 7; // pixel of the block, channel quad
            if (p0 + pl < a.HW)
                *reinterpret_cast<float4 *>(a.out + ((size_t)v * a.HW + p0 + pl) * kCo + cb * 32 + 4 * q) = tile[pl * 9 + q];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (the tile is overwritten by the next channel block)
        // over the wave's 32 pixels x 2 channel halves: a fixed butterfly
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                s1[q] += __shfl_xor(s1[q], d);
                s2[q] += __shfl_xor(s2[q], d);
            }
        }
        if (lane < 2) {
            const int g = cb * 2 + lane;
            double *pp = a.partial + (((size_t)v * kGroups + g) * (size_t)parts + (size_t)blk * PXW + pbw) * 2;
            pp[0] = lane == 0 ? s1[0] : s1[1];
            pp[1] = lane == 0 ? s2[0] : s2[1];
        }
    }
}

template <int PXW>
__global__ __launch_bounds__(kThreads, 2) void lateral_conv_kernel(LateralArgs a)
{
    __shared__ float4 s_out[kThreads / 64][32 * 9]; // [wave][pixel][8 channel quads + 1 pad]
    lateral_body<PXW>(a, blockIdx.y, blockIdx.x, s_out);
}
// The scales of a frame in ONE launch (vfa_lateral_convs_f32): workgroup id -> (map, view, block), the maps by DESCENDING K -- the
// deep maps are few workgroups with long chains (K = 512: 32 chunks each; 35 us as a launch of their own, latency-bound): they start
// first and the many short workgroups of the large shallow map run beside and behind them.  Same body, same bits as the per-map
// launches.
struct LateralBatch { LateralArgs m[kMaxMaps]; int pxw[kMaxMaps]; int end[kMaxMaps]; };
__global__ __launch_bounds__(kThreads, 2) void lateral_conv_batched_kernel(LateralBatch b)
{
    __shared__ float4 s_out[kThreads / 64][32 * 9];
    const int id = blockIdx.x;
    const int map = id >= b.end[1] ? 2 : (id >= b.end[0] ? 1 : 0);
    const int local = id - (map == 0 ? 0 : (map == 1 ? b.end[0] : b.end[1]));
    const LateralArgs a = map == 0 ? b.m[0] : (map == 1 ? b.m[1] : b.m[2]);
    const int pxw = map == 0 ? b.pxw[0] : (map == 1 ? b.pxw[1] : b.pxw[2]);
    const int v = local / a.blocks, blk = local - v * a.blocks;
    // (blocks_per_workgroup chooses 2 or 1; the 4-block body -- 242 registers, two waves per SIMD -- stays out of this kernel: 160, three)
    if (pxw == 2) lateral_body<2>(a, v, blk, s_out);
    else lateral_body<1>(a, v, blk, s_out);
}

struct FinalArgs {
    const double *partial;
    const float *gamma, *beta; // (256)
    float *scale, *shift;      // (n_views, 256)
    int parts;                 // 32-pixel blocks, rounded up to whole workgroups
    double count;              // H * W * 16
    float eps;
};
// one wave per (view, group): lane j adds the partial sums j, j + 64, ... (64 chains in flight: ONE chain of 452 dependent loads per
// thread took 200 us on the bench frame, 16 chains per group 15), then a fixed butterfly (no atomics: the same bits on every run), then
// lanes 0..15 write scale = gamma * rstd, shift = beta - mean * scale of the group's channels (the affine of nn.GroupNorm,
// vfanet.py:40-42)
__device__ __forceinline__ void lateral_stats(const FinalArgs &a)
{
    const int v = blockIdx.x / kGroups, g = blockIdx.x % kGroups, j = threadIdx.x;
    const double *pp = a.partial + ((size_t)v * kGroups + g) * (size_t)a.parts * 2;
    double s1 = 0.0, s2 = 0.0;
    for (int i = j; i < a.parts; i += kWave) {
        s1 += pp[2 * i];
        s2 += pp[2 * i + 1];
    }
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        s1 += __shfl_xor(s1, d);
        s2 += __shfl_xor(s2, d);
    }
    if (j >= kCo / kGroups) return;
    const int c = g * (kCo / kGroups) + j;
    const double mean = s1 / a.count;
    double var = s2 / a.count - mean * mean; // (biased, like nn.GroupNorm)
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)a.eps);
    const float sc = (float)((double)a.gamma[c] * rstd);
    a.scale[(size_t)v * kCo + c] = sc;
    a.shift[(size_t)v * kCo + c] = (float)((double)a.beta[c] - mean * (double)a.gamma[c] * rstd);
}
__global__ __launch_bounds__(kWave) void lateral_stats_kernel(FinalArgs a) { lateral_stats(a); }
struct FinalBatch { FinalArgs m[kMaxMaps]; };
__global__ __launch_bounds__(kWave) void lateral_stats_batched_kernel(FinalBatch b) // blockIdx.y = map
{
    const FinalArgs a = blockIdx.y == 0 ? b.m[0] : (blockIdx.y == 1 ? b.m[1] : b.m[2]);
    lateral_stats(a);
}

} // namespace

extern "C" {

// 32-pixel blocks per workgroup.  Measured on the three maps of the bench frame (rocprofv3, 7 cameras): stride 8 (K = 128) 79 / 68 /
// 72 us at 4 / 2 / 1 blocks, strides 16 and 32 (K = 256, 512) 28-32 us at 1 block against 33-36 at 2: two blocks where that still
// gives the launch a thousand workgroups (more waves per SIMD hide more than the pixel operand loaded twice costs), else one.
static int blocks_per_workgroup(int n_views, int H, int W)
{
    const long long px32 = ((long long)H * W + 31) / 32;
    return (long long)n_views * ((px32 + 1) / 2) >= 1024 ? 2 : 1;
}

size_t vfa_lateral_conv_workspace_bytes(int n_views, int H, int W)
{
    if (n_views < 0 || H <= 0 || W <= 0) return 0;
    const size_t parts = ((size_t)H * W + 31) / 32 + 4; // (32-pixel blocks, rounded up to whole workgroups)
    // GroupNorm partial sums + the split weight (three bf16 planes of up to kMaxK input channels)
    return ((size_t)n_views * kGroups * parts * 2 * sizeof(double) + 255) / 256 * 256 + (size_t)kCo * kMaxK * 3 * 2;
}

static int lateral_check(const float *feat, const float *weight, const float *bias, const float *gamma, const float *beta,
                         float *out_hwc, float *scale, float *shift, void *workspace, size_t workspace_bytes, int n_views, int K, int H, int W)
{
    if (!feat || !weight || !bias || !gamma || !beta || !out_hwc || !scale || !shift || n_views < 0 || K <= 0 || H <= 0 || W <= 0)
        return VFA_ERR_BAD_ARGUMENT;
    if (K % (2 * kKc) != 0 || K > kMaxK || ((reinterpret_cast<uintptr_t>(weight) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(out_hwc)) & 15) != 0)
        return VFA_ERR_UNSUPPORTED; // (ResNet laterals: K = 128, 256, 512; 16-byte loads / stores)
    if ((long long)H * W >= (1ll << 31) - kTilePx || n_views > 65535) return VFA_ERR_UNSUPPORTED;
    if (n_views > 0 && (!workspace || workspace_bytes < vfa_lateral_conv_workspace_bytes(n_views, H, W))) return VFA_ERR_BAD_ARGUMENT;
    return 0;
}
static uint4 *lateral_frag_of(void *workspace, int n_views, int H, int W)
{
    const size_t parts = ((size_t)H * W + 31) / 32 + 4;
    return reinterpret_cast<uint4 *>(reinterpret_cast<unsigned char *>(workspace) + ((size_t)n_views * kGroups * parts * 2 * sizeof(double) + 255) / 256 * 256);
}

int vfa_lateral_convs_f32(int n_maps, const float *const *feats, const float *const *weights, const float *const *biases,
                          const float *const *gammas, const float *const *betas, const float *eps, float *const *outs_hwc,
                          float *const *scales, float *const *shifts, void *const *workspaces, const size_t *workspace_bytes, int n_views,
                          const int *Ks, const int *feat_hw, void *stream)
{
    if (n_maps < 1 || n_maps > kMaxMaps || !feats || !weights || !biases || !gammas || !betas || !eps || !outs_hwc || !scales || !shifts ||
        !workspaces || !workspace_bytes || !Ks || !feat_hw)
        return VFA_ERR_BAD_ARGUMENT;
    for (int m = 0; m < n_maps; ++m) {
        const int st = lateral_check(feats[m], weights[m], biases[m], gammas[m], betas[m], outs_hwc[m], scales[m], shifts[m], workspaces[m],
                                     workspace_bytes[m], n_views, Ks[m], feat_hw[2 * m], feat_hw[2 * m + 1]);
        if (st) return st;
    }
    if (n_views == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    SplitBatch sb;
    LateralBatch lb;
    FinalBatch fb;
    int max_split = 0, total = 0;
    int order[kMaxMaps] = {0, 1, 2}; // dispatch order: descending K (stable)
    for (int i = 1; i < n_maps; ++i)
        for (int j = i; j > 0 && Ks[order[j]] > Ks[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    for (int m = 0; m < kMaxMaps; ++m) {
        const int q = m < n_maps ? order[m] : order[0];
        const int H = feat_hw[2 * q], W = feat_hw[2 * q + 1], K = Ks[q];
        const int pxw = blocks_per_workgroup(n_views, H, W);
        uint4 *frag = lateral_frag_of(workspaces[q], n_views, H, W);
        sb.w[m] = weights[q]; sb.frag[m] = frag; sb.K[m] = K;
        LateralArgs &a = lb.m[m];
        a.feat = feats[q]; a.weight = weights[q]; a.bias = biases[q]; a.out = outs_hwc[q]; a.partial = reinterpret_cast<double *>(workspaces[q]);
        a.wfrag = frag; a.K = K; a.HW = H * W; a.blocks = (a.HW + 32 * pxw - 1) / (32 * pxw);
        lb.pxw[m] = pxw;
        if (m < n_maps) {
            if ((long long)total + (long long)a.blocks * n_views >= (1ll << 31)) return VFA_ERR_UNSUPPORTED;
            total += a.blocks * n_views;
            const int sblk = ((K / 16) * 8 * 64 + 255) / 256;
            max_split = sblk > max_split ? sblk : max_split;
        }
        lb.end[m] = total; // (maps beyond n_maps: empty ranges)
        FinalArgs &f = fb.m[m];
        f.partial = a.partial; f.gamma = gammas[q]; f.beta = betas[q]; f.scale = scales[q]; f.shift = shifts[q];
        f.parts = a.blocks * pxw; f.count = (double)a.HW * (kCo / kGroups); f.eps = eps[q];
    }
    hipLaunchKernelGGL(lateral_split_weight_batched_kernel, dim3((unsigned)max_split, (unsigned)n_maps), dim3(256), 0, s, sb);
    int e = (int)hipGetLastError();
    if (e) return e;
    hipLaunchKernelGGL(lateral_conv_batched_kernel, dim3((unsigned)total), dim3(kThreads), 0, s, lb);
    e = (int)hipGetLastError();
    if (e) return e;
    hipLaunchKernelGGL(lateral_stats_batched_kernel, dim3((unsigned)n_views * kGroups, (unsigned)n_maps), dim3(kWave), 0, s, fb);
    return (int)hipGetLastError();
}

int vfa_lateral_conv_f32(const float *feat, const float *weight, const float *bias, const float *gamma, const float *beta, float eps,
                         float *out_hwc, float *scale, float *shift, void *workspace, size_t workspace_bytes, int n_views, int K,
                         int H, int W, void *stream)
{
    {
        const int st = lateral_check(feat, weight, bias, gamma, beta, out_hwc, scale, shift, workspace, workspace_bytes, n_views, K, H, W);
        if (st) return st;
    }
    if (n_views == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int pxw = blocks_per_workgroup(n_views, H, W);
    LateralArgs a;
    a.feat = feat; a.weight = weight; a.bias = bias; a.out = out_hwc; a.partial = reinterpret_cast<double *>(workspace);
    a.K = K; a.HW = H * W; a.blocks = (a.HW + 32 * pxw - 1) / (32 * pxw);
    {
        uint4 *frag = lateral_frag_of(workspace, n_views, H, W);
        hipLaunchKernelGGL(lateral_split_weight_kernel, dim3((unsigned)((K / 16) * 8 * 64 + 255) / 256), dim3(256), 0, s, weight, frag, K);
        const int e0 = (int)hipGetLastError();
        if (e0) return e0;
        a.wfrag = frag;
    }
    const dim3 grid((unsigned)a.blocks, (unsigned)n_views);
    if (pxw == 4) hipLaunchKernelGGL(lateral_conv_kernel<4>, grid, dim3(kThreads), 0, s, a);
    else if (pxw == 2) hipLaunchKernelGGL(lateral_conv_kernel<2>, grid, dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL(lateral_conv_kernel<1>, grid, dim3(kThreads), 0, s, a);
    int e = (int)hipGetLastError();
    if (e) return e;
    FinalArgs f;
    f.partial = a.partial; f.gamma = gamma; f.beta = beta; f.scale = scale; f.shift = shift;
    f.parts = a.blocks * pxw; f.count = (double)a.HW * (kCo / kGroups); f.eps = eps;
    hipLaunchKernelGGL(lateral_stats_kernel, dim3((unsigned)n_views * kGroups), dim3(kWave), 0, s, f);
    return (int)hipGetLastError();
}

} // extern "C"
