// vfa_lateral.hip -- the producer of the path's input: the lateral branch of one feature scale,
//     relu(GroupNorm16(conv1x1(feat)))            reference vfa/model/vfanet.py:37-42 (lat8/16/32, bn8/16/32), :72-74
// as far as the integral image needs it: the 1 x 1 convolution written CHANNELS-LAST with the GroupNorm statistics gathered in
// its epilogue, and the per-(view, channel) affine (gamma * rstd, beta - mean * gamma * rstd) that the row scan of the integral
// image applies together with the ReLU (vfa_integral.hip, rows_hwc_kernel).  The normalised lateral map itself is never stored,
// and no NCHW copy of the convolution output exists.
//
// The convolution is a (pixels x K) . (K x 256) product in fp32 ON THE MATRIX PIPE: v_mfma_f32_32x32x2_f32 is an exact fp32
// FMA chain over k (ascending), i.e. the arithmetic of a plain fp32 dot product -- sgemm-class by construction, no split
// operands.  It is 1/16 of the bf16 rate (155 TFLOP/s), which for K = 128 .. 512 -> 256 still sits at the HBM time of the
// operands (bench frame: 11.5 GFLOP = 75 us against 225 MB = 28 us), so nothing cheaper is worth its rounding.
//
// Tiling: a workgroup of four waves takes 128 consecutive pixels of one view; wave w owns pixels 32 w .. 32 w + 31 and ALL 256
// output channels (eight 32 x 32 accumulators = 128 registers; the CHANNELS are the rows of the MFMA, so that a lane ends up with
// four consecutive channels of one pixel per register quad: 16-byte channels-last stores) -- or, on the small maps, 64 / 32 pixels with the channels split
// over two / four waves (lateral_conv_kernel<PXW>).  A operands come straight from HBM: the NCHW input is
// k-major, so lane (pixel p, k half) loads feat[k][p] -- 32 consecutive floats per half wave, no LDS, no transpose.  B operands
// (the weight, reference layout (256, K)) are staged through LDS in chunks of 32 k, transposed on the way in ([k][co], so that a
// lane group reads 32 consecutive output channels), double-buffered.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vfa_hip.h"
#include "vfa_geom.h"

namespace {
using namespace vfa_dev;

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kCo = 256;        // output channels (the path's channel count)
constexpr int kCoPad = kCo + 1;
constexpr int kGroups = 16;     // GroupNorm(16, 256): 16 channels per group
constexpr int kTilePx = 128;    // pixels per workgroup
constexpr int kKc = 32;         // k per LDS chunk
constexpr int kThreads = 256;

struct LateralArgs {
    const float *feat;    // (n_views, K, H * W)
    const float *weight;  // (256, K)
    const float *bias;    // (256)
    float *out;           // (n_views, H * W, 256)
    double *partial;      // (n_views, 16 groups, parts, 2): sum, sum of squares of 32 pixels x 16 channels; parts = blocks * PXW
    int K, HW, blocks;
};

// PXW = 32-pixel blocks per workgroup (4, 2 or 1): wave w owns pixel block w % PXW and 8 PXW / 4 of the eight 32-channel
// blocks.  PXW = 4: a wave holds all 256 channels of its pixels (the A operand is loaded once); the small maps of strides 16
// and 32 take 2 or 1 so that the launch still has a few hundred workgroups.
template <int PXW>
__global__ __launch_bounds__(kThreads, 2) void lateral_conv_kernel(LateralArgs a)
{
    constexpr int CBW = 8 * PXW / 4; // channel blocks per wave: 8, 4, 2
    __shared__ float s_w[2][kKc][kCoPad]; // [buffer][k][co]: 64.25 KB (rows padded by one bank: the transposing stores)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int col = lane & 31, kh = lane >> 5;
    const int v = blockIdx.y, blk = blockIdx.x;
    const int pbw = wave % PXW, cb0 = (wave / PXW) * CBW;
    const int p0 = (blk * PXW + pbw) * 32;
    const int px = min(p0 + col, a.HW - 1); // (tail: a clamped pixel is computed and thrown away)
    const float *src = a.feat + (size_t)v * a.K * a.HW + px;
    // weight staging: eight lanes read one 128-byte piece of a weight row (32 k of one output channel), a wave eight rows
    const int wk = (tid & 7) * 4, wco = tid >> 3; // k offset inside the chunk, output channel 32 j + wco
    const float *wrow = a.weight + (size_t)wco * a.K + wk;

    f32x16 acc[CBW];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[cb][i] = 0.0f;

    float4 wreg[8];
    float areg[16], anext[16];
    auto load_w = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) wreg[j] = *reinterpret_cast<const float4 *>(wrow + (size_t)(32 * j) * a.K + k0);
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s_w[buf][wk + 0][32 * j + wco] = wreg[j].x; s_w[buf][wk + 1][32 * j + wco] = wreg[j].y;
            s_w[buf][wk + 2][32 * j + wco] = wreg[j].z; s_w[buf][wk + 3][32 * j + wco] = wreg[j].w;
        }
    };
    auto load_a = [&](float *dst, int k0) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) dst[kk] = src[(size_t)(k0 + 2 * kk + kh) * a.HW];
    };

    const int chunks = a.K / kKc;
    load_w(0);
    load_a(areg, 0);
    store_w(0);
    __syncthreads();
    for (int c = 0; c < chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < chunks) { // the next chunk's operands: requested now, used after this chunk's MFMAs
            load_w((c + 1) * kKc);
            load_a(anext, (c + 1) * kKc);
        }
        // B operands one k pair ahead of the MFMAs that take them (an LDS round trip in front of every pair of MFMAs otherwise)
        const float *wl = &s_w[buf][kh][cb0 * 32 + col];
        float bcur[CBW], bnxt[CBW];
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) bcur[cb] = wl[cb * 32];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            if (kk + 1 < 16) {
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb) bnxt[cb] = wl[(2 * (kk + 1)) * kCoPad + cb * 32];
            }
            __builtin_amdgcn_sched_barrier(0); // (the scheduler otherwise sinks every read to just in front of its MFMA)
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bcur[cb], areg[kk], acc[cb], 0, 0, 0); // D[channel][pixel]
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) bcur[cb] = bnxt[cb];
        }
        if (c + 1 < chunks) {
            store_w(buf ^ 1); // (the buffer last read two chunks ago: every wave passed the barrier of chunk c - 1 since)
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) areg[kk] = anext[kk];
        }
        __syncthreads();
    }

    // epilogue: + bias, channels-last store, GroupNorm partial sums (double: the variance is a difference of two of them).
    // The weight is the MFMA's row operand, so lane (col, kh) holds PIXEL p0 + col and register i of block cb is channel
    // 32 cb + (i & 3) + 8 (i >> 2) + 4 kh: four consecutive channels per i >> 2 -- one 16-byte store each (with the pixels as rows a
    // lane held one channel of 16 pixels and stored 4 bytes at a time: 70 of the 255 us of the stride-8 map).
    const int p = p0 + col;
    const bool on = p < a.HW;
    float *orow = a.out + ((size_t)v * a.HW + (on ? p : 0)) * kCo;
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
            if (on) {
                *reinterpret_cast<float4 *>(orow + c) = y;
                s1[g >> 1] += ((double)y.x + (double)y.y) + ((double)y.z + (double)y.w);
                s2[g >> 1] += ((double)y.x * (double)y.x + (double)y.y * (double)y.y) + ((double)y.z * (double)y.z + (double)y.w * (double)y.w);
            }
        }
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
__global__ __launch_bounds__(kWave) void lateral_stats_kernel(FinalArgs a)
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

} // namespace

extern "C" {

// 32-pixel blocks per workgroup: the largest of 4, 2, 1 that still gives the launch two workgroups per CU
static int blocks_per_workgroup(int n_views, int H, int W)
{
    const long long px32 = ((long long)H * W + 31) / 32;
    for (int pxw = 4; pxw > 1; pxw >>= 1)
        if ((long long)n_views * ((px32 + pxw - 1) / pxw) >= 512) return pxw;
    return 1;
}

size_t vfa_lateral_conv_workspace_bytes(int n_views, int H, int W)
{
    if (n_views < 0 || H <= 0 || W <= 0) return 0;
    const size_t parts = ((size_t)H * W + 31) / 32 + 4; // (32-pixel blocks, rounded up to whole workgroups)
    return (size_t)n_views * kGroups * parts * 2 * sizeof(double);
}

int vfa_lateral_conv_f32(const float *feat, const float *weight, const float *bias, const float *gamma, const float *beta, float eps,
                         float *out_hwc, float *scale, float *shift, void *workspace, size_t workspace_bytes, int n_views, int K,
                         int H, int W, void *stream)
{
    if (!feat || !weight || !bias || !gamma || !beta || !out_hwc || !scale || !shift || n_views < 0 || K <= 0 || H <= 0 || W <= 0)
        return VFA_ERR_BAD_ARGUMENT;
    if (K % kKc != 0 || ((reinterpret_cast<uintptr_t>(weight) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(out_hwc)) & 15) != 0)
        return VFA_ERR_UNSUPPORTED; // (ResNet laterals: K = 128, 256, 512; 16-byte loads / stores)
    if (n_views == 0) return 0;
    if ((long long)H * W >= (1ll << 31) - kTilePx || n_views > 65535) return VFA_ERR_UNSUPPORTED;
    const size_t need = vfa_lateral_conv_workspace_bytes(n_views, H, W);
    if (!workspace || workspace_bytes < need) return VFA_ERR_BAD_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    const int pxw = blocks_per_workgroup(n_views, H, W);
    LateralArgs a;
    a.feat = feat; a.weight = weight; a.bias = bias; a.out = out_hwc; a.partial = reinterpret_cast<double *>(workspace);
    a.K = K; a.HW = H * W; a.blocks = (a.HW + 32 * pxw - 1) / (32 * pxw);
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
