// vfa_grad.hip -- the weight gradient of `collapse` for the training step:
//     g_w (256, K) (+)= g_lin^T (256, M) . vox (M, K)        autograd of nn.Linear's weight: vfa/model/vfa_op.py:59, :123 under
//                                                            vfa/trainer.py:41 (loss.backward())
// M = views x cells of a chunk (hundreds of thousands of rows), K = n_layers * 256: a product whose REDUCTION index is the long one.
// The library's fp32 GEMM runs it at 57 TFLOP/s on the bench frame (643 us per scale, 27 % of a training step); here it is six bf16
// MFMA products of a three-piece split of both operands with fp32 accumulation -- x = p0 + p1 + p2 to 2^-25 |x|, the products p0 q0,
// p0 q1, p1 q0, p0 q2, p2 q0, p1 q1 (everything down to 2^-16 of the largest; what is dropped is <= 2^-23 of a product): the width of an
// sgemm, no scale needed (bf16 has fp32's exponent range) -- the arithmetic of vfa_lateral.hip and of VFA_FLAG_TERMS 6.
//
// One persistent 512-thread workgroup per (slab of rows, 256 columns of K): wave w owns output rows 32 w .. 32 w + 31 and all 256
// columns (eight 32 x 32 accumulators).  Both operands are row-major with the reduction index as the ROW: a thread loads eight rows of
// one column (4-byte loads, a wave reads 2 x 128 contiguous bytes per instruction), splits them and writes the three 16-byte MFMA
// fragments into LDS, 16 rows of both operands per step, double-buffered; the waves read their fragments back with ds_read_b128.
// Every workgroup writes its 256 x 256 partial sum; a second kernel adds the partials in a fixed order: the same bits on every run.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vfa_hip.h"
#include "vfa_geom.h"

namespace {
using namespace vfa_dev;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kN = 256;                 // output rows = channels of g_lin (the path's channel count)
constexpr int kKTile = 256;             // columns of K per workgroup
constexpr int kStepRows = 16;           // reduction rows per step (one v_mfma_f32_32x32x16_bf16 deep)
constexpr int kThreads = 512;
constexpr int kFragBytes = 16;          // 8 bf16
// LDS image of one operand tile of a step: [plane 3][column block 8][row half 2][column 32] fragments
constexpr int kOperandBytes = 3 * 8 * 2 * 32 * kFragBytes; // 24 KiB
constexpr int kMaxParts = 1024;         // row slabs per launch at most

struct GradWArgs {
    const float *g_lin;   // (rows, 256)
    const float *vox;     // (rows, K)
    float *partial;       // (parts, K / 256, 256, 256)
    long long rows;
    int K, parts;
};

// A workgroup barrier that orders LDS accesses only: `__syncthreads()` also drains the vector-memory queue -- here the loads of the
// step after the next, requested a moment ago: a full memory round trip at every step.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 &p0, bf16x8 &p1, bf16x8 &p2)
{
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 q0 = (__bf16)x[j];
        const float r1 = x[j] - (float)q0;
        const __bf16 q1 = (__bf16)r1;
        p0[j] = q0; p1[j] = q1; p2[j] = (__bf16)(r1 - (float)q1);
    }
}

__global__ __launch_bounds__(kThreads) void grad_weight_kernel(GradWArgs a)
{
    __shared__ __align__(16) unsigned char s_a[2][kOperandBytes]; // g_lin tile: columns = output rows o
    __shared__ __align__(16) unsigned char s_b[2][kOperandBytes]; // vox tile: columns = k
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int part = blockIdx.x, kt = blockIdx.y;
    // the slab of rows of this workgroup, in whole steps
    const long long steps_all = (a.rows + kStepRows - 1) / kStepRows;
    const long long s0 = steps_all * part / a.parts, s1 = steps_all * (part + 1) / a.parts;

    // staging role: thread (column c = tid & 255, row half kh = tid >> 8) of both tiles
    const int c = tid & 255, kh = tid >> 8;
    const float *src_a = a.g_lin + c;
    const float *src_b = a.vox + (size_t)kt * kKTile + c;
    const int frag_off = (((c >> 5) * 2 + kh) * 32 + (c & 31)) * kFragBytes; // inside a plane (8 x 2 x 32 fragments = 8 KiB)
    constexpr int kPlaneBytes = 8 * 2 * 32 * kFragBytes;
    float xa[8], xb[8];
    // this thread's eight rows of step s: pointers that advance by a step (no per-load row arithmetic); only the very last step of the
    // operands can be ragged -- its rows beyond the end are read clamped and zeroed
    const float *pa = src_a + (size_t)(s0 * kStepRows + 8 * kh) * kN;
    const float *pb = src_b + (size_t)(s0 * kStepRows + 8 * kh) * a.K;
    const long long s_ragged = (a.rows % kStepRows) ? steps_all - 1 : -1;
    auto load_step = [&](long long s) {
        if (__builtin_expect(s == s_ragged, 0)) {
            const long long m0 = s * kStepRows + 8 * kh;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long long m = m0 + j;
                const long long mc = m < a.rows ? m : a.rows - 1;
                xa[j] = m < a.rows ? a.g_lin[(size_t)mc * kN + c] : 0.0f;
                xb[j] = m < a.rows ? a.vox[(size_t)mc * a.K + (size_t)kt * kKTile + c] : 0.0f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xa[j] = pa[(size_t)j * kN];
                xb[j] = pb[(size_t)j * a.K];
            }
        }
        pa += (size_t)kStepRows * kN;
        pb += (size_t)kStepRows * a.K;
    };
    auto store_step = [&](int buf) {
        bf16x8 p0, p1, p2;
        split3(xa, p0, p1, p2);
        *reinterpret_cast<bf16x8 *>(s_a[buf] + 0 * kPlaneBytes + frag_off) = p0;
        *reinterpret_cast<bf16x8 *>(s_a[buf] + 1 * kPlaneBytes + frag_off) = p1;
        *reinterpret_cast<bf16x8 *>(s_a[buf] + 2 * kPlaneBytes + frag_off) = p2;
        split3(xb, p0, p1, p2);
        *reinterpret_cast<bf16x8 *>(s_b[buf] + 0 * kPlaneBytes + frag_off) = p0;
        *reinterpret_cast<bf16x8 *>(s_b[buf] + 1 * kPlaneBytes + frag_off) = p1;
        *reinterpret_cast<bf16x8 *>(s_b[buf] + 2 * kPlaneBytes + frag_off) = p2;
    };

    f32x16 acc[8];
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[cb][i] = 0.0f;

    if (s0 < s1) {
        load_step(s0);
        store_step(0);
        if (s0 + 1 < s1) load_step(s0 + 1);
        __syncthreads();
        const int n = lane & 31, half = lane >> 5;
        const int a_off = ((wave * 2 + half) * 32 + n) * kFragBytes; // this wave's fragment of the g_lin tile: column block = wave
        for (long long s = s0; s < s1; ++s) {
            const int buf = (int)((s - s0) & 1);
            const bf16x8 a0 = *reinterpret_cast<const bf16x8 *>(s_a[buf] + 0 * kPlaneBytes + a_off);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8 *>(s_a[buf] + 1 * kPlaneBytes + a_off);
            const bf16x8 a2 = *reinterpret_cast<const bf16x8 *>(s_a[buf] + 2 * kPlaneBytes + a_off);
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const int b_off = ((cb * 2 + half) * 32 + n) * kFragBytes;
                const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(s_b[buf] + 0 * kPlaneBytes + b_off);
                const bf16x8 b1 = *reinterpret_cast<const bf16x8 *>(s_b[buf] + 1 * kPlaneBytes + b_off);
                const bf16x8 b2 = *reinterpret_cast<const bf16x8 *>(s_b[buf] + 2 * kPlaneBytes + b_off);
                // D[o][k] += sum_m g_lin[m][o] vox[m][k]: small terms first
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[cb], 0, 0, 0);
            }
            // the next step's tiles: loaded a step ago, split and stored into the other buffer now; the step after it requested
            if (s + 1 < s1) {
                store_step(buf ^ 1);
                if (s + 2 < s1) load_step(s + 2);
            }
            lds_barrier();
        }
    }
    // partial[part][kt][o][k]: register i of lane (n, half) of block cb is row (i & 3) + 8 (i >> 2) + 4 half, column n
    {
        const int n = lane & 31, half = lane >> 5;
        float *dst = a.partial + ((size_t)part * gridDim.y + kt) * kN * kKTile;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * half;
                dst[(size_t)row * kKTile + cb * 32 + n] = acc[cb][i];
            }
    }
}

// g_w[o][kt * 256 + k] (+)= sum over the parts: four interleaved chains per element (parts g, g + 4, ...), then ((c0 + c1) + c2) + c3 --
// one fixed association; a single chain of 256 dependent loads per thread took 61 us
__global__ __launch_bounds__(1024) void grad_weight_reduce_kernel(const float *partial, float *g_w, int K, int parts, int accumulate)
{
    __shared__ float s_part[4][kKTile];
    const int kt = blockIdx.y, o = blockIdx.x, k = threadIdx.x & 255, g = threadIdx.x >> 8;
    const int ktiles = gridDim.y;
    const size_t stride = (size_t)ktiles * kN * kKTile;
    const float *src = partial + ((size_t)kt * kN + o) * kKTile + k;
    float s = 0.0f;
    int p = g;
    for (; p + 12 < parts; p += 16) { // four loads in flight per chain
        const float v0 = src[(size_t)p * stride], v1 = src[(size_t)(p + 4) * stride], v2 = src[(size_t)(p + 8) * stride], v3 = src[(size_t)(p + 12) * stride];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; p < parts; p += 4) s += src[(size_t)p * stride];
    s_part[g][k] = s;
    __syncthreads();
    if (g == 0) {
        const float t = ((s_part[0][k] + s_part[1][k]) + s_part[2][k]) + s_part[3][k];
        float *dst = g_w + (size_t)o * K + kt * kKTile + k;
        *dst = accumulate ? *dst + t : t;
    }
}

inline int parts_of(long long rows, int K)
{
    int n_cu = 256;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) n_cu = cus;
    }
    const long long steps = (rows + kStepRows - 1) / kStepRows;
    long long parts = n_cu / (K / kKTile); // one workgroup per CU over (parts, K tiles)
    if (parts < 1) parts = 1;
    if (parts > steps) parts = steps;
    if (parts > kMaxParts) parts = kMaxParts;
    return (int)(parts < 1 ? 1 : parts);
}

} // namespace

extern "C" {

size_t vfa_grad_weight_workspace_bytes(long long rows, int K)
{
    if (rows < 0 || K <= 0 || K % kKTile != 0) return 0;
    return (size_t)parts_of(rows, K) * (K / kKTile) * kN * kKTile * sizeof(float);
}

int vfa_grad_weight_f32(const float *g_lin, const float *vox, float *g_w, long long rows, int K, int accumulate, void *workspace,
                        size_t workspace_bytes, void *stream)
{
    if (!g_w || rows < 0 || K <= 0) return VFA_ERR_BAD_ARGUMENT;
    if (K % kKTile != 0) return VFA_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (rows == 0) {
        if (!accumulate) return (int)hipMemsetAsync(g_w, 0, (size_t)kN * K * sizeof(float), s);
        return 0;
    }
    if (!g_lin || !vox || !workspace || workspace_bytes < vfa_grad_weight_workspace_bytes(rows, K)) return VFA_ERR_BAD_ARGUMENT;
    GradWArgs a;
    a.g_lin = g_lin; a.vox = vox; a.partial = reinterpret_cast<float *>(workspace); a.rows = rows; a.K = K; a.parts = parts_of(rows, K);
    const int ktiles = K / kKTile;
    hipLaunchKernelGGL(grad_weight_kernel, dim3((unsigned)a.parts, (unsigned)ktiles), dim3(kThreads), 0, s, a);
    int e = (int)hipGetLastError();
    if (e) return e;
    hipLaunchKernelGGL(grad_weight_reduce_kernel, dim3(kN, (unsigned)ktiles), dim3(4 * kKTile), 0, s, a.partial, g_w, K, a.parts, accumulate);
    return (int)hipGetLastError();
}

} // extern "C"
