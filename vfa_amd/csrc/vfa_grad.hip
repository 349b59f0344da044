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
//
// ... and the gradient of its INPUT,  g_vox (M, K) = g_lin (M, 256) . w (256, K)  (autograd of nn.Linear's input, the same lines): the
// reduction index is the short one here.  The weight is split once per call into MFMA fragment order (grad_input_split_kernel); a
// workgroup walks blocks of 256 rows -- wave w owns rows 32 w .. 32 w + 31 of the block and all 256 columns of its K tile --, its waves
// load their own g_lin fragments straight from memory (eight consecutive floats per lane) and split them in registers, the weight
// fragments of a 16-deep step (24 KB) go through LDS once per workgroup and step.  Same six products.
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

// ---- g_vox = g_lin . w --------------------------------------------------------------------------------------------------------
// w (256, K) as three bf16 planes in MFMA b-fragment order:
//   frag[(((kt * 16 + s) * 8 + cb) * 3 + plane) * 64 + lane] (16 B) = w[o = 16 s + 8 (lane >> 5) + j][kt * 256 + 32 cb + (lane & 31)], j = 0..7
__global__ __launch_bounds__(256) void grad_input_split_kernel(const float *__restrict__ w, uint4 *__restrict__ frag, int K)
{
    const int idx = blockIdx.x * 256 + threadIdx.x; // (kt, s, cb, lane)
    if (idx >= (K / kKTile) * 16 * 8 * 64) return;
    const int lane = idx & 63, cb = (idx >> 6) & 7, st = (idx >> 9) & 15, kt = idx >> 13;
    const float *src = w + (size_t)(16 * st + 8 * (lane >> 5)) * K + kt * kKTile + 32 * cb + (lane & 31);
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = src[(size_t)j * K];
    bf16x8 p0, p1, p2;
    split3(x, p0, p1, p2);
    bf16x8 *o = reinterpret_cast<bf16x8 *>(frag) + ((size_t)((kt * 16 + st) * 8 + cb) * 3) * 64 + lane;
    o[0] = p0; o[64] = p1; o[128] = p2;
}

struct GradXArgs {
    const float *g_lin;   // (rows, 256)
    const uint4 *wfrag;   // grad_input_split_kernel output
    float *out;           // (rows, K)
    long long rows;
    int K;
};

__global__ __launch_bounds__(kThreads) void grad_input_kernel(GradXArgs a)
{
    __shared__ __align__(16) unsigned char s_b[2][kOperandBytes]; // weight fragments of a step: [cb][plane][lane] as in wfrag
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kt = blockIdx.y;
    const int n = lane & 31, half = lane >> 5;
    const long long blocks = (a.rows + 255) / 256;
    const long long b0 = blocks * blockIdx.x / gridDim.x, b1 = blocks * (blockIdx.x + 1) / gridDim.x;
    if (b0 >= b1) return;
    // staging role: thread t copies the three 16-byte pieces t, t + 512, t + 1024 of the step's 1536 fragments
    const uint4 *wsrc = a.wfrag + (size_t)kt * 16 * 8 * 3 * 64;
    // (native vectors in named registers: an array of HIP's uint4 structs copied to LDS is compiled as a memcpy through scratch memory)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 *wsrc4 = reinterpret_cast<const u32x4 *>(wsrc);
    u32x4 wr0, wr1, wr2;
    auto load_b = [&](int st) {
        const u32x4 *p = wsrc4 + (size_t)st * (8 * 3 * 64) + tid;
        wr0 = p[0]; wr1 = p[kThreads]; wr2 = p[2 * kThreads];
    };
    auto store_b = [&](int buf) {
        u32x4 *d = reinterpret_cast<u32x4 *>(s_b[buf]) + tid;
        d[0] = wr0; d[kThreads] = wr1; d[2 * kThreads] = wr2;
    };
    // this lane's row of the block and its eight consecutive reduction indices of a step
    float xa[8];
    auto load_a = [&](long long blk, int st) {
        long long m = blk * 256 + 32 * wave + n;
        m = m < a.rows ? m : a.rows - 1; // (rows beyond the end: computed and thrown away)
        const float4 *p = reinterpret_cast<const float4 *>(a.g_lin + (size_t)m * kN + 16 * st + 8 * half);
        const float4 u = p[0], v = p[1];
        xa[0] = u.x; xa[1] = u.y; xa[2] = u.z; xa[3] = u.w; xa[4] = v.x; xa[5] = v.y; xa[6] = v.z; xa[7] = v.w;
    };
    load_b(0);
    load_a(b0, 0);
    store_b(0);
    load_b(1);
    __syncthreads();
    for (long long blk = b0; blk < b1; ++blk) {
        f32x16 acc[8];
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][i] = 0.0f;
#pragma unroll 1
        for (int st = 0; st < 16; ++st) {
            const int buf = st & 1; // (16 steps per block: the buffer parity of step 0 is the same for every block)
            bf16x8 a0, a1, a2;
            split3(xa, a0, a1, a2);
            // the row fragment of the next step (of this block or of the next one)
            // (unconditional: behind the last step the last block's first fragment is read again and never used -- a load behind a branch
            // puts its destination registers into scratch memory)
            {
                const long long nb = st + 1 < 16 ? blk : (blk + 1 < b1 ? blk + 1 : blk);
                load_a(nb, (st + 1) & 15);
            }
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const bf16x8 *fp = reinterpret_cast<const bf16x8 *>(s_b[buf]) + (cb * 3) * 64 + lane;
                const bf16x8 w0 = fp[0], w1 = fp[64], w2 = fp[128];
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, w2, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, w0, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w1, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, w1, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w0, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, w0, acc[cb], 0, 0, 0);
            }
            // the weight fragments of the next step (the same sixteen for every block): requested a step ago, stored now; the ones
            // after them requested
            store_b(buf ^ 1);
            load_b((st + 2) & 15);
            lds_barrier();
        }
        // out[m][kt * 256 + 32 cb + n]: register i of lane (n, half) is row (i & 3) + 8 (i >> 2) + 4 half of the wave's 32
        const long long m0 = blk * 256 + 32 * wave;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const long long m = m0 + (i & 3) + 8 * (i >> 2) + 4 * half;
            if (m < a.rows) {
                float *o = a.out + (size_t)m * a.K + (size_t)kt * kKTile + n;
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) o[cb * 32] = acc[cb][i];
            }
        }
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

size_t vfa_grad_input_workspace_bytes(int K)
{
    if (K <= 0 || K % kKTile != 0) return 0;
    return (size_t)kN * K * 3 * 2; // the weight as three bf16 planes
}

int vfa_grad_input_f32(const float *g_lin, const float *w, float *g_vox, long long rows, int K, void *workspace, size_t workspace_bytes,
                       void *stream)
{
    if (rows < 0 || K <= 0) return VFA_ERR_BAD_ARGUMENT;
    if (K % kKTile != 0) return VFA_ERR_UNSUPPORTED;
    if (rows == 0) return 0;
    if (!g_lin || !w || !g_vox || !workspace || workspace_bytes < vfa_grad_input_workspace_bytes(K)) return VFA_ERR_BAD_ARGUMENT;
    if (((reinterpret_cast<uintptr_t>(g_lin)) & 15) != 0) return VFA_ERR_UNSUPPORTED; // (16-byte loads of the row fragments)
    hipStream_t s = (hipStream_t)stream;
    const int ktiles = K / kKTile;
    uint4 *frag = reinterpret_cast<uint4 *>(workspace);
    hipLaunchKernelGGL(grad_input_split_kernel, dim3((unsigned)(ktiles * 16 * 8 * 64 / 256)), dim3(256), 0, s, w, frag, K);
    int e = (int)hipGetLastError();
    if (e) return e;
    int n_cu = 256;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) n_cu = cus;
    }
    const long long blocks = (rows + 255) / 256;
    long long wgs = n_cu / ktiles;
    if (wgs < 1) wgs = 1;
    if (wgs > blocks) wgs = blocks;
    GradXArgs a;
    a.g_lin = g_lin; a.wfrag = frag; a.out = g_vox; a.rows = rows; a.K = K;
    hipLaunchKernelGGL(grad_input_kernel, dim3((unsigned)wgs, (unsigned)ktiles), dim3(kThreads), 0, s, a);
    return (int)hipGetLastError();
}

} // extern "C"
