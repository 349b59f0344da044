// vfa_eval.hip -- the two consumer-side kernels behind the path:
//   * sort_vertices   the reference's only native code, a CUDA op of its AP/AOS metric (vfa/evaluation/pyeval/cuda_op/
//                     sort_vert_kernel.cu:42-134, called from IoU.py through cuda_ext.py and hard-wired to device('cuda') in
//                     evaluateAPAOS.py:79-83): orders the <= 8 valid vertices of a rectangle-rectangle intersection polygon
//                     anticlockwise for the shoelace area.  Rewritten for wave64: ONE LANE per polygon (the polygons are
//                     independent, 24 candidate vertices each, all state in registers), 256-thread blocks over b * n.
//   * bev_nms         sigmoid + 5 x 5 max-pool non-maximum suppression of the heat map (vfa/data/encoder.py:230-232, first
//                     lines of decode3d / decode2d :238, :278): conf = (maxpool5(s) == s) * s with s = sigmoid(heatmap).
// Both are index / comparison work; the arithmetic inside the comparisons keeps the reference's operand types (float
// products, double constants).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vfa_hip.h"

namespace {

constexpr int kMaxVertIdx = 9;       // MAX_NUM_VERT_IDX   sort_vert_kernel.cu:6
constexpr int kIntersectionOffset = 8; // INTERSECTION_OFFSET :7
constexpr double kEps = 1e-8;        // EPSILON            :8 (a double literal in the reference)

// "vertex 1 comes before vertex 2", vertices normalised around (0, 0): smallest on the positive x axis, growing anticlockwise
// (sort_vert_kernel.cu:15-40; the reference falls off the end -- undefined -- when a y is exactly 0: false here).
// Kept out of line: inlined twice into the selection loop, hipcc 7.2 at -O1 and above folds the second call to "false"
// (every pick after the first stayed 0 on gfx950; -O0 and the out-of-line call agree with the CPU restatement).
__device__ __noinline__ bool before(float x1, float y1, float x2, float y2)
{
    if ((double)fabsf(x1 - x2) < kEps && (double)fabsf(y2 - y1) < kEps) return false;
    if (y1 > 0 && y2 < 0) return true;
    if (y1 < 0 && y2 > 0) return false;
    const float n1 = (float)((double)(x1 * x1 + y1 * y1) + kEps);
    const float n2 = (float)((double)(x2 * x2 + y2 * y2) + kEps);
    const float d = fabsf(x1) * x1 / n1 - fabsf(x2) * x2 / n2;
    if (y1 > 0 && y2 > 0) return (double)d > kEps;
    if (y1 < 0 && y2 < 0) return (double)d < kEps;
    return false;
}

__global__ __launch_bounds__(256) void sort_vertices_kernel(const float *__restrict__ vertices, const uint8_t *__restrict__ mask,
                                                            const int *__restrict__ num_valid, int *__restrict__ idx, long long total,
                                                            int m)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x; // polygon
    if (i >= total) return;
    const float *v = vertices + i * m * 2;
    const uint8_t *mk = mask + i * m;
    int *out = idx + i * kMaxVertIdx;
    const int nv = num_valid[i];
    int pad = m - 1; // an arbitrary INVALID intersection point (the reference leaves it uninitialised when there is none)
    for (int j = kIntersectionOffset; j < m; ++j)
        if (!mk[j]) { pad = j; break; }
    if (nv < 3) { // not enough vertices
#pragma unroll
        for (int j = 0; j < kMaxVertIdx; ++j) out[j] = pad;
        return;
    }
    int order[kMaxVertIdx];
#pragma unroll
    for (int j = 0; j < kMaxVertIdx; ++j) order[j] = pad;
    // selection sort: the j-th vertex is the smallest one that is larger than the (j - 1)-th          (:68-93)
    float px = 0.0f, py = 0.0f; // previous pick
    for (int j = 0; j < nv && j < kMaxVertIdx - 1; ++j) {
        float x_min = 1.0f, y_min = (float)-kEps;
        int take = 0;
        for (int k = 0; k < m; ++k) {
            const float x = v[2 * k], y = v[2 * k + 1];
            if (mk[k] && before(x, y, x_min, y_min) && (j == 0 || before(px, py, x, y))) {
                x_min = x; y_min = y; take = k;
            }
        }
        order[j] = take;
        px = v[2 * take];
        py = v[2 * take + 1];
    }
    const int nvc = nv < kMaxVertIdx - 1 ? nv : kMaxVertIdx - 1;
    order[nvc] = order[0]; // duplicate the first index                                                (:96)
    // two identical boxes: the four corners of box 1 equal those of box 2                              (:107-121)
    if (nv == 8) {
        int counter = 0;
        for (int j = 0; j < 4; ++j)
            for (int k = 4; k < kIntersectionOffset; ++k)
                if (order[k] == order[j]) ++counter;
        if (counter == 4) {
            order[4] = order[0];
            for (int j = 5; j < kMaxVertIdx; ++j) order[j] = pad;
        }
    }
#pragma unroll
    for (int j = 0; j < kMaxVertIdx; ++j) out[j] = order[j];
}

// conf[l, w] = s if s == max over the 5 x 5 neighbourhood (padding 2, -inf outside) else 0, s = 1 / (1 + exp(-heatmap[l, w]))
__global__ __launch_bounds__(256) void bev_nms_kernel(const float *__restrict__ heat, float *__restrict__ conf, int L, int W)
{
    const int w = blockIdx.x * 32 + (threadIdx.x & 31), l = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (l >= L || w >= W) return;
    auto sig = [](float x) { return 1.0f / (1.0f + expf(-x)); };
    const float s = sig(heat[(size_t)l * W + w]);
    float mx = s;
    for (int dl = -2; dl <= 2; ++dl)
        for (int dw = -2; dw <= 2; ++dw) {
            const int ll = l + dl, ww = w + dw;
            if (ll >= 0 && ll < L && ww >= 0 && ww < W) mx = fmaxf(mx, sig(heat[(size_t)ll * W + ww]));
        }
    conf[(size_t)l * W + w] = (mx == s) ? s : 0.0f;
}

} // namespace

extern "C" {

int vfa_sort_vertices_f32(const float *vertices, const uint8_t *mask, const int *num_valid, int *idx, int b, int n, int m, void *stream)
{
    if (b < 0 || n < 0 || m <= kIntersectionOffset) return VFA_ERR_BAD_ARGUMENT;
    const long long total = (long long)b * n;
    if (total == 0) return 0;
    if ((total + 255) / 256 >= (1ll << 31)) return VFA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(sort_vertices_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vertices, mask,
                       num_valid, idx, total, m);
    return (int)hipGetLastError();
}

int vfa_bev_nms_f32(const float *heatmap, float *conf, int L, int W, void *stream)
{
    if (L < 0 || W < 0) return VFA_ERR_BAD_ARGUMENT;
    if (L == 0 || W == 0) return 0;
    hipLaunchKernelGGL(bev_nms_kernel, dim3((W + 31) / 32, (L + 7) / 8), dim3(256), 0, (hipStream_t)stream, heatmap, conf, L, W);
    return (int)hipGetLastError();
}

} // extern "C"
