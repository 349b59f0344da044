// vfa_geom.h -- device helpers shared by the HIP translation units of the path: the box geometry of the reference
// (vfa/model/vfa_op.py:64-88, 104-106; vfa/utils.py:56-59) in its exact fp32 rounding sequence.
// Every file that includes this is compiled with -ffp-contract=off: an FMA appears only where fmaf() is written.
#ifndef VFA_GEOM_H
#define VFA_GEOM_H
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

#include "vfa_hip.h"

namespace vfa_dev {

constexpr int kWave = 64;

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
// torch.clamp / min / max propagate NaN; ordered comparisons do that for free.
__device__ __forceinline__ float clamp_t(float v, float lo, float hi)
{
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}
__device__ __forceinline__ float min_t(float a, float b) { return (a != a || a < b) ? a : b; }
__device__ __forceinline__ float max_t(float a, float b) { return (a != a || a > b) ? a : b; }

__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).  Give every XCD one
// contiguous eighth of the logical work so that neighbouring boxes -- whose image footprints overlap --
// are served by the same L2.  Speed only; any placement is correct.
__device__ __forceinline__ long long xcd_contiguous(long long block, long long per_xcd)
{
    return (block & 7) * per_xcd + (block >> 3);
}

// ------------------------------------------------------------------------------------------------
// box parameters                                           reference vfa_op.py:64-88, 104-106; utils.py:56-59
// ------------------------------------------------------------------------------------------------
struct BoxGeom {
    const float *calibs;     // (n_views, 12)
    const float *grid;       // (n_cells, 3)
    const float *z_layers;   // (nl)
    const float *corner_off; // (8, 3)
    int conv_kind;
    float img_w, img_h;
    float cmin, cmax;
};

// Normalised image coordinates of cube corner k of (cell, layer) seen by `P` (3x4, row-major).
// Every operation is a separately rounded fp32 op in the reference's order.
__device__ __forceinline__ void project_corner(const BoxGeom &g, const float *__restrict__ P, float gx, float gy,
                                               float gz, int k, float &nu, float &nv)
{
    float x = gx + g.corner_off[k * 3 + 0];
    float y = gy + g.corner_off[k * 3 + 1];
    float z = gz + g.corner_off[k * 3 + 2];
    if (g.conv_kind == VFA_CONV_MULTIVIEWX) {
        x = x / 40.0f; y = y / 40.0f; z = z / 40.0f;
    } else if (g.conv_kind == VFA_CONV_WILDTRACK) {
        x = x * 2.5f; x = x - 300.0f;
        y = y * 2.5f; y = y - 900.0f;
        z = z * 2.5f;
    }
    float h[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float a0 = P[r * 4 + 0] * x, a1 = P[r * 4 + 1] * y, a2 = P[r * 4 + 2] * z;
        float s = a0 + a1;
        s = s + a2;
        h[r] = s + P[r * 4 + 3];
    }
    const float u = h[0] / h[2], w = h[1] / h[2];
    nu = (2.0f * u) / g.img_w; nu = nu - 1.0f; nu = clamp_t(nu, g.cmin, g.cmax);
    nv = (2.0f * w) / g.img_h; nv = nv - 1.0f; nv = clamp_t(nv, g.cmin, g.cmax);
}

__device__ __forceinline__ float box_area(float l, float t, float r, float b, int Hf, int Wf)
{
    const float dx = r - l, dy = b - t;
    float a = dx * dy;
    a = a * (float)Hf;
    a = a * (float)Wf;
    a = a + (float)1e-6;
    return a;
}
__device__ __forceinline__ bool box_visible(float a, int Hf, int Wf)
{
    return (a > (float)1e-6) && (a < (float)((double)(Hf * Wf) * 0.3));
}

// One axis of F.grid_sample's bilinear set-up (align_corners=False): pixel coordinate by ONE fma,
// i0 = floor, hi = weight of tap i0+1, lo = weight of tap i0.              SURVEY.md A.5
struct Axis { int i0; float hi, lo; };
__device__ __forceinline__ Axis make_axis(float g, int size)
{
    const float X = fmaf(g + 1.0f, (float)size / 2.0f, -0.5f);
    const float f = floorf(X);
    Axis a;
    a.i0 = (int)f;
    a.hi = X - f;
    a.lo = 1.0f - a.hi;
    return a;
}

__device__ __forceinline__ void bilinear_weights(float (&w)[4], const Axis &ax, const Axis &ay)
{
    w[0] = ay.lo * ax.lo; // nw
    w[1] = ay.lo * ax.hi; // ne
    w[2] = ay.hi * ax.lo; // sw
    w[3] = ay.hi * ax.hi; // se
}

// (((lt + rb) - rt) - lb) / area                                                         (A.6)
// The quotient must be the correctly rounded IEEE quotient (the reference divides).  All channels of a box divide by
// the same area, so the reciprocal r = RN(1/area) is formed once per box and each channel runs two Markstein
// corrections:  q0 = RN(v r); q1 = RN(q0 + (v - area q0) r); q = RN(q1 + (v - area q1) r), residuals exact by FMA.
// q1 is within half an ulp (+ o(ulp)) of v/area, i.e. faithful, and for a faithful q1 and r = RN(1/area) the last
// step returns RN(v/area) (Markstein's theorem).  Holds while no intermediate leaves the normal range: v = 0 or
// 2^-100 < |v / area| < 2^100, always true for feature maps (checked against true division: tools/check_division.c).
__device__ __forceinline__ float box_mean(float lt, float rb, float rt, float lb, float area, float rcp)
{
    float v = lt + rb;
    v = v - rt;
    v = v - lb;
    const float q0 = v * rcp;
    const float q1 = fmaf(fmaf(-area, q0, v), rcp, q0);
    return fmaf(fmaf(-area, q1, v), rcp, q1);
}

// The same quotient with a power of two folded in (the fused frame kernels: the fp16 operand split wants v / area * 2^k):
// rs = RN(1 / area) 2^k and as = area 2^-k are exact scalings, every intermediate is 2^k times the one of `box_mean`, so the
// result is RN(v / area) 2^k bit for bit (no intermediate leaves the normal range for |k| <= 80 on feature maps).
// A masked box passes rs = its masked value (0, or NaN for a NaN box): q0 = 0, the residuals multiply by 0 -> 0 (or NaN).
__device__ __forceinline__ float box_quotient_scaled(float v, float as, float rs)
{
    const float q0 = v * rs;
    const float q1 = fmaf(fmaf(-as, q0, v), rs, q0);
    return fmaf(fmaf(-as, q1, v), rs, q1);
}

// How far rounding noise can lift a voxel feature above the feature map's largest value (the fp16 operand split of the fused
// kernels, vfa_split.h, needs a bound on |vox| before pooling).  vox = N / area with N = ((lt + rb) - rt) - lb (vfa_op.py:
// 118-119), area = 4 x pixel area + 1e-6 (:104-105).  In exact arithmetic |N| <= absmax x pixel area, i.e. |N / area| <= absmax / 4.
// In fp32, with u = 2^-24 and S = sum |feature| <= absmax Hf Wf:  an integral-image entry is off by <= 2 u S (two cumsums, each
// element rounded once: vfa_op.py:172-173), a bilinear sample by <= 8 u S (its taps, its rounded weights, its four roundings),
// the combination by <= 6 u S more:  |error of N| <= 38 u S < 2^-18 S.  So
//     |vox| / absmax  <=  1/4 + 2^-18 Hf Wf / area              (an honest box: ~1/4; a sliver of area 1e-5 on a 90 x 160 map: 5 500)
// and the reference KEEPS such slivers (visible = area > 1e-6, :106).  `sliver_shift` is the number of binary places the
// frame kernels take out of a (tile, view, scale) item that holds such a box -- 0 unless the bound reaches 1, else
// floor(log2(bound)) + 1 -- so that the scaled voxel features of the item stay below 2^(kExpA + 1) whatever the noise does.
__device__ __forceinline__ int sliver_shift(float area, int Hf, int Wf)
{
    const float bound = 0.25f + ((float)Hf * (float)Wf * 0x1p-18f) / area;
    if (!(bound >= 1.0f)) return 0; // (also a NaN area: such a box is not visible)
    const int e = (int)((__float_as_uint(bound) >> 23) & 0xffu) - 127 + 1;
    return e > 48 ? 48 : e;
}

// Zero `bytes` (a multiple of 16, 16-byte aligned) with a KERNEL on the stream.  The frame entry points clear their view masks,
// tickets and counters with this instead of hipMemsetAsync: inside a captured hipGraph the runtime's memset node was seen to run
// unordered against the kernel node behind it once the process had used a few more streams (ROCm 7.2: masks that
// frame_records_kernel had already OR-ed were zeroed again -- tiles vanished from every later replay of the graph; found by
// tests/test_pipe_frame.py::test_captured_frames_keep_a_workspace_of_their_own).  Kernel -> kernel edges of a graph hold.
static __global__ __launch_bounds__(256) void zero_fill_kernel(uint4 *p, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
inline hipError_t zero_fill(void *p, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return hipSuccess;
    if ((bytes & 15u) || ((size_t)p & 15u)) return hipMemsetAsync(p, 0, bytes, s); // (never the case for the workspaces: 256-byte regions)
    const size_t n16 = bytes / 16;
    const unsigned blocks = (unsigned)((n16 + 255) / 256 < 1024 ? (n16 + 255) / 256 : 1024);
    hipLaunchKernelGGL(zero_fill_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<uint4 *>(p), n16);
    return hipGetLastError();
}

} // namespace vfa_dev
#endif // VFA_GEOM_H
