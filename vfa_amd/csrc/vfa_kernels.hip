// vfa_kernels.hip -- hand-written HIP (gfx950 / CDNA4, wave64) kernels of the multiview feature -> voxel
// projection + aggregation path, behind the C ABI of include/vfa_hip.h.
//
// Reference behaviour reproduced (file:line relative to Jiahao-Ma/VFA):
//   vfa/model/vfa_op.py:61-125   VFA.forward  (cube corners, projection, clamped box, integral-image
//                                box pooling at 4 bilinear-sampled corners, visibility mask, collapse)
//   vfa/utils.py:50-59           project
//   vfa/model/vfanet.py:79, 82   scale sum, view sum
//
// Numerics: fp32 with the rounding sequence of the reference's PyTorch CPU path.
// The file is compiled with -ffp-contract=off; an FMA appears only where fmaf() is written.
//
// Data layout in HBM (see include/vfa_hip.h): the integral image is stored channels-last with a one-pixel
// zero border, (n_views, Hf+2, Wf+2, C).  One bilinear tap of one box is then C contiguous floats (1 KiB at
// C = 256 = one 16 B/lane wave64 load), and grid_sample's zeros padding is a plain load of the border.
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

#include "vfa_hip.h"

#include "vfa_geom.h"

namespace {
using namespace vfa_dev;

// ------------------------------------------------------------------------------------------------
// integral image, pass 1: cumsum along W.                                reference vfa_op.py:173 (inner)
// One wave owns 64 channels of one image row.  The row is staged through LDS in 32-column chunks so that
// the NCHW reads are row-contiguous and the channels-last writes are channel-contiguous (256 B per
// wave store).  Each lane scans one channel sequentially with a double accumulator rounded to fp32 at
// every element -- ATen's CPU cumsum -- so the result is bit-identical to the reference.
// grid = (Hf, ceil(C/64), n_views), block = 64.
// ------------------------------------------------------------------------------------------------
constexpr int kRowChunk = 32;

// AFFINE (producer fusion): the input is the lateral 1x1-conv output and the kernel applies the
// GroupNorm affine + ReLU of reference vfanet.py:72-74 while scanning: f = relu(x * scale[v, c] + shift[v, c]), two
// separately rounded fp32 operations, scale = gamma * rstd, shift = beta - mean * scale per (view, channel).  The lateral map
// itself is never written.
template <bool AFFINE>
__global__ __launch_bounds__(kWave) void integral_rows_kernel(const float *__restrict__ feat,
                                                              float *__restrict__ out, int C, int H, int W,
                                                              const float *__restrict__ scale, const float *__restrict__ shift)
{
    // tile[channel][x]: row stride 33 floats -> the per-lane scans (lane = channel) are bank-conflict free
    __shared__ __align__(16) float tile[kWave][kRowChunk + 1];
    const int lane = threadIdx.x;
    const int y = blockIdx.x, c0 = blockIdx.y * kWave, v = blockIdx.z;
    const int nch = min(kWave, C - c0);
    const size_t plane = (size_t)H * W;
    const float *src = feat + ((size_t)v * C + c0) * plane + (size_t)y * W;
    const int Wp = W + 2;
    float *dst = out + (((size_t)v * (H + 2) + (y + 1)) * Wp) * C + c0; // padded row y+1, padded col 0
    const bool active = lane < nch;
    if (active) {
        dst[lane] = 0.0f;                        // left border
        dst[(size_t)(W + 1) * C + lane] = 0.0f;  // right border
    }
    // 16-byte paths need aligned rows (loads) and channel quads (stores)
    const bool vec_load = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(feat) & 15) == 0);
    const bool vec_store = (C % 4 == 0) && (nch == kWave) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    const float sa = (AFFINE && active) ? scale[(size_t)v * C + c0 + lane] : 1.0f;
    const float sb = (AFFINE && active) ? shift[(size_t)v * C + c0 + lane] : 0.0f;
    auto act = [&](float x) {
        if (!AFFINE) return x;
        float t = x * sa;
        t = t + sb;
        return (t < 0.0f) ? 0.0f : t; // NaN stays NaN
    };
    double acc = 0.0;
    for (int x0 = 0; x0 < W; x0 += kRowChunk) {
        const int nx = min(kRowChunk, W - x0);
        if (vec_load && nx == kRowChunk) {
            // 8 lanes x float4 cover one 32-column row piece: 8 channels per load instruction, 1 KiB per wave
            const int q = lane & 7;
            for (int r = lane >> 3; r < nch; r += 8) {
                const float4 t = *reinterpret_cast<const float4 *>(src + (size_t)r * plane + x0 + 4 * q);
                tile[r][4 * q + 0] = t.x; tile[r][4 * q + 1] = t.y; tile[r][4 * q + 2] = t.z; tile[r][4 * q + 3] = t.w;
            }
        } else {
            const int j = lane & (kRowChunk - 1);
            for (int r = lane >> 5; r < nch; r += 2)
                if (j < nx) tile[r][j] = src[(size_t)r * plane + x0 + j];
        }
        __syncthreads();
        if (active) {
            if (vec_store) {
                for (int k = 0; k < nx; ++k) { // scan in place; the stores follow as 16-byte accesses
                    acc += (double)act(tile[lane][k]);
                    tile[lane][k] = (float)acc;
                }
            } else {
                for (int k = 0; k < nx; ++k) {
                    acc += (double)act(tile[lane][k]);
                    dst[(size_t)(x0 + k + 1) * C + lane] = (float)acc;
                }
            }
        }
        __syncthreads();
        if (vec_store) {
            // lane -> (x = lane / 16, channel quad = lane % 16): 4 x-positions x 256 B per store instruction
            const int cq = lane & 15;
            for (int k = lane >> 4; k < nx; k += 4) {
                const float4 t = make_float4(tile[4 * cq + 0][k], tile[4 * cq + 1][k], tile[4 * cq + 2][k], tile[4 * cq + 3][k]);
                *reinterpret_cast<float4 *>(dst + (size_t)(x0 + k + 1) * C + 4 * cq) = t;
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// integral image, pass 2: cumsum along H, in place on the channels-last buffer, plus the zero top and
// bottom border rows.                                                   reference vfa_op.py:173 (outer)
// One thread owns VEC channels of one padded column; loads of a column are independent of the running
// sum, so they are issued eight rows ahead.
// ------------------------------------------------------------------------------------------------
template <int VEC> struct vec_of;
template <> struct vec_of<1> { using type = float; };
template <> struct vec_of<4> { using type = float4; };

template <int VEC> __device__ __forceinline__ void scan_step(double (&acc)[VEC], typename vec_of<VEC>::type &v);
template <> __device__ __forceinline__ void scan_step<1>(double (&acc)[1], float &v)
{
    acc[0] += (double)v;
    v = (float)acc[0];
}
template <> __device__ __forceinline__ void scan_step<4>(double (&acc)[4], float4 &v)
{
    acc[0] += (double)v.x; v.x = (float)acc[0];
    acc[1] += (double)v.y; v.y = (float)acc[1];
    acc[2] += (double)v.z; v.z = (float)acc[2];
    acc[3] += (double)v.w; v.w = (float)acc[3];
}
template <int VEC> __device__ __forceinline__ typename vec_of<VEC>::type vzero();
template <> __device__ __forceinline__ float vzero<1>() { return 0.0f; }
template <> __device__ __forceinline__ float4 vzero<4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }

template <int VEC>
__global__ __launch_bounds__(256) void integral_cols_kernel(float *__restrict__ io, int H, size_t row_vecs,
                                                            size_t total_vecs)
{
    using V = typename vec_of<VEC>::type;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total_vecs) return;
    const size_t v = i / row_vecs, r = i % row_vecs;
    V *p = reinterpret_cast<V *>(io) + v * (size_t)(H + 2) * row_vecs + r;
    p[0] = vzero<VEC>();
    p[(size_t)(H + 1) * row_vecs] = vzero<VEC>();
    p += row_vecs; // first interior row
    double acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.0;
    constexpr int U = 8;
    int y = 0;
    for (; y + U <= H; y += U) {
        V t[U];
#pragma unroll
        for (int k = 0; k < U; ++k) t[k] = p[(size_t)(y + k) * row_vecs];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            scan_step<VEC>(acc, t[k]);
            p[(size_t)(y + k) * row_vecs] = t[k];
        }
    }
    for (; y < H; ++y) {
        V t = p[(size_t)y * row_vecs];
        scan_step<VEC>(acc, t);
        p[(size_t)y * row_vecs] = t;
    }
}

// one thread per (view, layer, cell)
__global__ __launch_bounds__(256) void box_params_kernel(BoxGeom g, int n_cells, int nl, size_t total, int Hf,
                                                         int Wf, float4 *__restrict__ box, float *__restrict__ area,
                                                         uint8_t *__restrict__ visible)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cell = (int)(idx % n_cells);
    const size_t vl = idx / n_cells;
    const int layer = (int)(vl % nl), view = (int)(vl / nl);
    const float *P = g.calibs + (size_t)view * 12;
    const float gx = g.grid[cell * 3 + 0] + 0.0f; // + the int64 zeros of z_corners (vfa_op.py:52, :64)
    const float gy = g.grid[cell * 3 + 1] + 0.0f;
    const float gz = g.grid[cell * 3 + 2] + g.z_layers[layer];
    float l = 0, t = 0, r = 0, b = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float nu, nv;
        project_corner(g, P, gx, gy, gz, k, nu, nv);
        if (k == 0) { l = r = nu; t = b = nv; }
        else { l = min_t(l, nu); r = max_t(r, nu); t = min_t(t, nv); b = max_t(b, nv); }
    }
    const float a = box_area(l, t, r, b, Hf, Wf);
    box[idx] = make_float4(l, t, r, b);
    area[idx] = a;
    visible[idx] = box_visible(a, Hf, Wf) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// box pooling                                                               reference vfa_op.py:112-120
// ------------------------------------------------------------------------------------------------
template <int VEC> __device__ __forceinline__ typename vec_of<VEC>::type vmul(typename vec_of<VEC>::type a, float w);
template <> __device__ __forceinline__ float vmul<1>(float a, float w) { return a * w; }
template <> __device__ __forceinline__ float4 vmul<4>(float4 a, float w)
{
    return make_float4(a.x * w, a.y * w, a.z * w, a.w * w);
}
template <int VEC>
__device__ __forceinline__ typename vec_of<VEC>::type vfma(typename vec_of<VEC>::type a, float w,
                                                           typename vec_of<VEC>::type c);
template <> __device__ __forceinline__ float vfma<1>(float a, float w, float c) { return fmaf(a, w, c); }
template <> __device__ __forceinline__ float4 vfma<4>(float4 a, float w, float4 c)
{
    return make_float4(fmaf(a.x, w, c.x), fmaf(a.y, w, c.y), fmaf(a.z, w, c.z), fmaf(a.w, w, c.w));
}

template <int VEC>
__device__ __forceinline__ typename vec_of<VEC>::type vbox_mean(typename vec_of<VEC>::type lt, typename vec_of<VEC>::type rb,
                                                                typename vec_of<VEC>::type rt, typename vec_of<VEC>::type lb,
                                                                float area, float rcp);
template <> __device__ __forceinline__ float vbox_mean<1>(float lt, float rb, float rt, float lb, float area, float rcp)
{
    return box_mean(lt, rb, rt, lb, area, rcp);
}
template <>
__device__ __forceinline__ float4 vbox_mean<4>(float4 lt, float4 rb, float4 rt, float4 lb, float area, float rcp)
{
    return make_float4(box_mean(lt.x, rb.x, rt.x, lb.x, area, rcp), box_mean(lt.y, rb.y, rt.y, lb.y, area, rcp),
                       box_mean(lt.z, rb.z, rt.z, lb.z, area, rcp), box_mean(lt.w, rb.w, rt.w, lb.w, area, rcp));
}

// Voxel features are written once and read once by the next kernel: non-temporal stores keep their 287 MB per scale from
// displacing the integral image in L2 / Infinity Cache and let the write-back overlap this (not bandwidth-bound) kernel
// instead of the consumer's reads (pooling kernel 162 -> 144 us at stride 8; frame 0.86 -> 0.84 ms).
typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_nt(void *p, float4 v)
{
    nt_f32x4 x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<nt_f32x4 *>(p));
}
__device__ __forceinline__ void store_nt(void *p, float v) { __builtin_nontemporal_store(v, reinterpret_cast<float *>(p)); }

// Unsigned 32-bit division by a launch constant as multiply-high + shifts (Granlund-Montgomery round-up form, exact
// for every 32-bit dividend): the tile decomposition of the tap-cache kernels is wave-uniform SALU work at the head of
// every tile's dependent chain, and hipcc's 64-bit '/' and '%' cost ~100 scalar instructions apiece there.
struct FastDiv {
    unsigned mul, sh1, sh2, d;
};
inline FastDiv make_fastdiv(unsigned d)
{
    FastDiv f;
    f.d = d;
    unsigned L = 0;
    while (L < 32 && (1ull << L) < d) ++L; // ceil(log2 d)
    f.mul = (unsigned)(((1ull << 32) * ((1ull << L) - d)) / d + 1);
    f.sh1 = L < 1 ? L : 1;
    f.sh2 = L > 0 ? L - 1 : 0;
    return f;
}
__device__ __forceinline__ unsigned fast_div(unsigned n, const FastDiv &f)
{
    const unsigned t = __umulhi(n, f.mul);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

struct GatherDims {
    int C, Hf, Wf, nl, n_cells, cell_begin, cell_count, vox_layout;
    long long n_boxes;    // n_views * cell_count * nl
    long long per_xcd;    // blocks per XCD
    int ws_chunk;         // two-kernel form: boxes per pooling wave (power of two <= 64)
    // tap-cache kernels: tile = (view, cell block, layer), layer fastest
    unsigned n_tiles;     // n_views * tiles_per_view
    FastDiv tiles_per_view, layers;
    // backward tap cache: patches of 4 x 8 cells (grid_w > 0: cells per row of the ground grid; tile_row0 = first patch row the
    // processed range touches, tiles_x = patches per grid row) or 32 cells in a line (grid_w == 0)
    int grid_w, tile_row0, tiles_x;
};

// Per-box record staged in LDS by phase 1 of the gather kernel (32 words = 8 x ds_read_b128, broadcast to the wave).
// Tap offsets are BYTES relative to the view's padded image (unsigned 32-bit: one padded image is < 4 GiB); any
// out-of-image tap is redirected to the zero border.  The 16 bilinear weights are formed here, once per box, in the
// reference's rounding (each a single rounded product), so that phase 2 spends its VALU on the channel arithmetic.
struct alignas(16) BoxHdr {  // what a wave needs before it can issue the tap loads of a run
    int flags;       // bit 0 visible, bits 1-2 DXC, bits 3-4 DYC
    int view;
    int run_len;     // boxes from this one on (inside the wave's chunk) that are visible and share its tap set
    float masked;    // value of a masked voxel: area * 0 (0, or NaN when the box itself is NaN)
    unsigned col[4]; // xl, xl+1, xr, xr+1
    unsigned row[4]; // yt, yt+1, yb, yb+1
};
struct alignas(16) BoxWeights { // what the channel arithmetic of one box needs
    float lt[4];     // nw, ne, sw, se of sample (left, top)
    float rb[4];
    float rt[4];
    float lb[4];
    float area;
    float rcp;       // RN(1 / area): the per-channel divisions share it (see box_mean)
    unsigned out_row; // reference layout only: (view * cell_count + cell_local), column base = layer
    int layer;
};
struct alignas(16) BoxRec {
    BoxHdr h;
    BoxWeights w;
};
static_assert(sizeof(BoxRec) == 128, "BoxRec must stay 8 x 16 bytes");

constexpr int kTileBoxes = 128;           // boxes per workgroup
constexpr int kPerWave = kTileBoxes / 4;  // contiguous boxes per wave in phase 2 (a power of two <= 64)

__device__ __forceinline__ void fill_record(BoxRec &rc, int view, float l, float t, float r, float b, float area, bool vis,
                                            const GatherDims &d, unsigned &key_x, unsigned &key_y)
{
    const Axis xl = make_axis(l, d.Wf), xr = make_axis(r, d.Wf);
    const Axis yt = make_axis(t, d.Hf), yb = make_axis(b, d.Hf);
    const int dx = xr.i0 - xl.i0, dy = yb.i0 - yt.i0;
    const int dxc = dx == 0 ? 0 : (dx == 1 ? 1 : 2), dyc = dy == 0 ? 0 : (dy == 1 ? 1 : 2);
    rc.h.flags = (vis ? 1 : 0) | (dxc << 1) | (dyc << 3);
    rc.h.view = view;
    rc.w.area = area;
    rc.w.rcp = 1.0f / area; // correctly rounded (hipcc's default fp32 division)
    rc.h.masked = area * 0.0f;
    const unsigned Wp = d.Wf + 2;
    const int xs[4] = {xl.i0, xl.i0 + 1, xr.i0, xr.i0 + 1};
    const int ys[4] = {yt.i0, yt.i0 + 1, yb.i0, yb.i0 + 1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        rc.h.col[k] = (unsigned)(min(max(xs[k], -1), d.Wf) + 1) * (unsigned)d.C * 4u;
        rc.h.row[k] = (unsigned)(min(max(ys[k], -1), d.Hf) + 1) * Wp * (unsigned)d.C * 4u;
    }
    // the tap set of a box is named by (view, flags, clamped tap origins)
    key_x = (unsigned)(min(max(xs[0], -1), 0xfffe) + 1) | ((unsigned)(min(max(xs[2], -1), 0xfffe) + 1) << 16);
    key_y = (unsigned)(min(max(ys[0], -1), 0xfffe) + 1) | ((unsigned)(min(max(ys[2], -1), 0xfffe) + 1) << 16);
    bilinear_weights(rc.w.lt, xl, yt);
    bilinear_weights(rc.w.rb, xr, yb);
    bilinear_weights(rc.w.rt, xr, yt);
    bilinear_weights(rc.w.lb, xl, yb);
}

// bilinear sample from the four rounded weights, taps in the order nw, ne, sw, se: one product, three FMAs (A.5)
template <int VEC>
__device__ __forceinline__ typename vec_of<VEC>::type bilinear_w(typename vec_of<VEC>::type nw, typename vec_of<VEC>::type ne,
                                                                 typename vec_of<VEC>::type sw, typename vec_of<VEC>::type se,
                                                                 const float (&w)[4])
{
    auto v = vmul<VEC>(nw, w[0]);
    v = vfma<VEC>(ne, w[1], v);
    v = vfma<VEC>(sw, w[2], v);
    v = vfma<VEC>(se, w[3], v);
    return v;
}


// The 16 taps of a box are the product {top rows yt, yt+1, bottom rows yb, yb+1} x {left cols xl, xl+1, right cols
// xr, xr+1}.  When the right pair coincides with / overlaps the left pair (DXC = xr - xl = 0 or 1; boxes are often
// narrower than a feature pixel) the shared columns are loaded once, and likewise for rows (DYC).  Every (DYC, DXC)
// variant is a straight-line body with its own register patch: NR x NC unique taps, all loads issued first.
// `img` is wave-uniform (SGPR pair); `lane_off` and the record offsets are 32-bit VGPRs, so each tap costs one
// v_add3_u32 and one global_load_dwordx4 with scalar base.
template <int VEC, int DYC, int DXC>
__device__ __forceinline__ void load_patch(typename vec_of<VEC>::type (&P)[4][4], const char *__restrict__ img,
                                           unsigned lane_off, const unsigned (&row)[4], const unsigned (&col)[4])
{
    using V = typename vec_of<VEC>::type;
    constexpr int NR = DYC == 0 ? 2 : (DYC == 1 ? 3 : 4), NC = DXC == 0 ? 2 : (DXC == 1 ? 3 : 4);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int rs = (DYC == 1 && r == 2) ? 3 : r; // unique rows: {0,1}, {0,1,3} or {0,1,2,3}
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int cs = (DXC == 1 && c == 2) ? 3 : c;
            P[r][c] = *reinterpret_cast<const V *>(img + (row[rs] + col[cs] + lane_off));
        }
    }
}

template <int VEC, int DYC, int DXC>
__device__ __forceinline__ typename vec_of<VEC>::type pool_patch(const typename vec_of<VEC>::type (&P)[4][4],
                                                                 const BoxWeights &w)
{
    constexpr int RB0 = DYC == 0 ? 0 : (DYC == 1 ? 1 : 2), RB1 = RB0 + 1; // bottom row pair inside the patch
    constexpr int CR0 = DXC == 0 ? 0 : (DXC == 1 ? 1 : 2), CR1 = CR0 + 1; // right column pair inside the patch
    const auto lt = bilinear_w<VEC>(P[0][0], P[0][1], P[1][0], P[1][1], w.lt);
    const auto rb = bilinear_w<VEC>(P[RB0][CR0], P[RB0][CR1], P[RB1][CR0], P[RB1][CR1], w.rb);
    const auto rt = bilinear_w<VEC>(P[0][CR0], P[0][CR1], P[1][CR0], P[1][CR1], w.rt);
    const auto lb = bilinear_w<VEC>(P[RB0][0], P[RB0][1], P[RB1][0], P[RB1][1], w.lb);
    return vbox_mean<VEC>(lt, rb, rt, lb, w.area, w.rcp);
}

// Phase 1 of the gather kernels: thread i of the workgroup computes the parameters of box tile0 + i (FUSED: projects
// the eight cube corners itself; otherwise reads box/area/visible) and stages its BoxRec in LDS.
template <bool FUSED>
__device__ __forceinline__ void stage_box_records(BoxRec *recs, long long tile0, int nb, int lane,
                                                  const float4 *__restrict__ box, const float *__restrict__ area_in,
                                                  const uint8_t *__restrict__ visible_in, const BoxGeom &g,
                                                  const GatherDims &d, int chunk = kPerWave)
{
    if ((int)threadIdx.x < nb) {
        const long long wid = tile0 + threadIdx.x;
        const int layer = (int)(wid % d.nl);
        const long long vc = wid / d.nl;
        const int cell = d.cell_begin + (int)(vc % d.cell_count), view = (int)(vc / d.cell_count);
        float l, t, r, b, area;
        bool vis;
        if constexpr (FUSED) {
            const float *P = g.calibs + (size_t)view * 12;
            const float gx = g.grid[cell * 3 + 0] + 0.0f; // + the int64 zeros of z_corners (vfa_op.py:52, :64)
            const float gy = g.grid[cell * 3 + 1] + 0.0f;
            const float gz = g.grid[cell * 3 + 2] + g.z_layers[layer];
            l = t = r = b = 0.0f;
#pragma unroll 1
            for (int k = 0; k < 8; ++k) {
                float nu, nv;
                project_corner(g, P, gx, gy, gz, k, nu, nv);
                if (k == 0) { l = r = nu; t = b = nv; }
                else { l = min_t(l, nu); r = max_t(r, nu); t = min_t(t, nv); b = max_t(b, nv); }
            }
            area = box_area(l, t, r, b, d.Hf, d.Wf);
            vis = box_visible(area, d.Hf, d.Wf);
        } else {
            const size_t bidx = ((size_t)view * d.nl + layer) * d.n_cells + cell;
            const float4 bx = box[bidx];
            l = bx.x; t = bx.y; r = bx.z; b = bx.w;
            area = area_in[bidx];
            vis = visible_in[bidx] != 0;
        }
        BoxRec &rc = recs[threadIdx.x];
        unsigned key_x, key_y;
        fill_record(rc, view, l, t, r, b, area, vis, d, key_x, key_y);
        rc.w.out_row = (unsigned)vc;
        rc.w.layer = layer;
        // Runs of boxes with one tap set: lanes are consecutive boxes, a wave's chunk in phase 2 is `chunk` of them.
        // cont = "this box continues the run of the previous lane"; run_len counts the set bits that follow.
        const int tag = (view << 5) | rc.h.flags;
        // shuffles first, unconditionally: behind a short-circuit a neighbour that skipped the shuffle would read as 0
        const int tag_prev = __shfl_up(tag, 1);
        const unsigned kx_prev = __shfl_up(key_x, 1), ky_prev = __shfl_up(key_y, 1);
        const bool cont = vis && (lane & (chunk - 1)) != 0 && tag_prev == tag && kx_prev == key_x && ky_prev == key_y;
        const unsigned long long mask = __ballot(cont);
        const unsigned long long after = lane == 63 ? 0ull : (mask >> (lane + 1));
        const int follow = after == ~0ull ? 64 : __builtin_ctzll(~after);
        rc.h.run_len = 1 + min(follow, chunk - 1 - (lane & (chunk - 1)));
    }
}

// Projection + box pooling.  A workgroup owns a tile of kTileBoxes consecutive boxes in (view, cell, layer) order.
//   phase 1: one thread per box computes the box parameters (FUSED: projects the eight cube corners itself;
//            otherwise reads box/area/visible) and stages a BoxRec in LDS;
//   phase 2: the four waves walk the tile, one box per wave at a time, lanes = channels (VEC = 4: 64 lanes x float4 =
//            256 channels = one 1 KiB load per tap).  The record is wave-uniform: masked boxes and duplicate taps are
//            skipped by scalar branches.
template <int VEC, bool FUSED>
__global__ __launch_bounds__(256, 4) void gather_kernel(const float *__restrict__ integral, const float4 *__restrict__ box,
                                                     const float *__restrict__ area_in,
                                                     const uint8_t *__restrict__ visible_in, BoxGeom g, GatherDims d,
                                                     float *__restrict__ vox)
{
    using V = typename vec_of<VEC>::type;
    __shared__ BoxRec recs[kTileBoxes];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = uniform_i(threadIdx.x >> 6);
    const long long blk = xcd_contiguous(blockIdx.x, d.per_xcd);
    const long long tile0 = blk * kTileBoxes; // first box of the tile; box id = (view, cell_local, layer), layer fastest
    if (tile0 >= d.n_boxes) return;
    const int nb = (int)min((long long)kTileBoxes, d.n_boxes - tile0);

    stage_box_records<FUSED>(recs, tile0, nb, lane, box, area_in, visible_in, g, d);
    __syncthreads();

    const size_t img_stride = (size_t)(d.Hf + 2) * (d.Wf + 2) * d.C * sizeof(float);
    const bool layer_major = d.vox_layout == VFA_VOX_LAYER_MAJOR;
    const unsigned c_bytes = (unsigned)d.C * 4u;
    // Each wave walks a contiguous quarter of the tile, so consecutive boxes are neighbouring cells / layers.  Their
    // tap sets often coincide (a cell is a fraction of a feature pixel wide): a run of such boxes loads its register
    // patch once and pools it with each box's weights.
    const int j_end = min(nb, (wave + 1) * kPerWave);
    int j = wave * kPerWave;
    while (j < j_end) {
        const BoxHdr &h = recs[j].h;
        const int flags = uniform_i(h.flags);
        char *out0 = reinterpret_cast<char *>(vox) + (size_t)(tile0 + j) * c_bytes; // layer-major address of box j
        if (!(flags & 1)) {
            const float z = h.masked;
            char *out = layer_major ? out0
                                    : reinterpret_cast<char *>(vox + (size_t)uniform_i(recs[j].w.out_row) * d.C * d.nl +
                                                               uniform_i(recs[j].w.layer));
#pragma unroll 1
            for (int c = lane * VEC; c < d.C; c += kWave * VEC) {
                if (layer_major) {
                    if constexpr (VEC == 4) store_nt(out + c * 4, make_float4(z, z, z, z));
                    else store_nt(out + c * 4, z);
                } else {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) reinterpret_cast<float *>(out)[(size_t)(c + k) * d.nl] = z;
                }
            }
            ++j;
            continue;
        }
        const int run = uniform_i(h.run_len);
        const char *img = reinterpret_cast<const char *>(integral) + (size_t)uniform_i(h.view) * img_stride;
        unsigned col[4], row[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            col[k] = h.col[k];
            row[k] = h.row[k];
        }
#pragma unroll 1
        for (int c = lane * VEC; c < d.C; c += kWave * VEC) {
            const unsigned lane_off = (unsigned)c * 4u;
#define VFA_VARIANT(DY, DX)                                                                                     \
    {                                                                                                           \
        V P[4][4];                                                                                              \
        load_patch<VEC, DY, DX>(P, img, lane_off, row, col);                                                    \
        for (int k = 0; k < run; ++k) {                                                                         \
            const BoxWeights w = recs[j + k].w;                                                                 \
            const V res = pool_patch<VEC, DY, DX>(P, w);                                                        \
            if (layer_major) {                                                                                  \
                store_nt(out0 + (size_t)k * c_bytes + lane_off, res);                                           \
            } else {                                                                                            \
                float *o = vox + (size_t)uniform_i(w.out_row) * d.C * d.nl + uniform_i(w.layer);                \
                const float *rs = reinterpret_cast<const float *>(&res);                                        \
                _Pragma("unroll") for (int q = 0; q < VEC; ++q) o[(size_t)(c + q) * d.nl] = rs[q];             \
            }                                                                                                   \
        }                                                                                                       \
    }                                                                                                           \
    break;
            switch (flags >> 1) { // DXC | DYC << 2
            case 0: VFA_VARIANT(0, 0)
            case 1: VFA_VARIANT(0, 1)
            case 2: VFA_VARIANT(0, 2)
            case 4: VFA_VARIANT(1, 0)
            case 5: VFA_VARIANT(1, 1)
            case 6: VFA_VARIANT(1, 2)
            case 8: VFA_VARIANT(2, 0)
            case 9: VFA_VARIANT(2, 1)
            default: VFA_VARIANT(2, 2)
            }
#undef VFA_VARIANT
        }
        j += run;
    }
}

// ------------------------------------------------------------------------------------------------
// Projection + box pooling through an LDS tap cache (C = 256, layer-major output).
// The direct kernel above is bound by the vector-L1 / texture-address path: every 1 KiB tap load costs a CU ~20 TA
// cycles and a box issues 5-9 of them, although neighbouring boxes share most taps (distinct taps per box: 3.0 / 1.4 /
// 0.65 at stride 8 / 16 / 32 on the bench workload).  Here ONE WAVE owns 8 consecutive boxes of a view:
//   1. lane (box, corner) projects one cube corner; an 8-lane xor-shuffle min/max gives the box; lane 0 of each box
//      stages the BoxRec in LDS;
//   2. the 8 x 16 tap keys are inserted into a 64-entry LDS hash set (ds_cmpst), giving each DISTINCT tap a slot id;
//   3. per 64-channel quarter: every distinct tap is loaded from memory once (16 lanes x float4 = 256 B, four taps per
//      load instruction) into its LDS slot, then 16-lane groups -- four boxes per wave instruction -- read their 16
//      taps with ds_read_b128 (bank = channel quad, conflict-free across the instruction's lane groups), run the
//      reference's FMA chains and store 256 B per box.
// Everything is private to the wave (its own ~10 KiB of LDS, workgroup = 64 threads), so no workgroup barriers.
// A tile with more distinct taps than slots falls back to the direct per-box path.
// ------------------------------------------------------------------------------------------------
constexpr int kCacheBoxes = 8;
constexpr int kCacheSlots = 32;
constexpr int kCacheHash = 64;
constexpr unsigned kEmptyKey = 0xffffffffu;
constexpr int kCacheQuads = 16;                      // float4 lanes per box: 16 -> 64 channels per pass, 4 boxes per instruction
                                                     // (8 -> 32 channels, 8 boxes: smaller LDS, 5 waves/SIMD, measured no faster)
constexpr int kCacheGroups = kWave / kCacheQuads;    // boxes pooled per wave instruction
constexpr int kCachePasses = 256 / (4 * kCacheQuads);

struct CachedLds {
    BoxRec recs[kCacheBoxes];
    unsigned tab[kCacheHash];                   // tap key (byte offset inside the view's padded image) or kEmptyKey
    unsigned slot_key[kCacheSlots];             // dense slot id -> tap key
    unsigned char ids[kCacheHash];              // hash entry -> dense slot id
    unsigned char box_slots[kCacheBoxes][16];   // [box][tap = row index * 4 + col index] -> slot id
    float4 slot_data[kCacheSlots][kCacheQuads]; // the current channel slice (kCacheQuads x 4 channels) of every distinct tap
};

__global__ __launch_bounds__(kWave) void gather_cached_kernel(const float *__restrict__ integral, BoxGeom g, GatherDims d,
                                                             float *__restrict__ vox)
{
    __shared__ CachedLds L;
    const int lane = threadIdx.x;
    // A tile is 8 consecutive CELLS of one (view, layer): neighbouring cells of a layer are the boxes that share taps.
    // Tiles are numbered (view, cell block, layer) so that tiles running side by side touch the same image columns.
    const unsigned tile = (blockIdx.x & 7u) * (unsigned)d.per_xcd + (blockIdx.x >> 3); // xcd_contiguous, 32-bit
    if (tile >= d.n_tiles) return;
    const int view = (int)fast_div(tile, d.tiles_per_view);
    const unsigned tv = tile - (unsigned)view * d.tiles_per_view.d;
    const unsigned cb = fast_div(tv, d.layers);
    const int layer = (int)(tv - cb * d.layers.d);
    const int cell0 = (int)cb * kCacheBoxes; // first cell (local to the processed range) of the tile
    const int nb = min(kCacheBoxes, d.cell_count - cell0);
    const size_t img_stride = (size_t)(d.Hf + 2) * (d.Wf + 2) * d.C * sizeof(float);
    const char *img = reinterpret_cast<const char *>(integral) + (size_t)view * img_stride;
    // layer-major output: box (cell, layer) owns 1 KiB at ((view * cells + cell) * nl + layer) * 1024
    const size_t box_pitch = (size_t)d.nl * 1024;
    char *out_tile = reinterpret_cast<char *>(vox) + (((size_t)view * d.cell_count + cell0) * d.nl + layer) * 1024;

    // ---- 1. box parameters: lane = (box, corner)
    const int b = lane >> 3, corner = lane & 7;
    const bool valid = b < nb;
    float l, t, r, bt, area = 0.0f;
    bool vis = false;
    {
        const int cell = d.cell_begin + cell0 + (valid ? b : 0);
        const float *P = g.calibs + (size_t)view * 12;
        const float gx = g.grid[cell * 3 + 0] + 0.0f; // + the int64 zeros of z_corners (vfa_op.py:52, :64)
        const float gy = g.grid[cell * 3 + 1] + 0.0f;
        const float gz = g.grid[cell * 3 + 2] + g.z_layers[layer];
        float nu, nv;
        project_corner(g, P, gx, gy, gz, corner, nu, nv);
        l = r = nu; t = bt = nv;
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) { // exact, order-insensitive (NaN propagates either way)
            l = min_t(l, __shfl_xor(l, m));
            r = max_t(r, __shfl_xor(r, m));
            t = min_t(t, __shfl_xor(t, m));
            bt = max_t(bt, __shfl_xor(bt, m));
        }
        area = box_area(l, t, r, bt, d.Hf, d.Wf);
        vis = valid && box_visible(area, d.Hf, d.Wf);
    }
    unsigned key_x, key_y;
    BoxRec rec;
    fill_record(rec, view, l, t, r, bt, area, vis, d, key_x, key_y);
    if (!valid) { rec.h.flags = 0; rec.h.masked = 0.0f; }
    if (__ballot(vis) == 0ull) { // nothing visible in the tile: eight masked rows, one 1 KiB store each
        for (int j = 0; j < nb; ++j) {
            const float z = __shfl(rec.h.masked, 8 * j);
            store_nt(out_tile + (size_t)j * box_pitch + lane * 16, make_float4(z, z, z, z));
        }
        return;
    }
    if (corner == 0) L.recs[b] = rec;
    L.tab[lane] = kEmptyKey;
    __syncthreads();

    // ---- 2. distinct taps of the tile: lane (box, corner) owns taps 2*corner and 2*corner + 1 of its box
    int myh[2] = {0, 0};
    bool overflow = false;
    if (vis) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // tap = 2 * corner + i: row index corner >> 1, column index 2 * (corner & 1) + i (selects, not indexing:
            // a runtime index into the record would put it in scratch)
            const int ri = corner >> 1, ci = 2 * (corner & 1) + i;
            const unsigned rsel = ri == 0 ? rec.h.row[0] : (ri == 1 ? rec.h.row[1] : (ri == 2 ? rec.h.row[2] : rec.h.row[3]));
            const unsigned csel = ci == 0 ? rec.h.col[0] : (ci == 1 ? rec.h.col[1] : (ci == 2 ? rec.h.col[2] : rec.h.col[3]));
            const unsigned key = rsel + csel;
            unsigned h = ((key >> 10) * 2654435761u) >> 26;
            int probes = 0;
#pragma unroll 1
            for (; probes < kCacheHash; ++probes) { // bounded: a full table means "too many distinct taps"
                const unsigned old = atomicCAS(&L.tab[h], kEmptyKey, key);
                if (old == kEmptyKey || old == key) break;
                h = (h + 1) & (kCacheHash - 1);
            }
            if (probes == kCacheHash) overflow = true;
            myh[i] = (int)h;
        }
    }
    __syncthreads();
    const unsigned mine = L.tab[lane];
    const bool occ = mine != kEmptyKey;
    const unsigned long long occ_mask = __ballot(occ);
    const int n_slots = __popcll(occ_mask);
    const int my_id = __popcll(occ_mask & ((1ull << lane) - 1ull));
    const bool cached = n_slots <= kCacheSlots && __ballot(overflow) == 0ull;
    if (occ && cached) {
        L.ids[lane] = (unsigned char)my_id;
        L.slot_key[my_id] = mine;
    }
    __syncthreads();
    if (vis && cached) {
        L.box_slots[b][2 * corner + 0] = L.ids[myh[0]];
        L.box_slots[b][2 * corner + 1] = L.ids[myh[1]];
    }
    __syncthreads();

    if (cached) {
        // ---- 3. per 64-channel quarter: fill the slots, pool four boxes per wave instruction.
        // The tap addresses of a lane's two boxes are quarter-invariant and stay in registers; the loads of quarter
        // q+1 are issued (into registers) before quarter q is pooled, so their latency hides behind the arithmetic.
        // LDS operations of one wave execute in order, so phases need no waits -- only compiler fences.
        const int grp = lane / kCacheQuads, cq = lane % kCacheQuads;
        constexpr int kIters = (kCacheBoxes + kCacheGroups - 1) / kCacheGroups;
        constexpr int kLoads = kCacheSlots / kCacheGroups; // slots per lane group
        constexpr int kSliceBytes = kCacheQuads * 16;
        unsigned taps[kIters][16]; // slot index of each of the 16 taps (row index * 4 + column index)
        bool bvis[kIters], bval[kIters];
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int bb = it * kCacheGroups + grp;
            bval[it] = bb < nb;
            const BoxRec &rc = L.recs[bval[it] ? bb : 0];
            bvis[it] = bval[it] && (rc.h.flags & 1);
            const uint4 sl4 = *reinterpret_cast<const uint4 *>(L.box_slots[bval[it] ? bb : 0]);
            const unsigned sw[4] = {sl4.x, sl4.y, sl4.z, sl4.w}; // sw[row index] = the 4 column slots of that row
#pragma unroll
            for (int tp = 0; tp < 16; ++tp)
                taps[it][tp] = bvis[it] ? ((sw[tp >> 2] >> (8 * (tp & 3))) & 0xffu) : 0u;
        }
        unsigned src[kLoads]; // byte offset of this lane's piece of slot grp + kCacheGroups k, slice 0
        float4 pre[kLoads];
#pragma unroll
        for (int k = 0; k < kLoads; ++k) {
            const int i = grp + kCacheGroups * k;
            src[k] = L.slot_key[i < n_slots ? i : 0] + cq * 16;
            pre[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n_slots) pre[k] = *reinterpret_cast<const float4 *>(img + src[k]);
        }
        for (int q = 0; q < kCachePasses; ++q) {
#pragma unroll
            for (int k = 0; k < kLoads; ++k)
                if (grp + kCacheGroups * k < n_slots) L.slot_data[grp + kCacheGroups * k][cq] = pre[k];
            if (q + 1 < kCachePasses) {
#pragma unroll
                for (int k = 0; k < kLoads; ++k)
                    if (grp + kCacheGroups * k < n_slots)
                        pre[k] = *reinterpret_cast<const float4 *>(img + (src[k] + (q + 1) * kSliceBytes));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int it = 0; it < kIters; ++it) {
                if (!bval[it]) continue;
                const int bb = it * kCacheGroups + grp;
                const BoxRec &rc = L.recs[bb];
                float4 res;
                if (bvis[it]) {
                    const BoxWeights w = rc.w;
                    auto T = [&](int tp) { return L.slot_data[taps[it][tp]][cq]; };
                    const float4 lt = bilinear_w<4>(T(0), T(1), T(4), T(5), w.lt);
                    const float4 rb = bilinear_w<4>(T(10), T(11), T(14), T(15), w.rb);
                    const float4 rt = bilinear_w<4>(T(2), T(3), T(6), T(7), w.rt);
                    const float4 lb = bilinear_w<4>(T(8), T(9), T(12), T(13), w.lb);
                    res = vbox_mean<4>(lt, rb, rt, lb, w.area, w.rcp);
                } else {
                    const float z = rc.h.masked;
                    res = make_float4(z, z, z, z);
                }
                store_nt(out_tile + (size_t)bb * box_pitch + q * kSliceBytes + cq * 16, res);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        return;
    }

    // ---- fallback: more distinct taps than slots -> direct per-box path (lanes = 256 channels)
    const unsigned lane_off = (unsigned)lane * 16u;
    for (int j = 0; j < nb; ++j) {
        const BoxRec &rc = L.recs[j];
        const int flags = uniform_i(rc.h.flags);
        float4 res;
        if (flags & 1) {
            unsigned col[4], row[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { col[k] = rc.h.col[k]; row[k] = rc.h.row[k]; }
            const BoxWeights w = rc.w;
            float4 P[4][4];
            load_patch<4, 2, 2>(P, img, lane_off, row, col); // all 16 taps (rare path: no dedupe)
            res = pool_patch<4, 2, 2>(P, w);
        } else {
            const float z = rc.h.masked;
            res = make_float4(z, z, z, z);
        }
        store_nt(out_tile + (size_t)j * box_pitch + lane_off, res);
    }
}

// Backward twin of gather_cached_kernel: the same records and hash set.  d integral[tap] of a tile of boxes is
//   sum_box coef[tap][box] * grad_vox[box],   coef = the signed bilinear weights / area of the box's taps that hit `tap`,
// an (n_taps x boxes) matrix built once per tile in LDS.  A lane owns channels lane + 64 q: per 64-channel slice it keeps the
// gradients of the tile's boxes in registers, forms each distinct tap's row with one fma per box and channel and issues ONE 256-byte
// atomic row per distinct tap and slice -- no LDS traffic in the channel loop beyond the broadcast reads of a tap's coefficients.
// The kernel runs at the chip's rate of float atomic rows (~1.3 TB/s), so what counts is how many distinct taps a tile has per box.
// Round 6: a tile is a PATCH of 4 x 8 cells of the ground grid (vfa_project_gather_backward_grid_f32 tells the grid's width) instead
// of 8 consecutive cells: neighbours share taps in both directions.  Bench frame, counted with the oracle's boxes: 1.74 / 0.89 / 0.49
// distinct taps per box on strides 8 / 16 / 32 against 4.48 / 2.70 / 1.65 for 1 x 8 (32 cells in a line: 3.48 / 1.80 / 0.91).  The
// hash set has 128 entries; a patch with more distinct taps (7 % on stride 8: boxes right in front of a camera) falls back to its
// four rows of 8 (64 entries), those to the per-box scatter.  Without a grid width: 32 cells in a line.
constexpr int kBwdBoxes = 32, kBwdHash = 128;
struct BackwardLds {
    BoxRec recs[kBwdBoxes];
    unsigned tab[kBwdHash];
    unsigned slot_key[kBwdHash];
    unsigned char ids[kBwdHash];
    float coef[kBwdHash][kBwdBoxes];
    unsigned vis_bits;  // what wave 0 found for the other wave of the tile: visible boxes,
    int n_slots;        // distinct taps,
    int status;         // 0 = go, 1 = nothing visible, 2 = more distinct taps than hash entries
};
// waves per tile: wave 0 builds the tile's tables, all run the channel loop, a 64-channel slice each (the tile's 21 KB of LDS leave seven
// tiles per CU: with one wave each the scatter ran at half the rate -- bench frame 600 / 335 / 224 us per scale, now 550 / 215 / 160)
constexpr int kBwdWaves = 4;
// (wave 0's steps follow each other through LDS: instructions of one wave reach the LDS in order; the compiler must keep them so)
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// where the boxes of a tile sit: box b = (row b >> tw_shift, column b & (2^tw_shift - 1)) of the tile; its cell, local to the
// processed range, is base_local + row * row_stride + column -- valid inside [0, cell_count) and left of the grid's right edge
struct BwdTile {
    long long base_local;
    int row_stride, tw_shift, col0, col_lim;
    __device__ __forceinline__ long long local_of(int b, int cell_count) const
    {
        const int ry = b >> tw_shift, cx = b & ((1 << tw_shift) - 1);
        const long long lc = base_local + (long long)ry * row_stride + cx;
        return (col0 + cx < col_lim && lc >= 0 && lc < cell_count) ? lc : -1;
    }
};

// the boxes of one tile of (view, layer); false: more distinct taps than hash entries, nothing was added
template <int NB, int HASH>
__device__ __forceinline__ bool gather_backward_tile(BackwardLds &L, const BoxGeom &g, const GatherDims &d, int view, int layer, const BwdTile &tl,
                                                     char *gimg, const char *gv_view, size_t box_pitch)
{
    constexpr int kPasses = NB / 8; // (box, corner) passes of the wave over the tile
    static_assert(HASH == 64 || HASH == 128, "one or two hash entries per lane");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b8 = lane >> 3, corner = lane & 7;
    __syncthreads(); // (the LDS of a tile before this one is done with)
    if (wave == 0) {
    // ---- 1. box parameters: lane = (box, corner), eight boxes per pass; the records go to LDS
    unsigned vis_bits = 0u; // bit b: box b is visible (wave-uniform)
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
        const int b = 8 * p + b8;
        const long long lc = tl.local_of(b, d.cell_count);
        const bool valid = lc >= 0;
        float l, t, r, bt, area = 0.0f;
        bool vis = false;
        {
            const int cell = d.cell_begin + (int)(valid ? lc : 0);
            const float *P = g.calibs + (size_t)view * 12;
            const float gx = g.grid[cell * 3 + 0] + 0.0f; // + the int64 zeros of z_corners (vfa_op.py:52, :64)
            const float gy = g.grid[cell * 3 + 1] + 0.0f;
            const float gz = g.grid[cell * 3 + 2] + g.z_layers[layer];
            float nu, nv;
            project_corner(g, P, gx, gy, gz, corner, nu, nv);
            l = r = nu; t = bt = nv;
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) { // exact, order-insensitive (NaN propagates either way)
                l = min_t(l, __shfl_xor(l, m));
                r = max_t(r, __shfl_xor(r, m));
                t = min_t(t, __shfl_xor(t, m));
                bt = max_t(bt, __shfl_xor(bt, m));
            }
            area = box_area(l, t, r, bt, d.Hf, d.Wf);
            vis = valid && box_visible(area, d.Hf, d.Wf);
        }
        unsigned key_x, key_y;
        BoxRec rec;
        fill_record(rec, view, l, t, r, bt, area, vis, d, key_x, key_y);
        if (!valid) { rec.h.flags = 0; rec.h.masked = 0.0f; }
        if (corner == 0) L.recs[b] = rec;
        const unsigned long long bal = __ballot(vis); // lanes 8 k .. 8 k + 7 = box 8 p + k
#pragma unroll
        for (int k = 0; k < 8; ++k) vis_bits |= (unsigned)((bal >> (8 * k)) & 1ull) << (8 * p + k);
    }
    int status = vis_bits == 0u ? 1 : 0; // (masked voxels pass no gradient)
    int n_slots = 0;
    unsigned myh = 0u, myh2 = 0u; // hash entry of tap i of pass p: byte 2 p + i of myh (p < 2) / byte 2 (p - 2) + i of myh2
    bool overflow = false;
    if (status == 0) {
#pragma unroll
    for (int e = 0; e < HASH / kWave; ++e) L.tab[lane + kWave * e] = kEmptyKey;
    wave_lds_fence();

    // ---- 2. distinct taps of the tile: lane (box, corner) owns taps 2*corner and 2*corner + 1 of its box
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
        const int b = 8 * p + b8;
        if ((vis_bits >> b) & 1u) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                // tap = 2 * corner + i: row index corner >> 1, column index 2 * (corner & 1) + i
                const int ri = corner >> 1, ci = 2 * (corner & 1) + i;
                const unsigned key = L.recs[b].h.row[ri] + L.recs[b].h.col[ci];
                unsigned h = ((key >> 10) * 2654435761u) >> (HASH == 64 ? 26 : 25);
                int probes = 0;
#pragma unroll 1
                for (; probes < HASH; ++probes) { // bounded: a full table means "too many distinct taps"
                    const unsigned old = atomicCAS(&L.tab[h], kEmptyKey, key);
                    if (old == kEmptyKey || old == key) break;
                    h = (h + 1) & (HASH - 1);
                }
                if (probes == HASH) overflow = true;
                if (p < 2) myh |= h << (8 * (2 * p + i));
                else myh2 |= h << (8 * (2 * (p - 2) + i));
            }
        }
    }
    wave_lds_fence();
    if (__ballot(overflow) != 0ull) status = 2;
    }
    if (status == 0) {
#pragma unroll
    for (int e = 0; e < HASH / kWave; ++e) { // dense slot ids in entry order
        const unsigned mine = L.tab[lane + kWave * e];
        const bool occ = mine != kEmptyKey;
        const unsigned long long occ_mask = __ballot(occ);
        const int my_id = n_slots + __popcll(occ_mask & ((1ull << lane) - 1ull));
        if (occ) {
            L.ids[lane + kWave * e] = (unsigned char)my_id;
            L.slot_key[my_id] = mine;
        }
        n_slots += __popcll(occ_mask);
    }
    // ---- 3. coefficient matrix: lane (box, corner) adds its two taps (ds_add_f32)
#pragma unroll
    for (int e = 0; e < HASH / kWave; ++e)
#pragma unroll
        for (int k = 0; k < NB; ++k) L.coef[lane + kWave * e][k] = 0.0f;
    wave_lds_fence();
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
        const int b = 8 * p + b8;
        if ((vis_bits >> b) & 1u) {
            const BoxWeights &w = L.recs[b].w;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ri = corner >> 1, ci = 2 * (corner & 1) + i, k = (ri & 1) * 2 + (ci & 1);
                const bool top = ri < 2, left = ci < 2;
                const float *arr = top ? (left ? w.lt : w.rt) : (left ? w.lb : w.rb);
                const float wt = arr[k];
                const unsigned h = ((p < 2 ? myh >> (8 * (2 * p + i)) : myh2 >> (8 * (2 * (p - 2) + i))) & 0xffu);
                atomicAdd(&L.coef[L.ids[h]][b], (top == left ? wt : -wt) / w.area);
            }
        }
    }
    }
    if (lane == 0) { L.vis_bits = vis_bits; L.n_slots = n_slots; L.status = status; }
    } // (wave 0)
    __syncthreads();
    const int status = uniform_i(L.status), n_slots = uniform_i(L.n_slots);
    const unsigned vis_bits = (unsigned)uniform_i((int)L.vis_bits);
    if (status) return status == 1;

    // ---- 4. lane = channel (+ 64 q): the gradients of a 64-channel slice in registers, one atomic row per distinct tap
    const char *gv_layer = gv_view + (size_t)layer * 1024;
    for (int q = wave; q < kCachePasses; q += kBwdWaves) {
        float gv[NB];
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) { // masked voxels pass no gradient (and may hold anything)
            gv[bb] = 0.0f;
            if ((vis_bits >> bb) & 1u)
                gv[bb] = *reinterpret_cast<const float *>(gv_layer + (size_t)tl.local_of(bb, d.cell_count) * box_pitch + q * 256 + lane * 4);
        }
        for (int i = 0; i < n_slots; ++i) {
            float a = 0.0f;
#pragma unroll
            for (int k4 = 0; k4 < NB / 4; ++k4) {
                const float4 cc = *reinterpret_cast<const float4 *>(&L.coef[i][4 * k4]);
                a = fmaf(cc.x, gv[4 * k4 + 0], a); a = fmaf(cc.y, gv[4 * k4 + 1], a);
                a = fmaf(cc.z, gv[4 * k4 + 2], a); a = fmaf(cc.w, gv[4 * k4 + 3], a);
            }
            unsafeAtomicAdd(reinterpret_cast<float *>(gimg + L.slot_key[i]) + lane + q * kWave, a);
        }
    }
    return true;
}

__global__ __launch_bounds__(kWave * kBwdWaves) void gather_backward_cached_kernel(const float *__restrict__ grad_vox, BoxGeom g,
                                                                                  GatherDims d, float *__restrict__ grad_integral)
{
    __shared__ BackwardLds L;
    // Tiles are numbered (view, cell block, layer) so that tiles running side by side touch the same image columns.
    const unsigned tile = (blockIdx.x & 7u) * (unsigned)d.per_xcd + (blockIdx.x >> 3); // xcd_contiguous, 32-bit
    if (tile >= d.n_tiles) return;
    const int view = (int)fast_div(tile, d.tiles_per_view);
    const unsigned tv = tile - (unsigned)view * d.tiles_per_view.d;
    const unsigned cb = fast_div(tv, d.layers);
    const int layer = (int)(tv - cb * d.layers.d);
    BwdTile tl;
    if (d.grid_w > 0) { // a patch of 4 x 8 cells of the ground grid; cell block = (tile row of the range, tile column)
        const unsigned tr = cb / (unsigned)d.tiles_x, tc = cb - tr * (unsigned)d.tiles_x;
        tl.base_local = ((long long)d.tile_row0 + tr) * 4 * d.grid_w + 8ll * tc - d.cell_begin;
        tl.row_stride = d.grid_w; tl.tw_shift = 3; tl.col0 = 8 * (int)tc; tl.col_lim = d.grid_w;
    } else {            // 32 consecutive cells
        tl.base_local = (long long)cb * kBwdBoxes;
        tl.row_stride = 0; tl.tw_shift = 5; tl.col0 = 0; tl.col_lim = 0x7fffffff;
    }
    const size_t img_stride = (size_t)(d.Hf + 2) * (d.Wf + 2) * d.C * sizeof(float);
    char *gimg = reinterpret_cast<char *>(grad_integral) + (size_t)view * img_stride;
    // layer-major input: box (cell, layer) owns 1 KiB at ((view * cells + cell) * nl + layer) * 1024
    const size_t box_pitch = (size_t)d.nl * 1024;
    const char *gv_view = reinterpret_cast<const char *>(grad_vox) + (size_t)view * d.cell_count * box_pitch;
    if (gather_backward_tile<kBwdBoxes, kBwdHash>(L, g, d, view, layer, tl, gimg, gv_view, box_pitch)) return;
    for (int sub = 0; sub < kBwdBoxes / kCacheBoxes; ++sub) { // too many distinct taps: eight boxes at a time (a row of the patch)
        BwdTile ts = tl;
        ts.base_local = d.grid_w > 0 ? tl.base_local + (long long)sub * d.grid_w : tl.base_local + sub * kCacheBoxes;
        ts.row_stride = 0; ts.tw_shift = 3;
        if (gather_backward_tile<kCacheBoxes, kCacheHash>(L, g, d, view, layer, ts, gimg, gv_view, box_pitch)) continue;
        // ---- still too many: per-box global atomics, lanes = 64 channels x 4 sweeps (the records of the eight are in LDS)
        for (int j = 0; j < kCacheBoxes; ++j) {
            const BoxRec &rc = L.recs[j];
            if (!(uniform_i(rc.h.flags) & 1)) continue;
            const BoxWeights w = rc.w;
            const float *gvox = reinterpret_cast<const float *>(gv_view + (size_t)ts.local_of(j, d.cell_count) * box_pitch + (size_t)layer * 1024);
            for (int c = threadIdx.x; c < d.C; c += kWave * kBwdWaves) {
                const float gv = gvox[c] / w.area;
                auto add = [&](int ry, int cx, float wt) {
                    unsafeAtomicAdd(reinterpret_cast<float *>(gimg + (rc.h.row[ry] + rc.h.col[cx])) + c, gv * wt);
                };
                add(0, 0, w.lt[0]); add(0, 1, w.lt[1]); add(1, 0, w.lt[2]); add(1, 1, w.lt[3]);
                add(2, 2, w.rb[0]); add(2, 3, w.rb[1]); add(3, 2, w.rb[2]); add(3, 3, w.rb[3]);
                add(0, 2, -w.rt[0]); add(0, 3, -w.rt[1]); add(1, 2, -w.rt[2]); add(1, 3, -w.rt[3]);
                add(2, 0, -w.lb[0]); add(2, 1, -w.lb[1]); add(3, 0, -w.lb[2]); add(3, 1, -w.lb[3]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Two-kernel form of the projection + box pooling: box records through HBM, scalar-loaded by the pooling waves.
// ------------------------------------------------------------------------------------------------
// records kernel: one thread per box, 256 boxes per workgroup; writes the same BoxRec the tiled kernel stages in LDS
// (runs are delimited per 32-box chunk = the unit of work of one pooling wave).
template <bool FUSED>
__global__ __launch_bounds__(256) void box_records_kernel(BoxRec *__restrict__ recs, const float4 *__restrict__ box,
                                                          const float *__restrict__ area_in,
                                                          const uint8_t *__restrict__ visible_in, BoxGeom g, GatherDims d)
{
    const long long tile0 = (long long)blockIdx.x * 256;
    if (tile0 >= d.n_boxes) return;
    const int nb = (int)min(256ll, d.n_boxes - tile0);
    stage_box_records<FUSED>(recs + tile0, tile0, nb, threadIdx.x & (kWave - 1), box, area_in, visible_in, g, d,
                             d.ws_chunk);
}

// Scalar-memory prefetch of box records.  hipcc sinks a plain load to its first use, which would put the record fetch
// of box k+1 behind the arithmetic of box k; the loads are therefore issued by hand (s_load into SGPRs) one step
// ahead and waited for (lgkmcnt(0), scalar loads return out of order) right before the registers are consumed.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct ScalarWeights { f32x16 w; f32x4 t; };  // BoxRec::w: 16 tap weights | area, rcp, out_row, layer
struct ScalarHeader { i32x8 a; i32x4 b; };    // BoxRec::h: flags, view, run_len, masked, col[4] | row[4]
static_assert(offsetof(BoxRec, w) == 48 && sizeof(BoxHdr) == 48 && sizeof(BoxWeights) == 80, "record layout");

// (Compiler-issued scalar loads through the constant address space.  The first form, `s_load` in inline asm with the wait a
// step later, is invisible to the register allocator: with scalar registers spilled in between it would save the destination
// before the data lands and restore garbage -- which is what happened to the fused kernel of round 2.)
__device__ __forceinline__ void sload_weights(ScalarWeights &o, const BoxRec *p)
{
    const char *b = reinterpret_cast<const char *>(p);
    o.w = *reinterpret_cast<const __attribute__((address_space(4))) f32x16 *>((size_t)(b + 0x30));
    o.t = *reinterpret_cast<const __attribute__((address_space(4))) f32x4 *>((size_t)(b + 0x70));
}
__device__ __forceinline__ void swait_weights(ScalarWeights &) {}
__device__ __forceinline__ void sload_header(ScalarHeader &o, const BoxRec *p)
{
    const char *b = reinterpret_cast<const char *>(p);
    o.a = *reinterpret_cast<const __attribute__((address_space(4))) i32x8 *>((size_t)b);
    o.b = *reinterpret_cast<const __attribute__((address_space(4))) i32x4 *>((size_t)(b + 0x20));
}
__device__ __forceinline__ void swait_header(ScalarHeader &) {}
__device__ __forceinline__ BoxWeights unpack(const ScalarWeights &s)
{
    BoxWeights w;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        w.lt[q] = s.w[q]; w.rb[q] = s.w[4 + q]; w.rt[q] = s.w[8 + q]; w.lb[q] = s.w[12 + q];
    }
    w.area = s.t[0];
    w.rcp = s.t[1];
    w.out_row = __float_as_uint(s.t[2]);
    w.layer = __float_as_int(s.t[3]);
    return w;
}

// pooling kernel: one wave (= one workgroup) per chunk of kPerWave consecutive boxes.  Records are wave-uniform and
// fetched with scalar loads: offsets, weights and control live in SGPRs, the SALU forms tap addresses, and the VALU is
// left with the channel arithmetic.  No LDS, no barrier.
template <int VEC>
__global__ __launch_bounds__(kWave) void gather_records_kernel(const float *__restrict__ integral,
                                                               const BoxRec *__restrict__ recs, GatherDims d,
                                                               float *__restrict__ vox)
{
    using V = typename vec_of<VEC>::type;
    const int lane = threadIdx.x;
    const long long chunk = xcd_contiguous(blockIdx.x, d.per_xcd);
    const long long j0 = chunk * d.ws_chunk;
    if (j0 >= d.n_boxes) return;
    const long long j_end = min(d.n_boxes, j0 + d.ws_chunk);
    const size_t img_stride = (size_t)(d.Hf + 2) * (d.Wf + 2) * d.C * sizeof(float);
    const bool layer_major = d.vox_layout == VFA_VOX_LAYER_MAJOR;
    const unsigned c_bytes = (unsigned)d.C * 4u;
    long long j = j0;
    ScalarHeader sh;
    sload_header(sh, recs + j);
    swait_header(sh);
    while (j < j_end) {
        const int flags = sh.a[0];
        const bool vis = (flags & 1) != 0;
        const int run = vis ? sh.a[2] : 1;
        const int view = sh.a[1];
        const float masked = __int_as_float(sh.a[3]);
        const unsigned col[4] = {(unsigned)sh.a[4], (unsigned)sh.a[5], (unsigned)sh.a[6], (unsigned)sh.a[7]};
        const unsigned row[4] = {(unsigned)sh.b[0], (unsigned)sh.b[1], (unsigned)sh.b[2], (unsigned)sh.b[3]};
        ScalarHeader sh_next;
        sload_header(sh_next, recs + j + run); // the workspace holds one spare record past the last box
        char *out0 = reinterpret_cast<char *>(vox) + (size_t)j * c_bytes; // layer-major address of box j
        if (!vis) {
            char *out = layer_major ? out0
                                    : reinterpret_cast<char *>(vox + (size_t)recs[j].w.out_row * d.C * d.nl + recs[j].w.layer);
#pragma unroll 1
            for (int c = lane * VEC; c < d.C; c += kWave * VEC) {
                if (layer_major) {
                    if constexpr (VEC == 4) *reinterpret_cast<float4 *>(out + c * 4) = make_float4(masked, masked, masked, masked);
                    else *reinterpret_cast<float *>(out + c * 4) = masked;
                } else {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) reinterpret_cast<float *>(out)[(size_t)(c + k) * d.nl] = masked;
                }
            }
        } else {
            const char *img = reinterpret_cast<const char *>(integral) + (size_t)view * img_stride;
#pragma unroll 1
            for (int c = lane * VEC; c < d.C; c += kWave * VEC) {
                const unsigned lane_off = (unsigned)c * 4u;
#define VFA_VARIANT(DY, DX)                                                                                     \
    {                                                                                                           \
        ScalarWeights sw;                                                                                       \
        sload_weights(sw, recs + j);                                                                            \
        V P[4][4];                                                                                              \
        load_patch<VEC, DY, DX>(P, img, lane_off, row, col);                                                    \
        swait_weights(sw);                                                                                      \
        for (int k = 0; k < run; ++k) {                                                                         \
            ScalarWeights sw_next;                                                                              \
            sload_weights(sw_next, recs + j + k + 1);                                                           \
            const BoxWeights w = unpack(sw);                                                                    \
            const V res = pool_patch<VEC, DY, DX>(P, w);                                                        \
            if (layer_major) {                                                                                  \
                *reinterpret_cast<V *>(out0 + (size_t)k * c_bytes + lane_off) = res;                            \
            } else {                                                                                            \
                float *o = vox + (size_t)w.out_row * d.C * d.nl + w.layer;                                      \
                const float *rs = reinterpret_cast<const float *>(&res);                                        \
                _Pragma("unroll") for (int q = 0; q < VEC; ++q) o[(size_t)(c + q) * d.nl] = rs[q];             \
            }                                                                                                   \
            swait_weights(sw_next);                                                                             \
            sw = sw_next;                                                                                       \
        }                                                                                                       \
    }                                                                                                           \
    break;
                switch (flags >> 1) { // DXC | DYC << 2
                case 0: VFA_VARIANT(0, 0)
                case 1: VFA_VARIANT(0, 1)
                case 2: VFA_VARIANT(0, 2)
                case 4: VFA_VARIANT(1, 0)
                case 5: VFA_VARIANT(1, 1)
                case 6: VFA_VARIANT(1, 2)
                case 8: VFA_VARIANT(2, 0)
                case 9: VFA_VARIANT(2, 1)
                default: VFA_VARIANT(2, 2)
                }
#undef VFA_VARIANT
            }
        }
        j += run;
        swait_header(sh_next);
        sh = sh_next;
    }
}

// ------------------------------------------------------------------------------------------------
// Fused projection + box pooling + collapse (fp32 MFMA):  lin[view, cell, :] = sum_layer vox[view, cell, layer, :] . W_layer^T
//                                               replaces vfa_op.py:64-123 without the vox round trip through HBM.
// One persistent workgroup of 12 waves per CU.  A tile is 64 cells of one view; a sub-tile is one z-layer of it: a
// 64 x 256 block of voxel features, i.e. the A operand of a 64 x 256 x 256 product with W_layer^T.
//   waves 4-11 (producers, VALU): project + pool 8 boxes each -- the same arithmetic as gather_kernel, bit for bit --
//                                 and write the rows into the LDS A buffer of the NEXT sub-tile;
//   waves 0-3 (consumers, MFMA):  v_mfma_f32_32x32x2_f32 over the CURRENT A buffer; wave w owns all 64 rows and columns
//                                 64 w.. as 2 x 2 blocks of 32 x 32 (W streamed from L2 once per wave and sub-tile).
// The matrix pipe and the VALU are separate, so the two wave groups of a SIMD overlap; one barrier per sub-tile swaps
// the double-buffered A tile.  MFMA fp32 is a k-ordered fmaf chain, so results are within the collapse tolerance of
// the reference GEMM (not bitwise; no GEMM order is).
// ------------------------------------------------------------------------------------------------
constexpr int kFusedRows = 64;            // cells per tile
constexpr int kFusedK = 256;              // channels per layer == K of one sub-tile (C must equal this)
constexpr int kFusedN = 256;              // output channels
constexpr int kProducers = 8;             // producer waves per workgroup (waves 4..4+kProducers-1)
constexpr int kRowsPerProducer = kFusedRows / kProducers;

struct FusedDims {
    int Hf, Wf, nl, n_cells, n_views;
    int tiles_per_view; // ceil(n_cells / 64)
    int n_tiles;        // n_views * tiles_per_view
};

typedef float mfma_acc_t __attribute__((ext_vector_type(16)));

// position of element (row, k) inside a 64 x 256 A buffer: XOR swizzle so that the 32 rows read by one MFMA operand
// fetch (fixed k, rows r..r+31) fall into 32 different banks
__device__ __forceinline__ int a_index(int row, int k) { return row * kFusedK + (k ^ (row & 31)); }

__global__ __launch_bounds__(64 * (4 + kProducers)) void fused_collapse_kernel(const float *__restrict__ integral, BoxGeom g, FusedDims fd,
                                                                const float *__restrict__ w_t, // (nl*256, 256): [l*256+k][n]
                                                                float *__restrict__ lin)      // (n_views, n_cells, 256)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *a_buf = reinterpret_cast<float *>(smem);                                         // 2 x 64 x 256 floats
    BoxRec *recs = reinterpret_cast<BoxRec *>(smem + 2 * kFusedRows * kFusedK * sizeof(float)); // 2 x 64 records
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = uniform_i(threadIdx.x >> 6);
    const bool producer = wave >= 4;

    // contiguous range of tiles for this workgroup (neighbouring workgroups of an XCD get neighbouring ranges)
    const int nblk = gridDim.x;
    const int lb = (int)xcd_contiguous(blockIdx.x, (nblk + 7) / 8);
    if (lb >= nblk) return; // (grid is a multiple of 8 by construction)
    const int t_begin = (int)((long long)fd.n_tiles * lb / nblk), t_end = (int)((long long)fd.n_tiles * (lb + 1) / nblk);
    const int total_sub = (t_end - t_begin) * fd.nl;
    if (total_sub == 0) return;

    GatherDims d;
    d.C = kFusedK; d.Hf = fd.Hf; d.Wf = fd.Wf; d.nl = fd.nl; d.n_cells = fd.n_cells; d.cell_begin = 0;
    d.cell_count = fd.n_cells; d.vox_layout = VFA_VOX_LAYER_MAJOR; d.n_boxes = 0; d.per_xcd = 0;
    const size_t img_stride = (size_t)(fd.Hf + 2) * (fd.Wf + 2) * kFusedK * sizeof(float);

    // ---- producer: sub-tile `sub` -> a_buf[sub & 1]
    auto produce = [&](int sub) {
        const int tile = t_begin + sub / fd.nl, layer = sub % fd.nl;
        const int view = tile / fd.tiles_per_view, cell0 = (tile % fd.tiles_per_view) * kFusedRows;
        const int pw = wave - 4;
        float *A = a_buf + (size_t)(sub & 1) * kFusedRows * kFusedK;
        BoxRec *rr = recs + (sub & 1) * kFusedRows + pw * kRowsPerProducer;
        // records of this wave's 16 rows (lanes 0..15)
        {
            const int cell = cell0 + pw * kRowsPerProducer + lane;
            const bool valid = lane < kRowsPerProducer && cell < fd.n_cells;
            float l = 0.f, t = 0.f, r = 0.f, b = 0.f, area = 0.f;
            bool vis = false;
            if (valid) {
                const float *P = g.calibs + (size_t)view * 12;
                const float gx = g.grid[cell * 3 + 0] + 0.0f;
                const float gy = g.grid[cell * 3 + 1] + 0.0f;
                const float gz = g.grid[cell * 3 + 2] + g.z_layers[layer];
#pragma unroll 1
                for (int k = 0; k < 8; ++k) {
                    float nu, nv;
                    project_corner(g, P, gx, gy, gz, k, nu, nv);
                    if (k == 0) { l = r = nu; t = b = nv; }
                    else { l = min_t(l, nu); r = max_t(r, nu); t = min_t(t, nv); b = max_t(b, nv); }
                }
                area = box_area(l, t, r, b, fd.Hf, fd.Wf);
                vis = box_visible(area, fd.Hf, fd.Wf);
            }
            unsigned key_x = 0, key_y = 0;
            int tag = -1 - lane;
            if (lane < kRowsPerProducer) {
                BoxRec &rc = rr[lane];
                fill_record(rc, view, l, t, r, b, area, vis, d, key_x, key_y);
                if (!valid) { rc.h.flags = 0; rc.h.masked = 0.0f; }
                tag = valid ? rc.h.flags : (-1 - lane);
            }
            const int tag_prev = __shfl_up(tag, 1); // unconditional: see stage_box_records
            const unsigned kx_prev = __shfl_up(key_x, 1), ky_prev = __shfl_up(key_y, 1);
            const bool cont = vis && lane != 0 && lane < kRowsPerProducer && tag_prev == tag && kx_prev == key_x && ky_prev == key_y;
            const unsigned long long mask = __ballot(cont);
            const unsigned long long after = lane == 63 ? 0ull : (mask >> (lane + 1));
            const int follow = after == ~0ull ? 64 : __builtin_ctzll(~after);
            if (lane < kRowsPerProducer) rr[lane].h.run_len = 1 + min(follow, kRowsPerProducer - 1 - lane);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): this wave's LDS record writes are visible to its own reads
        __builtin_amdgcn_wave_barrier();
        const char *img = reinterpret_cast<const char *>(integral) + (size_t)view * img_stride;
        const unsigned lane_off = (unsigned)lane * 16u;
        int j = 0;
        while (j < kRowsPerProducer) {
            const BoxHdr &h = rr[j].h;
            const int flags = uniform_i(h.flags);
            const int row0 = pw * kRowsPerProducer + j;
            if (!(flags & 1)) {
                const float z = h.masked;
#pragma unroll
                for (int q = 0; q < 4; ++q) A[a_index(row0, lane * 4 + q)] = z;
                ++j;
                continue;
            }
            const int run = uniform_i(h.run_len);
            unsigned col[4], row[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { col[k] = h.col[k]; row[k] = h.row[k]; }
#define VFA_VARIANT(DY, DX)                                                                     \
    {                                                                                           \
        float4 P[4][4];                                                                         \
        load_patch<4, DY, DX>(P, img, lane_off, row, col);                                      \
        for (int k = 0; k < run; ++k) {                                                         \
            const BoxWeights w = rr[j + k].w;                                                   \
            const float4 res = pool_patch<4, DY, DX>(P, w);                                     \
            const int rw = row0 + k;                                                            \
            A[a_index(rw, lane * 4 + 0)] = res.x;                                               \
            A[a_index(rw, lane * 4 + 1)] = res.y;                                               \
            A[a_index(rw, lane * 4 + 2)] = res.z;                                               \
            A[a_index(rw, lane * 4 + 3)] = res.w;                                               \
        }                                                                                       \
    }                                                                                           \
    break;
            switch (flags >> 1) {
            case 0: VFA_VARIANT(0, 0)
            case 1: VFA_VARIANT(0, 1)
            case 2: VFA_VARIANT(0, 2)
            case 4: VFA_VARIANT(1, 0)
            case 5: VFA_VARIANT(1, 1)
            case 6: VFA_VARIANT(1, 2)
            case 8: VFA_VARIANT(2, 0)
            case 9: VFA_VARIANT(2, 1)
            default: VFA_VARIANT(2, 2)
            }
#undef VFA_VARIANT
            j += run;
        }
    };

    // ---- consumer state: wave w owns all 64 rows x columns 64 w + 2 j + q: two row blocks x two column blocks of 32 x 32
    // (column 2j+q of the wave's strip belongs to MFMA column block q, so one 8-byte load of W feeds both column blocks
    // and one 8-byte store covers a row piece).  Every wave reads its own 64 columns of W exactly once per sub-tile.
    mfma_acc_t acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][q][e] = 0.0f;
    constexpr int U = 4;               // k-steps per software-pipeline group
    constexpr int G = kFusedK / 2 / U; // groups per sub-tile (16)
    float2 b0[U], b1[U];
    float a0[U][2], a1[U][2];
    const int khalf = lane >> 5;
    auto w_ptr = [&](int sub) {
        const int layer = sub % fd.nl;
        return reinterpret_cast<const float2 *>(w_t + ((size_t)layer * kFusedK + khalf) * kFusedN + 64 * wave) + (lane & 31);
    };
    auto load_b = [&](const float2 *wp, int grp, float2 (&bb)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) bb[u] = wp[(size_t)(2 * (grp * U + u)) * (kFusedN / 2)];
    };
    auto load_a = [&](const float *A, int grp, float (&aa)[U][2]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = 2 * (grp * U + u) + khalf;
            aa[u][0] = A[a_index(lane & 31, k)];
            aa[u][1] = A[a_index(32 + (lane & 31), k)];
        }
    };
    auto mfma_group = [&](const float2 (&bb)[U], const float (&aa)[U][2]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[u][0], bb[u].x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[u][0], bb[u].y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[u][1], bb[u].x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[u][1], bb[u].y, acc[1][1], 0, 0, 0);
        }
    };

    // W of the first group is already in b0 when consume() is entered (fetched before the barrier)
    auto consume = [&](int sub) {
        const int layer = sub % fd.nl;
        const float *A = a_buf + (size_t)(sub & 1) * kFusedRows * kFusedK;
        const float2 *wp = w_ptr(sub);
        load_a(A, 0, a0);
#pragma unroll 1
        for (int grp = 0; grp < G; grp += 2) {
            load_b(wp, grp + 1, b1);
            load_a(A, grp + 1, a1);
            mfma_group(b0, a0);
            if (grp + 2 < G) {
                load_b(wp, grp + 2, b0);
                load_a(A, grp + 2, a0);
            } else if (sub + 1 < total_sub) {
                load_b(w_ptr(sub + 1), 0, b0); // W of the next sub-tile does not depend on the barrier
            }
            mfma_group(b1, a1);
        }
        if (layer == fd.nl - 1) { // tile finished: write the 64 x 64 strip, reset the accumulators
            const int tile = t_begin + sub / fd.nl;
            const int view = tile / fd.tiles_per_view, cell0 = (tile % fd.tiles_per_view) * kFusedRows;
            float *out = lin + ((size_t)view * fd.n_cells + cell0) * kFusedN + 64 * wave + 2 * (lane & 31);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rloc = 32 * r + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    if (cell0 + rloc < fd.n_cells)
                        *reinterpret_cast<float2 *>(out + (size_t)rloc * kFusedN) = make_float2(acc[r][0][e], acc[r][1][e]);
                }
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[r][q][e] = 0.0f;
        }
    };

    // The two roles run separate loops (same number of barriers) so that neither carries the other's registers.
    if (producer) {
        __builtin_amdgcn_s_setprio(1); // VALU-heavy role first: an MFMA wave needs one issue slot per 64 cycles (+3 %)
        produce(0);
        __syncthreads();
        for (int i = 0; i < total_sub; ++i) {
            if (i + 1 < total_sub) produce(i + 1);
            __syncthreads();
        }
    } else {
        load_b(w_ptr(0), 0, b0);
        __syncthreads();
        for (int i = 0; i < total_sub; ++i) {
            consume(i);
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward of the box pooling: d vox -> d integral (scatter-add), then d integral -> d feature (reverse scans).
// The reference trains through this path with autograd (trainer.py:41); there is no reference code to mirror, only
// the derivative of vfa_op.py:112-119.  Not bit-reproducible: float atomics sum in arrival order.
// ------------------------------------------------------------------------------------------------
// Combined weight of every unique tap of a box: + lt + rb - rt - lb (the 1/area factor rides on the gradient).
// A run of consecutive boxes with one tap set (forward: register-patch reuse) is summed in registers first, so each
// distinct tap of the run receives ONE atomic per channel instead of one per box.
template <int DYC, int DXC>
__device__ __forceinline__ void scatter_run(float *__restrict__ gimg, int C, int lane, const BoxRec *__restrict__ rr, int run,
                                            const float *__restrict__ gvox)
{
    constexpr int NR = DYC == 0 ? 2 : (DYC == 1 ? 3 : 4), NC = DXC == 0 ? 2 : (DXC == 1 ? 3 : 4);
    constexpr int RB0 = DYC == 0 ? 0 : (DYC == 1 ? 1 : 2), RB1 = RB0 + 1;
    constexpr int CR0 = DXC == 0 ? 0 : (DXC == 1 ? 1 : 2), CR1 = CR0 + 1;
    const BoxHdr h = rr[0].h;
    for (int cb = 0; cb < C; cb += 4 * kWave) { // 256 channels per sweep: lane owns channels cb + lane + 64 q
        float T[NR][NC][4];
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) T[r][c][q] = 0.0f;
        for (int k = 0; k < run; ++k) {
            const BoxWeights w = rr[k].w;
            float W[NR][NC];
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int c = 0; c < NC; ++c) W[r][c] = 0.0f;
            W[0][0] += w.lt[0]; W[0][1] += w.lt[1]; W[1][0] += w.lt[2]; W[1][1] += w.lt[3];
            W[RB0][CR0] += w.rb[0]; W[RB0][CR1] += w.rb[1]; W[RB1][CR0] += w.rb[2]; W[RB1][CR1] += w.rb[3];
            W[0][CR0] -= w.rt[0]; W[0][CR1] -= w.rt[1]; W[1][CR0] -= w.rt[2]; W[1][CR1] -= w.rt[3];
            W[RB0][0] -= w.lb[0]; W[RB0][1] -= w.lb[1]; W[RB1][0] -= w.lb[2]; W[RB1][1] -= w.lb[3];
            float gv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = cb + lane + kWave * q;
                gv[q] = c < C ? gvox[(size_t)k * C + c] / w.area : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int q = 0; q < 4; ++q) T[r][c][q] = fmaf(gv[q], W[r][c], T[r][c][q]);
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int rs = (DYC == 1 && r == 2) ? 3 : r;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int cs = (DXC == 1 && c == 2) ? 3 : c;
                float *tap = gimg + ((h.row[rs] + h.col[cs]) >> 2) + cb + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (cb + lane + kWave * q < C) unsafeAtomicAdd(tap + kWave * q, T[r][c][q]); // 256 contiguous bytes per wave
            }
        }
    }
}

__global__ __launch_bounds__(256) void gather_backward_kernel(const float *__restrict__ grad_vox, BoxGeom g, GatherDims d,
                                                              float *__restrict__ grad_integral)
{
    __shared__ BoxRec recs[kTileBoxes];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = uniform_i(threadIdx.x >> 6);
    const long long blk = xcd_contiguous(blockIdx.x, d.per_xcd);
    const long long tile0 = blk * kTileBoxes;
    if (tile0 >= d.n_boxes) return;
    const int nb = (int)min((long long)kTileBoxes, d.n_boxes - tile0);
    stage_box_records<true>(recs, tile0, nb, lane, nullptr, nullptr, nullptr, g, d);
    __syncthreads();
    const size_t img_floats = (size_t)(d.Hf + 2) * (d.Wf + 2) * d.C;
    const int j_end = min(nb, (wave + 1) * kPerWave);
    int j = wave * kPerWave;
    while (j < j_end) {
        const BoxHdr &h = recs[j].h;
        const int flags = uniform_i(h.flags);
        if (!(flags & 1)) { ++j; continue; } // masked voxels pass no gradient
        const int run = uniform_i(h.run_len);
        float *gimg = grad_integral + (size_t)uniform_i(h.view) * img_floats;
        const float *gvox = grad_vox + (size_t)(tile0 + j) * d.C; // layer-major: box j + k at + k * C
        switch (flags >> 1) {
        case 0: scatter_run<0, 0>(gimg, d.C, lane, recs + j, run, gvox); break;
        case 1: scatter_run<0, 1>(gimg, d.C, lane, recs + j, run, gvox); break;
        case 2: scatter_run<0, 2>(gimg, d.C, lane, recs + j, run, gvox); break;
        case 4: scatter_run<1, 0>(gimg, d.C, lane, recs + j, run, gvox); break;
        case 5: scatter_run<1, 1>(gimg, d.C, lane, recs + j, run, gvox); break;
        case 6: scatter_run<1, 2>(gimg, d.C, lane, recs + j, run, gvox); break;
        case 8: scatter_run<2, 0>(gimg, d.C, lane, recs + j, run, gvox); break;
        case 9: scatter_run<2, 1>(gimg, d.C, lane, recs + j, run, gvox); break;
        default: scatter_run<2, 2>(gimg, d.C, lane, recs + j, run, gvox); break;
        }
        j += run;
    }
}

// d integral (n, Hf+2, Wf+2, C) -> reverse cumsum along H, in place on the interior (the border carries no gradient
// to the features: it is constant zero in the forward pass).
__global__ __launch_bounds__(256) void integral_cols_backward_kernel(float *__restrict__ io, int H, size_t row_elems,
                                                                     size_t total)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t v = i / row_elems, r = i % row_elems;
    float *p = io + v * (size_t)(H + 2) * row_elems + r + row_elems; // first interior row
    double acc = 0.0;
    for (int y = H - 1; y >= 0; --y) {
        acc += (double)p[(size_t)y * row_elems];
        p[(size_t)y * row_elems] = (float)acc;
    }
}

// reverse cumsum along W of the (already H-scanned) channels-last gradient, written as NCHW d feature.
// One wave owns 64 channels of one image row; grid = (Hf, ceil(C/64), n_views), block = 64.
__global__ __launch_bounds__(kWave) void integral_rows_backward_kernel(const float *__restrict__ gi, float *__restrict__ gf,
                                                                       int C, int H, int W)
{
    __shared__ float tile[kWave][kRowChunk + 1];
    const int lane = threadIdx.x;
    const int y = blockIdx.x, c0 = blockIdx.y * kWave, v = blockIdx.z;
    const int nch = min(kWave, C - c0);
    const size_t plane = (size_t)H * W;
    float *dst = gf + ((size_t)v * C + c0) * plane + (size_t)y * W;
    const float *src = gi + (((size_t)v * (H + 2) + (y + 1)) * (W + 2) + 1) * C + c0; // interior pixel (y, 0)
    const bool active = lane < nch;
    double acc = 0.0;
    for (int x1 = W; x1 > 0; x1 -= kRowChunk) { // chunks from the right
        const int x0 = max(0, x1 - kRowChunk), nx = x1 - x0;
        if (active)
            for (int k = nx - 1; k >= 0; --k) {
                acc += (double)src[(size_t)(x0 + k) * C + lane];
                tile[lane][k] = (float)acc;
            }
        __syncthreads();
        const int j = lane & (kRowChunk - 1);
        for (int r = lane >> 5; r < nch; r += 2)
            if (j < nx) dst[(size_t)r * plane + x0 + j] = tile[r][j];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// collapse epilogues                                            reference vfa_op.py:124; vfanet.py:79, 82
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float relu_t(float x) { return (x < 0.0f) ? 0.0f : x; } // NaN stays NaN

template <int VEC>
__global__ __launch_bounds__(256) void bias_relu_accumulate_kernel(const float *__restrict__ lin,
                                                                   const float *__restrict__ bias,
                                                                   float *__restrict__ out, int n_views, size_t MN,
                                                                   int N, int accumulate)
{
    const size_t stride = (size_t)gridDim.x * 256 * VEC;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC; i < MN; i += stride) {
        const int col = (int)(i % N);
        float acc[VEC], bb[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            bb[k] = bias ? bias[col + k] : 0.0f;
            acc[k] = accumulate ? out[i + k] : 0.0f;
        }
        for (int v = 0; v < n_views; ++v) {
            float x[VEC];
            if constexpr (VEC == 4) {
                const float4 q = *reinterpret_cast<const float4 *>(lin + (size_t)v * MN + i);
                x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
            } else {
                x[0] = lin[(size_t)v * MN + i];
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = acc[k] + relu_t(x[k] + bb[k]);
        }
        if constexpr (VEC == 4) *reinterpret_cast<float4 *>(out + i) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        else out[i] = acc[0];
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void scale_view_sum_kernel(const float *__restrict__ l8, const float *__restrict__ l16,
                                                             const float *__restrict__ l32, const float *__restrict__ b8,
                                                             const float *__restrict__ b16,
                                                             const float *__restrict__ b32, float *__restrict__ ortho,
                                                             int n_views, size_t MN, int N, int accumulate)
{
    const size_t stride = (size_t)gridDim.x * 256 * VEC;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC; i < MN; i += stride) {
        const int col = (int)(i % N);
        float acc[VEC], c8[VEC], c16[VEC], c32[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            c8[k] = b8 ? b8[col + k] : 0.0f;
            c16[k] = b16 ? b16[col + k] : 0.0f;
            c32[k] = b32 ? b32[col + k] : 0.0f;
            acc[k] = accumulate ? ortho[i + k] : 0.0f;
        }
        for (int v = 0; v < n_views; ++v) {
            float x8[VEC], x16[VEC], x32[VEC];
            const size_t o = (size_t)v * MN + i;
            if constexpr (VEC == 4) {
                const float4 q8 = *reinterpret_cast<const float4 *>(l8 + o);
                const float4 q16 = *reinterpret_cast<const float4 *>(l16 + o);
                const float4 q32 = *reinterpret_cast<const float4 *>(l32 + o);
                x8[0] = q8.x; x8[1] = q8.y; x8[2] = q8.z; x8[3] = q8.w;
                x16[0] = q16.x; x16[1] = q16.y; x16[2] = q16.z; x16[3] = q16.w;
                x32[0] = q32.x; x32[1] = q32.y; x32[2] = q32.z; x32[3] = q32.w;
            } else {
                x8[0] = l8[o]; x16[0] = l16[o]; x32[0] = l32[o];
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float s = relu_t(x8[k] + c8[k]) + relu_t(x16[k] + c16[k]); // vfanet.py:79
                s = s + relu_t(x32[k] + c32[k]);
                acc[k] = acc[k] + s;                                        // vfanet.py:82
            }
        }
        if constexpr (VEC == 4) *reinterpret_cast<float4 *>(ortho + i) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        else ortho[i] = acc[0];
    }
}

// backward of the collapse epilogues: glin[v] = grad * (lin[v] + bias > 0), gbias += column sums of glin.
// Fast path (4 | N, N | 1024): a thread keeps its four columns over the whole grid-stride loop, so the bias gradient
// is reduced in registers, then across the workgroup's waves in LDS, and costs N atomics per workgroup.
__global__ __launch_bounds__(256) void relu_mask_backward_kernel(const float *__restrict__ grad, const float *__restrict__ lin,
                                                                 const float *__restrict__ bias, float *__restrict__ glin,
                                                                 float *__restrict__ gbias, int n_views, size_t MN, int N)
{
    __shared__ float red[256 * 4];
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    const size_t first = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int col = (int)(first % N);
    float bb[4], cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) bb[k] = bias ? bias[col + k] : 0.0f;
    for (size_t i = first; i < MN; i += stride) {
        const float4 g = *reinterpret_cast<const float4 *>(grad + i);
        for (int v = 0; v < n_views; ++v) {
            const float4 x = *reinterpret_cast<const float4 *>(lin + (size_t)v * MN + i);
            float4 o;
            o.x = (x.x + bb[0] > 0.0f) ? g.x : 0.0f;
            o.y = (x.y + bb[1] > 0.0f) ? g.y : 0.0f;
            o.z = (x.z + bb[2] > 0.0f) ? g.z : 0.0f;
            o.w = (x.w + bb[3] > 0.0f) ? g.w : 0.0f;
            *reinterpret_cast<float4 *>(glin + (size_t)v * MN + i) = o;
            cs[0] += o.x; cs[1] += o.y; cs[2] += o.z; cs[3] += o.w;
        }
    }
    if (gbias) {
#pragma unroll
        for (int k = 0; k < 4; ++k) red[threadIdx.x * 4 + k] = cs[k];
        __syncthreads();
        // threads t and t + N/4 (mod 256) own the same columns
        const int per = N / 4; // threads per distinct column group
        if ((int)threadIdx.x < per) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a = 0.0f;
                for (int t = threadIdx.x; t < 256; t += per) a += red[t * 4 + k];
                unsafeAtomicAdd(gbias + col + k, a);
            }
        }
    }
}

inline int launch_status() { return (int)hipGetLastError(); }


inline unsigned elementwise_blocks(size_t n_items)
{
    const size_t want = (n_items + 255) / 256;
    const size_t cap = 256 * 8; // 256 CUs x 8 blocks, grid-stride beyond that
    return (unsigned)(want < cap ? (want ? want : 1) : cap);
}

template <bool FUSED>
int launch_gather(const float *integral, const float *box, const float *area, const uint8_t *visible, const BoxGeom &g,
                  float *vox, int n_views, int C, int Hf, int Wf, int nl, int n_cells, int cell_begin, int cell_count,
                  int vox_layout, hipStream_t s)
{
    const int kernel_choice = vox_layout & (VFA_VOX_KERNEL_DIRECT | VFA_VOX_KERNEL_TAP_CACHE);
    vox_layout &= ~(VFA_VOX_KERNEL_DIRECT | VFA_VOX_KERNEL_TAP_CACHE);
    if (n_views < 0 || C <= 0 || Hf <= 0 || Wf <= 0 || nl <= 0 || n_cells < 0 || cell_begin < 0 || cell_count < 0 ||
        cell_begin + cell_count > n_cells || (vox_layout != VFA_VOX_REFERENCE && vox_layout != VFA_VOX_LAYER_MAJOR))
        return VFA_ERR_BAD_ARGUMENT;
    GatherDims d;
    d.C = C; d.Hf = Hf; d.Wf = Wf; d.nl = nl; d.n_cells = n_cells; d.cell_begin = cell_begin;
    d.cell_count = cell_count; d.vox_layout = vox_layout;
    d.n_boxes = (long long)n_views * cell_count * nl;
    if (d.n_boxes == 0) return 0;
    // default choice: the tap cache pays when neighbouring cells share taps and few boxes are masked; measured faster on
    // single-layer grids, about equal on the multi-layer dataset configs -> cached for nl == 1 unless told otherwise
    const bool want_cached = kernel_choice == VFA_VOX_KERNEL_TAP_CACHE ? true
                             : kernel_choice == VFA_VOX_KERNEL_DIRECT  ? false
                                                                       : nl == 1;
    if (FUSED && C == 256 && vox_layout == VFA_VOX_LAYER_MAJOR && want_cached) {
        const long long tiles = (long long)n_views * nl * ((cell_count + kCacheBoxes - 1) / kCacheBoxes);
        if (tiles >= (1ll << 31) - 8) return VFA_ERR_BAD_ARGUMENT; // chunk the cells (the host side does, by vox bytes)
        d.per_xcd = (tiles + 7) / 8;
        d.n_tiles = (unsigned)tiles;
        d.tiles_per_view = make_fastdiv((unsigned)(nl * ((cell_count + kCacheBoxes - 1) / kCacheBoxes)));
        d.layers = make_fastdiv((unsigned)nl);
        hipLaunchKernelGGL(gather_cached_kernel, dim3((unsigned)(d.per_xcd * 8)), dim3(kWave), 0, s, integral, g, d, vox);
        return launch_status();
    }
    const long long blocks = (d.n_boxes + kTileBoxes - 1) / kTileBoxes;
    d.per_xcd = (blocks + 7) / 8;
    const dim3 grid((unsigned)(d.per_xcd * 8));
    if (C % 4 == 0)
        hipLaunchKernelGGL((gather_kernel<4, FUSED>), grid, dim3(256), 0, s, integral, (const float4 *)box, area, visible,
                           g, d, vox);
    else
        hipLaunchKernelGGL((gather_kernel<1, FUSED>), grid, dim3(256), 0, s, integral, (const float4 *)box, area, visible,
                           g, d, vox);
    return launch_status();
}

template <bool FUSED>
int launch_gather_ws(const float *integral, const float *box, const float *area, const uint8_t *visible, const BoxGeom &g,
                     float *vox, void *workspace, size_t workspace_bytes, int n_views, int C, int Hf, int Wf, int nl,
                     int n_cells, int cell_begin, int cell_count, int vox_layout, hipStream_t s)
{
    if (n_views < 0 || C <= 0 || Hf <= 0 || Wf <= 0 || nl <= 0 || n_cells < 0 || cell_begin < 0 || cell_count < 0 ||
        cell_begin + cell_count > n_cells || (vox_layout != VFA_VOX_REFERENCE && vox_layout != VFA_VOX_LAYER_MAJOR))
        return VFA_ERR_BAD_ARGUMENT;
    GatherDims d;
    d.C = C; d.Hf = Hf; d.Wf = Wf; d.nl = nl; d.n_cells = n_cells; d.cell_begin = cell_begin;
    d.cell_count = cell_count; d.vox_layout = vox_layout;
    d.n_boxes = (long long)n_views * cell_count * nl;
    if (d.n_boxes == 0) return 0;
    if (!workspace || workspace_bytes < ((size_t)d.n_boxes + 1) * sizeof(BoxRec)) return VFA_ERR_BAD_ARGUMENT;
    BoxRec *recs = reinterpret_cast<BoxRec *>(workspace);
    d.per_xcd = 0;
    d.ws_chunk = 16; // boxes per pooling wave
    hipLaunchKernelGGL((box_records_kernel<FUSED>), dim3((unsigned)((d.n_boxes + 255) / 256)), dim3(256), 0, s, recs,
                       (const float4 *)box, area, visible, g, d);
    int st = launch_status();
    if (st) return st;
    const long long chunks = (d.n_boxes + d.ws_chunk - 1) / d.ws_chunk;
    d.per_xcd = (chunks + 7) / 8;
    const dim3 grid((unsigned)(d.per_xcd * 8));
    if (C % 4 == 0)
        hipLaunchKernelGGL((gather_records_kernel<4>), grid, dim3(kWave), 0, s, integral, recs, d, vox);
    else
        hipLaunchKernelGGL((gather_records_kernel<1>), grid, dim3(kWave), 0, s, integral, recs, d, vox);
    return launch_status();
}

} // namespace

extern "C" {

int vfa_abi_version(void) { return VFA_ABI_VERSION; }


static int integral_image_impl(const float *feature, const float *scale, const float *shift, float *integral, int n_views, int C,
                               int Hf, int Wf, void *stream)
{
    if (n_views < 0 || C <= 0 || Hf <= 0 || Wf <= 0) return VFA_ERR_BAD_ARGUMENT;
    if (n_views == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (scale)
        hipLaunchKernelGGL((integral_rows_kernel<true>), dim3(Hf, (C + kWave - 1) / kWave, n_views), dim3(kWave), 0, s, feature,
                           integral, C, Hf, Wf, scale, shift);
    else
        hipLaunchKernelGGL((integral_rows_kernel<false>), dim3(Hf, (C + kWave - 1) / kWave, n_views), dim3(kWave), 0, s, feature,
                           integral, C, Hf, Wf, scale, shift);
    int st = launch_status();
    if (st) return st;
    if (C % 4 == 0) {
        const size_t row_vecs = (size_t)(Wf + 2) * C / 4, total = row_vecs * n_views;
        hipLaunchKernelGGL((integral_cols_kernel<4>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, integral, Hf,
                           row_vecs, total);
    } else {
        const size_t row_vecs = (size_t)(Wf + 2) * C, total = row_vecs * n_views;
        hipLaunchKernelGGL((integral_cols_kernel<1>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, integral, Hf,
                           row_vecs, total);
    }
    return launch_status();
}

int vfa_integral_image_f32(const float *feature, float *integral, int n_views, int C, int Hf, int Wf, void *stream)
{
    return integral_image_impl(feature, nullptr, nullptr, integral, n_views, C, Hf, Wf, stream);
}

int vfa_affine_relu_integral_image_f32(const float *x, const float *scale, const float *shift, float *integral, int n_views, int C,
                                       int Hf, int Wf, void *stream)
{
    if (!scale || !shift) return VFA_ERR_BAD_ARGUMENT;
    return integral_image_impl(x, scale, shift, integral, n_views, C, Hf, Wf, stream);
}

int vfa_box_params_f32(const float *calibs, const float *grid, const float *z_layers, const float *corner_off,
                       int n_views, int n_cells, int nl, int conv_kind, float img_w, float img_h, int Hf, int Wf,
                       float cmin, float cmax, float *box, float *area, uint8_t *visible, void *stream)
{
    if (n_views < 0 || n_cells < 0 || nl <= 0 || Hf <= 0 || Wf <= 0 || conv_kind < 0 || conv_kind > 2)
        return VFA_ERR_BAD_ARGUMENT;
    const size_t total = (size_t)n_views * nl * n_cells;
    if (total == 0) return 0;
    BoxGeom g{calibs, grid, z_layers, corner_off, conv_kind, img_w, img_h, cmin, cmax};
    hipLaunchKernelGGL(box_params_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g,
                       n_cells, nl, total, Hf, Wf, (float4 *)box, area, visible);
    return launch_status();
}

int vfa_gather_f32(const float *integral, const float *box, const float *area, const uint8_t *visible, float *vox,
                   int n_views, int C, int Hf, int Wf, int nl, int n_cells, int cell_begin, int cell_count,
                   int vox_layout, void *stream)
{
    BoxGeom g{};
    return launch_gather<false>(integral, box, area, visible, g, vox, n_views, C, Hf, Wf, nl, n_cells, cell_begin,
                                cell_count, vox_layout, (hipStream_t)stream);
}

int vfa_project_gather_f32(const float *integral, const float *calibs, const float *grid, const float *z_layers,
                           const float *corner_off, float *vox, int n_views, int C, int Hf, int Wf, int nl,
                           int n_cells, int cell_begin, int cell_count, int conv_kind, float img_w, float img_h,
                           float cmin, float cmax, int vox_layout, void *stream)
{
    if (conv_kind < 0 || conv_kind > 2) return VFA_ERR_BAD_ARGUMENT;
    BoxGeom g{calibs, grid, z_layers, corner_off, conv_kind, img_w, img_h, cmin, cmax};
    return launch_gather<true>(integral, nullptr, nullptr, nullptr, g, vox, n_views, C, Hf, Wf, nl, n_cells, cell_begin,
                               cell_count, vox_layout, (hipStream_t)stream);
}

size_t vfa_gather_workspace_bytes(int n_views, int nl, int cell_count)
{
    return ((size_t)n_views * (size_t)nl * (size_t)cell_count + 1) * sizeof(BoxRec); // + 1: prefetch past the end
}

int vfa_project_gather_ws_f32(const float *integral, const float *calibs, const float *grid, const float *z_layers,
                              const float *corner_off, float *vox, void *workspace, size_t workspace_bytes, int n_views,
                              int C, int Hf, int Wf, int nl, int n_cells, int cell_begin, int cell_count, int conv_kind,
                              float img_w, float img_h, float cmin, float cmax, int vox_layout, void *stream)
{
    if (conv_kind < 0 || conv_kind > 2) return VFA_ERR_BAD_ARGUMENT;
    BoxGeom g{calibs, grid, z_layers, corner_off, conv_kind, img_w, img_h, cmin, cmax};
    return launch_gather_ws<true>(integral, nullptr, nullptr, nullptr, g, vox, workspace, workspace_bytes, n_views, C, Hf,
                                  Wf, nl, n_cells, cell_begin, cell_count, vox_layout, (hipStream_t)stream);
}

int vfa_project_collapse_f32(const float *integral, const float *calibs, const float *grid, const float *z_layers,
                             const float *corner_off, const float *weight_t, float *lin, int n_views, int C, int Hf, int Wf,
                             int nl, int n_cells, int c_out, int conv_kind, float img_w, float img_h, float cmin, float cmax,
                             void *stream)
{
    if (n_views < 0 || Hf <= 0 || Wf <= 0 || nl <= 0 || n_cells < 0 || conv_kind < 0 || conv_kind > 2)
        return VFA_ERR_BAD_ARGUMENT;
    if (C != kFusedK || c_out != kFusedN) return VFA_ERR_UNSUPPORTED; // the fused tile is built for 256 -> 256 channels
    if (n_views == 0 || n_cells == 0) return 0;
    BoxGeom g{calibs, grid, z_layers, corner_off, conv_kind, img_w, img_h, cmin, cmax};
    FusedDims fd;
    fd.Hf = Hf; fd.Wf = Wf; fd.nl = nl; fd.n_cells = n_cells; fd.n_views = n_views;
    fd.tiles_per_view = (n_cells + kFusedRows - 1) / kFusedRows;
    fd.n_tiles = n_views * fd.tiles_per_view;
    const size_t lds = 2 * kFusedRows * kFusedK * sizeof(float) + 2 * kFusedRows * sizeof(BoxRec);
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute((const void *)fused_collapse_kernel,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    int n_cu = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n_cu = prop.multiProcessorCount;
    }
    int nblk = n_cu < fd.n_tiles ? n_cu : fd.n_tiles;
    nblk = (nblk + 7) / 8 * 8;
    hipLaunchKernelGGL(fused_collapse_kernel, dim3(nblk), dim3(64 * (4 + kProducers)), lds, (hipStream_t)stream, integral, g, fd, weight_t, lin);
    return launch_status();
}

int vfa_project_gather_backward_f32(const float *grad_vox, const float *calibs, const float *grid, const float *z_layers,
                                    const float *corner_off, float *grad_integral, int n_views, int C, int Hf, int Wf,
                                    int nl, int n_cells, int cell_begin, int cell_count, int conv_kind, float img_w,
                                    float img_h, float cmin, float cmax, int flags, void *stream)
{
    return vfa_project_gather_backward_grid_f32(grad_vox, calibs, grid, z_layers, corner_off, grad_integral, n_views, C, Hf, Wf, nl, n_cells,
                                                cell_begin, cell_count, 0, conv_kind, img_w, img_h, cmin, cmax, flags, stream);
}

int vfa_project_gather_backward_grid_f32(const float *grad_vox, const float *calibs, const float *grid, const float *z_layers,
                                         const float *corner_off, float *grad_integral, int n_views, int C, int Hf, int Wf,
                                         int nl, int n_cells, int cell_begin, int cell_count, int grid_w, int conv_kind, float img_w,
                                         float img_h, float cmin, float cmax, int flags, void *stream)
{
    const int accumulate = flags & VFA_BWD_ACCUMULATE;
    if (grid_w < 0 || (grid_w > 0 && n_cells % grid_w != 0)) return VFA_ERR_BAD_ARGUMENT;
    if (flags & ~(VFA_BWD_ACCUMULATE | VFA_VOX_KERNEL_DIRECT | VFA_VOX_KERNEL_TAP_CACHE)) return VFA_ERR_BAD_ARGUMENT;
    if (n_views < 0 || C <= 0 || Hf <= 0 || Wf <= 0 || nl <= 0 || n_cells < 0 || cell_begin < 0 || cell_count < 0 ||
        cell_begin + cell_count > n_cells || conv_kind < 0 || conv_kind > 2)
        return VFA_ERR_BAD_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) {
        const size_t bytes = (size_t)n_views * (Hf + 2) * (Wf + 2) * C * sizeof(float);
        if (bytes) {
            const hipError_t e = hipMemsetAsync(grad_integral, 0, bytes, s);
            if (e != hipSuccess) return (int)e;
        }
    }
    BoxGeom g{calibs, grid, z_layers, corner_off, conv_kind, img_w, img_h, cmin, cmax};
    GatherDims d;
    d.C = C; d.Hf = Hf; d.Wf = Wf; d.nl = nl; d.n_cells = n_cells; d.cell_begin = cell_begin;
    d.cell_count = cell_count; d.vox_layout = VFA_VOX_LAYER_MAJOR;
    d.n_boxes = (long long)n_views * cell_count * nl;
    if (d.n_boxes == 0) return 0;
    if (C == 256 && !(flags & VFA_VOX_KERNEL_DIRECT)) {
        long long blocks_per_layer = (cell_count + kBwdBoxes - 1) / kBwdBoxes; // cell blocks of a (view, layer)
        d.grid_w = grid_w; d.tile_row0 = 0; d.tiles_x = 0;
        if (grid_w > 0 && cell_count > 0) { // patches of 4 x 8 cells: every patch the range [cell_begin, cell_begin + cell_count) touches
            const int row_a = cell_begin / grid_w, row_b = (cell_begin + cell_count - 1) / grid_w;
            d.tile_row0 = row_a / 4;
            d.tiles_x = (grid_w + 7) / 8;
            blocks_per_layer = (long long)(row_b / 4 - d.tile_row0 + 1) * d.tiles_x;
        }
        const long long tiles = (long long)n_views * nl * blocks_per_layer;
        if (tiles >= (1ll << 31) - 8 || blocks_per_layer * nl >= (1ll << 31)) return VFA_ERR_BAD_ARGUMENT;
        d.per_xcd = (tiles + 7) / 8;
        d.n_tiles = (unsigned)tiles;
        d.tiles_per_view = make_fastdiv((unsigned)(nl * blocks_per_layer));
        d.layers = make_fastdiv((unsigned)nl);
        hipLaunchKernelGGL(gather_backward_cached_kernel, dim3((unsigned)(d.per_xcd * 8)), dim3(kWave * kBwdWaves), 0, s, grad_vox, g, d,
                           grad_integral);
        return launch_status();
    }
    const long long blocks = (d.n_boxes + kTileBoxes - 1) / kTileBoxes;
    d.per_xcd = (blocks + 7) / 8;
    hipLaunchKernelGGL(gather_backward_kernel, dim3((unsigned)(d.per_xcd * 8)), dim3(256), 0, s, grad_vox, g, d,
                       grad_integral);
    return launch_status();
}

int vfa_integral_image_backward_f32(float *grad_integral, float *grad_feature, int n_views, int C, int Hf, int Wf,
                                    void *stream)
{
    if (n_views < 0 || C <= 0 || Hf <= 0 || Wf <= 0) return VFA_ERR_BAD_ARGUMENT;
    if (n_views == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const size_t row_elems = (size_t)(Wf + 2) * C, total = row_elems * n_views;
    hipLaunchKernelGGL(integral_cols_backward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, grad_integral,
                       Hf, row_elems, total);
    int st = launch_status();
    if (st) return st;
    hipLaunchKernelGGL(integral_rows_backward_kernel, dim3(Hf, (C + kWave - 1) / kWave, n_views), dim3(kWave), 0, s,
                       grad_integral, grad_feature, C, Hf, Wf);
    return launch_status();
}

int vfa_relu_mask_backward_f32(const float *grad, const float *lin, const float *bias, float *grad_lin, float *grad_bias,
                               int n_views, size_t M, int N, void *stream)
{
    if (n_views < 0 || N <= 0 || N % 4 != 0 || 1024 % N != 0) return N > 0 && n_views >= 0 ? VFA_ERR_UNSUPPORTED : VFA_ERR_BAD_ARGUMENT;
    const size_t MN = M * (size_t)N;
    hipStream_t s = (hipStream_t)stream;
    if (grad_bias) {
        const hipError_t e = hipMemsetAsync(grad_bias, 0, (size_t)N * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
    }
    if (MN == 0 || n_views == 0) return 0;
    hipLaunchKernelGGL(relu_mask_backward_kernel, dim3(elementwise_blocks(MN / 4)), dim3(256), 0, s, grad, lin, bias, grad_lin,
                       grad_bias, n_views, MN, N);
    return launch_status();
}

int vfa_bias_relu_accumulate_f32(const float *lin, const float *bias, float *out, int n_views, size_t M, int N,
                                 int accumulate, void *stream)
{
    if (n_views < 0 || N <= 0) return VFA_ERR_BAD_ARGUMENT;
    const size_t MN = M * (size_t)N;
    if (MN == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (N % 4 == 0)
        hipLaunchKernelGGL((bias_relu_accumulate_kernel<4>), dim3(elementwise_blocks(MN / 4)), dim3(256), 0, s, lin, bias,
                           out, n_views, MN, N, accumulate);
    else
        hipLaunchKernelGGL((bias_relu_accumulate_kernel<1>), dim3(elementwise_blocks(MN)), dim3(256), 0, s, lin, bias, out,
                           n_views, MN, N, accumulate);
    return launch_status();
}

int vfa_scale_view_sum_f32(const float *lin8, const float *lin16, const float *lin32, const float *bias8,
                           const float *bias16, const float *bias32, float *ortho, int n_views, size_t M, int N,
                           int accumulate, void *stream)
{
    if (n_views < 0 || N <= 0) return VFA_ERR_BAD_ARGUMENT;
    const size_t MN = M * (size_t)N;
    if (MN == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (N % 4 == 0)
        hipLaunchKernelGGL((scale_view_sum_kernel<4>), dim3(elementwise_blocks(MN / 4)), dim3(256), 0, s, lin8, lin16, lin32,
                           bias8, bias16, bias32, ortho, n_views, MN, N, accumulate);
    else
        hipLaunchKernelGGL((scale_view_sum_kernel<1>), dim3(elementwise_blocks(MN)), dim3(256), 0, s, lin8, lin16, lin32,
                           bias8, bias16, bias32, ortho, n_views, MN, N, accumulate);
    return launch_status();
}

} // extern "C"
