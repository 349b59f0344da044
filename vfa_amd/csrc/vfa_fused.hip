// vfa_fused.hip -- the inference hot path of single-layer grids (nl = 1, C = 256) as TWO launches per frame:
//
//   1. frame_records_kernel   geometry ONCE per frame: every (view, BEV cell) cube is projected once (reference
//                             vfa/model/vfa_op.py:64-88, vfa/utils.py:50-59) and, per feature scale, turned into a 96-byte
//                             box record (the 16 bilinear tap weights, 1 / area, visibility, tap coordinates: vfa_op.py:
//                             104-106 and the set-up half of :112-115) plus, per (view, 8 x 4-cell tile, scale), the window
//                             of the integral image that holds all of the tile's taps.
//   2. pool_collapse_kernel   one persistent 512-thread workgroup per CU walks tiles; per (tile, scale, view) it brings the
//                             tile's tap window into LDS by LDS-DMA (each DISTINCT tap is read from L2 / HBM once), pools
//                             the 32 boxes with the reference's FMA chains (vfa_op.py:112-119) straight into bf16 hi/lo
//                             planes in LDS, multiplies them with `collapse.weight` on the matrix cores (vfa_op.py:123, three
//                             bf16 MFMA products per fp32 product, fp32 accumulation), adds bias + ReLU (:124) and sums
//                             views and scales in registers (vfa/model/vfanet.py:79, 82).  The voxel features never touch
//                             HBM and the BEV map is written exactly once.
//
// Numerics: tap chains and the box sum are the reference's exact fp32 sequence (same device code as the bit-exact pooling
// kernels); the quotient is v * RN(1 / area) (<= 1.5 ulp from the reference's division -- far below the 2^-17 of the bf16
// split that follows) and the product is the bf16-split MFMA arithmetic of vfa_collapse.hip: within the path's post-GEMM
// tolerance (rtol 1e-4, atol 1e-5 max|ref|), not bitwise -- no GEMM order is.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vfa_geom.h"

namespace {
using namespace vfa_dev;

constexpr int kTileW = 8, kTileL = 4, kTileBoxes = kTileW * kTileL; // 32 cells = one 32-row MFMA block
constexpr int kC = 256;                                             // channels in = channels out
constexpr int kRecBytes = 96, kHdrBytes = 32;
constexpr int kMaxScales = 3;
constexpr int kSlotBytes = kC * 4;                                  // one tap = 256 fp32
constexpr int kMaxSlots = 126;                                      // LDS tap window of a (tile, view, scale)
constexpr int kThreads = 512;
constexpr int kRowBytes = 2 * kC;                                   // one bf16 plane row; 16-byte chunks XOR-swizzled with (row & 15)
constexpr int kPlane = kTileBoxes * kRowBytes;                      // 16 KiB
constexpr int kSteps = kC / 16;                                     // k-steps of v_mfma_f32_32x32x16_bf16

// record flags
constexpr int kVis = 1, kCont = 1 << 8; // bits 1-2 DXC, bits 3-4 DYC; kCont: same tap set as the previous box of the 4-box chunk
// tile header flags
constexpr int kTileLive = 1, kTileDirect = 2;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct ScaleDims { int Hf, Wf; };

struct RecordArgs {
    BoxGeom g;
    int n_views, L, W, tiles_w, n_tiles, n_scales;
    ScaleDims dims[kMaxScales];
    unsigned *live[kMaxScales];      // (n_tiles) bit v = view v has a visible box in the tile
    unsigned char *hdrs[kMaxScales]; // (n_views, n_tiles, 32 B)
    unsigned char *recs[kMaxScales]; // (n_views, n_tiles, 32 boxes, 96 B)
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// ------------------------------------------------------------------------------------------------
// 1. geometry of the frame: one half-wave (32 lanes) per (view, tile), lane = cell of the tile (4 rows of 8)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void frame_records_kernel(RecordArgs a)
{
    const int lane = threadIdx.x, half = lane >> 5, b = lane & 31;
    const long long pair = (long long)blockIdx.x * 2 + half;
    const bool pair_ok = pair < (long long)a.n_views * a.n_tiles;
    const int view = pair_ok ? (int)(pair / a.n_tiles) : 0, tile = pair_ok ? (int)(pair % a.n_tiles) : 0;
    const int tl = tile / a.tiles_w, tw = tile - tl * a.tiles_w;
    const int cl = tl * kTileL + (b >> 3), cw = tw * kTileW + (b & 7);
    const bool valid = pair_ok && cl < a.L && cw < a.W;
    const int cell = valid ? cl * a.W + cw : 0;

    // the cube once per (view, cell): scale-independent                     vfa_op.py:64-88, utils.py:56-59
    float l, t, r, bt;
    {
        const float *P = a.g.calibs + (size_t)view * 12;
        const float gx = a.g.grid[cell * 3 + 0] + 0.0f; // + the int64 zeros of z_corners (vfa_op.py:52, :64)
        const float gy = a.g.grid[cell * 3 + 1] + 0.0f;
        const float gz = a.g.grid[cell * 3 + 2] + a.g.z_layers[0];
        l = t = r = bt = 0.0f;
#pragma unroll 1
        for (int k = 0; k < 8; ++k) {
            float nu, nv;
            project_corner(a.g, P, gx, gy, gz, k, nu, nv);
            if (k == 0) { l = r = nu; t = bt = nv; }
            else { l = min_t(l, nu); r = max_t(r, nu); t = min_t(t, nv); bt = max_t(bt, nv); }
        }
    }
#pragma unroll 1
    for (int s = 0; s < a.n_scales; ++s) {
        const int Hf = a.dims[s].Hf, Wf = a.dims[s].Wf;
        const float area = box_area(l, t, r, bt, Hf, Wf);                                     // vfa_op.py:104-105
        const bool vis = valid && box_visible(area, Hf, Wf);                                  // :106
        const float masked = valid ? area * 0.0f : 0.0f; // value of a masked voxel: 0, or NaN when the box itself is NaN
        const bool live_box = vis || (valid && masked != masked);
        const Axis xl = make_axis(l, Wf), xr = make_axis(r, Wf), yt = make_axis(t, Hf), yb = make_axis(bt, Hf);
        const int dx = xr.i0 - xl.i0, dy = yb.i0 - yt.i0;
        const int dxc = dx == 0 ? 0 : (dx == 1 ? 1 : 2), dyc = dy == 0 ? 0 : (dy == 1 ? 1 : 2);
        // tap coordinates, out-of-image taps redirected to the zero border (coordinate -1 or Hf / Wf)
        const int xs[4] = {clampi(xl.i0, -1, Wf), clampi(xl.i0 + 1, -1, Wf), clampi(xr.i0, -1, Wf), clampi(xr.i0 + 1, -1, Wf)};
        const int ys[4] = {clampi(yt.i0, -1, Hf), clampi(yt.i0 + 1, -1, Hf), clampi(yb.i0, -1, Hf), clampi(yb.i0 + 1, -1, Hf)};
        // window of the tile over its VISIBLE boxes: columns [x0, x1], top rows [t0, t1], bottom rows [b0, b1]
        constexpr int kBig = 1 << 20;
        int x0 = vis ? min(xs[0], xs[2]) : kBig, x1 = vis ? max(xs[1], xs[3]) : -kBig;
        int t0 = vis ? ys[0] : kBig, t1 = vis ? ys[1] : -kBig, b0 = vis ? ys[2] : kBig, b1 = vis ? ys[3] : -kBig;
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) {
            x0 = min(x0, __shfl_xor(x0, m, 32)); x1 = max(x1, __shfl_xor(x1, m, 32));
            t0 = min(t0, __shfl_xor(t0, m, 32)); t1 = max(t1, __shfl_xor(t1, m, 32));
            b0 = min(b0, __shfl_xor(b0, m, 32)); b1 = max(b1, __shfl_xor(b1, m, 32));
        }
        const unsigned long long vis_all = __ballot(vis), live_all = __ballot(live_box);
        const bool any_vis = ((vis_all >> (32 * half)) & 0xffffffffull) != 0ull;
        const bool any_live = ((live_all >> (32 * half)) & 0xffffffffull) != 0ull;
        int cwid = 0, top_rows = 0, bot_rows = 0, n_slots = 0;
        if (any_vis) {
            cwid = x1 - x0 + 1;
            if (b0 <= t1 + 1) { // the bands touch or overlap: one band [t0, max(t1, b1)]
                top_rows = max(t1, b1) - t0 + 1;
                bot_rows = 0;
                b0 = t0 + top_rows; // rows >= b0 would start the (empty) second band
            } else {
                top_rows = t1 - t0 + 1;
                bot_rows = b1 - b0 + 1;
            }
            n_slots = cwid * (top_rows + bot_rows);
        }
        const bool direct = n_slots > kMaxSlots;
        auto slot_row = [&](int y) { return y < t0 + top_rows ? y - t0 : top_rows + (y - b0); };
        unsigned rows[4], cols[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (direct) { rows[k] = (unsigned)(ys[k] + 1); cols[k] = (unsigned)(xs[k] + 1); }
            else { rows[k] = (unsigned)(slot_row(ys[k]) * cwid); cols[k] = (unsigned)(xs[k] - x0); }
        }
        // boxes that continue the tap set of their left neighbour inside a 4-box chunk reuse its register patch
        const int tag = (vis ? 1 : 0) | (dxc << 1) | (dyc << 3);
        const unsigned kx = (unsigned)(xs[0] + 1) | ((unsigned)(xs[2] + 1) << 16), ky = (unsigned)(ys[0] + 1) | ((unsigned)(ys[2] + 1) << 16);
        const int tag_p = __shfl_up(tag, 1);
        const unsigned kx_p = __shfl_up(kx, 1), ky_p = __shfl_up(ky, 1);
        const bool cont = vis && (b & 3) != 0 && tag_p == tag && kx_p == kx && ky_p == ky;

        float w[16];
        {
            float q[4];
            bilinear_weights(q, xl, yt); w[0] = q[0]; w[1] = q[1]; w[2] = q[2]; w[3] = q[3];       // lt
            bilinear_weights(q, xr, yb); w[4] = q[0]; w[5] = q[1]; w[6] = q[2]; w[7] = q[3];       // rb
            bilinear_weights(q, xr, yt); w[8] = q[0]; w[9] = q[1]; w[10] = q[2]; w[11] = q[3];     // rt
            bilinear_weights(q, xl, yb); w[12] = q[0]; w[13] = q[1]; w[14] = q[2]; w[15] = q[3];   // lb
        }
        if (pair_ok) {
            uint4 *rec = reinterpret_cast<uint4 *>(a.recs[s] + (((size_t)view * a.n_tiles + tile) * kTileBoxes + b) * kRecBytes);
            rec[0] = make_uint4(__float_as_uint(w[0]), __float_as_uint(w[1]), __float_as_uint(w[2]), __float_as_uint(w[3]));
            rec[1] = make_uint4(__float_as_uint(w[4]), __float_as_uint(w[5]), __float_as_uint(w[6]), __float_as_uint(w[7]));
            rec[2] = make_uint4(__float_as_uint(w[8]), __float_as_uint(w[9]), __float_as_uint(w[10]), __float_as_uint(w[11]));
            rec[3] = make_uint4(__float_as_uint(w[12]), __float_as_uint(w[13]), __float_as_uint(w[14]), __float_as_uint(w[15]));
            const float rcp = 1.0f / area; // correctly rounded
            rec[4] = make_uint4(__float_as_uint(rcp), (unsigned)tag | (cont ? (unsigned)kCont : 0u), rows[0] | (rows[1] << 16),
                                rows[2] | (rows[3] << 16));
            rec[5] = make_uint4(cols[0] | (cols[1] << 16), cols[2] | (cols[3] << 16), __float_as_uint(masked), __float_as_uint(area));
            if (b == 0) {
                uint4 *hdr = reinterpret_cast<uint4 *>(a.hdrs[s] + ((size_t)view * a.n_tiles + tile) * kHdrBytes);
                const int inv = cwid > 0 ? (65536 + cwid - 1) / cwid : 0; // floor(s / cwid) == (s * inv) >> 16 for s < 128
                hdr[0] = make_uint4((any_live ? kTileLive : 0) | (direct ? kTileDirect : 0), (unsigned)n_slots, (unsigned)cwid, (unsigned)inv);
                hdr[1] = make_uint4((unsigned)x0, (unsigned)t0, (unsigned)top_rows, (unsigned)b0);
                if (any_live) atomicOr(a.live[s] + tile, 1u << view);
            }
        }
    }
}

// collapse.weight (N = 256, K = 256) fp32 -> bf16 hi / lo planes in MFMA B-fragment order:
//   out[((wave * 16 + s) * 2 + plane) * 64 + lane] (16 B) = W[n = 32 wave + (lane & 31)][k = 16 s + 8 (lane >> 5) + j], j = 0..7
__global__ __launch_bounds__(256) void split_weight_frag_kernel(const float *__restrict__ w, uint4 *__restrict__ out)
{
    const int idx = blockIdx.x * 256 + threadIdx.x; // (wave, s, lane)
    if (idx >= 8 * kSteps * 64) return;
    const int lane = idx & 63, s = (idx >> 6) & 15, wave = idx >> 10;
    const float *src = w + (size_t)(wave * 32 + (lane & 31)) * kC + 16 * s + 8 * (lane >> 5);
    union { __bf16 b[8]; uint4 u; } hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = src[j];
        hi.b[j] = (__bf16)x;
        lo.b[j] = (__bf16)(x - (float)hi.b[j]);
    }
    out[((size_t)(wave * kSteps + s) * 2 + 0) * 64 + lane] = hi.u;
    out[((size_t)(wave * kSteps + s) * 2 + 1) * 64 + lane] = lo.u;
}

// ------------------------------------------------------------------------------------------------
// 2. pooling + collapse + bias + ReLU + view / scale sum
// ------------------------------------------------------------------------------------------------
struct FusedScale {
    const float *integral;          // (n_views, Hf+2, Wf+2, 256) zero-bordered channels-last
    const float *bias;              // (256) or NULL
    const uint4 *wfrag;             // split_weight_frag_kernel output
    const unsigned *live;           // (n_tiles)
    const unsigned char *hdrs, *recs;
    int Hf, Wf;
};
struct FusedArgs {
    FusedScale sc[kMaxScales];
    int n_scales, n_views, L, W, tiles_w, n_tiles;
    float *out;                     // (L * W, 256)
    int accumulate;
};

struct Frag { bf16x8 hi, lo; };

__device__ __forceinline__ float relu_t(float x) { return (x < 0.0f) ? 0.0f : x; } // NaN stays NaN

// one box record in SGPRs (wave-uniform): dwords 0-15 the tap weights, 16-23 rcp, flags, rows, cols, masked, area
struct SRec { f32x16 w; i32x8 t; };
__device__ __forceinline__ void sload_rec(SRec &o, const unsigned char *p)
{
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, 0x40" : "=&s"(o.w), "=&s"(o.t) : "s"(p));
}
__device__ __forceinline__ void swait_rec(SRec &o) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(o.w), "+s"(o.t)); }

__device__ __forceinline__ float4 mul4(float4 a, float w) { return make_float4(a.x * w, a.y * w, a.z * w, a.w * w); }
__device__ __forceinline__ float4 fma4(float4 a, float w, float4 c)
{
    return make_float4(fmaf(a.x, w, c.x), fmaf(a.y, w, c.y), fmaf(a.z, w, c.z), fmaf(a.w, w, c.w));
}
// bilinear sample from the four rounded weights, taps in the order nw, ne, sw, se: one product, three FMAs
__device__ __forceinline__ float4 sample4(float4 nw, float4 ne, float4 sw, float4 se, float w0, float w1, float w2, float w3)
{
    float4 v = mul4(nw, w0);
    v = fma4(ne, w1, v);
    v = fma4(sw, w2, v);
    v = fma4(se, w3, v);
    return v;
}
__device__ __forceinline__ float box_quot(float lt, float rb, float rt, float lb, float rcp)
{
    float v = lt + rb; // (((lt + rb) - rt) - lb): the reference's order (A.6)
    v = v - rt;
    v = v - lb;
    return v * rcp;
}

// The 16 taps of a box are {top rows yt, yt+1, bottom rows yb, yb+1} x {left cols xl, xl+1, right cols xr, xr+1}; when the pairs
// coincide or overlap (DYC / DXC = 0 or 1) the shared rows / columns are loaded once.  The two samples of the top rows (lt,
// rt) are formed first, then the bottom rows replace the top ones in registers (at most 2 rows x 4 columns of taps are live:
// the W fragments hold half of the register file).  LDS: `lds` = tap window, offsets in slots; DIRECT: `img` = the view's
// padded integral image, offsets in pixels.  Every variant is a straight-line body.
template <bool DIRECT>
__device__ __forceinline__ float4 tap_at(const float4 *lds, const char *img, unsigned row, unsigned col, int lane)
{
    if constexpr (DIRECT) return *reinterpret_cast<const float4 *>(img + ((size_t)(row + col) * kSlotBytes + lane * 16));
    else return lds[(row + col) * 64 + lane];
}
template <bool DIRECT, int DXC>
__device__ __forceinline__ void load_row(float4 (&R)[4], const float4 *lds, const char *img, unsigned row, const unsigned (&col)[4], int lane)
{
    constexpr int NC = DXC == 0 ? 2 : (DXC == 1 ? 3 : 4);
#pragma unroll
    for (int c = 0; c < NC; ++c) R[c] = tap_at<DIRECT>(lds, img, row, col[(DXC == 1 && c == 2) ? 3 : c], lane); // unique cols: {0,1}, {0,1,3}, {0,1,2,3}
}
template <bool DIRECT, int DYC, int DXC>
__device__ __forceinline__ float4 pool_box(const float4 *lds, const char *img, const unsigned (&row)[4], const unsigned (&col)[4],
                                           int lane, const f32x16 &w, float rcp)
{
    constexpr int CR0 = DXC == 0 ? 0 : (DXC == 1 ? 1 : 2), CR1 = CR0 + 1; // right column pair inside a loaded row
    float4 A[4], B[4];
    load_row<DIRECT, DXC>(A, lds, img, row[0], col, lane);
    load_row<DIRECT, DXC>(B, lds, img, row[1], col, lane);
    const float4 lt = sample4(A[0], A[1], B[0], B[1], w[0], w[1], w[2], w[3]);
    const float4 rt = sample4(A[CR0], A[CR1], B[CR0], B[CR1], w[8], w[9], w[10], w[11]);
    float4 lb, rb;
    if constexpr (DYC == 0) { // bottom rows = top rows
        lb = sample4(A[0], A[1], B[0], B[1], w[12], w[13], w[14], w[15]);
        rb = sample4(A[CR0], A[CR1], B[CR0], B[CR1], w[4], w[5], w[6], w[7]);
    } else if constexpr (DYC == 1) { // bottom rows = (yt + 1, yb + 1): the second top row stays
        load_row<DIRECT, DXC>(A, lds, img, row[3], col, lane);
        lb = sample4(B[0], B[1], A[0], A[1], w[12], w[13], w[14], w[15]);
        rb = sample4(B[CR0], B[CR1], A[CR0], A[CR1], w[4], w[5], w[6], w[7]);
    } else {
        load_row<DIRECT, DXC>(A, lds, img, row[2], col, lane);
        load_row<DIRECT, DXC>(B, lds, img, row[3], col, lane);
        lb = sample4(A[0], A[1], B[0], B[1], w[12], w[13], w[14], w[15]);
        rb = sample4(A[CR0], A[CR1], B[CR0], B[CR1], w[4], w[5], w[6], w[7]);
    }
    return make_float4(box_quot(lt.x, rb.x, rt.x, lb.x, rcp), box_quot(lt.y, rb.y, rt.y, lb.y, rcp),
                       box_quot(lt.z, rb.z, rt.z, lb.z, rcp), box_quot(lt.w, rb.w, rt.w, lb.w, rcp));
}

// x = hi + lo + r exactly in fp32 arithmetic: hi = RNE bf16(x), lo = RNE bf16(x - hi); row `row` of the A tile, channels
// 4 lane .. 4 lane + 3, written into the XOR-swizzled hi / lo planes
__device__ __forceinline__ void store_row(unsigned char *planes, int row, int lane, float4 v)
{
    const float x[4] = {v.x, v.y, v.z, v.w};
    union { __bf16 b[4]; uint2 u; } hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi.b[j] = (__bf16)x[j];
        lo.b[j] = (__bf16)(x[j] - (float)hi.b[j]);
    }
    const int off = row * kRowBytes + ((((lane >> 1) ^ (row & 15)) << 4) | ((lane & 1) << 3));
    *reinterpret_cast<uint2 *>(planes + off) = hi.u;
    *reinterpret_cast<uint2 *>(planes + kPlane + off) = lo.u;
}

struct Item { int tile, scale, view; unsigned rest; bool valid; }; // rest: live views of (tile, scale) above `view`

template <int TERMS>
__global__ __launch_bounds__(kThreads) void pool_collapse_kernel(FusedArgs a)
{
    __shared__ float4 s_taps[kMaxSlots * 64];                       // 126 KiB: the tap window of the current item
    __shared__ __align__(16) unsigned char s_planes[2 * kPlane];    // 32 KiB: bf16 hi / lo planes of the 32 x 256 A tile
    const int tid = threadIdx.x, wave = uniform_i(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;

    // contiguous range of tiles for this workgroup; neighbouring ranges share an XCD (their tap windows overlap)
    const int nblk = gridDim.x;
    const int lb = (int)xcd_contiguous(blockIdx.x, (nblk + 7) / 8);
    if (lb >= nblk) return;
    const int t_begin = (int)((long long)a.n_tiles * lb / nblk), t_end = (int)((long long)a.n_tiles * (lb + 1) / nblk);
    if (t_begin >= t_end) return;
    const unsigned view_mask = a.n_views >= 32 ? 0xffffffffu : ((1u << a.n_views) - 1u);

    auto live_of = [&](int tile, int scale) { return (unsigned)uniform_i((int)(a.sc[scale].live[tile] & view_mask)); };
    // first live item at or after (tile, scale) with view bits `rest`
    auto seek = [&](int tile, int scale, unsigned rest) {
        Item it;
        it.valid = false; it.tile = tile; it.scale = scale; it.view = 0; it.rest = 0;
        while (tile < t_end) {
            if (rest) {
                it.view = __builtin_ctz(rest);
                it.rest = rest & (rest - 1u);
                it.tile = tile; it.scale = scale; it.valid = true;
                return it;
            }
            if (++scale == a.n_scales) { scale = 0; ++tile; }
            if (tile < t_end) rest = live_of(tile, scale);
        }
        return it;
    };

    float bcol[kMaxScales], brelu[kMaxScales];
#pragma unroll
    for (int s = 0; s < kMaxScales; ++s) {
        bcol[s] = (s < a.n_scales && a.sc[s].bias) ? a.sc[s].bias[wave * 32 + r] : 0.0f;
        brelu[s] = relu_t(bcol[s]);
    }

    // output rows of a tile: register i of lane (r, h) is row (i & 3) + 8 (i >> 2) + 4 h of the 32 x 32 MFMA block, column r
    auto write_tile = [&](int tile, const f32x16 &sum, bool have_sum) {
        const int tl = tile / a.tiles_w, tw = tile - tl * a.tiles_w;
        float extra = 0.0f; // fully masked (view, scale) items of this tile: vox = 0 -> relu(bias)
#pragma unroll
        for (int s = 0; s < kMaxScales; ++s)
            if (s < a.n_scales) extra += (float)(a.n_views - __popc(live_of(tile, s))) * brelu[s];
        float *ocol = a.out + wave * 32 + r;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
            const int cl = tl * kTileL + (row >> 3), cw = tw * kTileW + (row & 7);
            if (cl < a.L && cw < a.W) {
                float *o = ocol + (size_t)(cl * a.W + cw) * kC;
                float v = (have_sum ? sum[i] : 0.0f) + extra;
                if (a.accumulate) v += *o; // the workgroup owns these rows: a plain read-modify-write
                *o = v;
            }
        }
    };

    // LDS-DMA of the item's tap window: slot s = (window row, window column), 1 KiB per wave instruction
    auto header_of = [&](const Item &it, uint4 &h0, uint4 &h1) {
        const uint4 *hdr = reinterpret_cast<const uint4 *>(a.sc[it.scale].hdrs + ((size_t)it.view * a.n_tiles + it.tile) * kHdrBytes);
        h0 = hdr[0];
        h1 = hdr[1];
    };
    auto issue_fills = [&](const Item &it, const uint4 &h0, const uint4 &h1) {
        const FusedScale &sc = a.sc[it.scale];
        const int flags = uniform_i((int)h0.x), n_slots = uniform_i((int)h0.y), cwid = uniform_i((int)h0.z), inv = uniform_i((int)h0.w);
        if (flags & kTileDirect) return;
        const int x0 = uniform_i((int)h1.x), t0 = uniform_i((int)h1.y), top_rows = uniform_i((int)h1.z), b0 = uniform_i((int)h1.w);
        const int Wp = sc.Wf + 2;
        const char *img = reinterpret_cast<const char *>(sc.integral) + (size_t)it.view * (sc.Hf + 2) * Wp * kSlotBytes;
        for (int s = wave; s < n_slots; s += kThreads / 64) {
            const int wr = (s * inv) >> 16, wc = s - wr * cwid;
            const int y = wr < top_rows ? t0 + wr : b0 + (wr - top_rows), x = x0 + wc;
            const char *src = img + ((size_t)(y + 1) * Wp + (x + 1)) * kSlotBytes + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(s_taps + s * 64), 16, 0, 0);
        }
    };

    // pool this wave's four boxes of the item into rows 4 wave .. 4 wave + 3 of the A tile
    auto pool = [&](const Item &it, int tflags) {
        const FusedScale &sc = a.sc[it.scale];
        const unsigned char *recs = sc.recs + (((size_t)it.view * a.n_tiles + it.tile) * kTileBoxes + 4 * wave) * kRecBytes;
        const bool direct = (tflags & kTileDirect) != 0;
        const unsigned Wp = (unsigned)sc.Wf + 2u;
        const char *img = reinterpret_cast<const char *>(sc.integral) + (size_t)it.view * (sc.Hf + 2) * Wp * kSlotBytes;
        SRec rc;
        sload_rec(rc, recs);
        swait_rec(rc);
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            SRec nx;
            sload_rec(nx, recs + (j + 1) * kRecBytes); // one spare record follows the last tile of the workspace
            const int flags = rc.t[1];
            const int row = 4 * wave + j;
            if (!(flags & kVis)) {
                const float z = __int_as_float(rc.t[6]);
                store_row(s_planes, row, lane, make_float4(z, z, z, z));
            } else {
                const float rcp = __int_as_float(rc.t[0]);
                unsigned rw[4] = {(unsigned)rc.t[2] & 0xffffu, (unsigned)rc.t[2] >> 16, (unsigned)rc.t[3] & 0xffffu, (unsigned)rc.t[3] >> 16};
                const unsigned cl[4] = {(unsigned)rc.t[4] & 0xffffu, (unsigned)rc.t[4] >> 16, (unsigned)rc.t[5] & 0xffffu, (unsigned)rc.t[5] >> 16};
                if (direct) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) rw[k] *= Wp;
                }
                float4 res;
#define VFA_VARIANT(DY, DX)                                                                          \
    res = direct ? pool_box<true, DY, DX>(s_taps, img, rw, cl, lane, rc.w, rcp)                      \
                 : pool_box<false, DY, DX>(s_taps, img, rw, cl, lane, rc.w, rcp);                    \
    break;
                switch ((flags >> 1) & 15) { // DXC | DYC << 2
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
                store_row(s_planes, row, lane, res);
            }
            swait_rec(nx);
            rc = nx;
        }
    };

    Frag w[kSteps];
    auto load_weights = [&](int scale) {
        const uint4 *src = a.sc[scale].wfrag + (size_t)wave * kSteps * 2 * 64 + lane;
#pragma unroll
        for (int s = 0; s < kSteps; ++s) {
            const uint4 uh = src[(s * 2 + 0) * 64], ul = src[(s * 2 + 1) * 64];
            w[s].hi = *reinterpret_cast<const bf16x8 *>(&uh);
            w[s].lo = *reinterpret_cast<const bf16x8 *>(&ul);
        }
    };

    const int key = r & 15;
    const int frag_base = r * kRowBytes + ((h ^ (key & 1)) << 4);

    // tiles in front of the first live item are fully masked
    Item cur = seek(t_begin, 0, live_of(t_begin, 0));
    {
        f32x16 none;
        for (int t2 = t_begin; t2 < (cur.valid ? cur.tile : t_end); ++t2) write_tile(t2, none, false);
    }
    if (!cur.valid) return;
    uint4 nh0, nh1;
    header_of(cur, nh0, nh1);
    issue_fills(cur, nh0, nh1);
    int cur_flags = uniform_i((int)nh0.x);
    int w_scale = -1;
    f32x16 sum;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum[i] = 0.0f;

    while (cur.valid) {
        const Item nxt = seek(cur.tile, cur.scale, cur.rest);
        if (nxt.valid) header_of(nxt, nh0, nh1); // needed after the MFMA-side barrier: in flight until then
        if (cur.scale != w_scale) { // W of this scale: lands while the boxes are pooled
            load_weights(cur.scale);
            w_scale = cur.scale;
        }
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0): this wave's share of the tap window has landed
        __syncthreads();                    // ... and everybody else's
        pool(cur, cur_flags);
        __syncthreads();                    // A tile complete; the tap window is free again
        if (nxt.valid) issue_fills(nxt, nh0, nh1); // in flight under the MFMAs below

        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        const unsigned char *pa = s_planes + frag_base;
#pragma unroll
        for (int c = 0; c < kSteps / 2; ++c) {
            const int off0 = ((2 * c) ^ (key >> 1)) << 5, off1 = ((2 * c + 1) ^ (key >> 1)) << 5;
            const bf16x8 h0 = *reinterpret_cast<const bf16x8 *>(pa + off0);
            const bf16x8 h1 = *reinterpret_cast<const bf16x8 *>(pa + off1);
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
        const float bc = cur.scale == 0 ? bcol[0] : (cur.scale == 1 ? bcol[1] : bcol[2]);
#pragma unroll
        for (int i = 0; i < 16; ++i) sum[i] = sum[i] + relu_t(acc[i] + bc); // vfa_op.py:124; vfanet.py:79, 82

        if (!nxt.valid || nxt.tile != cur.tile) {
            write_tile(cur.tile, sum, true);
#pragma unroll
            for (int i = 0; i < 16; ++i) sum[i] = 0.0f;
            for (int t2 = cur.tile + 1; t2 < (nxt.valid ? nxt.tile : t_end); ++t2) write_tile(t2, sum, false);
        }
        cur = nxt;
        cur_flags = uniform_i((int)nh0.x);
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WorkspaceLayout {
    size_t live[kMaxScales], hdrs[kMaxScales], recs[kMaxScales], wfrag[kMaxScales], total;
    int tiles_l, tiles_w, n_tiles;
};
inline WorkspaceLayout layout_of(int n_views, int L, int W, int n_scales)
{
    WorkspaceLayout w;
    w.tiles_l = (L + kTileL - 1) / kTileL;
    w.tiles_w = (W + kTileW - 1) / kTileW;
    w.n_tiles = w.tiles_l * w.tiles_w;
    size_t off = 0;
    for (int s = 0; s < kMaxScales; ++s) {
        const bool on = s < n_scales;
        w.live[s] = off;  off = align_up(off + (on ? (size_t)w.n_tiles * 4 : 0), 256);
        w.hdrs[s] = off;  off = align_up(off + (on ? (size_t)n_views * w.n_tiles * kHdrBytes : 0), 256);
        w.recs[s] = off;  off = align_up(off + (on ? ((size_t)n_views * w.n_tiles * kTileBoxes + 1) * kRecBytes : 0), 256);
        w.wfrag[s] = off; off = align_up(off + (on ? (size_t)8 * kSteps * 2 * 64 * 16 : 0), 256);
    }
    w.total = off;
    return w;
}

} // namespace

extern "C" {

size_t vfa_frame_workspace_bytes(int n_views, int L, int W, int n_scales)
{
    if (n_views < 0 || L < 0 || W < 0 || n_scales < 1 || n_scales > kMaxScales) return 0;
    return layout_of(n_views, L, W, n_scales).total;
}

int vfa_frame_records_f32(const float *calibs, const float *grid, const float *z_layers, const float *corner_off, int n_views, int L,
                          int W, int conv_kind, float img_w, float img_h, float cmin, float cmax, int n_scales,
                          const int *feat_hw, const float *const *weights, void *workspace, size_t workspace_bytes, void *stream)
{
    if (n_views < 0 || L < 0 || W < 0 || n_scales < 1 || n_scales > kMaxScales || conv_kind < 0 || conv_kind > 2 || !feat_hw)
        return VFA_ERR_BAD_ARGUMENT;
    if (n_views > 32) return VFA_ERR_UNSUPPORTED; // live-view masks are 32 bits wide
    const WorkspaceLayout lay = layout_of(n_views, L, W, n_scales);
    if (lay.n_tiles == 0 || n_views == 0) return 0;
    if ((long long)n_views * lay.n_tiles >= (1ll << 31) - 2) return VFA_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < lay.total) return VFA_ERR_BAD_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    RecordArgs a;
    a.g = BoxGeom{calibs, grid, z_layers, corner_off, conv_kind, img_w, img_h, cmin, cmax};
    a.n_views = n_views; a.L = L; a.W = W; a.tiles_w = lay.tiles_w; a.n_tiles = lay.n_tiles; a.n_scales = n_scales;
    for (int k = 0; k < kMaxScales; ++k) {
        a.dims[k].Hf = k < n_scales ? feat_hw[2 * k] : 1;
        a.dims[k].Wf = k < n_scales ? feat_hw[2 * k + 1] : 1;
        if (a.dims[k].Hf <= 0 || a.dims[k].Wf <= 0 || a.dims[k].Hf > 65533 || a.dims[k].Wf > 65533) return VFA_ERR_BAD_ARGUMENT;
        a.live[k] = reinterpret_cast<unsigned *>(ws + lay.live[k]);
        a.hdrs[k] = ws + lay.hdrs[k];
        a.recs[k] = ws + lay.recs[k];
    }
    for (int k = 0; k < n_scales; ++k) {
        hipError_t e = hipMemsetAsync(ws + lay.live[k], 0, (size_t)lay.n_tiles * 4, s);
        if (e == hipSuccess) // the spare record behind the last tile (prefetched, never used)
            e = hipMemsetAsync(ws + lay.recs[k] + (size_t)n_views * lay.n_tiles * kTileBoxes * kRecBytes, 0, kRecBytes, s);
        if (e != hipSuccess) return (int)e;
    }
    const long long pairs = (long long)n_views * lay.n_tiles;
    hipLaunchKernelGGL(frame_records_kernel, dim3((unsigned)((pairs + 1) / 2)), dim3(kWave), 0, s, a);
    int st = (int)hipGetLastError();
    if (st) return st;
    if (weights)
        for (int k = 0; k < n_scales; ++k) {
            if (!weights[k]) return VFA_ERR_BAD_ARGUMENT;
            hipLaunchKernelGGL(split_weight_frag_kernel, dim3(8 * kSteps * 64 / 256), dim3(256), 0, s, weights[k],
                               reinterpret_cast<uint4 *>(ws + lay.wfrag[k]));
            st = (int)hipGetLastError();
            if (st) return st;
        }
    return 0;
}

int vfa_pool_collapse_relu_sum_f32(const float *const *integrals, const float *const *biases, const void *workspace,
                                   size_t workspace_bytes, float *out, int n_views, int L, int W, int n_scales, const int *feat_hw,
                                   int accumulate, int flags, void *stream)
{
    const int terms = flags & VFA_FLAG_TERMS_MASK, reserved_cus = (flags >> 8) & 0xff;
    if (flags & ~(VFA_FLAG_TERMS_MASK | 0xff00)) return VFA_ERR_BAD_ARGUMENT;
    if (n_views < 0 || L < 0 || W < 0 || n_scales < 1 || n_scales > kMaxScales || !feat_hw || !integrals ||
        (terms != 0 && terms != 3 && terms != 4))
        return VFA_ERR_BAD_ARGUMENT;
    if (n_views > 32) return VFA_ERR_UNSUPPORTED;
    const WorkspaceLayout lay = layout_of(n_views, L, W, n_scales);
    if (lay.n_tiles == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (n_views == 0) {
        if (!accumulate) return (int)hipMemsetAsync(out, 0, (size_t)L * W * kC * sizeof(float), s);
        return 0;
    }
    if (!workspace || workspace_bytes < lay.total) return VFA_ERR_BAD_ARGUMENT;
    const unsigned char *ws = reinterpret_cast<const unsigned char *>(workspace);
    FusedArgs a;
    for (int k = 0; k < kMaxScales; ++k) {
        const int q = k < n_scales ? k : 0;
        a.sc[k].integral = integrals[q];
        a.sc[k].bias = biases ? biases[q] : nullptr;
        a.sc[k].wfrag = reinterpret_cast<const uint4 *>(ws + lay.wfrag[q]);
        a.sc[k].live = reinterpret_cast<const unsigned *>(ws + lay.live[q]);
        a.sc[k].hdrs = ws + lay.hdrs[q];
        a.sc[k].recs = ws + lay.recs[q];
        a.sc[k].Hf = feat_hw[2 * q];
        a.sc[k].Wf = feat_hw[2 * q + 1];
        if (!a.sc[k].integral) return VFA_ERR_BAD_ARGUMENT;
    }
    a.n_scales = n_scales; a.n_views = n_views; a.L = L; a.W = W; a.tiles_w = lay.tiles_w; a.n_tiles = lay.n_tiles;
    a.out = out; a.accumulate = accumulate;
    int n_cu = 256;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            cus > 0)
            n_cu = cus;
    }
    if (reserved_cus > 0 && n_cu - reserved_cus >= 8) n_cu -= reserved_cus;
    int nblk = lay.n_tiles < n_cu ? lay.n_tiles : n_cu;
    nblk = (nblk + 7) / 8 * 8; // xcd_contiguous deals whole eighths; surplus blocks find an empty range and leave
    if (terms == 4)
        hipLaunchKernelGGL((pool_collapse_kernel<4>), dim3(nblk), dim3(kThreads), 0, s, a);
    else
        hipLaunchKernelGGL((pool_collapse_kernel<3>), dim3(nblk), dim3(kThreads), 0, s, a);
    return (int)hipGetLastError();
}

} // extern "C"
