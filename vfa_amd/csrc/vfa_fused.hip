// vfa_fused.hip -- the inference hot path of single-layer grids (nl = 1, C = 256) as TWO launches per frame:
//
//   1. frame_records_kernel   geometry ONCE per frame: every (view, BEV cell) cube is projected once (reference
//                             vfa/model/vfa_op.py:64-88, vfa/utils.py:50-59) and, per feature scale, turned into a 96-byte
//                             box record (the 16 bilinear tap weights, 1 / area, visibility, tap coordinates: vfa_op.py:
//                             104-106 and the set-up half of :112-115) plus, per (view, 8 x 4-cell tile, scale), the window
//                             of the integral image that holds all of the tile's taps.
//   2. pool_collapse_kernel   one persistent 512-thread workgroup per CU walks tiles; per (tile, scale, view) it brings the
//                             tile's tap window into LDS by LDS-DMA (each DISTINCT tap is read from L2 / HBM once), pools
//                             the 32 boxes with the reference's FMA chains (vfa_op.py:112-119) straight into two 16-bit
//                             planes in LDS, multiplies them with `collapse.weight` on the matrix cores (vfa_op.py:123, three
//                             MFMA products per fp32 product, fp32 accumulation), adds bias + ReLU (:124) and sums
//                             views and scales in registers (vfa/model/vfanet.py:79, 82).  The voxel features never touch
//                             HBM and the BEV map is written exactly once.
//      (+ pool_rows_kernel, a pre-pass over the few items whose window exceeds LDS, and an empty second launch for those of them
//       that found no row slot)
//
// Numerics: tap chains and the box sum are the reference's exact fp32 sequence (same device code as the bit-exact pooling
// kernels); the quotient is the reference's correctly rounded v / area (vfa_op.py:118-119), formed as RN(1 / area) and two Markstein
// corrections (box_quotient_scaled, vfa_geom.h): VFA_FLAG_DUMP_VOX stores the rows this code forms and tests/test_fused_frame.py
// finds them bit for bit the reference's voxel features; the product, default (TERMS 2): both operands scaled by a power of two and split into two
// fp16 pieces, hi.lo + hi.hi + lo.hi -- the width of the reference's fp32 nn.Linear (vfa_split.h) --; TERMS 3 / 4: two bf16
// pieces (16-bit operands).  Within the path's post-GEMM tolerance (rtol 1e-4, atol 1e-5 max|ref|), not bitwise -- no GEMM order is.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "vfa_geom.h"
#include "vfa_split.h"

#ifndef VFA_TICKET_ORDER
// The hand-off ticket of a tile cut between workgroups (flush): a RELAXED agent-scope add by one lane behind a workgroup barrier.
// Every handed-off byte is stored `sc1` behind the storing wave's vmcnt(0) and read `sc1` behind the return of the add: the guide's
// measured-valid form (MI355X_MICROARCH.md, inter-workgroup visibility, first row).  Until round 6 the add was acquire-release: a
// `buffer_wbl2 sc1` (the XCD's L2 written back) + `buffer_inv sc1` per hand-off that the sc1 discipline makes redundant
// (vfa_pipe.hip has the measurement).  -DVFA_TICKET_ORDER=__ATOMIC_ACQ_REL restores it.
#define VFA_TICKET_ORDER __ATOMIC_RELAXED
#endif

namespace {
using namespace vfa_dev;

constexpr int kTileW = 8, kTileL = 4, kTileBoxes = kTileW * kTileL; // 32 cells = one 32-row MFMA block
constexpr int kC = 256;                                             // channels in = channels out
constexpr int kRecBytes = 96, kHdrBytes = 32;
constexpr int kMaxScales = 3;
constexpr int kSlotBytes = kC * 4;                                  // one tap = 256 fp32
constexpr int kMaxSlots = 123;                                      // LDS tap window of a (tile, view, scale)
// cost estimate of an item of the persistent kernel in units of 16 cycles (tile_chunks_kernel): base + 1 per window slot
constexpr unsigned kItemCost = 704, kRowItemCost = 522; // (unit: 16 cycles -- refitted in the second session of round 5, see tile_chunks_kernel)
constexpr int kRecSlots = 3;                                        // + the 32 box records of the item (3 KiB) behind it
constexpr int kThreads = 512;
constexpr int kRowBytes = 2 * kC;                                   // one bf16 plane row; 16-byte chunks XOR-swizzled with (row & 15)
constexpr int kPlane = kTileBoxes * kRowBytes;                      // 16 KiB
constexpr int kSteps = kC / 16;                                     // k-steps of v_mfma_f32_32x32x16_bf16

// record flags
constexpr int kVis = 1, kCont = 1 << 8; // bits 1-2 DXC, bits 3-4 DYC; kCont: same tap set as the previous box of the 4-box chunk
// tile header flags
constexpr int kTileLive = 1, kTileDirect = 2, kTileRows = 4; // kTileRows: a direct item whose pooled rows the pre-pass leaves in the workspace (header word 2 = its slot there)
constexpr int kTileShiftAt = 8; // bits 8-15: the item's sliver shift (vfa_geom.h): binary places its voxel features are scaled down by in the fp16 split

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct ScaleDims { int Hf, Wf; };

struct RecordArgs {
    BoxGeom g;
    int n_views, L, W, tiles_w, n_tiles, n_scales;
    ScaleDims dims[kMaxScales];
    unsigned *live[kMaxScales];      // (n_tiles) bit v = view v has a visible box in the tile
    unsigned *direct[kMaxScales];    // (n_tiles) bit v = ... and the tile's tap window does not fit LDS (subset of live)
    unsigned *overflow[kMaxScales];  // (n_tiles) bit v = ... and there was no row slot left for it (subset of direct)
    unsigned *row_counter;           // direct items numbered so far (all scales)
    unsigned rows_cap;               // row slots in the workspace
    unsigned *row_list;              // (rows_cap) slot -> scale << 30 | view << 25 | tile
    unsigned char *hdrs[kMaxScales]; // (n_views, n_tiles, 32 B)
    unsigned char *recs[kMaxScales]; // (n_views, n_tiles, 32 boxes, 96 B)
    unsigned short *item_w;          // (n_tiles, kMaxScales, views_pad): cost estimate of every item, 0 = none (tile_chunks_kernel)
    int views_pad;                   // n_views rounded up to 8
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// ------------------------------------------------------------------------------------------------
// 1. geometry of the frame: one half-wave (32 lanes) per (view, tile), lane = cell of the tile (4 rows of 8)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void frame_records_kernel(RecordArgs a)
{
    // the 32 records of a (view, tile) are 3 KiB of consecutive memory: staged here so that the half-wave writes them as six
    // 512-byte rows instead of 32 x 6 scattered 16-byte pieces (the kernel was bound by those stores)
    __shared__ uint4 stage[2][kTileBoxes * 6];
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
        // binary places the fp16 split of the item gives up for its noisiest visible box (vfa_geom.h: sliver_shift; 0 for honest boxes)
        int shift = vis ? sliver_shift(area, Hf, Wf) : 0;
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) {
            x0 = min(x0, __shfl_xor(x0, m, 32)); x1 = max(x1, __shfl_xor(x1, m, 32));
            t0 = min(t0, __shfl_xor(t0, m, 32)); t1 = max(t1, __shfl_xor(t1, m, 32));
            b0 = min(b0, __shfl_xor(b0, m, 32)); b1 = max(b1, __shfl_xor(b1, m, 32));
            shift = max(shift, __shfl_xor(shift, m, 32));
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
            uint4 *st = stage[half] + b * 6;
            st[0] = make_uint4(__float_as_uint(w[0]), __float_as_uint(w[1]), __float_as_uint(w[2]), __float_as_uint(w[3]));
            st[1] = make_uint4(__float_as_uint(w[4]), __float_as_uint(w[5]), __float_as_uint(w[6]), __float_as_uint(w[7]));
            st[2] = make_uint4(__float_as_uint(w[8]), __float_as_uint(w[9]), __float_as_uint(w[10]), __float_as_uint(w[11]));
            st[3] = make_uint4(__float_as_uint(w[12]), __float_as_uint(w[13]), __float_as_uint(w[14]), __float_as_uint(w[15]));
            const float rcp = 1.0f / area; // correctly rounded
            st[4] = make_uint4(__float_as_uint(rcp), (unsigned)tag | (cont ? (unsigned)kCont : 0u), rows[0] | (rows[1] << 16),
                               rows[2] | (rows[3] << 16));
            st[5] = make_uint4(cols[0] | (cols[1] << 16), cols[2] | (cols[3] << 16), __float_as_uint(masked), __float_as_uint(area));
        }
        __syncthreads(); // (one wave: orders the LDS writes above against the reads below)
        if (pair_ok) {
            uint4 *rec = reinterpret_cast<uint4 *>(a.recs[s] + ((size_t)view * a.n_tiles + tile) * kTileBoxes * kRecBytes);
#pragma unroll
            for (int k = 0; k < 6; ++k) rec[k * 32 + b] = stage[half][k * 32 + b];
            if (b == 0) {
                uint4 *hdr = reinterpret_cast<uint4 *>(a.hdrs[s] + ((size_t)view * a.n_tiles + tile) * kHdrBytes);
                const int inv = cwid > 0 ? (65536 + cwid - 1) / cwid : 0; // floor(s / cwid) == (s * inv) >> 16 for s < 128
                unsigned hflags = (any_live ? kTileLive : 0) | (direct ? kTileDirect : 0) | ((unsigned)shift << kTileShiftAt), word2 = (unsigned)cwid;
                if (any_live) atomicOr(a.live[s] + tile, 1u << view);
                if (any_live && direct) {
                    atomicOr(a.direct[s] + tile, 1u << view);
                    const unsigned slot = atomicAdd(a.row_counter, 1u); // (the order is arbitrary: the slot only names scratch space)
                    if (slot < a.rows_cap) { // (a direct item has no use for the window width)
                        hflags |= kTileRows; word2 = slot;
                        a.row_list[slot] = ((unsigned)s << 30) | ((unsigned)view << 25) | (unsigned)tile;
                    }
                    else atomicOr(a.overflow[s] + tile, 1u << view);
                }
                const bool main_item = any_live && !(direct && !(hflags & kTileRows));
                a.item_w[((size_t)tile * kMaxScales + s) * a.views_pad + view] =
                    (unsigned short)(!main_item ? 0u : (direct ? kRowItemCost : kItemCost + (unsigned)n_slots));
                hdr[0] = make_uint4(hflags, (unsigned)n_slots, word2, (unsigned)inv);
                hdr[1] = make_uint4((unsigned)x0, (unsigned)t0, (unsigned)top_rows, (unsigned)b0);
            }
        }
        __syncthreads(); // the stage is reused by the next scale
        if (blockIdx.x == 0 && lane < 6) // the spare record behind the table (prefetched by the consumers, never used)
            reinterpret_cast<uint4 *>(a.recs[s] + (size_t)a.n_views * a.n_tiles * kTileBoxes * kRecBytes)[lane] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// The sliver shifts of a frame as the BACKWARD of training needs them (vfa_collapse_gemm_relu_backward_f16_f32): per row of the
// recomputed product the binary places its item was scaled down by in the forward.  Same projection, same area, same sliver_shift as
// frame_records_kernel / pipe_records_kernel.  per_item: the serial kernel's unit (view, tile, scale) -> out[view][cell]; else the
// pipelined kernel's (tile, scale) over all views and layers -> tile_max[tile] (atomicMax), expanded by sliver_expand_kernel.
struct ShiftArgs {
    BoxGeom g;
    int n_views, L, W, tiles_w, n_tiles, nl, Hf, Wf, per_item;
    unsigned char *out;   // per_item: (n_views, L * W)
    unsigned *tile_max;   // else: (n_tiles), zeroed
};
__global__ __launch_bounds__(kWave) void sliver_shift_kernel(ShiftArgs a)
{
    const int lane = threadIdx.x, half = lane >> 5, b = lane & 31;
    const long long pair = (long long)blockIdx.x * 2 + half;
    const bool pair_ok = pair < (long long)a.n_views * a.n_tiles;
    const int view = pair_ok ? (int)(pair / a.n_tiles) : 0, tile = pair_ok ? (int)(pair % a.n_tiles) : 0;
    const int tl = tile / a.tiles_w, tw = tile - tl * a.tiles_w;
    const int cl = tl * kTileL + (b >> 3), cw = tw * kTileW + (b & 7);
    const bool valid = pair_ok && cl < a.L && cw < a.W;
    const int cell = valid ? cl * a.W + cw : 0;
    const float *P = a.g.calibs + (size_t)view * 12;
    int shift = 0;
    for (int layer = 0; layer < a.nl; ++layer) {
        const float gx = a.g.grid[cell * 3 + 0] + 0.0f, gy = a.g.grid[cell * 3 + 1] + 0.0f, gz = a.g.grid[cell * 3 + 2] + a.g.z_layers[layer];
        float l = 0.0f, t = 0.0f, r = 0.0f, bt = 0.0f;
#pragma unroll 1
        for (int k = 0; k < 8; ++k) {
            float nu, nv;
            project_corner(a.g, P, gx, gy, gz, k, nu, nv);
            if (k == 0) { l = r = nu; t = bt = nv; }
            else { l = min_t(l, nu); r = max_t(r, nu); t = min_t(t, nv); bt = max_t(bt, nv); }
        }
        const float area = box_area(l, t, r, bt, a.Hf, a.Wf);
        if (valid && box_visible(area, a.Hf, a.Wf)) shift = max(shift, sliver_shift(area, a.Hf, a.Wf));
    }
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) shift = max(shift, __shfl_xor(shift, m, 32));
    if (a.per_item) {
        if (valid) a.out[(size_t)view * a.L * a.W + cell] = (unsigned char)shift;
    } else if (pair_ok && b == 0 && shift > 0) {
        atomicMax(a.tile_max + tile, (unsigned)shift);
    }
}
__global__ __launch_bounds__(256) void sliver_expand_kernel(const unsigned *tile_max, unsigned char *out, int L, int W, int tiles_w)
{
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= L * W) return;
    const int cl = cell / W, cw = cell - cl * W;
    out[cell] = (unsigned char)tile_max[(cl / kTileL) * tiles_w + cw / kTileW];
}

// collapse.weight (N = 256, K = 256) fp32 -> bf16 hi / lo planes in MFMA B-fragment order:
//   out[((wave * 16 + s) * 2 + plane) * 64 + lane] (16 B) = W[n = 32 wave + (lane & 31)][k = 16 s + 8 (lane >> 5) + j], j = 0..7
// F16 (VFA_FLAG_TERMS 2, the default): the two-piece fp16 split of vfa_split.h, scaled by 2^ew with max|W| 2^ew in [2^14, 2^15)
// -- every splitting block finds the maximum for itself --, the exponent is left in wexp[scale] for the frame kernel.
constexpr int kWmaxParts = 32; // (a reserved region of the workspace: the partial maxima of rounds 3-4)
struct SplitArgs { const float *w[kMaxScales]; uint4 *out[kMaxScales]; unsigned *wmax; int *wexp; int f16; };
// The split runs in the spare blocks of the work-cuts launch (round 5): tile_chunks_kernel is ONE workgroup walking dependent memory
// round trips for ~30 us; as two launches of their own behind it (rounds 3-4: partial maxima, then the split: 20 us of launches for
// 0.8 MB of weights) the weights were the tail of the geometry stream, and with the integral images at 87 us the geometry had become
// the longer of the two chains in front of the frame kernel.
// Block b = 1 + scale * kSplitBlocks + j (1024 threads): the maximum of |W[scale]| over all 65 536 weights (every block for itself:
// 64 loads per thread out of L2), then fragments [1024 j, 1024 j + 1024) of the scale.
constexpr int kSplitBlocks = 8 * kSteps * 64 / 1024;
__device__ __forceinline__ void split_block(const SplitArgs &sa, int b, unsigned long long *part)
{
    const int scale = b / kSplitBlocks, j = b - scale * kSplitBlocks, tid = threadIdx.x;
    const float *__restrict__ w = sa.w[scale];
    uint4 *__restrict__ out = sa.out[scale];
    int ew = 0;
    if (sa.f16) {
        unsigned m = 0u;
        for (int i = tid; i < kC * kC; i += 1024) m = max(m, __float_as_uint(w[i]) & 0x7fffffffu);
        m = wave_max_u32(m);
        if ((tid & 63) == 0) part[tid >> 6] = m;
        __syncthreads();
        unsigned mm = 0u;
        for (int i = 0; i < 16; ++i) mm = max(mm, (unsigned)part[i]);
        ew = split_exponent(mm, kExpW);
    }
    const int idx = j * 1024 + tid; // (wave, s, lane)
    const int lane = idx & 63, st = (idx >> 6) & 15, wave = idx >> 10;
    const float *src = w + (size_t)(wave * 32 + (lane & 31)) * kC + 16 * st + 8 * (lane >> 5);
    if (idx == 0 && scale == 0) sa.wexp[kMaxScales] = sa.f16 ? 2 : 3; // the arithmetic these fragments are for: checked by the frame kernel
    uint4 uh, ul;
    if (sa.f16) {
        fp16_saturate_mode(true);
        if (idx == 0) sa.wexp[scale] = ew;
        const float sc = pow2f(ew);
        uint2 h0, l0, h1, l1;
        split_f16x4(src[0] * sc, src[1] * sc, src[2] * sc, src[3] * sc, h0, l0);
        split_f16x4(src[4] * sc, src[5] * sc, src[6] * sc, src[7] * sc, h1, l1);
        uh = make_uint4(h0.x, h0.y, h1.x, h1.y);
        ul = make_uint4(l0.x, l0.y, l1.x, l1.y);
    } else {
        union { __bf16 b[8]; uint4 u; } hi, lo;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float x = src[k];
            hi.b[k] = (__bf16)x;
            lo.b[k] = (__bf16)(x - (float)hi.b[k]);
        }
        uh = hi.u; ul = lo.u;
    }
    out[((size_t)(wave * kSteps + st) * 2 + 0) * 64 + lane] = uh;
    out[((size_t)(wave * kSteps + st) * 2 + 1) * 64 + lane] = ul;
}

// Work balance of the persistent kernel: the item sequence (tile, scale, view) is cut into kChunks pieces of equal estimated COST;
// a workgroup takes a contiguous run of chunks.  Cost model, fitted to the per-workgroup cycle counts of the diagnostic build on
// the bench frame (tools/bench_fused.py; unit = 32 cycles): an item 330 + 1 per slot of its tap window (a row item -50), + 110
// for the first item of a (tile, scale) (W and bias are reloaded), + 380 per tile (the tile store and its masked-item term);
// a tile without items 16.  A cut may fall INSIDE a tile: chunk c starts at the chunk_rank[c]-th live item of tile
// chunk_start[c] (items in (scale, view) order; rank 0 = the tile's beginning) -- whole tiles only left the slowest workgroup 19 %
// above the mean on the bench frame (4.9 tiles of ~18 items per workgroup), equal item counts 7 %.  A tile cut this way is
// finished by the workgroup that holds its beginning, which gets the partial sums of the other one through the workspace (see
// `flush`).  One workgroup, an LDS scan over per-thread sums.
constexpr int kChunks = 8192; // (fine enough that a launch with any number of workgroups gets pieces within 3 % of each other)
constexpr int kMaxBlocks = 512; // workgroups of the persistent kernel
struct ChunkArgs {
    const unsigned *live[kMaxScales], *overflow[kMaxScales];
    const unsigned short *item_w; // (n_tiles, kMaxScales, views_pad): written by frame_records_kernel
    int n_scales, n_tiles, n_views, views_pad;
    int *chunk_start, *chunk_rank;
    SplitArgs split; // n_split_blocks > 0: the blocks behind block 0 split the collapse weights (nothing of the frame in them)
    int n_split_blocks;
};
template <int G> struct TileItemsT { unsigned m[kMaxScales]; uint4 q[kMaxScales][G]; }; // item masks and their cost estimates (8 views per uint4)
// G = 1: up to 8 views, the first two tiles of a thread cached in registers; G = 4: up to 32 views, tiles re-read for the second walk
template <int G>
__global__ __launch_bounds__(1024) void tile_chunks_kernel(ChunkArgs a)
{
    using TileItems = TileItemsT<G>;
    constexpr bool kCache = G == 1;
    __shared__ unsigned long long part[1024];
    if (blockIdx.x > 0) { // (the weight split rides in this launch: split_block)
        split_block(a.split, (int)blockIdx.x - 1, part);
        return;
    }
    const int tid = threadIdx.x, n_tiles = a.n_tiles;
    const unsigned view_mask = a.n_views >= 32 ? 0xffffffffu : ((1u << a.n_views) - 1u);
    const int per = (n_tiles + 1023) / 1024, t0 = min(n_tiles, tid * per), t1 = min(n_tiles, t0 + per);
    // (refit of round 5, second session, least squares over the 256 workgroups of the bench frame, diagnostic build: cycles ~ 11 255 per
    // item + 18.6 per slot - 2 894 for a row item + 4 699 per scale change + 5 271 per tile; in units of 16 cycles.  The round-4 constants
    // -- 330 + slots, 280, 110, 380 in units of 32 -- priced a tile at 12 160 cycles and left the heaviest workgroup 2.7 % above the mean.)
    constexpr unsigned kScale = 294, kTile = 330, kEmpty = 32;
    // This kernel runs beside the bandwidth-bound integral-image kernels, where every dependent memory round trip costs
    // microseconds: everything a tile needs is requested at once and -- for the first two tiles of a thread, i.e. all of
    // them up to 2048 tiles -- kept in registers for the second walk.
    // (Every load UNCONDITIONAL, from a clamped scale / view group / tile, and masked afterwards: written under `s < n_scales` and
    // `t0 < t1` the loads of a thread came out as a chain -- masks of scale 0, wait, costs of scale 0, masks of scale 1, ... for the
    // first tile, then the same for the second: twelve dependent round trips beside a bandwidth-bound kernel, most of this kernel's
    // 33 us -- round 6)
    struct RawTile { unsigned lv[kMaxScales], ov[kMaxScales]; uint4 q[kMaxScales][G]; };
    auto load_raw = [&](int t) {
        RawTile rt;
#pragma unroll
        for (int s = 0; s < kMaxScales; ++s) {
            const bool on = s < a.n_scales;
            const unsigned *lp = on ? a.live[s] : a.live[0], *op = on ? a.overflow[s] : a.overflow[0];
            rt.lv[s] = lp[t]; rt.ov[s] = op[t];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int gg = g * 8 < a.n_views ? g : 0;
                rt.q[s][g] = *reinterpret_cast<const uint4 *>(a.item_w + ((size_t)t * kMaxScales + (on ? s : 0)) * a.views_pad + 8 * gg);
            }
        }
        return rt;
    };
    auto finish_tile = [&](const RawTile &rt) {
        TileItems ti;
#pragma unroll
        for (int s = 0; s < kMaxScales; ++s) {
            const bool on = s < a.n_scales;
            ti.m[s] = on ? rt.lv[s] & ~rt.ov[s] & view_mask : 0u; // (the items of the main launch: live, minus the direct ones without a row slot)
#pragma unroll
            for (int g = 0; g < G; ++g) ti.q[s][g] = (!on || g * 8 >= a.n_views) ? make_uint4(0u, 0u, 0u, 0u) : rt.q[s][g];
        }
        return ti;
    };
    auto load_tile = [&](int t) { return finish_tile(load_raw(t)); };
    // walks the items of a tile in kernel order; `visit(k, w0, w1)`: item k covers the positions [w0, w1) of the tile's weight
    auto walk = [&](const TileItems &ti, auto &&visit) -> unsigned {
        unsigned w = 0;
        int k = 0;
#pragma unroll
        for (int s = 0; s < kMaxScales; ++s) {
            bool first = true;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                unsigned mg = (ti.m[s] >> (8 * g)) & 0xffu;
                const uint4 q = ti.q[s][g];
                const unsigned long long lo = (unsigned long long)q.x | ((unsigned long long)q.y << 32), hi = (unsigned long long)q.z | ((unsigned long long)q.w << 32);
                while (mg) {
                    const int v = __builtin_ctz(mg);
                    mg &= mg - 1u;
                    unsigned wi = (unsigned)(((v < 4 ? lo : hi) >> (16 * (v & 3))) & 0xffffull);
                    if (first) wi += kScale;
                    if (k == 0) wi += kTile;
                    first = false;
                    visit(k, w, w + wi);
                    w += wi;
                    ++k;
                }
            }
        }
        return k == 0 ? kEmpty : w;
    };
    TileItems c0 = {}, c1 = {};
    if (kCache && n_tiles > 0) { // (clamped: a thread without tiles reads the last one and never looks at it; both tiles' loads first)
        const RawTile r0 = load_raw(min(t0, n_tiles - 1)), r1 = load_raw(min(t0 + 1, n_tiles - 1));
        __builtin_amdgcn_sched_barrier(0);
        c0 = finish_tile(r0);
        c1 = finish_tile(r1);
    }
    auto tile_of = [&](int t) { return (kCache && t == t0) ? c0 : (kCache && t == t0 + 1) ? c1 : load_tile(t); };
    unsigned long long local = 0;
    for (int t = t0; t < t1; ++t) local += walk(tile_of(t), [](int, unsigned, unsigned) {});
    part[tid] = local;
    for (int c = tid; c <= kChunks; c += 1024) { a.chunk_start[c] = n_tiles; a.chunk_rank[c] = 0; } // (positions beyond the last tile)
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { // inclusive Hillis-Steele scan
        const unsigned long long v = tid >= d ? part[tid - d] : 0ull;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const unsigned long long total = part[1023];
    unsigned long long before = part[tid] - local; // weight of all tiles in front of tile t
    auto pos_of = [&](long long cc) { return (total * (unsigned long long)cc + kChunks - 1) / kChunks; };
    // chunk c starts at position p_c = ceil(total c / kChunks): at the item of tile t whose weight interval holds p_c - before(t)
    // (a position inside an item's interval rounds to the nearer end; the end of the last item = the next tile's beginning).
    // The cuts of a tile are found in ONE walk: they come in increasing position.  A chunk index falls into one
    // tile's interval at most (stores of different threads never meet).
    for (int t = t0; t < t1; ++t) {
        const unsigned long long tb = before;
        long long c = tb > 0 ? (long long)((tb - 1) * kChunks / total) : 0; // p_c >= before  <=>  c > (before - 1) K / total
        while (c < kChunks && pos_of(c) < tb) ++c;
        const TileItems ti = tile_of(t);
        int n_items = 0;
        walk(ti, [&](int kk, unsigned, unsigned) { n_items = kk + 1; });
        const unsigned w = walk(ti, [&](int kk, unsigned w0, unsigned w1) {
            while (c < kChunks) {
                const unsigned long long pc = pos_of(c);
                if (pc >= tb + w1) break;
                // (pc >= tb + w0: the cuts in front of this item were taken by its predecessors)
                const int k = ((unsigned)(pc - tb) - w0) * 2 < (w1 - w0) ? kk : kk + 1;
                if (k >= n_items) { a.chunk_start[c] = t + 1 < n_tiles ? t + 1 : n_tiles; a.chunk_rank[c] = 0; } // = the next tile's beginning
                else { a.chunk_start[c] = t; a.chunk_rank[c] = k; }
                ++c;
            }
        });
        if (n_items == 0)
            for (; c < kChunks && pos_of(c) < tb + w; ++c) { a.chunk_start[c] = t; a.chunk_rank[c] = 0; }
        before += w;
    }
}

// ------------------------------------------------------------------------------------------------
// 2. pooling + collapse + bias + ReLU + view / scale sum
// ------------------------------------------------------------------------------------------------
struct FusedScale {
    const float *integral;          // (n_views, Hf+2, Wf+2, 256) zero-bordered channels-last
    const float *bias;              // (256) or NULL
    const uint4 *wfrag;             // the split collapse weights (split_block: the spare blocks of the work-cuts launch)
    const unsigned *live, *direct, *overflow; // (n_tiles) each
    const unsigned char *hdrs, *recs;
    int Hf, Wf;
    const unsigned *amax;           // amax_n partial maxima of |feature| (fp32 bits): the scale of the fp16 split (vfa_split.h)
    int amax_n;
};
struct FusedArgs {
    FusedScale sc[kMaxScales];
    int n_scales, n_views, L, W, tiles_w, n_tiles;
    float *out;                     // (L * W, 256)
    const int *chunk_start, *chunk_rank; // (kChunks + 1) each: tile_chunks_kernel
    float *partial;                 // (kMaxBlocks, 2) x 8 waves x 16 registers x 64 lanes: a workgroup's part of a tile it shares (first / last tile)
    unsigned *tickets;              // (n_tiles): parts of a shared tile that have arrived (zero between launches: see layout_of)
    const float *rows;              // pooled rows of the direct items (pool_rows_kernel): slot x 32 boxes x 256 channels
    const unsigned *row_counter;    // direct items of the frame
    int rows_cap;                   // row slots in the workspace
    int accumulate;
    const int *wexp;                // (kMaxScales) scale exponent of the split collapse weight (split_block; fp16 form); [kMaxScales]: 2 = fp16 fragments, 3 = bf16
    int debug;                      // diagnostic build only: ablation mask (kDbg*), results are then meaningless
    unsigned long long *diag;       // diagnostic build only: per workgroup 8 cycle counters
};
// diagnostic ablations (VFA_FLAG_DEBUG(mask), pool_collapse_kernel<TERMS, true> only)
constexpr int kDbgNoFills = 1, kDbgNoPool = 2, kDbgNoMfma = 4, kDbgOneW = 8, kDbgNoRecords = 16, kDbgNoExtra = 32, kDbgStamps = 128; // (64: only the direct-item launch)
// VFA_FLAG_DUMP_VOX: with ONE view and ONE scale, `out` receives the pooled fp32 voxel features (cell, channel) exactly as the
// pooling code of THIS kernel forms them in front of the operand split -- RN((((lt + rb) - rt) - lb) / area), masked boxes their
// masked value -- instead of the map: what tests/test_fused_frame.py compares with the reference's voxel features.
constexpr int kDbgDumpVox = 0x1000; // (set by VFA_FLAG_DUMP_VOX; bits 0-11 are VFA_FLAG_DEBUG's)

struct Frag { bf16x8 hi, lo; };

__device__ __forceinline__ float relu_t(float x) { return (x < 0.0f) ? 0.0f : x; } // NaN stays NaN

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
// Pooling layout: a wave owns four boxes of the tile and pools them SIDE BY SIDE -- 16 lanes x float4 = 64 channels of one
// box, four boxes per wave instruction, four channel quarters per item.  Box parameters are per-lane registers (the box
// record, read from the LDS slots behind the tap window, where LDS-DMA put it), control flow is the same for every box (always 16 taps: the
// duplicates of narrow boxes hit the same LDS words), so the quarter loop is one straight-line body that the compiler
// software-pipelines.  A 16-lane group reads 256 contiguous bytes of a tap slot: conflict-free for any mix of slots.
struct LRec { uint4 v[6]; }; // one box record per lane: v[0..3] the 16 tap weights, v[4] rcp, flags, rows; v[5] cols, masked, area

// x = hi + lo + r exactly in fp32 arithmetic: hi = RNE bf16(x), lo = RNE bf16(x - hi); row `row` of the A tile, channels
// 4 c4 .. 4 c4 + 3, written into the XOR-swizzled hi / lo planes
// F16: the two-piece fp16 split of vfa_split.h (v already carries the scale 2^ea)
template <bool F16>
__device__ __forceinline__ void store_quad_at(unsigned char *planes, int off, float4 v)
{
    uint2 hu, lu;
    if constexpr (F16) {
        split_f16x4(v.x, v.y, v.z, v.w, hu, lu);
    } else {
        const float x[4] = {v.x, v.y, v.z, v.w};
        union { __bf16 b[4]; uint2 u; } hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            hi.b[j] = (__bf16)x[j];
            lo.b[j] = (__bf16)(x[j] - (float)hi.b[j]);
        }
        hu = hi.u; lu = lo.u;
    }
    *reinterpret_cast<uint2 *>(planes + off) = hu;
    *reinterpret_cast<uint2 *>(planes + kPlane + off) = lu;
}
template <bool F16>
__device__ __forceinline__ void store_quad(unsigned char *planes, int row, int c4, float4 v)
{
    store_quad_at<F16>(planes, row * kRowBytes + ((((c4 >> 1) ^ (row & 15)) << 4) | ((c4 & 1) << 3)), v);
}

struct Item { int tile, scale, view; unsigned rest; bool valid; int rank; }; // rest: live views of (tile, scale) above `view`; rank: index among the tile's live items

// DIRECT = false: every item whose tap window fits LDS (all but the tiles right in front of a camera); DIRECT = true: a second
// launch for exactly the others -- taps read straight from the integral image, contribution ADDED to the map.
template <int TERMS, bool DIAG, bool DIRECT>
__global__ __launch_bounds__(kThreads) void pool_collapse_kernel(FusedArgs a)
{
    __shared__ float4 s_taps[(kMaxSlots + kRecSlots) * 64];         // 123 + 3 KiB: the tap window of the current item + its box records
    __shared__ __align__(16) unsigned char s_planes[2 * kPlane];    // 32 KiB: bf16 hi / lo planes of the 32 x 256 A tile
    __shared__ uint4 s_hdr[2][16];                                  // tile headers of the next two items (32 B used of each 256)
    const int tid = threadIdx.x, wave = uniform_i(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;
    constexpr bool F16 = TERMS == 2; // two fp16 pieces per operand (vfa_split.h) instead of two bf16 pieces

    // Main launch: a contiguous range of tiles of equal COST for this workgroup (tile_chunks_kernel); neighbouring ranges
    // share an XCD (their tap windows overlap).  Direct-item launch: tiles dealt round-robin -- those items sit in clusters
    // in front of the cameras.
    if (DIRECT && uniform_i((int)*a.row_counter) <= a.rows_cap) return; // every direct item of this frame got a row slot
    const int nblk = gridDim.x;
    // The weight fragments in the workspace were split for ONE arithmetic (the geometry call's flags); a launch that asks for the other
    // would read fp16 pieces as bf16 ones: fail loudly -- a map of NaNs -- instead of returning plausible garbage.
    if (uniform_i(a.wexp[kMaxScales]) != (F16 ? 2 : 3)) {
        if (!DIRECT)
            for (size_t i = (size_t)blockIdx.x * kThreads + tid; i < (size_t)a.L * a.W * kC; i += (size_t)nblk * kThreads)
                a.out[i] = __uint_as_float(0x7fc00000u);
        return;
    }
    const int lb = (int)xcd_contiguous(blockIdx.x, (nblk + 7) / 8);
    if (lb >= nblk) return;
    const int t_step = DIRECT ? nblk : 1;
    // main launch: the items from the k_begin-th live item of tile t_begin up to (not including) the k_end-th of tile t_end
    auto range_of = [&](int wg, int &tb, int &kb, int &te, int &ke) {
        const int c0 = (int)((long long)kChunks * wg / nblk), c1 = (int)((long long)kChunks * (wg + 1) / nblk);
        tb = uniform_i(a.chunk_start[c0]); kb = uniform_i(a.chunk_rank[c0]);
        te = uniform_i(a.chunk_start[c1]); ke = uniform_i(a.chunk_rank[c1]);
    };
    int t_begin = lb, k_begin = 0, t_end = a.n_tiles, k_end = 0;
    if (!DIRECT) range_of(lb, t_begin, k_begin, t_end, k_end);
    if (t_begin > t_end || (t_begin == t_end && k_begin >= k_end)) return;
    const int t_lim = (!DIRECT && k_end > 0) ? t_end + 1 : t_end; // tiles this workgroup looks at
    // A tile cut between workgroups is finished by whichever of them arrives LAST (a ticket per tile): nobody ever waits for
    // another workgroup, so the launch makes progress whatever subset of its workgroups is resident (a spin on the neighbour's
    // flag needed all of them on the chip at once).  The partners of the two tiles this workgroup may share, found once:
    __shared__ unsigned s_ticket;
    auto share_of = [&](int tile, int &first, int &last, int &parts) {
        first = lb; last = lb; parts = 1;
        for (int j = lb - 1; j >= 0; --j) {
            int tb, kb, te, ke;
            range_of(j, tb, kb, te, ke);
            if (te < tile || (te == tile && ke == 0)) break;
            if (tb > te || (tb == te && kb >= ke)) continue; // (empty range)
            first = j; ++parts;
        }
        for (int j = lb + 1; j < nblk; ++j) {
            int tb, kb, te, ke;
            range_of(j, tb, kb, te, ke);
            if (tb > tile) break;
            if (tb > te || (tb == te && kb >= ke)) continue;
            last = j; ++parts;
        }
    };
    int sh_b_first = lb, sh_b_last = lb, sh_b_parts = 1, sh_e_first = lb, sh_e_last = lb, sh_e_parts = 1;
    if (!DIRECT && k_begin > 0) share_of(t_begin, sh_b_first, sh_b_last, sh_b_parts);
    if (!DIRECT && k_end > 0) share_of(t_end, sh_e_first, sh_e_last, sh_e_parts);
    const unsigned view_mask = a.n_views >= 32 ? 0xffffffffu : ((1u << a.n_views) - 1u);

    // fp16 form: per scale, the factor 2^ea of the voxel features (from the largest |feature| the integral-image kernels saw) and
    // 2^(ea + ew) / 2^-(ea + ew) for the bias in the accumulator / the epilogue (powers of two: exact)
    float f_a[kMaxScales] = {1.0f, 1.0f, 1.0f}, f_a_inv[kMaxScales] = {1.0f, 1.0f, 1.0f}, f_fwd[kMaxScales] = {1.0f, 1.0f, 1.0f}, f_inv[kMaxScales] = {1.0f, 1.0f, 1.0f};
    if constexpr (F16) {
        __shared__ unsigned s_amax[kMaxScales];
        if (tid < kMaxScales) s_amax[tid] = 0u;
        __syncthreads();
#pragma unroll
        for (int s2 = 0; s2 < kMaxScales; ++s2) {
            if (s2 < a.n_scales) {
                unsigned m = 0u;
                for (int i = tid; i < a.sc[s2].amax_n; i += kThreads) m = max(m, a.sc[s2].amax[i]);
                m = wave_max_u32(m);
                if (lane == 0) atomicMax(&s_amax[s2], m);
            }
        }
        __syncthreads();
#pragma unroll
        for (int s2 = 0; s2 < kMaxScales; ++s2) {
            if (s2 < a.n_scales) {
                const int ea = split_exponent((unsigned)uniform_i((int)s_amax[s2]), kExpA), ew = uniform_i(a.wexp[s2]);
                f_a[s2] = pow2f(ea); f_a_inv[s2] = pow2f(-ea); f_fwd[s2] = pow2f(ea + ew); f_inv[s2] = pow2f(-(ea + ew));
            }
        }
    }
    auto pick = [&](const float (&t)[kMaxScales], int scale) { return scale == 0 ? t[0] : (scale == 1 ? t[1] : t[2]); };

    // view masks come by SCALAR loads (constant address space: the records kernel finished before this launch): a vector load
    // here would queue behind the window DMA of the next item, which is in flight whenever the item sequence is advanced
    auto mask_at = [&](const unsigned *p, int tile) {
        return *reinterpret_cast<const __attribute__((address_space(4))) unsigned *>((size_t)(p + tile));
    };
    auto live_all = [&](int tile, int scale) { return mask_at(a.sc[scale].live, tile) & view_mask; };
    // the views of (tile, scale) this launch works on
    // the views of (tile, scale) this launch works on (the ranks of the chunk cuts count exactly these)
    auto live_of = [&](int tile, int scale) {
        const unsigned dm = mask_at(a.sc[scale].overflow, tile) & view_mask;
        return DIRECT ? dm : (mask_at(a.sc[scale].live, tile) & view_mask & ~dm);
    };
    // first item at or after (tile, scale) with view bits `rest`; `rank` = its index among the items of its tile
    auto seek = [&](int tile, int scale, unsigned rest, int rank) {
        Item it;
        it.valid = false; it.tile = tile; it.scale = scale; it.view = 0; it.rest = 0; it.rank = rank;
        while (tile < t_lim) {
            if (rest) {
                if (!DIRECT && tile == t_end && rank >= k_end) return it; // the next workgroup's part of the tile
                it.view = __builtin_ctz(rest);
                it.rest = rest & (rest - 1u);
                it.tile = tile; it.scale = scale; it.rank = rank; it.valid = true;
                return it;
            }
            if (++scale == a.n_scales) { scale = 0; tile += t_step; rank = 0; }
            if (tile < t_lim) rest = live_of(tile, scale);
        }
        return it;
    };

    auto bias_of = [&](int scale) { return a.sc[scale].bias ? a.sc[scale].bias[wave * 32 + r] : 0.0f; };
    // relu(bias) of this lane's column for every scale: what a fully masked (view, scale) item adds to a cell.  Kept in registers:
    // loading the three biases inside `write_tile` put three dependent memory round trips in front of every tile store.
    float rbias[kMaxScales];
#pragma unroll
    for (int s2 = 0; s2 < kMaxScales; ++s2) rbias[s2] = s2 < a.n_scales ? relu_t(bias_of(s2)) : 0.0f;

    // output rows of a tile: register i of lane (r, h) is row (i & 3) + 8 (i >> 2) + 4 h of the 32 x 32 MFMA block, column r
    auto write_tile = [&](int tile, const f32x16 &sum, bool have_sum) {
        if (DIAG && (a.debug & kDbgDumpVox)) return; // (the output buffer holds the dumped voxel features)
        const int tl = tile / a.tiles_w, tw = tile - tl * a.tiles_w;
        float extra = 0.0f; // fully masked (view, scale) items of this tile: vox = 0 -> relu(bias)
        if (!DIRECT && !(DIAG && (a.debug & kDbgNoExtra))) {
#pragma unroll
            for (int s = 0; s < kMaxScales; ++s)
                if (s < a.n_scales) extra += (float)(a.n_views - __popc(live_all(tile, s))) * rbias[s];
        }
        // (opaque copies: the per-lane row offsets below are loop invariants the compiler would otherwise keep in VGPRs for the
        // whole kernel, pushing other values into scratch)
        int h2 = h, r2 = r;
        asm volatile("" : "+v"(h2), "+v"(r2));
        float *ocol = a.out + wave * 32 + r2;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * h2;
            const int cl = tl * kTileL + (row >> 3), cw = tw * kTileW + (row & 7);
            if (cl < a.L && cw < a.W) {
                float *o = ocol + (size_t)(cl * a.W + cw) * kC;
                float v = (have_sum ? sum[i] : 0.0f) + extra;
                if (DIRECT || a.accumulate) v += *o; // the workgroup owns these rows: a plain read-modify-write
                *o = v;
            }
        }
    };

    // LDS-DMA of the item's tap window: slot s = (window row, window column), 1 KiB per wave instruction
    // Tile headers end up in SGPRs, but travel by LDS-DMA like the tap window, two items ahead: one wave requests the 32 bytes
    // (a 256-byte piece, the DMA granule) into one of two small LDS buffers; after the `vmcnt(0)` + barrier at the head of the
    // loop every wave reads them back (a broadcast read) into scalar registers.  In flight the header costs no register at all.
    // The two scalar forms both failed: an asm s_load is invisible to the register allocator, which spilled its destination
    // registers before the data landed and restored garbage as soon as scalar pressure rose; a compiler-issued scalar load is
    // waited for at the next lgkmcnt(0) -- scalar loads return out of order -- i.e. at once (+1 300 cycles per item).
    auto header_of = [&](const Item &it, int buf) {
        if (wave == 0) { // (an older wave: see begin_fills)
            const unsigned char *p = a.sc[it.scale].hdrs + ((size_t)it.view * a.n_tiles + it.tile) * kHdrBytes + lane * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                             (__attribute__((address_space(3))) void *)(&s_hdr[buf][0]), 4, 0, 0);
        }
    };
    auto header_get = [&](i32x8 &hd, int buf) {
        const uint4 h0 = s_hdr[buf][0], h1 = s_hdr[buf][1];
        hd[0] = uniform_i((int)h0.x); hd[1] = uniform_i((int)h0.y); hd[2] = uniform_i((int)h0.z); hd[3] = uniform_i((int)h0.w);
        hd[4] = uniform_i((int)h1.x); hd[5] = uniform_i((int)h1.y); hd[6] = uniform_i((int)h1.z); hd[7] = uniform_i((int)h1.w);
    };
    // The window is brought in by LDS-DMA, 1 KiB (one slot) per wave instruction, slot s = (window row, window column).  The
    // instructions are issued ONE AT A TIME between groups of MFMAs: eight waves issuing their whole share at once queue up
    // behind the texture addresser for ~1000 cycles and enter the MFMA phase skewed by as much.
    // The source offsets of a wave's fills (slots wave, wave + 4, ...: at most 32) are computed ONCE per item, lane j that of
    // fill j, with a handful of vector instructions; the MFMA loop then reads them back with v_readlane: 5 instructions per
    // fill where the closed form took ~20 dependent scalar ones, in a wave that issues in order, in front of its next MFMA.
    constexpr int kFillWaves = 4; // waves 0-3 bring the window in (see begin_fills; 2 or 8 waves measured 2.5 % slower in round 5)
    int f_slot = 0, f_n = 0;
    unsigned f_off = 0; // lane j: byte offset of this wave's j-th slot inside the view's padded integral image (< 4 GiB)
    const char *f_img = nullptr;
    auto begin_fills = [&](const Item &it, const i32x8 &hd) {
        // Only the OLDER wave of every SIMD (waves 0-3) issues DMA: the arbiter favours it, it leaves the MFMA phase ~1 000
        // cycles before its partner, and the scalar address arithmetic of this stage -- identical in every wave, on the CU's one
        // scalar unit -- is then done by four waves instead of eight.
        f_slot = wave;
        f_n = 0;
        if (wave >= kFillWaves) return;
        const FusedScale &sc = a.sc[it.scale];
        const int slot = wave + kFillWaves * (lane & 31);
        if (!DIRECT && (hd[0] & kTileRows)) {
            // a direct item with a row slot: its "window" is its 32 pooled rows (pool_rows_kernel), slot b = row of box b
            f_n = kTileBoxes;
            f_off = (unsigned)slot * kSlotBytes;
            f_img = reinterpret_cast<const char *>(a.rows) + (size_t)hd[2] * kTileBoxes * kSlotBytes;
            return;
        }
        f_n = (DIRECT || (hd[0] & kTileDirect)) ? 0 : hd[1];
        const int cw = hd[2], inv = hd[3], x0 = hd[4], t0 = hd[5], top = hd[6], b0 = hd[7], wp = sc.Wf + 2;
        const int wr = (slot * inv) >> 16, wc = slot - wr * cw;
        const int y = wr < top ? t0 + wr : b0 + (wr - top), x = x0 + wc;
        f_off = (unsigned)((y + 1) * wp + (x + 1)) * kSlotBytes;
        f_img = reinterpret_cast<const char *>(sc.integral) + (size_t)it.view * (sc.Hf + 2) * wp * kSlotBytes;
    };
    auto fill_at = [&](int j) { // (j: wave-uniform)
        const char *src = f_img + (size_t)(unsigned)__builtin_amdgcn_readlane((int)f_off, j) + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(s_taps + f_slot * 64), 16, 0, 0);
        f_slot += kFillWaves;
    };
    auto one_fill = [&](int j) { // the j-th fill of this wave (j a compile-time constant where the loop is unrolled)
        if (f_slot < f_n) fill_at(j);
    };
    auto rest_fills = [&]() { // every fill that is left
        while (f_slot < f_n) fill_at((f_slot - wave) / kFillWaves);
    };

    const int grp = lane >> 4, cq = lane & 15;
    LRec rec; // record of THIS lane's box (direct-item launch: loaded one item ahead; main launch: read from LDS when pooling)
    // Main launch: the 32 box records of an item (3 KiB, contiguous) travel by LDS-DMA like its window, into the three slots
    // behind it -- three wave instructions per item, nothing held in registers across the MFMA phase.  As per-lane global loads
    // into registers that live across the loop they made the compiler wait for them (`vmcnt(0)`) right where they were issued:
    // a full memory latency per item in front of the MFMA phase.
    auto load_record = [&](const Item &it, int it_flags, int it_slot) {
        if constexpr (DIRECT) {
            const uint4 *p = reinterpret_cast<const uint4 *>(a.sc[it.scale].recs + (((size_t)it.view * a.n_tiles + it.tile) * kTileBoxes + 4 * wave + grp) * kRecBytes);
#pragma unroll
            for (int k = 0; k < 6; ++k) rec.v[k] = p[k];
        } else if (!(it_flags & kTileRows) && wave >= 1 && wave <= kRecSlots) { // (older waves: see begin_fills)
            const int k = wave - 1;
            const unsigned char *p = a.sc[it.scale].recs + ((size_t)it.view * a.n_tiles + it.tile) * kTileBoxes * kRecBytes + k * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                             (__attribute__((address_space(3))) void *)(s_taps + (kMaxSlots + k) * 64), 16, 0, 0);
        }
    };
    // diagnostic (kDbgDumpVox): the voxel features of box `row` of `tile`, channels 4 c4 .. 4 c4 + 3, to the output buffer
    auto dump_vox = [&](int tile, int row, int c4, float4 v) {
        const int tl = tile / a.tiles_w, tw = tile - tl * a.tiles_w;
        const int cl = tl * kTileL + (row >> 3), cw = tw * kTileW + (row & 7);
        if (cl < a.L && cw < a.W) *reinterpret_cast<float4 *>(a.out + (size_t)(cl * a.W + cw) * kC + 4 * c4) = v;
    };
    auto pool = [&](const Item &it, int it_flags, int it_word1, float s_a, float s_a_inv) {
        if (!DIRECT && (it_flags & kTileRows)) {
            // a direct item of the main launch: its pooled rows arrived as its window (slot b = row of box b); only the
            // split remains
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = s_taps[(4 * wave + grp) * 64 + q * 16 + cq];
                if (DIAG && (a.debug & kDbgDumpVox)) dump_vox(it.tile, 4 * wave + grp, q * 16 + cq, v);
                if constexpr (F16) v = mul4(v, s_a); // (the pre-pass divided exactly: box_mean)
                store_quad<F16>(s_planes, 4 * wave + grp, q * 16 + cq, v);
            }
            return;
        }
        const FusedScale &sc = a.sc[it.scale];
        const unsigned Wp = (unsigned)sc.Wf + 2u;
        const char *img = reinterpret_cast<const char *>(sc.integral) + (size_t)it.view * (sc.Hf + 2) * Wp * kSlotBytes;
        if constexpr (!DIRECT) { // this lane's box record, out of the slots behind the window (a broadcast read per 16-lane group)
            const uint4 *rp = reinterpret_cast<const uint4 *>(s_taps + kMaxSlots * 64) + (4 * wave + grp) * (kRecBytes / 16);
#pragma unroll
            for (int k = 0; k < 6; ++k) rec.v[k] = rp[k];
        }
        const float wt[16] = {__uint_as_float(rec.v[0].x), __uint_as_float(rec.v[0].y), __uint_as_float(rec.v[0].z), __uint_as_float(rec.v[0].w),
                              __uint_as_float(rec.v[1].x), __uint_as_float(rec.v[1].y), __uint_as_float(rec.v[1].z), __uint_as_float(rec.v[1].w),
                              __uint_as_float(rec.v[2].x), __uint_as_float(rec.v[2].y), __uint_as_float(rec.v[2].z), __uint_as_float(rec.v[2].w),
                              __uint_as_float(rec.v[3].x), __uint_as_float(rec.v[3].y), __uint_as_float(rec.v[3].z), __uint_as_float(rec.v[3].w)};
        const float rcp = __uint_as_float(rec.v[4].x), masked = __uint_as_float(rec.v[5].z), area = __uint_as_float(rec.v[5].w);
        const bool vis = (rec.v[4].y & (unsigned)kVis) != 0u;
        // tap positions in float4 units (LDS: slot inside the window; DIRECT: pixel inside the view's padded image): row part +
        // column part.  The coordinates of a masked box are meaningless: point them at slot / pixel 0 (the value is discarded).
        unsigned rw[4] = {rec.v[4].z & 0xffffu, rec.v[4].z >> 16, rec.v[4].w & 0xffffu, rec.v[4].w >> 16};
        unsigned cl[4] = {rec.v[5].x & 0xffffu, rec.v[5].x >> 16, rec.v[5].y & 0xffffu, rec.v[5].y >> 16};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            rw[k] = (vis ? (DIRECT ? rw[k] * Wp : rw[k]) : 0u) * 64u + (unsigned)cq;
            cl[k] = (vis ? cl[k] : 0u) * 64u;
        }
        const int row = 4 * wave + grp;
        // a 64-channel quarter at a time, its taps requested eight at a time (two samples; the W fragments hold half of the
        // register file): the partner wave of the SIMD covers the two LDS latencies per quarter
        auto pair = [&](int q, int ra, int ca, int rb2, int cb, const float *wa, const float *wb, float4 &sa, float4 &sb) {
            float4 t[2][4]; // sample a = rows (ra, ra + 1) x cols (ca, ca + 1), sample b likewise: nw, ne, sw, se each
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned oa = rw[ra + (k >> 1)] + cl[ca + (k & 1)] + (unsigned)(q * 16);
                const unsigned ob = rw[rb2 + (k >> 1)] + cl[cb + (k & 1)] + (unsigned)(q * 16);
                if constexpr (DIRECT) {
                    t[0][k] = *reinterpret_cast<const float4 *>(img + (size_t)oa * 16);
                    t[1][k] = *reinterpret_cast<const float4 *>(img + (size_t)ob * 16);
                } else {
                    t[0][k] = s_taps[oa];
                    t[1][k] = s_taps[ob];
                }
            }
            sa = sample4(t[0][0], t[0][1], t[0][2], t[0][3], wa[0], wa[1], wa[2], wa[3]);
            sb = sample4(t[1][0], t[1][1], t[1][2], t[1][3], wb[0], wb[1], wb[2], wb[3]);
        };
        // a masked box reads slot / pixel 0 (finite) and multiplies by its masked value (0, or NaN for a NaN box) instead of
        // selecting afterwards; the A-plane address of quarter q is that of quarter 0 with one bit pair flipped
        // (((8 q + a) ^ r) << 4 = ((a ^ r) << 4) ^ (q << 7) for a < 8): both keep VALU work out of the quarter passes
        // The quotient is the reference's correctly rounded division (vfa_op.py:118-119) times the item's power of two 2^(ea - shift)
        // of the fp16 split (1 in the bf16 forms): box_quotient_scaled with rs = RN(1 / area) 2^k, as = area 2^-k (vfa_geom.h).
        const float rs = (vis ? rcp : masked) * s_a, as = area * s_a_inv;
        const int plane0 = row * kRowBytes + (((((cq >> 1) ^ (row & 15)) << 4)) | ((cq & 1) << 3));
        auto finish = [&](int q, float4 lt, float4 rb, float4 rt, float4 lb) {
            // RN((((lt + rb) - rt) - lb) / area)                                                  (A.6)
            float4 v = make_float4(lt.x + rb.x, lt.y + rb.y, lt.z + rb.z, lt.w + rb.w);
            v = make_float4(v.x - rt.x, v.y - rt.y, v.z - rt.z, v.w - rt.w);
            v = make_float4(v.x - lb.x, v.y - lb.y, v.z - lb.z, v.w - lb.w);
            v = make_float4(box_quotient_scaled(v.x, as, rs), box_quotient_scaled(v.y, as, rs), box_quotient_scaled(v.z, as, rs),
                            box_quotient_scaled(v.w, as, rs));
            if (DIAG && (a.debug & kDbgDumpVox)) // (without the power-of-two factor of the fp16 split: the same bits / 2^k)
                dump_vox(it.tile, row, q * 16 + cq, make_float4(v.x * s_a_inv, v.y * s_a_inv, v.z * s_a_inv, v.w * s_a_inv));
            store_quad_at<F16>(s_planes, plane0 ^ (q << 7), v);
        };
        if constexpr (DIRECT) {
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
                float4 lt, rb, rt, lb;
                pair(q, 0, 0, 2, 2, wt + 0, wt + 4, lt, rb);
                __builtin_amdgcn_sched_barrier(0);
                pair(q, 0, 2, 2, 0, wt + 8, wt + 12, rt, lb);
                finish(q, lt, rb, rt, lb);
            }
        } else {
            // the 16 tap addresses of the box once per item; the quarter is an immediate offset of the LDS read (the loop is
            // unrolled): no address arithmetic left in the quarter passes, which are bound by VALU issue (365 VALU instructions
            // per item and wave instead of ~600: pooling 4 600 -> 3 850 cycles per item)
            unsigned tb[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) tb[i][j] = rw[i] + cl[j];
            // (Tried in round 5, slower: a ROLLING form -- four taps at a time into two register quads, the reads of sample k + 1 in front of
            // the FMA chain of sample k, the next quarter's first sample in front of this quarter's quotient / split / stores -- 480 us
            // against 445 per launch in an A/B on one device: 256 registers, eight of them spilled inside the loop, and the scheduling
            // barriers it needs keep the compiler from interleaving reads and arithmetic on its own.)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 lt, rb, rt, lb;
                {
                    const float4 t0 = s_taps[tb[0][0] + q * 16], t1 = s_taps[tb[0][1] + q * 16], t2 = s_taps[tb[1][0] + q * 16], t3 = s_taps[tb[1][1] + q * 16];
                    const float4 u0 = s_taps[tb[2][2] + q * 16], u1 = s_taps[tb[2][3] + q * 16], u2 = s_taps[tb[3][2] + q * 16], u3 = s_taps[tb[3][3] + q * 16];
                    lt = sample4(t0, t1, t2, t3, wt[0], wt[1], wt[2], wt[3]);
                    rb = sample4(u0, u1, u2, u3, wt[4], wt[5], wt[6], wt[7]);
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    const float4 t0 = s_taps[tb[0][2] + q * 16], t1 = s_taps[tb[0][3] + q * 16], t2 = s_taps[tb[1][2] + q * 16], t3 = s_taps[tb[1][3] + q * 16];
                    const float4 u0 = s_taps[tb[2][0] + q * 16], u1 = s_taps[tb[2][1] + q * 16], u2 = s_taps[tb[3][0] + q * 16], u3 = s_taps[tb[3][1] + q * 16];
                    rt = sample4(t0, t1, t2, t3, wt[8], wt[9], wt[10], wt[11]);
                    lb = sample4(u0, u1, u2, u3, wt[12], wt[13], wt[14], wt[15]);
                }
                __builtin_amdgcn_sched_barrier(0);
                finish(q, lt, rb, rt, lb);
            }
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
    Item cur;
    if (DIRECT || k_begin == 0) {
        cur = seek(t_begin, 0, live_of(t_begin, 0), 0);
    } else { // the first k_begin live items of the tile are the previous workgroup's
        int sc0 = 0, left = k_begin;
        unsigned rest0 = live_of(t_begin, 0);
        while (left > 0 && (rest0 || sc0 + 1 < a.n_scales)) {
            if (rest0) { rest0 &= rest0 - 1u; --left; }
            else rest0 = live_of(t_begin, ++sc0);
        }
        cur = seek(t_begin, sc0, rest0, k_begin);
    }
    {
        f32x16 none = {};
        if (!DIRECT)
            for (int t2 = t_begin; t2 < (cur.valid ? cur.tile : t_end); ++t2) write_tile(t2, none, false);
    }
    if (!cur.valid) return;
    const int dbg = DIAG ? a.debug : 0;
    unsigned long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
    auto tick = [&](int k) { // diagnostic build: cycles of this wave between consecutive marks
        if (DIAG && (dbg & kDbgStamps)) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            stamp[k] += now - t_prev;
            t_prev = now;
        }
    };
    i32x8 nh;
    int hbuf = 0; // LDS buffer holding the header of `nxt`
    header_of(cur, 1);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    header_get(nh, 1);
    int cur_flags = nh[0], cur_word1 = nh[2], nxt_flags = 0, nxt_word1 = 0; // header words 0, 2 of the item being pooled / the next
    load_record(cur, cur_flags, cur_word1);
    if (!(dbg & kDbgNoFills)) {
        begin_fills(cur, nh);
        rest_fills();
    }
    Item nxt = seek(cur.tile, cur.scale, cur.rest, cur.rank + 1);
    if (nxt.valid) header_of(nxt, hbuf);
    int w_scale = -1;
    float bc = 0.0f, sa_cur = 1.0f, sa_inv_cur = 1.0f, inv_cur = 1.0f;
    f32x16 sum;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum[i] = 0.0f;
    // a finished tile keeps its sums in registers until the NEXT barrier: its stores then drain under a whole item
    int pend_tile = -1, pend_next = 0;
    auto flush = [&]() {
        if (pend_tile < 0) return;
        if (!DIRECT) {
            const bool at_begin = pend_tile == t_begin && k_begin > 0, at_end = pend_tile == t_end && k_end > 0;
            if (!(at_begin || at_end)) {
                write_tile(pend_tile, sum, true);
            } else {
                // every part goes to the workspace (sc1 stores, every storing wave drained, then one ticket per workgroup at agent
                // scope -- VFA_TICKET_ORDER at the head of this file --: vfa_pipe.hip, finish_run); the last arriver adds the parts in workgroup order -- one
                // fixed association -- and stores the tile
                const int which = at_begin ? 0 : 1;
                float *pp = a.partial + (((size_t)lb * 2 + which) * 8 + wave) * 16 * 64 + lane;
#pragma unroll
                for (int i = 0; i < 16; ++i) __hip_atomic_store(pp + i * 64, sum[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) s_ticket = __hip_atomic_fetch_add(a.tickets + pend_tile, 1u, VFA_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                const int first = at_begin ? sh_b_first : sh_e_first, last = at_begin ? sh_b_last : sh_e_last;
                const int parts = at_begin ? sh_b_parts : sh_e_parts;
                if (uniform_i((int)s_ticket) == parts - 1) {
                    // (the last arriver: nobody else touches this ticket in this launch -- clear it for the next one)
                    if (tid == 0) __hip_atomic_store(a.tickets + pend_tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    f32x16 tot;
#pragma unroll
                    for (int i = 0; i < 16; ++i) tot[i] = 0.0f;
                    for (int j = first; j <= last; ++j) {
                        if (j == lb) {
#pragma unroll
                            for (int i = 0; i < 16; ++i) tot[i] += sum[i];
                            continue;
                        }
                        int tb, kb, te, ke;
                        range_of(j, tb, kb, te, ke);
                        if (tb > te || (tb == te && kb >= ke)) continue;
                        const int wj = (pend_tile == tb && kb > 0) ? 0 : 1;
                        const float *qq = a.partial + (((size_t)j * 2 + wj) * 8 + wave) * 16 * 64 + lane;
#pragma unroll
                        for (int i = 0; i < 16; ++i) tot[i] += __hip_atomic_load(qq + i * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    write_tile(pend_tile, tot, true);
                }
            }
        } else {
            write_tile(pend_tile, sum, true);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) sum[i] = 0.0f;
        if (!DIRECT)
            for (int t2 = pend_tile + 1; t2 < pend_next; ++t2) write_tile(t2, sum, false); // fully masked tiles in between
        pend_tile = -1;
    };
    if (DIAG) t_prev = __builtin_amdgcn_s_memtime();

    // Per item:  wait for its tap window | pool -> A tile | issue the NEXT item's window and box records, fetch the header
    // after that | MFMAs of this item (the DMA and the loads land underneath).  Headers run two items ahead.
    while (cur.valid) {
        Item nn;
        nn.valid = false;
        if (nxt.valid) nn = seek(nxt.tile, nxt.scale, nxt.rest, nxt.rank + 1);
        tick(0);
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0): this wave's share of the tap window has landed
        tick(1);
        __syncthreads();                    // ... and everybody else's
        if (nxt.valid) header_get(nh, hbuf); // (the header of the next item landed with that wait)
        tick(2);
        flush();
        // (Tried in round 5, no gain: W of the NEXT scale requested at the end of the last item of this one -- behind the fills, waited
        // for with a counted vmcnt(32) at the head of the loop -- 446-448 us against 443 per launch in an A/B on one device.  Also
        // without effect: the younger wave of every SIMD delayed by 128 - 512 cycles at the head of pooling (434-438 against 435), or
        // raised to priority 1 / 3 for the pooling pass (438.0 / 437.4-438.4 against 437.8);
        // the window fills issued by 2 or 8 waves instead of 4: 453 / 455 against 442.)
        if (cur.scale != w_scale && !((dbg & kDbgOneW) && w_scale >= 0)) { // W and bias of this scale: land while the boxes are pooled
            load_weights(cur.scale);
            bc = bias_of(cur.scale);
            if constexpr (F16) { bc *= pick(f_fwd, cur.scale); sa_cur = pick(f_a, cur.scale); sa_inv_cur = pick(f_a_inv, cur.scale); inv_cur = pick(f_inv, cur.scale); }
            w_scale = cur.scale;
        }
        // (the conversions of the fp16 split saturate instead of overflowing; the MFMAs below need the default mode: vfa_split.h)
        // the item's own power of two: 2^ea of its scale, less the binary places its noisiest box asks for (header bits 8-15: sliver_shift)
        const int shift = F16 ? (cur_flags >> kTileShiftAt) & 0xff : 0;
        const float up = pow2f(shift), down = pow2f(-shift);
        if constexpr (F16) fp16_saturate_mode(true);
        if (!(dbg & kDbgNoPool)) pool(cur, cur_flags, cur_word1, sa_cur * down, sa_inv_cur * up);
        if constexpr (F16) fp16_saturate_mode(false);
        tick(3);
        // W, the bias and the tile stores of `flush` were issued a pooling pass ago.  Waiting for them HERE, explicitly, is
        // what keeps the compiler from doing it in front of the first MFMA, behind the DMA instructions issued below: its
        // in-order vmcnt would then cover those too -- a full memory round trip per item before the MFMA phase.
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();                    // A tile complete; the tap window is free again
        tick(4);
        f_n = 0;
        if (nxt.valid) {
            nxt_flags = nh[0]; nxt_word1 = nh[2];
            if (!(dbg & kDbgNoFills)) begin_fills(nxt, nh); // issued between the MFMAs below
            if (!(dbg & kDbgNoRecords)) load_record(nxt, nxt_flags, nxt_word1); // this lane's box of the next item: lands under the MFMAs
        }
        hbuf ^= 1;
        if (nn.valid) header_of(nn, hbuf);
        tick(5);

        // ONE accumulator chain: back-to-back dependent v_mfma_f32_32x32x16_bf16 issue at the pipe rate (32 cycles), so extra
        // chains buy nothing and their registers are better spent on running the A fragments ahead of the MFMAs
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = bc * down; // (the bias rides in the accumulator, in the item's units)
        if (!(dbg & kDbgNoMfma)) {
            int key2 = key, fb = frag_base; // (opaque: keeps the 16 swizzled fragment offsets out of long-lived registers)
            asm volatile("" : "+v"(key2), "+v"(fb));
            // The A fragments run two k-steps ahead of the MFMAs.  Their LDS reads and the waits for them are written in
            // assembly: LDS reads return in order, so `lgkmcnt(4)` = everything but the two youngest pairs has arrived, but
            // whenever the compiler's own bookkeeping needs a wait here it emits `lgkmcnt(0)` -- a drain that includes the
            // reads just issued -- in 5 of the 14 steps (with or without the explicit counted waits next to it).  The reads
            // are invisible to it this way; the `sched_barrier`s keep its MFMAs behind the hand-written waits.
            const unsigned pa = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)s_planes + (unsigned)fb;
            auto frags = [&](int k, bf16x8 &hh, bf16x8 &ll) {
                const unsigned addr = pa + (unsigned)((k ^ (key2 >> 1)) << 5);
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3" : "=&v"(hh), "=&v"(ll) : "v"(addr), "n"(kPlane) : "memory");
            };
            bf16x8 fh[3], fl[3];
            frags(0, fh[0], fl[0]);
            frags(1, fh[1], fl[1]);
#pragma unroll
            for (int k = 0; k < kSteps; ++k) {
                one_fill(k); // (wave-uniform branch; nothing to issue once the wave's share is on its way)
                if (k + 2 < kSteps) {
                    frags(k + 2, fh[(k + 2) % 3], fl[(k + 2) % 3]);
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fh[k % 3]), "+v"(fl[k % 3]));
                } else if (k + 1 < kSteps) {
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fh[k % 3]), "+v"(fl[k % 3]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fh[k % 3]), "+v"(fl[k % 3]));
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (F16) {
                    const f16x8 ah = __builtin_bit_cast(f16x8, fh[k % 3]), al = __builtin_bit_cast(f16x8, fl[k % 3]);
                    const f16x8 wh = __builtin_bit_cast(f16x8, w[k].hi), wl = __builtin_bit_cast(f16x8, w[k].lo);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh, acc, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[k % 3], w[k].lo, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[k % 3], w[k].hi, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[k % 3], w[k].hi, acc, 0, 0, 0);
                    if (TERMS >= 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[k % 3], w[k].lo, acc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        rest_fills(); // windows of more than 64 slots
#pragma unroll
        for (int i = 0; i < 16; ++i) { // vfa_op.py:124; vfanet.py:79, 82
            if constexpr (F16) sum[i] = fmaf(relu_t(acc[i]), inv_cur * up, sum[i]); // (relu(acc) 2^-(ea+ew-shift) is exact: the same bits as multiply, then add)
            else sum[i] = sum[i] + relu_t(acc[i]);
        }
        tick(6);

        if (!nxt.valid || nxt.tile != cur.tile) {
            pend_tile = cur.tile;
            pend_next = nxt.valid ? nxt.tile : t_end;
        }
        if (DIAG) stamp[7] += 1;
        cur = nxt;
        cur_flags = nxt_flags; cur_word1 = nxt_word1;
        nxt = nn;
    }
    flush();
    if (DIAG && (dbg & kDbgStamps) && a.diag && tid == ((dbg >> 8) & 7) * 64) // (bits 8..10: the wave that reports)
        for (int k = 0; k < 8; ++k) a.diag[(size_t)blockIdx.x * 8 + k] = stamp[k];
}

// ------------------------------------------------------------------------------------------------
// 3. box pooling alone, through LDS tap windows, with the voxel features written to HBM bit-exactly  (reference vfa_op.py:112-120)
// The same records and windows as the fused kernel, but as a plain high-occupancy kernel: a 256-thread workgroup owns ONE
// 64-channel quarter of one (view, tile): its window slice (256 B per tap, <= 31.5 KiB) arrives by LDS-DMA, then every wave pools
// 8 boxes (two instructions' worth: 16 lanes x float4 per box), divides exactly (box_mean) and streams 256 B per box to `vox`.
// Four workgroups per CU hide each other's DMA and LDS latencies; nothing is pipelined by hand.  The kernel is bound by the
// HBM write of the voxel features.  Items whose window does not fit read their taps from the image itself.
// ------------------------------------------------------------------------------------------------
struct PoolArgs {
    const float *integral;  // (n_views, Hf+2, Wf+2, 256)
    const unsigned char *hdrs, *recs;
    const unsigned *direct; // (n_tiles) views whose window does not fit LDS
    float *vox;             // (n_views, L * W, 256)
    int n_views, L, W, tiles_w, n_tiles, Hf, Wf;
    long long per_xcd;      // blocks per XCD
};

// Persistent form: 512-thread workgroups, two per CU, each walking a contiguous run of (view, tile, quarter) units with the
// window slice and the 32 box records of unit u + 1 arriving by LDS-DMA (into the other LDS buffer) while unit u is pooled.
//   wave w, lane (grp = lane >> 4, cq = lane & 15): box 4 w + grp of the tile, channels 64 q + 4 cq .. + 3.
__global__ __launch_bounds__(512, 2) void pool_windows_kernel(PoolArgs a)
{
    // two buffers as separate objects: hipcc drains every LDS-DMA in flight before an LDS read it cannot prove disjoint from it
    // (256 B per slot; a DMA instruction writes four slots, so the capacity is rounded up to a multiple of four)
    __shared__ float4 s_win0[(kMaxSlots + 3) / 4 * 4 * 16], s_win1[(kMaxSlots + 3) / 4 * 4 * 16];
    __shared__ __align__(16) unsigned char s_rec0[kTileBoxes * kRecBytes], s_rec1[kTileBoxes * kRecBytes];
    const int tid = threadIdx.x, wave = uniform_i(tid >> 6), lane = tid & 63, grp = lane >> 4, cq = lane & 15;
    const long long n_units = (long long)a.n_views * a.n_tiles * 4;
    const int nblk = gridDim.x;
    const int lb = (int)xcd_contiguous(blockIdx.x, (nblk + 7) / 8);
    if (lb >= nblk) return;
    const long long u_begin = n_units * lb / nblk, u_end = n_units * (lb + 1) / nblk;
    if (u_begin >= u_end) return;
    const int Wp = a.Wf + 2;
    const size_t img_stride = (size_t)(a.Hf + 2) * Wp * kSlotBytes;

    // (a compiler-issued scalar load: an asm s_load is invisible to the register allocator, which may spill the destination
    // registers before the data lands -- see the persistent kernel above)
    auto header_of = [&](long long u, i32x8 &hd) {
        const unsigned char *p = a.hdrs + (size_t)(u >> 2) * kHdrBytes;
        hd = *reinterpret_cast<const __attribute__((address_space(4))) i32x8 *>((size_t)p);
    };
    auto header_wait = [&](i32x8 &) {};
    // position of a unit, advanced by counting (the CU has ONE scalar unit: divisions per unit and wave made the kernel
    // scalar-bound)
    struct Pos { int item, q, view, tl, tw; };
    auto pos_of = [&](long long u) {
        Pos p;
        p.item = (int)(u >> 2); p.q = (int)(u & 3);
        p.view = p.item / a.n_tiles;
        const int tile = p.item - p.view * a.n_tiles;
        p.tl = tile / a.tiles_w; p.tw = tile - p.tl * a.tiles_w;
        return p;
    };
    auto advance = [&](Pos &p) {
        if (++p.q == 4) {
            p.q = 0; ++p.item;
            if (++p.tw == a.tiles_w) {
                p.tw = 0;
                if (++p.tl * a.tiles_w >= a.n_tiles) { p.tl = 0; ++p.view; }
            }
        }
    };
    // LDS-DMA of unit u into buffer B: the 256-byte quarter of every window slot (4 slots per instruction, 16 lanes each)
    // and the 3 KiB of box records (waves 0-2)
    auto fetch = [&](auto buf_tag, const Pos &ps, const i32x8 &hd) {
        constexpr int B = decltype(buf_tag)::value;
        const int item = ps.item, q = ps.q, view = ps.view;
        const int flags = hd[0], n_slots = hd[1], cwid = hd[2], inv = hd[3], x0 = hd[4], t0 = hd[5], top_rows = hd[6], b0 = hd[7];
        if (wave < 3) {
            const unsigned char *src = a.recs + (size_t)item * kTileBoxes * kRecBytes + wave * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)((B ? s_rec1 : s_rec0) + wave * 1024), 16, 0, 0);
        }
        if ((flags & kTileLive) && !(flags & kTileDirect)) { // (direct items: pool_direct_kernel)
            const char *img = reinterpret_cast<const char *>(a.integral) + view * img_stride + q * 256 + cq * 16;
            for (int s = 4 * wave; s < n_slots; s += 32) {
                const int sl = min(s + grp, n_slots - 1);
                const int wr = (sl * inv) >> 16, wc = sl - wr * cwid;
                const int y = wr < top_rows ? t0 + wr : b0 + (wr - top_rows), x = x0 + wc;
                const char *src = img + ((size_t)(y + 1) * Wp + (x + 1)) * kSlotBytes;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)((B ? s_win1 : s_win0) + s * 16), 16, 0, 0);
            }
        }
    };
    // The result of a unit is stored one unit LATER, right behind the wait at the head of the loop: the store then has a whole
    // unit to drain before the next `vmcnt(0)` (which must only wait for the window that was requested a unit ago).
    float4 keep = make_float4(0.f, 0.f, 0.f, 0.f);
    float *keep_at = nullptr;
    auto flush = [&]() {
        if (keep_at) {
            typedef float nt4 __attribute__((ext_vector_type(4)));
            const nt4 x = {keep.x, keep.y, keep.z, keep.w};
            __builtin_nontemporal_store(x, reinterpret_cast<nt4 *>(keep_at)); // written once, read once by the collapse kernel
        }
    };
    // pool unit u out of buffer B
    auto pool = [&](auto buf_tag, const Pos &ps, int flags) {
        constexpr int B = decltype(buf_tag)::value;
        const int q = ps.q, view = ps.view, tl = ps.tl, tw = ps.tw;
        if (flags & kTileDirect) { // left to pool_direct_kernel
            keep_at = nullptr;
            return;
        }
        const int b = 4 * wave + grp;
        const int cl = tl * kTileL + (b >> 3), cw = tw * kTileW + (b & 7);
        // LDS reads by inline asm: the window of the NEXT unit is arriving by DMA in the other buffer, and hipcc puts a
        // `vmcnt(0)` in front of every LDS read it generates while an LDS-DMA is in flight (it would undo the prefetch); the
        // barrier at the head of the loop made this buffer visible.
        uint4 r[6];
        {
            const unsigned ra = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)(B ? s_rec1 : s_rec0) + (unsigned)(b * kRecBytes);
            asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:32\n\t"
                         "ds_read_b128 %3, %6 offset:48\n\tds_read_b128 %4, %6 offset:64\n\tds_read_b128 %5, %6 offset:80\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]) : "v"(ra) : "memory");
        }
        const float wt[16] = {__uint_as_float(r[0].x), __uint_as_float(r[0].y), __uint_as_float(r[0].z), __uint_as_float(r[0].w),
                              __uint_as_float(r[1].x), __uint_as_float(r[1].y), __uint_as_float(r[1].z), __uint_as_float(r[1].w),
                              __uint_as_float(r[2].x), __uint_as_float(r[2].y), __uint_as_float(r[2].z), __uint_as_float(r[2].w),
                              __uint_as_float(r[3].x), __uint_as_float(r[3].y), __uint_as_float(r[3].z), __uint_as_float(r[3].w)};
        const float rcp = __uint_as_float(r[4].x), masked = __uint_as_float(r[5].z), area = __uint_as_float(r[5].w);
        const bool vis = (r[4].y & (unsigned)kVis) != 0u;
        // tap coordinates of a masked box are meaningless: point them at slot / pixel 0 (the value is discarded)
        unsigned rws[4] = {r[4].z & 0xffffu, r[4].z >> 16, r[4].w & 0xffffu, r[4].w >> 16};
        unsigned cls[4] = {r[5].x & 0xffffu, r[5].x >> 16, r[5].y & 0xffffu, r[5].y >> 16};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            rws[k] = vis ? rws[k] : 0u;
            cls[k] = vis ? cls[k] : 0u;
        }
        typedef float nf4 __attribute__((ext_vector_type(4))); // (a native vector: inline asm cannot tie HIP's float4 struct)
        nf4 t[4][4];
        {
            const unsigned wa = (unsigned)(size_t)(__attribute__((address_space(3))) float4 *)(B ? s_win1 : s_win0) + (unsigned)(cq * 16);
            unsigned ad[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) ad[i][j] = wa + (rws[i] + cls[j]) * 256u;
#pragma unroll
            for (int i = 0; i < 4; i += 2) // eight reads per statement; the wait names all sixteen destinations
                asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"
                             "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15"
                             : "=&v"(t[i][0]), "=&v"(t[i][1]), "=&v"(t[i][2]), "=&v"(t[i][3]), "=&v"(t[i + 1][0]), "=&v"(t[i + 1][1]),
                               "=&v"(t[i + 1][2]), "=&v"(t[i + 1][3])
                             : "v"(ad[i][0]), "v"(ad[i][1]), "v"(ad[i][2]), "v"(ad[i][3]), "v"(ad[i + 1][0]), "v"(ad[i + 1][1]),
                               "v"(ad[i + 1][2]), "v"(ad[i + 1][3])
                             : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(t[0][0]), "+v"(t[0][1]), "+v"(t[0][2]), "+v"(t[0][3]), "+v"(t[1][0]), "+v"(t[1][1]), "+v"(t[1][2]), "+v"(t[1][3]),
                           "+v"(t[2][0]), "+v"(t[2][1]), "+v"(t[2][2]), "+v"(t[2][3]), "+v"(t[3][0]), "+v"(t[3][1]), "+v"(t[3][2]), "+v"(t[3][3])
                         :: "memory");
        }
        auto f4 = [](const nf4 &v) { return make_float4(v.x, v.y, v.z, v.w); };
        const float4 lt = sample4(f4(t[0][0]), f4(t[0][1]), f4(t[1][0]), f4(t[1][1]), wt[0], wt[1], wt[2], wt[3]);
        const float4 rb = sample4(f4(t[2][2]), f4(t[2][3]), f4(t[3][2]), f4(t[3][3]), wt[4], wt[5], wt[6], wt[7]);
        const float4 rt = sample4(f4(t[0][2]), f4(t[0][3]), f4(t[1][2]), f4(t[1][3]), wt[8], wt[9], wt[10], wt[11]);
        const float4 lb2 = sample4(f4(t[2][0]), f4(t[2][1]), f4(t[3][0]), f4(t[3][1]), wt[12], wt[13], wt[14], wt[15]);
        float4 res = make_float4(box_mean(lt.x, rb.x, rt.x, lb2.x, area, rcp), box_mean(lt.y, rb.y, rt.y, lb2.y, area, rcp),
                                 box_mean(lt.z, rb.z, rt.z, lb2.z, area, rcp), box_mean(lt.w, rb.w, rt.w, lb2.w, area, rcp));
        if (!vis) res = make_float4(masked, masked, masked, masked);
        const bool inside = cl < a.L && cw < a.W; // 256 B per box
        keep = res;
        keep_at = inside ? a.vox + ((size_t)view * a.L * a.W + (size_t)cl * a.W + cw) * kC + q * 64 + cq * 4 : nullptr;
    };

    i32x8 hc, hn;
    header_of(u_begin, hc);
    header_wait(hc);
    Pos pc = pos_of(u_begin), pn = pc; // the unit being pooled, the unit being fetched
    fetch(std::integral_constant<int, 0>{}, pc, hc);
    advance(pn);
    if (u_begin + 1 < u_end) header_of(u_begin + 1, hn);
    // the loop is unrolled by two so that the buffer of every LDS access is a compile-time fact
    for (long long u = u_begin; u < u_end; u += 2) {
        // ---- even: pool buffer 0, fetch into buffer 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the window of unit u has landed (and the store of unit u - 2)
        __builtin_amdgcn_s_barrier();                    // ... for every wave, and everybody is done with buffer 1
        asm volatile("" ::: "memory");
        flush();
        if (u + 1 < u_end) {
            header_wait(hn);
            fetch(std::integral_constant<int, 1>{}, pn, hn);
        }
        pool(std::integral_constant<int, 0>{}, pc, hc[0]);
        if (u + 1 >= u_end) break;
        hc = hn;
        pc = pn;
        advance(pn);
        if (u + 2 < u_end) header_of(u + 2, hn);
        // ---- odd: pool buffer 1, fetch into buffer 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        flush();
        if (u + 2 < u_end) {
            header_wait(hn);
            fetch(std::integral_constant<int, 0>{}, pn, hn);
        }
        pool(std::integral_constant<int, 1>{}, pc, hc[0]);
        hc = hn;
        pc = pn;
        advance(pn);
        if (u + 3 < u_end) header_of(u + 3, hn);
    }
    flush();
}

// The items the window kernel leaves out (tap window larger than the LDS slice: boxes right in front of a camera): one
// 512-thread workgroup per (tile, quarter) walks the views flagged in the direct mask and reads every tap from the image.
__global__ __launch_bounds__(512) void pool_direct_kernel(PoolArgs a)
{
    const int tid = threadIdx.x, wave = uniform_i(tid >> 6), lane = tid & 63, grp = lane >> 4, cq = lane & 15;
    const int tile = blockIdx.x >> 2, q = blockIdx.x & 3;
    unsigned mask = (unsigned)uniform_i((int)a.direct[tile]);
    if (a.n_views < 32) mask &= (1u << a.n_views) - 1u;
    const int tl = tile / a.tiles_w, tw = tile - tl * a.tiles_w;
    const int b = 4 * wave + grp;
    const int cl = tl * kTileL + (b >> 3), cw = tw * kTileW + (b & 7);
    const int Wp = a.Wf + 2;
    while (mask) {
        const int view = __builtin_ctz(mask);
        mask &= mask - 1u;
        const int item = view * a.n_tiles + tile;
        const uint4 *rp = reinterpret_cast<const uint4 *>(a.recs + ((size_t)item * kTileBoxes + b) * kRecBytes);
        uint4 r[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) r[k] = rp[k];
        const float wt[16] = {__uint_as_float(r[0].x), __uint_as_float(r[0].y), __uint_as_float(r[0].z), __uint_as_float(r[0].w),
                              __uint_as_float(r[1].x), __uint_as_float(r[1].y), __uint_as_float(r[1].z), __uint_as_float(r[1].w),
                              __uint_as_float(r[2].x), __uint_as_float(r[2].y), __uint_as_float(r[2].z), __uint_as_float(r[2].w),
                              __uint_as_float(r[3].x), __uint_as_float(r[3].y), __uint_as_float(r[3].z), __uint_as_float(r[3].w)};
        const float rcp = __uint_as_float(r[4].x), masked = __uint_as_float(r[5].z), area = __uint_as_float(r[5].w);
        const bool vis = (r[4].y & (unsigned)kVis) != 0u;
        // records of a direct item hold pixel coordinates (+ 1: the zero border); a masked box reads pixel 0 and discards it
        const unsigned rws[4] = {vis ? r[4].z & 0xffffu : 0u, vis ? r[4].z >> 16 : 0u, vis ? r[4].w & 0xffffu : 0u, vis ? r[4].w >> 16 : 0u};
        const unsigned cls[4] = {vis ? r[5].x & 0xffffu : 0u, vis ? r[5].x >> 16 : 0u, vis ? r[5].y & 0xffffu : 0u, vis ? r[5].y >> 16 : 0u};
        const char *img = reinterpret_cast<const char *>(a.integral) + (size_t)view * (a.Hf + 2) * Wp * kSlotBytes + q * 256 + cq * 16;
        float4 t[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t[i][j] = *reinterpret_cast<const float4 *>(img + (size_t)(rws[i] * Wp + cls[j]) * kSlotBytes);
        // (the sixteen loads stay in front of the arithmetic: left to itself the compiler kept 62 registers and two or three loads in
        // flight, eight round trips per unit instead of one -- round 6)
        __builtin_amdgcn_sched_barrier(0);
        const float4 lt = sample4(t[0][0], t[0][1], t[1][0], t[1][1], wt[0], wt[1], wt[2], wt[3]);
        const float4 rb = sample4(t[2][2], t[2][3], t[3][2], t[3][3], wt[4], wt[5], wt[6], wt[7]);
        const float4 rt = sample4(t[0][2], t[0][3], t[1][2], t[1][3], wt[8], wt[9], wt[10], wt[11]);
        const float4 lb2 = sample4(t[2][0], t[2][1], t[3][0], t[3][1], wt[12], wt[13], wt[14], wt[15]);
        float4 res = make_float4(box_mean(lt.x, rb.x, rt.x, lb2.x, area, rcp), box_mean(lt.y, rb.y, rt.y, lb2.y, area, rcp),
                                 box_mean(lt.z, rb.z, rt.z, lb2.z, area, rcp), box_mean(lt.w, rb.w, rt.w, lb2.w, area, rcp));
        if (!vis) res = make_float4(masked, masked, masked, masked);
        if (cl < a.L && cw < a.W) {
            typedef float nt4 __attribute__((ext_vector_type(4)));
            const nt4 x = {res.x, res.y, res.z, res.w};
            __builtin_nontemporal_store(x, reinterpret_cast<nt4 *>(a.vox + ((size_t)view * a.L * a.W + (size_t)cl * a.W + cw) * kC + q * 64 + cq * 4));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 2a. pre-pass of the fused path: the direct items (tap window larger than LDS: boxes right in front of a camera) are pooled
// here, at full occupancy and with every tap load of a box in flight at once, into fp32 rows in the workspace.  Inside the
// persistent kernel those loads had nothing to hide behind (W takes half of the register file: 8 taps per round trip,
// ~53 000 cycles per item).  Same arithmetic as the persistent kernel's own pooling (the correctly rounded v / area: box_quotient_scaled).
// A fixed grid walks the list of direct items the geometry pass appended to (a grid over all tiles spent its 27 us launching
// 15 000 workgroups, 95 % of which had nothing to do); one wave = 4 boxes x 64 channels.
// ------------------------------------------------------------------------------------------------
struct RowsArgs {
    const float *integral[kMaxScales];
    const unsigned char *recs[kMaxScales];
    int Hf[kMaxScales], Wf[kMaxScales];
    const unsigned *row_list;    // slot -> scale << 30 | view << 25 | tile (frame_records_kernel)
    const unsigned *row_counter; // direct items of the frame
    float *rows;
    int n_tiles, rows_cap;
};
__global__ __launch_bounds__(512) void pool_rows_kernel(RowsArgs a)
{
    const int tid = threadIdx.x, wave = uniform_i(tid >> 6), lane = tid & 63, grp = lane >> 4, cq = lane & 15;
    const int count = min(uniform_i((int)*a.row_counter), a.rows_cap);
    const int b = 4 * wave + grp;
    // unit = (direct item, 64-channel quarter); the items come from the list the geometry pass appended to
    for (int u = blockIdx.x; u < 4 * count; u += gridDim.x) {
        const int slot = u >> 2, q = u & 3;
        const unsigned e = (unsigned)uniform_i((int)a.row_list[slot]);
        const int s = (int)(e >> 30), view = (int)((e >> 25) & 31u), tile = (int)(e & 0x1ffffffu);
        const int Wp = a.Wf[s] + 2;
        const int item = view * a.n_tiles + tile;
        const uint4 *rp = reinterpret_cast<const uint4 *>(a.recs[s] + ((size_t)item * kTileBoxes + b) * kRecBytes);
        uint4 r[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) r[k] = rp[k];
        const float wt[16] = {__uint_as_float(r[0].x), __uint_as_float(r[0].y), __uint_as_float(r[0].z), __uint_as_float(r[0].w),
                              __uint_as_float(r[1].x), __uint_as_float(r[1].y), __uint_as_float(r[1].z), __uint_as_float(r[1].w),
                              __uint_as_float(r[2].x), __uint_as_float(r[2].y), __uint_as_float(r[2].z), __uint_as_float(r[2].w),
                              __uint_as_float(r[3].x), __uint_as_float(r[3].y), __uint_as_float(r[3].z), __uint_as_float(r[3].w)};
        const float rcp = __uint_as_float(r[4].x), masked = __uint_as_float(r[5].z), area = __uint_as_float(r[5].w);
        const bool vis = (r[4].y & (unsigned)kVis) != 0u;
        // records of a direct item hold pixel coordinates (+ 1: the zero border); a masked box reads pixel 0 and discards it
        const unsigned rws[4] = {vis ? r[4].z & 0xffffu : 0u, vis ? r[4].z >> 16 : 0u, vis ? r[4].w & 0xffffu : 0u, vis ? r[4].w >> 16 : 0u};
        const unsigned cls[4] = {vis ? r[5].x & 0xffffu : 0u, vis ? r[5].x >> 16 : 0u, vis ? r[5].y & 0xffffu : 0u, vis ? r[5].y >> 16 : 0u};
        const char *img = reinterpret_cast<const char *>(a.integral[s]) + (size_t)view * (a.Hf[s] + 2) * Wp * kSlotBytes + q * 256 + cq * 16;
        float4 t[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t[i][j] = *reinterpret_cast<const float4 *>(img + (size_t)(rws[i] * Wp + cls[j]) * kSlotBytes);
        // (the sixteen loads stay in front of the arithmetic: left to itself the compiler kept 62 registers and two or three loads in
        // flight, eight round trips per unit instead of one -- round 6)
        __builtin_amdgcn_sched_barrier(0);
        const float4 lt = sample4(t[0][0], t[0][1], t[1][0], t[1][1], wt[0], wt[1], wt[2], wt[3]);
        const float4 rb = sample4(t[2][2], t[2][3], t[3][2], t[3][3], wt[4], wt[5], wt[6], wt[7]);
        const float4 rt = sample4(t[0][2], t[0][3], t[1][2], t[1][3], wt[8], wt[9], wt[10], wt[11]);
        const float4 lb2 = sample4(t[2][0], t[2][1], t[3][0], t[3][1], wt[12], wt[13], wt[14], wt[15]);
        // RN((((lt + rb) - rt) - lb) / area): the reference's correctly rounded quotient (vfa_geom.h: box_mean)
        float4 res = make_float4(box_mean(lt.x, rb.x, rt.x, lb2.x, area, rcp), box_mean(lt.y, rb.y, rt.y, lb2.y, area, rcp),
                                 box_mean(lt.z, rb.z, rt.z, lb2.z, area, rcp), box_mean(lt.w, rb.w, rt.w, lb2.w, area, rcp));
        if (!vis) res = make_float4(masked, masked, masked, masked);
        reinterpret_cast<float4 *>(a.rows)[((size_t)slot * kTileBoxes + b) * (kC / 4) + q * 16 + cq] = res;
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WorkspaceLayout {
    size_t live[kMaxScales], direct[kMaxScales], overflow[kMaxScales], counter, hdrs[kMaxScales], recs[kMaxScales], wfrag[kMaxScales],
        masks_bytes, chunks, ranks, diag, rows, row_list, partial, tickets, item_w, wmax, wexp, amax, total;
    int tiles_l, tiles_w, n_tiles, rows_cap, views_pad;
};
inline WorkspaceLayout layout_of(int n_views, int L, int W, int n_scales)
{
    WorkspaceLayout w;
    w.tiles_l = (L + kTileL - 1) / kTileL;
    w.tiles_w = (W + kTileW - 1) / kTileW;
    w.n_tiles = w.tiles_l * w.tiles_w;
    size_t off = 0;
    for (int s = 0; s < kMaxScales; ++s) { // the view masks of all scales first, contiguous: zeroed by ONE memset
        const bool on = s < n_scales;
        w.live[s] = off;  off = align_up(off + (on ? (size_t)w.n_tiles * 4 : 0), 256);
        w.direct[s] = off; off = align_up(off + (on ? (size_t)w.n_tiles * 4 : 0), 256);
        w.overflow[s] = off; off = align_up(off + (on ? (size_t)w.n_tiles * 4 : 0), 256);
    }
    w.counter = off; off += 256; // direct items numbered so far
    // hand-off of tiles cut between workgroups of the persistent kernel: a ticket per tile (two 32 KiB parts per workgroup: `partial`).
    // Zeroed with the masks by the geometry call; the workgroup that draws a tile's last ticket puts it back to zero, so every later
    // call on the same workspace finds them clear without a memset of its own (two fill kernels, ~13 us, in front of every launch).
    w.tickets = off; off = align_up(off + (size_t)w.n_tiles * sizeof(unsigned), 256);
    w.masks_bytes = off;
    for (int s = 0; s < kMaxScales; ++s) {
        const bool on = s < n_scales;
        w.hdrs[s] = off;  off = align_up(off + (on ? (size_t)n_views * w.n_tiles * kHdrBytes : 0), 256);
        w.recs[s] = off;  off = align_up(off + (on ? ((size_t)n_views * w.n_tiles * kTileBoxes + 1) * kRecBytes : 0), 256);
        w.wfrag[s] = off; off = align_up(off + (on ? (size_t)8 * kSteps * 2 * 64 * 16 : 0), 256);
    }
    w.chunks = off;
    off = align_up(off + (kChunks + 1) * sizeof(int), 256);
    w.ranks = off;
    off = align_up(off + (kChunks + 1) * sizeof(int), 256);
    w.partial = off;
    off = align_up(off + (size_t)kMaxBlocks * 2 * 8 * 16 * 64 * sizeof(float), 256);
    w.diag = off;
    off = align_up(off + 512 * 8 * sizeof(unsigned long long), 256); // diagnostic build: 8 counters per workgroup
    // pooled rows of the direct items: room for a quarter of all (view, tile, scale) items, at most 256 MiB; the rest (none on the
    // BASELINE frames: 3 % of the items of the bench frame are direct) goes through the second launch
    // Round 6: where a row slot for EVERY item costs at most 1 GiB (the bench frame: 26 250 items, 860 MB of address space of which
    // the 3-4 % direct items are ever touched) the workspace has them all: no item can be left without one, and the frame call
    // drops the second launch (4 us of an empty kernel + a dependent-launch gap per frame).
    const size_t items = (size_t)n_views * w.n_tiles * n_scales;
    size_t cap = items / 4 + 64;
    if (cap > 8192) cap = 8192;
    if (items * (size_t)(kTileBoxes * kC * sizeof(float)) <= ((size_t)1 << 30)) cap = items;
    if (cap > items) cap = items;
    w.rows_cap = (int)cap;
    w.row_list = off;
    off = align_up(off + cap * sizeof(unsigned), 256);
    w.wmax = off; off = align_up(off + (size_t)kMaxScales * kWmaxParts * sizeof(unsigned), 256); // fp16 split: partial maxima of |W| per scale,
    w.wexp = off; off = align_up(off + (kMaxScales + 1) * sizeof(int), 256);                            // ... the weight exponents,
    w.amax = off; off = align_up(off + (size_t)kMaxScales * kFallbackStats * sizeof(unsigned), 256); // ... and feature statistics made here for callers that pass none
    w.views_pad = (n_views + 7) / 8 * 8; // cost estimates of the items, per tile (tile_chunks_kernel)
    w.item_w = off;
    off = align_up(off + (size_t)w.n_tiles * kMaxScales * w.views_pad * sizeof(unsigned short) + 16, 256);
    w.rows = off;
    off = align_up(off + cap * kTileBoxes * kC * sizeof(float), 256);
    w.total = off;
    return w;
}

// row slots that fit a workspace of `bytes` (the rows are the last region; fewer slots only send more direct items to the
// second launch of the persistent kernel)
inline int rows_cap_of(const WorkspaceLayout &lay, size_t bytes)
{
    if (bytes <= lay.rows) return 0;
    const size_t fit = (bytes - lay.rows) / ((size_t)kTileBoxes * kC * sizeof(float));
    return fit < (size_t)lay.rows_cap ? (int)fit : lay.rows_cap;
}

} // namespace

extern "C" {

size_t vfa_frame_workspace_bytes(int n_views, int L, int W, int n_scales)
{
    if (n_views < 0 || L < 0 || W < 0 || n_scales < 1 || n_scales > kMaxScales) return 0;
    return layout_of(n_views, L, W, n_scales).total;
}

int vfa_frame_workspace_layout(int n_views, int L, int W, int n_scales, size_t *offsets, int *tiles)
{
    if (n_views < 0 || L < 0 || W < 0 || n_scales < 1 || n_scales > kMaxScales || !offsets || !tiles) return VFA_ERR_BAD_ARGUMENT;
    const WorkspaceLayout lay = layout_of(n_views, L, W, n_scales);
    for (int k = 0; k < kMaxScales; ++k) {
        offsets[5 * k + 0] = lay.live[k];
        offsets[5 * k + 1] = lay.direct[k];
        offsets[5 * k + 2] = lay.hdrs[k];
        offsets[5 * k + 3] = lay.recs[k];
        offsets[5 * k + 4] = lay.wfrag[k];
    }
    offsets[15] = lay.diag;
    offsets[16] = lay.total;
    for (int k = 0; k < kMaxScales; ++k) offsets[17 + k] = lay.overflow[k];
    offsets[20] = lay.counter;
    offsets[21] = lay.rows;
    offsets[22] = (size_t)lay.rows_cap;
    offsets[23] = lay.chunks;
    offsets[24] = lay.ranks;
    tiles[0] = lay.tiles_l;
    tiles[1] = lay.tiles_w;
    tiles[2] = kMaxSlots;
    tiles[3] = kChunks;
    return 0;
}

int vfa_frame_boxes_f32(const float *calibs, const float *grid, const float *z_layers, const float *corner_off, int n_views, int L,
                        int W, int conv_kind, float img_w, float img_h, float cmin, float cmax, int n_scales,
                        const int *feat_hw, void *workspace, size_t workspace_bytes, void *stream)
{
    if (n_views < 0 || L < 0 || W < 0 || n_scales < 1 || n_scales > kMaxScales || conv_kind < 0 || conv_kind > 2 || !feat_hw)
        return VFA_ERR_BAD_ARGUMENT;
    if (n_views > 32) return VFA_ERR_UNSUPPORTED; // live-view masks are 32 bits wide
    const WorkspaceLayout lay = layout_of(n_views, L, W, n_scales);
    if (lay.n_tiles == 0 || n_views == 0) return 0;
    if ((long long)n_views * lay.n_tiles >= (1ll << 31) - 2 || lay.n_tiles >= (1 << 25)) return VFA_ERR_UNSUPPORTED; // (row_list packs the tile in 25 bits)
    if (!workspace || workspace_bytes < lay.rows) return VFA_ERR_BAD_ARGUMENT; // (everything in front of the row slots is mandatory)
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
        a.direct[k] = reinterpret_cast<unsigned *>(ws + lay.direct[k]);
        a.overflow[k] = reinterpret_cast<unsigned *>(ws + lay.overflow[k]);
        a.hdrs[k] = ws + lay.hdrs[k];
        a.recs[k] = ws + lay.recs[k];
    }
    // one memset (all view masks) and the records kernel (which also clears the spare record behind each scale's table)
    a.row_counter = reinterpret_cast<unsigned *>(ws + lay.counter);
    a.rows_cap = (unsigned)rows_cap_of(lay, workspace_bytes);
    a.row_list = reinterpret_cast<unsigned *>(ws + lay.row_list);
    a.item_w = reinterpret_cast<unsigned short *>(ws + lay.item_w);
    a.views_pad = lay.views_pad;
    const hipError_t e = zero_fill(ws, lay.masks_bytes, s); // (a kernel, not hipMemsetAsync: see vfa_geom.h)
    if (e != hipSuccess) return (int)e;
    const long long pairs = (long long)n_views * lay.n_tiles;
    hipLaunchKernelGGL(frame_records_kernel, dim3((unsigned)((pairs + 1) / 2)), dim3(kWave), 0, s, a);
    return (int)hipGetLastError();
}

int vfa_frame_cuts_f32(int n_views, int L, int W, int n_scales, const float *const *weights, int flags, void *workspace,
                       size_t workspace_bytes, void *stream)
{
    const int terms = flags & VFA_FLAG_TERMS_MASK;
    if ((flags & ~VFA_FLAG_TERMS_MASK) || (terms != 0 && terms != 2 && terms != 3 && terms != 4)) return VFA_ERR_BAD_ARGUMENT;
    if (n_views < 0 || L < 0 || W < 0 || n_scales < 1 || n_scales > kMaxScales) return VFA_ERR_BAD_ARGUMENT;
    if (n_views > 32) return VFA_ERR_UNSUPPORTED;
    const WorkspaceLayout lay = layout_of(n_views, L, W, n_scales);
    if (lay.n_tiles == 0 || n_views == 0) return 0;
    if (!workspace || workspace_bytes < lay.rows) return VFA_ERR_BAD_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    ChunkArgs ca;
    for (int k = 0; k < kMaxScales; ++k) {
        const int q = k < n_scales ? k : 0;
        ca.live[k] = reinterpret_cast<const unsigned *>(ws + lay.live[q]);
        ca.overflow[k] = reinterpret_cast<const unsigned *>(ws + lay.overflow[q]);
    }
    ca.n_scales = n_scales; ca.n_tiles = lay.n_tiles; ca.n_views = n_views; ca.views_pad = lay.views_pad;
    ca.item_w = reinterpret_cast<const unsigned short *>(ws + lay.item_w);
    ca.chunk_start = reinterpret_cast<int *>(ws + lay.chunks);
    ca.chunk_rank = reinterpret_cast<int *>(ws + lay.ranks);
    ca.split = SplitArgs{};
    ca.n_split_blocks = 0;
    if (weights) { // the split of the collapse weights: the spare blocks of the same launch
        SplitArgs &sa = ca.split;
        for (int k = 0; k < kMaxScales; ++k) {
            sa.w[k] = weights[k < n_scales ? k : 0];
            sa.out[k] = reinterpret_cast<uint4 *>(ws + lay.wfrag[k < n_scales ? k : 0]);
            if (!sa.w[k]) return VFA_ERR_BAD_ARGUMENT;
        }
        sa.wmax = reinterpret_cast<unsigned *>(ws + lay.wmax);
        sa.wexp = reinterpret_cast<int *>(ws + lay.wexp);
        sa.f16 = (terms == 0 || terms == 2) ? 1 : 0;
        ca.n_split_blocks = n_scales * kSplitBlocks;
    }
    if (n_views <= 8) hipLaunchKernelGGL(tile_chunks_kernel<1>, dim3(1 + ca.n_split_blocks), dim3(1024), 0, s, ca);
    else hipLaunchKernelGGL(tile_chunks_kernel<4>, dim3(1 + ca.n_split_blocks), dim3(1024), 0, s, ca);
    return (int)hipGetLastError();
}

// a single entry point = few launches: boxes (memset + records kernel), then the work cuts and one weight-split launch
int vfa_frame_records_f32(const float *calibs, const float *grid, const float *z_layers, const float *corner_off, int n_views, int L,
                          int W, int conv_kind, float img_w, float img_h, float cmin, float cmax, int n_scales,
                          const int *feat_hw, const float *const *weights, int flags, void *workspace, size_t workspace_bytes, void *stream)
{
    const int st = vfa_frame_boxes_f32(calibs, grid, z_layers, corner_off, n_views, L, W, conv_kind, img_w, img_h, cmin, cmax, n_scales, feat_hw,
                                       workspace, workspace_bytes, stream);
    if (st) return st;
    return vfa_frame_cuts_f32(n_views, L, W, n_scales, weights, flags, workspace, workspace_bytes, stream);
}

int vfa_pool_windows_f32(const float *integral, const void *workspace, size_t workspace_bytes, float *vox, int n_views, int L, int W,
                         int n_scales, int scale, int Hf, int Wf, void *stream)
{
    if (n_views < 0 || L < 0 || W < 0 || n_scales < 1 || n_scales > kMaxScales || scale < 0 || scale >= n_scales || Hf <= 0 || Wf <= 0)
        return VFA_ERR_BAD_ARGUMENT;
    if (n_views > 32) return VFA_ERR_UNSUPPORTED;
    const WorkspaceLayout lay = layout_of(n_views, L, W, n_scales);
    if (lay.n_tiles == 0 || n_views == 0) return 0;
    if (!workspace || workspace_bytes < lay.rows || !integral || !vox) return VFA_ERR_BAD_ARGUMENT;
    const unsigned char *ws = reinterpret_cast<const unsigned char *>(workspace);
    PoolArgs a;
    a.integral = integral; a.hdrs = ws + lay.hdrs[scale]; a.recs = ws + lay.recs[scale]; a.vox = vox;
    a.direct = reinterpret_cast<const unsigned *>(ws + lay.direct[scale]);
    a.n_views = n_views; a.L = L; a.W = W; a.tiles_w = lay.tiles_w; a.n_tiles = lay.n_tiles; a.Hf = Hf; a.Wf = Wf;
    const long long units = (long long)n_views * lay.n_tiles * 4;
    if (units >= (1ll << 31) - 8) return VFA_ERR_UNSUPPORTED;
    a.per_xcd = 0;
    int n_cu = 256;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            cus > 0)
            n_cu = cus;
    }
    long long nblk = 2ll * n_cu; // two persistent workgroups per CU (LDS: two double-buffered windows)
    if (nblk > units) nblk = units;
    nblk = (nblk + 7) / 8 * 8;
    hipLaunchKernelGGL(pool_windows_kernel, dim3((unsigned)nblk), dim3(512), 0, (hipStream_t)stream, a);
    int st = (int)hipGetLastError();
    if (st) return st;
    if ((long long)lay.n_tiles * 4 >= (1ll << 31)) return VFA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pool_direct_kernel, dim3((unsigned)(lay.n_tiles * 4)), dim3(512), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

int vfa_pool_collapse_relu_sum_f32(const float *const *integrals, const unsigned *const *feat_absmax, const float *const *biases,
                                   const void *workspace, size_t workspace_bytes, float *out, int n_views, int L, int W, int n_scales,
                                   const int *feat_hw, int accumulate, int flags, void *stream)
{
    const int terms = flags & VFA_FLAG_TERMS_MASK, reserved_cus = (flags >> 8) & 0xff;
    const int debug = ((flags >> 16) & 0xfff) | ((flags & VFA_FLAG_DUMP_VOX) ? kDbgDumpVox : 0);
    if (debug && terms != 0 && terms != 2) return VFA_ERR_BAD_ARGUMENT; // (the diagnostic build exists for the default arithmetic only)
    if (flags & ~(VFA_FLAG_TERMS_MASK | 0xfffff00 | VFA_FLAG_ROWS_ONLY | VFA_FLAG_SKIP_ROWS | VFA_FLAG_DUMP_VOX)) return VFA_ERR_BAD_ARGUMENT;
    if ((flags & VFA_FLAG_ROWS_ONLY) && (flags & VFA_FLAG_SKIP_ROWS)) return VFA_ERR_BAD_ARGUMENT;
    if (n_views < 0 || L < 0 || W < 0 || n_scales < 1 || n_scales > kMaxScales || !feat_hw || !integrals ||
        (terms != 0 && terms != 2 && terms != 3 && terms != 4))
        return VFA_ERR_BAD_ARGUMENT;
    const bool f16 = terms == 0 || terms == 2;
    if (n_views > 32) return VFA_ERR_UNSUPPORTED;
    const WorkspaceLayout lay = layout_of(n_views, L, W, n_scales);
    if (lay.n_tiles == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (n_views == 0) {
        if (!accumulate) return (int)zero_fill(out, (size_t)L * W * kC * sizeof(float), s);
        return 0;
    }
    if (!workspace || workspace_bytes < lay.rows) return VFA_ERR_BAD_ARGUMENT;
    const unsigned char *ws = reinterpret_cast<const unsigned char *>(workspace);
    const int rows_cap = rows_cap_of(lay, workspace_bytes); // (the same figure the records call derived from the same size)
    FusedArgs a;
    for (int k = 0; k < kMaxScales; ++k) {
        const int q = k < n_scales ? k : 0;
        a.sc[k].integral = integrals[q];
        a.sc[k].bias = biases ? biases[q] : nullptr;
        a.sc[k].wfrag = reinterpret_cast<const uint4 *>(ws + lay.wfrag[q]);
        a.sc[k].live = reinterpret_cast<const unsigned *>(ws + lay.live[q]);
        a.sc[k].direct = reinterpret_cast<const unsigned *>(ws + lay.direct[q]);
        a.sc[k].overflow = reinterpret_cast<const unsigned *>(ws + lay.overflow[q]);
        a.sc[k].hdrs = ws + lay.hdrs[q];
        a.sc[k].recs = ws + lay.recs[q];
        a.sc[k].Hf = feat_hw[2 * q];
        a.sc[k].Wf = feat_hw[2 * q + 1];
        if (!a.sc[k].integral) return VFA_ERR_BAD_ARGUMENT;
        a.sc[k].amax = nullptr; a.sc[k].amax_n = 0;
    }
    if (f16 && !(flags & VFA_FLAG_ROWS_ONLY)) {
        // the scale of the fp16 split: what the integral-image call left (feat_absmax), or one pass over the integral images here
        for (int k = 0; k < n_scales; ++k) {
            if (feat_absmax && feat_absmax[k]) {
                a.sc[k].amax = feat_absmax[k];
                a.sc[k].amax_n = (int)feature_stats_count(n_views, kC, a.sc[k].Hf);
            } else {
                unsigned *dst = reinterpret_cast<unsigned *>(const_cast<unsigned char *>(ws) + lay.amax) + (size_t)k * kFallbackStats;
                const int st = integral_absmax_folded(a.sc[k].integral, dst, n_views, kC, a.sc[k].Hf, a.sc[k].Wf, kFallbackStats, &a.sc[k].amax_n, stream);
                if (st) return st;
                a.sc[k].amax = dst;
            }
        }
    }
    a.wexp = reinterpret_cast<const int *>(ws + lay.wexp);
    a.n_scales = n_scales; a.n_views = n_views; a.L = L; a.W = W; a.tiles_w = lay.tiles_w; a.n_tiles = lay.n_tiles;
    a.out = out; a.accumulate = accumulate;
    a.chunk_start = reinterpret_cast<const int *>(ws + lay.chunks);
    a.chunk_rank = reinterpret_cast<const int *>(ws + lay.ranks);
    a.partial = reinterpret_cast<float *>(const_cast<unsigned char *>(ws) + lay.partial);
    a.tickets = reinterpret_cast<unsigned *>(const_cast<unsigned char *>(ws) + lay.tickets);
    a.debug = debug;
    a.diag = reinterpret_cast<unsigned long long *>(const_cast<unsigned char *>(ws) + lay.diag);
    if (debug & kDbgDumpVox) { // (one view, one scale: `out` has room for exactly one set of voxel features; masked tiles stay 0)
        if (n_views != 1 || n_scales != 1 || accumulate || !out) return VFA_ERR_BAD_ARGUMENT;
        const hipError_t e = hipMemsetAsync(out, 0, (size_t)L * W * kC * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
    }
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
    if (nblk > kMaxBlocks) nblk = kMaxBlocks;
    a.rows = reinterpret_cast<const float *>(ws + lay.rows);
    a.row_counter = reinterpret_cast<const unsigned *>(ws + lay.counter);
    a.rows_cap = rows_cap;
    if (!(debug & 64) && !(flags & VFA_FLAG_SKIP_ROWS)) { // pre-pass: pooled rows of the direct items (a block leaves at once where its tile has none)
        RowsArgs ra;
        for (int k = 0; k < kMaxScales; ++k) {
            ra.integral[k] = a.sc[k].integral; ra.recs[k] = a.sc[k].recs;
            ra.Hf[k] = a.sc[k].Hf; ra.Wf[k] = a.sc[k].Wf;
        }
        ra.row_list = reinterpret_cast<const unsigned *>(ws + lay.row_list);
        ra.row_counter = a.row_counter;
        ra.rows = reinterpret_cast<float *>(const_cast<unsigned char *>(ws) + lay.rows);
        ra.n_tiles = lay.n_tiles; ra.rows_cap = rows_cap;
        hipLaunchKernelGGL(pool_rows_kernel, dim3(4096), dim3(512), 0, s, ra); // (a unit per block for up to 1024 direct items: a unit is a chain of dependent round trips)
        const int st0 = (int)hipGetLastError();
        if (st0 || (flags & VFA_FLAG_ROWS_ONLY)) return st0;
    }
    // (the tickets are clear: zeroed by the geometry call, and put back by the last arriver of every earlier launch)
    if (debug & 64) { // diagnostic: only the second launch (direct items without a row slot)
        hipLaunchKernelGGL((pool_collapse_kernel<2, false, true>), dim3(nblk), dim3(kThreads), 0, s, a);
        return (int)hipGetLastError();
    }
    if (debug) // diagnostic build (ablations / cycle stamps) of the default arithmetic: never used by the product path
        hipLaunchKernelGGL((pool_collapse_kernel<2, true, false>), dim3(nblk), dim3(kThreads), 0, s, a);
    else if (terms == 4)
        hipLaunchKernelGGL((pool_collapse_kernel<4, false, false>), dim3(nblk), dim3(kThreads), 0, s, a);
    else if (terms == 3)
        hipLaunchKernelGGL((pool_collapse_kernel<3, false, false>), dim3(nblk), dim3(kThreads), 0, s, a);
    else
        hipLaunchKernelGGL((pool_collapse_kernel<2, false, false>), dim3(nblk), dim3(kThreads), 0, s, a);
    int st = (int)hipGetLastError();
    if (st || debug) return st;
    if ((size_t)rows_cap >= (size_t)n_views * lay.n_tiles * n_scales) return st; // (every item has a row slot: nothing can be left over)
    // the few items whose tap window does not fit LDS: same kernel, taps straight from the image, ADDED to the map (the
    // launch leaves at once where there are none)
    if (terms == 4)
        hipLaunchKernelGGL((pool_collapse_kernel<4, false, true>), dim3(nblk), dim3(kThreads), 0, s, a);
    else if (terms == 3)
        hipLaunchKernelGGL((pool_collapse_kernel<3, false, true>), dim3(nblk), dim3(kThreads), 0, s, a);
    else
        hipLaunchKernelGGL((pool_collapse_kernel<2, false, true>), dim3(nblk), dim3(kThreads), 0, s, a);
    return (int)hipGetLastError();
}

size_t vfa_sliver_shifts_scratch_bytes(int L, int W)
{
    if (L <= 0 || W <= 0) return 0;
    return ((size_t)((L + kTileL - 1) / kTileL) * ((W + kTileW - 1) / kTileW) * sizeof(unsigned) + 15) / 16 * 16;
}

int vfa_sliver_shifts_u8(const float *calibs, const float *grid, const float *z_layers, int n_layers, const float *corner_off, int n_views,
                         int L, int W, int conv_kind, float img_w, float img_h, float cmin, float cmax, int Hf, int Wf, int per_item,
                         unsigned char *shift, void *scratch, size_t scratch_bytes, void *stream)
{
    if (n_views < 0 || L < 0 || W < 0 || n_layers < 1 || Hf <= 0 || Wf <= 0 || conv_kind < 0 || conv_kind > 2 || !shift) return VFA_ERR_BAD_ARGUMENT;
    if (per_item && n_layers != 1) return VFA_ERR_BAD_ARGUMENT; // (the serial kernel's items: single-layer grids)
    if (n_views == 0 || L == 0 || W == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ShiftArgs a;
    a.g = BoxGeom{calibs, grid, z_layers, corner_off, conv_kind, img_w, img_h, cmin, cmax};
    a.n_views = n_views; a.L = L; a.W = W; a.tiles_w = (W + kTileW - 1) / kTileW; a.n_tiles = ((L + kTileL - 1) / kTileL) * a.tiles_w;
    a.nl = n_layers; a.Hf = Hf; a.Wf = Wf; a.per_item = per_item ? 1 : 0;
    a.out = shift; a.tile_max = reinterpret_cast<unsigned *>(scratch);
    if (!per_item) {
        if (!scratch || scratch_bytes < vfa_sliver_shifts_scratch_bytes(L, W)) return VFA_ERR_BAD_ARGUMENT;
        const hipError_t e = zero_fill(scratch, vfa_sliver_shifts_scratch_bytes(L, W), s);
        if (e != hipSuccess) return (int)e;
    }
    const long long pairs = (long long)n_views * a.n_tiles;
    hipLaunchKernelGGL(sliver_shift_kernel, dim3((unsigned)((pairs + 1) / 2)), dim3(kWave), 0, s, a);
    if (!per_item) hipLaunchKernelGGL(sliver_expand_kernel, dim3((unsigned)((L * W + 255) / 256)), dim3(256), 0, s, a.tile_max, shift, L, W, a.tiles_w);
    return (int)hipGetLastError();
}

} // extern "C"
