// vfa_pipe.hip -- the inference hot path for ANY number of z-layers (C = 256) as a producer / consumer pipeline inside one
// persistent workgroup per CU.  Supersedes the serial pool -> multiply kernel of vfa_fused.hip (single-layer grids only).
//
//   reference                                                              here
//   corners, convert, project, clamp, bbox, area, visibility               pipe_records_kernel: once per frame for all scales
//     vfa/model/vfa_op.py:64-88, 104-106; vfa/utils.py:50-59                 and ALL z-layers (box records + tap windows)
//   4 x grid_sample of the integral image, box mean, mask  :112-120        pooling waves 8-15 of pipe_kernel (fp32, the
//                                                                            reference's FMA chains, bit for bit)
//   collapse = Linear(C * nl -> C)                          :50-59, :123   matrix waves 0-7: split-operand MFMA, K = nl * 256
//   relu; f8 + f16 + f32; ortho += ...       :124; vfa/model/vfanet.py:79,82  in registers of the matrix waves
//
// Why this shape.  `relu` follows the sum over the whole K = nl * 256, so the 32 x 256 accumulator of a (tile, view, scale)
// has to stay on the CU while all nl layers are pooled and multiplied, and one layer of `collapse.weight` (256 KiB as two
// 16-bit planes) cannot stay in registers beside it.  A workgroup therefore keeps the accumulators of a GROUP of four views of a
// tile (128 rows) and streams the weight through registers in slices of 64 k x 256 n, each used for all 128 rows: 64 KiB of
// weight traffic per 32 x 256 x 256 product instead of 256.  Pooling (VALU + LDS) and the products (matrix pipe) run
// CONCURRENTLY on different waves: four waves per SIMD -- two matrix waves, two pooling waves, 128 registers each (the
// six-product variant: three per SIMD, 168) -- one barrier per step of 64 rows x 64 channels (vfa_pipe_seq.h has the step
// order).  Tap windows arrive by LDS-DMA one step ahead, issued by the pooling waves; boxes whose window does not fit (right in
// front of a camera) are pooled from L2 by the same pooling code (no pre-pass, no row scratch).  A tile cut between workgroups
// is finished by whichever of them arrives LAST (a ticket per tile: no workgroup ever waits for another; the last arriver
// puts the ticket back to zero).  Frames of one or two views run the four-step phase (template parameter SMALL).
//
// Numerics: pooling = the reference's exact fp32 sequence including the correctly rounded quotient v / area (box_quotient_scaled,
// vfa_geom.h; VFA_FLAG_DUMP_VOX stores the rows this code forms: bit for bit the reference's voxel features); product, default: both operands scaled by a power of two and split into two
// fp16 pieces, three MFMA products hi.lo + hi.hi + lo.hi with fp32 accumulation, k ascending (vfa_split.h: the width of the
// reference's fp32 nn.Linear); VFA_FLAG_TERMS 3 / 4: two bf16 pieces (16-bit operands), 6: three bf16 pieces, six products.  On a
// single-layer grid the product sequence is vfa_fused.hip's; the two kernels differ by the association of the view sum only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "vfa_geom.h"
#include "vfa_pipe_seq.h"
#include "vfa_split.h"

#ifndef VFA_TICKET_ORDER
// The hand-off ticket of a run cut between workgroups (finish_run): a RELAXED agent-scope add.  Every byte handed off is stored `sc1`
// by the wave that signals for it, behind that wave's own vmcnt(0), and read `sc1` by the wave whose add came last, behind the
// return of that add: the guide's measured-valid form (MI355X_MICROARCH.md, inter-workgroup visibility, first row of the table) with
// "workgroup" read as "wave" -- each matrix wave hands off its own 32 columns.  Rounds 3-5 drew ONE acquire-release ticket per
// workgroup; with a ticket per wave that order cost a `buffer_wbl2 sc1` + `buffer_inv sc1` (the XCD's L2 written back, the CU's L1
// invalidated) in each of the eight matrix waves of every workgroup at both ends of its range: 100 K cycles per workgroup on the
// one-camera five-layer frame (613 K -> 5xx K), the whole cost of "finish" there.  -DVFA_TICKET_ORDER=__ATOMIC_ACQ_REL restores it.
#define VFA_TICKET_ORDER __ATOMIC_RELAXED
#endif

namespace {
using namespace vfa_dev;
using namespace vfa_pipe;

constexpr int kTileW = 8, kTileL = 4, kTileBoxes = kTileW * kTileL;
constexpr int kC = 256;
constexpr int kRecBytes = 48, kHdrBytes = 32; // (box record: see pipe_records_kernel)
constexpr int kMaxScales = 3;
constexpr int kSlotBytes = kC * 4;              // one tap in the integral image: 256 fp32
constexpr int kQSlot = 256;                     // ... and the 64-channel quarter of it that a step needs
constexpr int kWinSlots = 112;                  // LDS tap window of a (tile, view, layer, scale), in quarter slots: four of them (two being
                                                // pooled, two arriving); 28 KiB each.  A multiple of 4: a fill instruction brings four
constexpr int kWinSlots3 = 96;                   // ... of the three-piece variant (VFA_FLAG_TERMS 6): a third bf16 plane takes 17 KiB of LDS
// 8 matrix waves + 8 pooling waves (four waves per SIMD, 128 registers); the six-product variant keeps 4 pooling waves (three per
// SIMD, 168 registers: its third weight plane and third fragment do not fit into 128, and it is bound by the matrix pipe anyway)
constexpr int kMatWaves = 8;
#ifndef VFA_PIPE_DMA_ON_MATRIX
#define VFA_PIPE_DMA_ON_MATRIX 0
#endif
// Sixteen waves: which role requests the tap windows (twelve: always the matrix waves).  Measured as an A/B in one process after
// every other change of the round: on the pooling waves 551 / 1 485 / 3 664 us (bench frame, MultiviewC x5, Wildtrack x8), on the
// matrix waves 565 / 1 530 / 3 727.
constexpr bool kDmaOnMatrix = VFA_PIPE_DMA_ON_MATRIX != 0;
constexpr int kGroupRing = 128; // group records in LDS (2 KB); a tile has at most 3 * 8 groups
constexpr int pool_waves_of(int terms) { return terms == 6 ? 4 : 8; }
constexpr int threads_of(int terms) { return 64 * (kMatWaves + pool_waves_of(terms)); }
constexpr int kStepRows = 64;                   // rows of a step: two sub-tiles
// A tile of a step in LDS, per bf16 plane: 8 chunks (16 bytes = 8 k) x 64 rows x 16 bytes, chunk stride padded by 32 bytes
// (the pooling waves' 8-byte stores of neighbouring chunks then fall into different banks)
constexpr int kChunkStride = kStepRows * 16 + 32;
constexpr int kPlaneBytes = 8 * kChunkStride;   // 8448
constexpr int kWPlanes = 3;                     // bf16 planes of the split collapse weight in the workspace: hi, mid (= lo of the two-piece split), lo
constexpr int kSteps = 16;                      // k-steps of v_mfma_f32_32x32x16_bf16 per layer
constexpr int kChunks = 8192, kMaxBlocks = 512;
// Balance state of a workspace (written by vfa_pipe_balance_f32, read by the frame kernel; nothing else touches it):
//   int bounds[kMaxBlocks + 1]   workgroup wg of a launch of `tag` workgroups takes the pieces [bounds[wg], bounds[wg + 1]) of the kChunks
//   int tag                      number of workgroups the bounds are for (0: none -- the uniform split)
//   int sig[2]                   total estimated cost of the frame the bounds were made for (the cuts of another geometry do not match)
//   u64 cycles[kMaxBlocks]       at byte 4096: cycles every workgroup of the LAST launch took
constexpr int kBalTag = kMaxBlocks + 1, kBalSig = kMaxBlocks + 2, kBalCyclesAt = 4096, kBalanceBytes = kBalCyclesAt + kMaxBlocks * 8;
constexpr int kSigAt = kChunks + 1; // the cuts kernel leaves the frame's total cost behind the last entry of chunk_start (2 ints)
constexpr int kVis = 1;
constexpr int kTileLive = 1, kTileDirect = 2;
// VFA_FLAG_DUMP_VOX (diagnostic build): with ONE view, ONE scale and ONE layer `out` receives the pooled fp32 voxel features (cell,
// channel) exactly as the pooling waves form them in front of the operand split, instead of the map (tests/test_pipe_frame.py)
constexpr int kDbgDumpVox = 0x1000; // (set by VFA_FLAG_DUMP_VOX; bits 0-11 are VFA_FLAG_DEBUG's)
// Compile-time ablation of the PRODUCTION kernels (tools/ablate_pipe.sh builds one library per mask; results are then meaningless):
// 1 no window fills, 2 no pooling, 4 no MFMAs, 64 loop + tables + barriers only.  The run-time masks of the diagnostic build
// (VFA_FLAG_DEBUG) time a kernel with other registers and a kernel-argument load per check: 17 % slower before anything is ablated.
#ifndef VFA_PIPE_ABLATE
#define VFA_PIPE_ABLATE 0
#endif
// wave priorities of the two roles (s_setprio): the pooling waves go first (measured: tools/ablate_pipe.sh EXTRA=-DVFA_PIPE_PRIO_...)
#ifndef VFA_PIPE_PRIO_POOL
#define VFA_PIPE_PRIO_POOL 1
#endif
#ifndef VFA_PIPE_PRIO_MAT
#define VFA_PIPE_PRIO_MAT 0
#endif
constexpr int kAblate = VFA_PIPE_ABLATE;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct ScaleDims { int Hf, Wf; };
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }
__device__ __forceinline__ float relu_t(float x) { return (x < 0.0f) ? 0.0f : x; } // NaN stays NaN

// ------------------------------------------------------------------------------------------------
// 1. geometry of the frame: one half-wave per (view, tile), lane = cell of the tile; every z-layer, every scale
//    records / headers of a scale: index ((tile * nl + layer) * n_views + view)
// ------------------------------------------------------------------------------------------------
struct RecordArgs {
    BoxGeom g;
    int n_views, L, W, tiles_w, n_tiles, n_scales, nl, win_slots;
    ScaleDims dims[kMaxScales];
    unsigned *live[kMaxScales];      // (n_tiles) bit v = view v has a live box in the tile, in any layer
    unsigned *globs;                 // (n_tiles) live (view, layer, scale) items whose tap window does not fit LDS
    unsigned *subcost[kMaxScales];   // (n_tiles, n_views) estimated cost of the sub-tile over its layers (vfa_pipe_seq.h: sub_layer_cost)
    unsigned *shift[kMaxScales];     // (n_tiles) sliver shift of (tile, scale): the largest over its views, layers and visible boxes (vfa_geom.h)
    unsigned char *hdrs[kMaxScales];
    unsigned char *recs[kMaxScales];
};

__global__ __launch_bounds__(kWave) void pipe_records_kernel(RecordArgs a)
{
    // a half-wave per (view, tile, LAYER) (round 5, second session: one per (view, tile) walked its layers one after the other -- 15
    // dependent (layer, scale) rounds of shuffles, LDS staging and stores on the 5-layer MultiviewC frame, 113 us beside the integral
    // images with 2.7 waves per SIMD: the geometry stream, not the integral images, decided when the frame kernel could start.  Now
    // 84 us; a half-wave per (view, tile, layer, SCALE) was slower again (107: 41 000 one-wave workgroups), and so was writing the
    // records without the LDS staging (85, and the integral images beside it 93 instead of 84))
    __shared__ uint4 stage[2][kTileBoxes * 3];
    const int lane = threadIdx.x, half = lane >> 5, b = lane & 31;
    const long long unit = (long long)blockIdx.x * 2 + half;
    const long long pair = unit / a.nl;
    const int layer0 = (int)(unit - pair * a.nl);
    const bool pair_ok = pair < (long long)a.n_views * a.n_tiles;
    const int view = pair_ok ? (int)(pair / a.n_tiles) : 0, tile = pair_ok ? (int)(pair % a.n_tiles) : 0;
    const int tl = tile / a.tiles_w, tw = tile - tl * a.tiles_w;
    const int cl = tl * kTileL + (b >> 3), cw = tw * kTileW + (b & 7);
    const bool valid = pair_ok && cl < a.L && cw < a.W;
    const int cell = valid ? cl * a.W + cw : 0;
    const float *P = a.g.calibs + (size_t)view * 12;
    const float g0 = a.g.grid[cell * 3 + 0], g1 = a.g.grid[cell * 3 + 1], g2 = a.g.grid[cell * 3 + 2];
    bool live_any[kMaxScales] = {false, false, false};
    int shift_max[kMaxScales] = {0, 0, 0};
    unsigned n_glob = 0;
    {
        const int layer = layer0;
        // the cube once per (view, cell, layer): scale-independent                     vfa_op.py:64-88, utils.py:56-59
        float l, t, r, bt;
        {
            const float gx = g0 + 0.0f, gy = g1 + 0.0f; // + the int64 zeros of z_corners (vfa_op.py:52, :64)
            const float gz = g2 + a.g.z_layers[layer];
            l = t = r = bt = 0.0f;
#pragma unroll 1
            for (int k = 0; k < 8; ++k) {
                float nu, nv;
                project_corner(a.g, P, gx, gy, gz, k, nu, nv);
                if (k == 0) { l = r = nu; t = bt = nv; }
                else { l = min_t(l, nu); r = max_t(r, nu); t = min_t(t, nv); bt = max_t(bt, nv); }
            }
        }
        const size_t item = ((size_t)tile * a.nl + layer) * a.n_views + view;
#pragma unroll 1
        for (int s = 0; s < a.n_scales; ++s) {
            const int Hf = a.dims[s].Hf, Wf = a.dims[s].Wf;
            const float area = box_area(l, t, r, bt, Hf, Wf);                                     // vfa_op.py:104-105
            const bool vis = valid && box_visible(area, Hf, Wf);                                  // :106
            const float masked = valid ? area * 0.0f : 0.0f; // value of a masked voxel: 0, or NaN when the box itself is NaN
            const bool live_box = vis || (valid && masked != masked);
            const Axis xl = make_axis(l, Wf), xr = make_axis(r, Wf), yt = make_axis(t, Hf), yb = make_axis(bt, Hf);
            // tap coordinates, out-of-image taps redirected to the zero border (coordinate -1 or Hf / Wf)
            const int xs[4] = {clampi(xl.i0, -1, Wf), clampi(xl.i0 + 1, -1, Wf), clampi(xr.i0, -1, Wf), clampi(xr.i0 + 1, -1, Wf)};
            const int ys[4] = {clampi(yt.i0, -1, Hf), clampi(yt.i0 + 1, -1, Hf), clampi(yb.i0, -1, Hf), clampi(yb.i0 + 1, -1, Hf)};
            // window of the tile over its VISIBLE boxes: columns [x0, x1], top rows [t0, t1], bottom rows [b0, b1]
            constexpr int kBig = 1 << 20;
            int x0 = vis ? min(xs[0], xs[2]) : kBig, x1 = vis ? max(xs[1], xs[3]) : -kBig;
            int t0 = vis ? ys[0] : kBig, t1 = vis ? ys[1] : -kBig, b0 = vis ? ys[2] : kBig, b1 = vis ? ys[3] : -kBig;
            int shift = vis ? sliver_shift(area, Hf, Wf) : 0; // (binary places the fp16 split gives up for a noise-dominated box: vfa_geom.h)
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) {
                x0 = min(x0, __shfl_xor(x0, m, 32)); x1 = max(x1, __shfl_xor(x1, m, 32));
                t0 = min(t0, __shfl_xor(t0, m, 32)); t1 = max(t1, __shfl_xor(t1, m, 32));
                b0 = min(b0, __shfl_xor(b0, m, 32)); b1 = max(b1, __shfl_xor(b1, m, 32));
                shift = max(shift, __shfl_xor(shift, m, 32));
            }
            shift_max[s] = max(shift_max[s], shift);
            const unsigned long long vis_all = __ballot(vis), live_all = __ballot(live_box);
            const bool any_vis = ((vis_all >> (32 * half)) & 0xffffffffull) != 0ull;
            const bool any_live = ((live_all >> (32 * half)) & 0xffffffffull) != 0ull;
            int cwid = 0, top_rows = 0, bot_rows = 0, n_slots = 0;
            if (any_vis) {
                cwid = x1 - x0 + 1;
                if (b0 <= t1 + 1) { // the bands touch or overlap: one band [t0, max(t1, b1)]
                    top_rows = max(t1, b1) - t0 + 1;
                    bot_rows = 0;
                    b0 = t0 + top_rows;
                } else {
                    top_rows = t1 - t0 + 1;
                    bot_rows = b1 - b0 + 1;
                }
                n_slots = cwid * (top_rows + bot_rows);
            }
            // pooled straight from the integral image (pixel coordinates in the record): the window does not fit, or the padded
            // image has 2^24 pixels or more (the window fetch of the frame kernel works out its addresses with 24-bit multiplies)
            const bool direct = n_slots > a.win_slots || (long long)(Hf + 2) * (Wf + 2) >= (1ll << 24);
            auto slot_row = [&](int y) { return y < t0 + top_rows ? y - t0 : top_rows + (y - b0); };
            unsigned rows[4], cols[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (direct) { rows[k] = (unsigned)(ys[k] + 1); cols[k] = (unsigned)(xs[k] + 1); }
                else { rows[k] = (unsigned)(slot_row(ys[k]) * cwid); cols[k] = (unsigned)(xs[k] - x0); }
            }
            // A box in 48 bytes: the upper bilinear fractions of its four axes (the sixteen tap weights are products of them and of
            // 1 - them: `unpack` of the frame kernel forms them with the same two rounded operations as `bilinear_weights`), the factor
            // (RN(1 / area), or the masked value), the visibility flag, four row and four column parts.  (96 bytes with the weights
            // spelled out: 19 GB per frame of the 512 x 512 x 32 stress config.)
            if (pair_ok) {
                uint4 *st = stage[half] + b * 3;
                const float rcp = 1.0f / area; // correctly rounded
                st[0] = make_uint4(__float_as_uint(xl.hi), __float_as_uint(xr.hi), __float_as_uint(yt.hi), __float_as_uint(yb.hi));
                st[1] = make_uint4(__float_as_uint(vis ? rcp : masked), vis ? (unsigned)kVis : 0u, rows[0] | (rows[1] << 16), rows[2] | (rows[3] << 16));
                st[2] = make_uint4(cols[0] | (cols[1] << 16), cols[2] | (cols[3] << 16), __float_as_uint(area), 0u);
            }
            __syncthreads(); // (one wave: orders the LDS writes above against the reads below)
            if (pair_ok) {
                uint4 *rec = reinterpret_cast<uint4 *>(a.recs[s] + item * kTileBoxes * kRecBytes);
#pragma unroll
                for (int k = 0; k < 3; ++k) rec[k * 32 + b] = stage[half][k * 32 + b];
                if (b == 0) {
                    uint4 *hdr = reinterpret_cast<uint4 *>(a.hdrs[s] + item * kHdrBytes);
                    const int inv = cwid > 0 ? (65536 + cwid - 1) / cwid : 0; // floor(s / cwid) == (s * inv) >> 16 for s < 128
                    const unsigned hflags = (any_live ? kTileLive : 0) | (direct ? kTileDirect : 0);
                    hdr[0] = make_uint4(hflags, (unsigned)n_slots, (unsigned)cwid, (unsigned)inv);
                    hdr[1] = make_uint4((unsigned)x0, (unsigned)t0, (unsigned)top_rows, (unsigned)b0);
                }
            }
            live_any[s] = live_any[s] || any_live;
            if (any_live && direct) ++n_glob;
            if (pair_ok && b == 0) atomicAdd(a.subcost[s] + (size_t)tile * a.n_views + view, sub_layer_cost(any_live, direct, n_slots)); // (the work cuts' weight)
            __syncthreads(); // the stage is reused
        }
    }
    if (pair_ok && b == 0)
        for (int s = 0; s < a.n_scales; ++s)
            if (live_any[s]) atomicOr(a.live[s] + tile, 1u << view);
    // (the accumulators of a (tile, scale) run over all layers and, inside a group, over its views: one shift for all of them)
    if (pair_ok && b == 0)
        for (int s = 0; s < a.n_scales; ++s)
            if (shift_max[s] > 0) atomicMax(a.shift[s] + tile, (unsigned)shift_max[s]);
    if (pair_ok && b == 0 && n_glob) atomicAdd(a.globs + tile, n_glob);
}

// collapse.weight of a scale, in the REFERENCE layout (256, 256 * nl), column = c * nl + layer (vfa_op.py:59, :120), as THREE bf16
// planes x = p0 + p1 + p2 (+ r, |r| <= 2^-25 |x|; p0 + p1 is the two-piece split) in MFMA B-fragment order, 384 KiB per layer:
//   out[(((layer * 8 + wave) * 16 + s) * 3 + plane) * 64 + lane] (16 B) = W[n = 32 wave + (lane & 31)][c = 16 s + 8 (lane >> 5) + j], j = 0..7
// F16 (VFA_FLAG_TERMS 2, the default): the two-piece fp16 split of vfa_split.h in planes 0 and 1, scaled by 2^ew with
// max|W| 2^ew in [2^14, 2^15) (kWmaxParts partial maxima per scale, left by the spare blocks of pipe_cuts_kernel); ew is left in wexp[scale].
constexpr int kWmaxParts = 32;
struct SplitArgs { const float *w[kMaxScales]; uint4 *out[kMaxScales]; int nl; unsigned *wmax; int *wexp; int f16; };
__global__ __launch_bounds__(256) void pipe_split_weight_kernel(SplitArgs sa)
{
    const int scale = blockIdx.y / sa.nl, layer = blockIdx.y - scale * sa.nl;
    const float *__restrict__ w = sa.w[scale];
    uint4 *__restrict__ out = sa.out[scale] + (size_t)layer * 8 * kSteps * kWPlanes * 64;
    const int idx = blockIdx.x * 256 + threadIdx.x; // (wave, s, lane)
    if (idx >= 8 * kSteps * 64) return;
    const int lane = idx & 63, s = (idx >> 6) & 15, wave = idx >> 10;
    const float *src = w + (size_t)(wave * 32 + (lane & 31)) * kC * sa.nl + (size_t)(16 * s + 8 * (lane >> 5)) * sa.nl + layer;
    if (idx == 0 && blockIdx.y == 0) sa.wexp[kMaxScales] = sa.f16 ? 2 : 3; // the arithmetic these fragments are for: checked by the frame kernel
    if (sa.f16) {
        fp16_saturate_mode(true);
        unsigned m = 0u;
        for (int i = 0; i < kWmaxParts; ++i) m = max(m, sa.wmax[scale * kWmaxParts + i]);
        const int ew = split_exponent(m, kExpW);
        if (idx == 0 && layer == 0) sa.wexp[scale] = ew;
        const float sc = pow2f(ew);
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = src[(size_t)j * sa.nl] * sc;
        uint2 h0, l0, h1, l1;
        split_f16x4(x[0], x[1], x[2], x[3], h0, l0);
        split_f16x4(x[4], x[5], x[6], x[7], h1, l1);
        out[((size_t)(wave * kSteps + s) * kWPlanes + 0) * 64 + lane] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        out[((size_t)(wave * kSteps + s) * kWPlanes + 1) * 64 + lane] = make_uint4(l0.x, l0.y, l1.x, l1.y);
        out[((size_t)(wave * kSteps + s) * kWPlanes + 2) * 64 + lane] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    union { __bf16 b[8]; uint4 u; } p0, p1, p2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = src[(size_t)j * sa.nl];
        p0.b[j] = (__bf16)x;
        const float r1 = x - (float)p0.b[j];
        p1.b[j] = (__bf16)r1;
        p2.b[j] = (__bf16)(r1 - (float)p1.b[j]);
    }
    out[((size_t)(wave * kSteps + s) * kWPlanes + 0) * 64 + lane] = p0.u;
    out[((size_t)(wave * kSteps + s) * kWPlanes + 1) * 64 + lane] = p1.u;
    out[((size_t)(wave * kSteps + s) * kWPlanes + 2) * 64 + lane] = p2.u;
}

// work cuts (vfa_pipe_seq.h: walk_run): one workgroup, an LDS scan over per-thread sums, then every thread places the cuts
// that fall into its RUNS of tiles (chunk_start counts runs, chunk_rank groups of the run)
struct CutArgs {
    const unsigned *live[kMaxScales];
    const unsigned *globs;
    const unsigned *subcost[kMaxScales]; // (n_tiles, n_views): pipe_records_kernel
    int n_scales, n_tiles, n_views, nl, rt; // rt: tiles of a run (vfa_pipe_seq.h: run_tiles_of)
    int *chunk_start, *chunk_rank;
    unsigned long long *chunk_cost; // (kChunks + 1): estimated cost of everything in front of the group boundary a piece starts at
    SplitArgs split;                // wmax_count > 0: the blocks behind block 0 leave the partial maxima of |W| (they need nothing of the frame)
    long long wmax_count;           // weights per scale
    int staged;                     // 1: the launch has LDS for the masks and sub-tile costs of the whole frame (n_scales * n_tiles * (1 + n_views) words)
};
#ifdef VFA_CUTS_STAMPS
__device__ unsigned long long g_cut_stamps[16];
#define CUT_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_cut_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define CUT_STAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(1024) void pipe_cuts_kernel(CutArgs a)
{
    __shared__ unsigned long long part[1024];
    CUT_STAMP(0);
    if (blockIdx.x > 0) {
        // the partial maxima of max|W| (block 1 + scale * kWmaxParts + part), riding in this launch: as a launch of its own behind the
        // cuts (rounds 3-4) it was 13 us of the geometry stream's tail
        const int scale = ((int)blockIdx.x - 1) / kWmaxParts, pi = ((int)blockIdx.x - 1) % kWmaxParts;
        const float *__restrict__ w = a.split.w[scale];
        unsigned m = 0u;
        for (long long i = (long long)pi * 1024 + threadIdx.x; i < a.wmax_count; i += (long long)kWmaxParts * 1024) m = max(m, __float_as_uint(w[i]) & 0x7fffffffu);
        m = wave_max_u32(m);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned mm = 0u;
            for (int i = 0; i < 16; ++i) mm = max(mm, (unsigned)part[i]);
            a.split.wmax[scale * kWmaxParts + pi] = mm;
        }
        return;
    }
    const int tid = threadIdx.x, n_tiles = a.n_tiles, rt = a.rt, n_runs = runs_of(n_tiles, rt);
    const unsigned view_mask = a.n_views >= 32 ? 0xffffffffu : ((1u << a.n_views) - 1u);
    // The masks and the sub-tile costs of the frame come into LDS first, all threads together, coalesced.  A thread walks its runs
    // three times and asks for a mask or a cost where the walk needs it: out of global memory that is a chain of ~100 dependent
    // round trips per run of four tiles -- 105 us for the Wildtrack frame, 54 for MultiviewX, 29 for 156 x 156 x 5 (round 6, when the
    // sub-tile costs arrived; the masks alone: 18).  Frames whose tables do not fit (512 x 512 x 32: 221 KB per band) walk global memory.
    extern __shared__ unsigned cut_stage[]; // [scale][tile] masks, then [scale][tile][view] costs
    const size_t cost_base = (size_t)a.n_scales * n_tiles;
    if (a.staged) {
        for (int s2 = 0; s2 < a.n_scales; ++s2) { // (a loop per table: one flat loop with the table picked per word measured slower)
            const unsigned *lv = s2 == 0 ? a.live[0] : (s2 == 1 ? a.live[1] : a.live[2]);
            const unsigned *sc = s2 == 0 ? a.subcost[0] : (s2 == 1 ? a.subcost[1] : a.subcost[2]);
            for (int t = tid; t < n_tiles; t += 1024) cut_stage[(size_t)s2 * n_tiles + t] = lv[t] & view_mask;
            const int n_cost = n_tiles * a.n_views;
            for (int i = tid; i < n_cost; i += 1024) cut_stage[cost_base + (size_t)s2 * n_cost + i] = sc[i];
        }
        __syncthreads();
    }
    CUT_STAMP(1);
    auto mask_of = [&](int r) {
        return [&, r](int s2, int off) -> unsigned {
            const int t = r * rt + off;
            if (t >= n_tiles) return 0u;
            if (a.staged) return cut_stage[(size_t)s2 * n_tiles + t];
            const unsigned *lv = s2 == 0 ? a.live[0] : (s2 == 1 ? a.live[1] : a.live[2]);
            return lv[t] & view_mask;
        };
    };
    auto cost_of = [&](int r) {
        return [&, r](int s2, int off, int v) -> unsigned {
            if (a.staged) return cut_stage[cost_base + ((size_t)s2 * n_tiles + (r * rt + off)) * a.n_views + v];
            const unsigned *sc = s2 == 0 ? a.subcost[0] : (s2 == 1 ? a.subcost[1] : a.subcost[2]);
            return sc[(size_t)(r * rt + off) * a.n_views + v];
        };
    };
    // entries = (run, scale) in the kernel's order; a thread takes a contiguous range of them
    const int n_entries = n_runs * a.n_scales, per_e = (n_entries + 1023) / 1024, e0 = min(n_entries, tid * per_e), e1 = min(n_entries, e0 + per_e);
    struct RunInfo { int groups_before, n_groups; bool first_of_run; };
    int info_run = -1, info_groups[kMaxScales] = {0, 0, 0}; // (the entries of a thread follow each other: mostly the same run)
    auto run_info = [&](int r, int s2) { // what entry (r, s2) needs to know of the other scales of its run: their group counts
        if (r != info_run) {
            info_run = r;
#pragma unroll
            for (int s3 = 0; s3 < kMaxScales; ++s3) {
                int cnt = 0;
                if (s3 < a.n_scales)
                    for (int off = 0; off < rt; ++off) cnt += seq_popc(mask_of(r)(s3, off));
                info_groups[s3] = groups_of_count(cnt);
            }
        }
        RunInfo ri = {0, 0, true};
#pragma unroll
        for (int s3 = 0; s3 < kMaxScales; ++s3) {
            const int g = info_groups[s3];
            if (s3 < s2) { ri.groups_before += g; if (g) ri.first_of_run = false; }
            ri.n_groups += g;
        }
        return ri;
    };
    unsigned long long local = 0;
    for (int e = e0; e < e1; ++e) {
        const int r = e / a.n_scales, s2 = e - r * a.n_scales, tiles = min(rt, n_tiles - r * rt);
        const RunInfo ri = run_info(r, s2);
        if (ri.n_groups == 0) local += s2 == 0 ? kEmptyCost * (unsigned)tiles : 0u; // (a run without items: its cost sits in its first entry)
        else local += walk_scale(rt, a.nl, tiles, s2, ri.first_of_run, mask_of(r), cost_of(r), [](int, unsigned, unsigned) {});
    }
    part[tid] = local;
    CUT_STAMP(2);
    for (int c = tid; c <= kChunks; c += 1024) { a.chunk_start[c] = n_runs; a.chunk_rank[c] = 0; a.chunk_cost[c] = ~0ull; }
    __syncthreads();
    CUT_STAMP(3);
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned long long v = tid >= d ? part[tid - d] : 0ull;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const unsigned long long total = part[1023];
    CUT_STAMP(4);
    if (tid == 0) { a.chunk_start[kSigAt] = (int)(unsigned)total; a.chunk_start[kSigAt + 1] = (int)(unsigned)(total >> 32); }
    // (pieces behind the last group keep chunk_cost = ~0: the balance kernel reads them as `total`)
    unsigned long long before = part[tid] - local;
    auto pos_of = [&](long long cc) { return (total * (unsigned long long)cc + kChunks - 1) / kChunks; };
    for (int e = e0; e < e1; ++e) {
        const int r = e / a.n_scales, s2 = e - r * a.n_scales, tiles = min(rt, n_tiles - r * rt);
        const RunInfo ri = run_info(r, s2);
        const unsigned long long tb = before; // cost of everything in front of this entry
        long long c = tb > 0 ? (long long)((tb - 1) * kChunks / total) : 0;
        while (c < kChunks && pos_of(c) < tb) ++c;
        unsigned w = 0;
        if (ri.n_groups == 0) {
            w = s2 == 0 ? kEmptyCost * (unsigned)tiles : 0u;
            for (; c < kChunks && pos_of(c) < tb + w; ++c) { a.chunk_start[c] = r; a.chunk_rank[c] = 0; a.chunk_cost[c] = tb; }
        } else {
            w = walk_scale(rt, a.nl, tiles, s2, ri.first_of_run, mask_of(r), cost_of(r), [&](int j, unsigned w0, unsigned w1) {
                const int kk = ri.groups_before + j; // rank of the group in its run
                while (c < kChunks) {
                    const unsigned long long pc = pos_of(c);
                    if (pc >= tb + w1) break;
                    const int k = ((unsigned)(pc - tb) - w0) * 2 < (w1 - w0) ? kk : kk + 1;
                    if (k >= ri.n_groups) { a.chunk_start[c] = r + 1 < n_runs ? r + 1 : n_runs; a.chunk_rank[c] = 0; }
                    else { a.chunk_start[c] = r; a.chunk_rank[c] = k; }
                    a.chunk_cost[c] = tb + (k == kk ? w0 : w1);
                    ++c;
                }
            });
        }
        before += w;
    }
    CUT_STAMP(5);
    __syncthreads();
    CUT_STAMP(6);
}

// ------------------------------------------------------------------------------------------------
// 2. the frame kernel
// ------------------------------------------------------------------------------------------------
struct PipeScale {
    const float *integral;          // (n_views, Hf+2, Wf+2, 256) zero-bordered channels-last
    const float *bias;              // (256) or NULL
    const uint4 *wfrag;             // pipe_split_weight_kernel output
    const unsigned *live;           // (n_tiles)
    const unsigned *shift;          // (n_tiles) sliver shift of (tile, scale): pipe_records_kernel
    const unsigned char *hdrs, *recs;
    int Hf, Wf;
    const unsigned *amax;           // amax_n partial maxima of |feature| (fp32 bits): the scale of the fp16 split (vfa_split.h)
    int amax_n;
};
struct PipeArgs {
    PipeScale sc[kMaxScales];
    int n_scales, n_views, nl, L, W, tiles_w, n_tiles, rt; // rt: tiles of a run (vfa_pipe_seq.h: run_tiles_of -- the cuts were made for the same)
    float *out;                     // (L * W, 256)
    const int *chunk_start, *chunk_rank;
    float *partial;                 // (kMaxBlocks, 2, kRunTiles) x 8 waves x 16 registers x 64 lanes: a workgroup's sums of the tiles of a run it shares (first / last run)
    float *slots;                   // (blocks, kRunTiles, kMaxScales, n_contrib) x 8 waves x 16 x 64: the contributions to the tiles of the run a workgroup is in (private to it)
    int n_contrib;                  // contributions a (tile, scale) can get from one workgroup (vfa_pipe_seq.h: contributions_of)
    unsigned *tickets;              // (runs, 8 matrix waves): parts of a shared run that have arrived, per wave (zero between launches: the geometry call, then the last arriver)
    int accumulate;
    unsigned long long *diag;       // per workgroup 8 counters (VFA_FLAG_DEBUG)
    int debug;
    int *balance;                   // balance state of the workspace (kBalanceBytes: see kBalTag)
    const int *wexp;                // (kMaxScales) scale exponent of the split collapse weight (fp16 form); [kMaxScales]: 2 = fp16 fragments, 3 = bf16
};

struct DevMasks {
    const unsigned *l0, *l1, *l2; // (no array: a dynamically indexed member would put the whole sequencer into scratch memory)
    unsigned view_mask;
    __device__ __forceinline__ unsigned operator()(int s, int t) const
    {
        // scalar loads (constant address space: the geometry kernels finished before this launch)
        const size_t p = (s == 0 ? (size_t)l0 : 0) | (s == 1 ? (size_t)l1 : 0) | (s == 2 ? (size_t)l2 : 0);
        return *reinterpret_cast<const __attribute__((address_space(4))) unsigned *>(p + (size_t)t * 4) & view_mask;
    }
};

struct Frag { bf16x8 hi, lo; };
struct Frag3 { bf16x8 hi, lo, lo2; };

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

// Tap reads of the pooling waves, by hand: four 16-byte taps (one corner of the box) per instruction group, counted waits.  Eight
// taps are in flight at most -- a corner's registers take the corner after the next once it is consumed --: 32 tap registers
// instead of 64 is what lets two boxes per lane live at 128 registers (four waves per SIMD).  LDS
// operations return in order; a scalar load the compiler puts in between only makes a counted wait more conservative.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_read4(f32x4 &t0, f32x4 &t1, f32x4 &t2, f32x4 &t3, unsigned p0, unsigned p1, unsigned p2, unsigned p3)
{
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7"
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait4(f32x4 &t0, f32x4 &t1, f32x4 &t2, f32x4 &t3)
{
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "n"(N));
}
// the same for a box pooled straight from the integral image in L2: 32-bit byte offsets from the image address in scalar registers
// (`s_nop 4`: the image address reaches its scalar registers by v_readfirstlane, and on gfx9 a VMEM instruction that reads an SGPR a
// VALU instruction has just written needs five wait states in between.  The compiler inserts them for its own instructions but
// cannot see into an asm statement: scheduled four instructions behind the v_readfirstlane, these loads went out with the OLD low
// half of the address -- a memory fault that came and went with unrelated edits, in the optimised build only.)
__device__ __forceinline__ void glob_read4(f32x4 &t0, f32x4 &t1, f32x4 &t2, f32x4 &t3, const char *img, unsigned p0, unsigned p1, unsigned p2, unsigned p3)
{
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %8\n\tglobal_load_dwordx4 %1, %5, %8\n\tglobal_load_dwordx4 %2, %6, %8\n\tglobal_load_dwordx4 %3, %7, %8"
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "s"(img) : "memory");
}
template <int N>
__device__ __forceinline__ void glob_wait4(f32x4 &t0, f32x4 &t1, f32x4 &t2, f32x4 &t3)
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "n"(N));
}
__device__ __forceinline__ float4 as_float4(f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); }

// box of one pooling lane for a whole layer: the 16 rounded tap weights, the scale (RN(1 / area), or the masked value) and the
// tap positions as 4 row + 4 column byte offsets (tap (i, k) sits at rowb[i] + colb[k] inside the tap window, or inside the
// view's integral image for a direct item): eight registers instead of sixteen -- two boxes per lane must fit beside sixteen
// taps in flight at 128 registers
struct LaneBox {
    float wt[16];
    float scl, asc; // RN(1 / area) 2^k (or the masked value) and area 2^-k: box_quotient_scaled (vfa_geom.h); k = ea - shift (fp16 form), else 0
    float back;     // 2^-k (diagnostic dump only)
    unsigned rowb[4], colb[4];
};

// A fragments of one k-step for the two row blocks of a step (hi / lo planes): immediate offsets from one LDS address.
// Assembly: the reads run one k-step ahead of the MFMAs with counted waits, and the compiler must not order them behind the
// LDS-DMA the wave has in flight (it would drain it in front of every LDS read it knows of).
template <int KS>
__device__ __forceinline__ void read_frags(unsigned pa, bf16x8 &h0, bf16x8 &l0, bf16x8 &h1, bf16x8 &l1)
{
    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\t"
                 "ds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                 : "=&v"(h0), "=&v"(l0), "=&v"(h1), "=&v"(l1)
                 : "v"(pa), "n"(KS * 2 * kChunkStride), "n"(KS * 2 * kChunkStride + kPlaneBytes), "n"(KS * 2 * kChunkStride + 512),
                   "n"(KS * 2 * kChunkStride + 512 + kPlaneBytes)
                 : "memory");
}
// one plane (0 = hi, 1 = lo) of both row blocks of a k-step, and the counted wait for a pair (LDS reads return in order)
template <int KS, int PLANE>
__device__ __forceinline__ void read_pair(unsigned pa, bf16x8 &f0, bf16x8 &f1)
{
    if constexpr ((VFA_PIPE_ABLATE & 2048) != 0) { // (ablation: the products without their fragment reads)
        asm volatile("" : "+v"(f0), "+v"(f1) : "v"(pa));
        return;
    }
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                 : "=&v"(f0), "=&v"(f1)
                 : "v"(pa), "n"(KS * 2 * kChunkStride + PLANE * kPlaneBytes), "n"(KS * 2 * kChunkStride + 512 + PLANE * kPlaneBytes)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void wait_pair(bf16x8 &f0, bf16x8 &f1)
{
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f0), "+v"(f1) : "n"(N));
}
template <int N>
__device__ __forceinline__ void wait_frags(bf16x8 &h0, bf16x8 &l0, bf16x8 &h1, bf16x8 &l1)
{
    if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(h0), "+v"(l0), "+v"(h1), "+v"(l1));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0), "+v"(l0), "+v"(h1), "+v"(l1));
}

// three planes of ONE row block (the three-piece variant keeps 48 weight registers: both row blocks' fragments would not fit)
template <int KS, int RB, int PLANE_BYTES>
__device__ __forceinline__ void read_frags3(unsigned pa, bf16x8 &p0, bf16x8 &p1, bf16x8 &p2)
{
    asm volatile("ds_read_b128 %0, %3 offset:%4\n\tds_read_b128 %1, %3 offset:%5\n\tds_read_b128 %2, %3 offset:%6\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(p0), "=&v"(p1), "=&v"(p2)
                 : "v"(pa), "n"(KS * 2 * kChunkStride + RB * 512), "n"(KS * 2 * kChunkStride + RB * 512 + PLANE_BYTES),
                   "n"(KS * 2 * kChunkStride + RB * 512 + 2 * PLANE_BYTES)
                 : "memory");
}

// SMALL: a frame of at most TWO views (a rank's share of a camera-sharded rig): no group has a second set, so a phase is FOUR steps
// -- one per channel quarter, all in the registers of set 0 -- instead of eight of which every other one is empty (a barrier, and the
// exposed landing of the next step's windows: ~3 100 cycles).  The LDS buffers alternate by quarter, the weight slice is reloaded
// behind every step.  Same arithmetic in the same order as the eight-step form.
// RT1: runs of ONE tile with three or more views (the order of rounds 3-5).  All sub-tiles of a group belong to one tile and a tile
// is finished before the next begins, so the tile's sum can do what it did then: wait in the workspace between the groups (one
// 32-row x 256-column slot per workgroup), come back into the registers of acc[1] in front of the group's LAST step -- set 0 is
// complete there and r0 + r1 has moved into acc[0] -- and leave behind it; nothing is read back when the tile ends.  A template
// parameter, not a branch on a.rt: the contribution code of the longer runs in the same loop body cost the step loop 3-5 %.
// RTC: the run length as a compile-time fact (2 or 4; 0 = read from the arguments): the generator's loops over the tiles of a run
// unroll and its state stays out of the pooling loops' registers (the tile-by-tile templates gained 2 % from the same).
template <int TERMS, bool DIAG, bool SMALL = false, bool RT1 = false, int RTC = 0>
__global__ __launch_bounds__(threads_of(TERMS)) void pipe_kernel(PipeArgs a)
{
    static_assert(!SMALL || TERMS != 6, "the four-step phase exists in the sixteen-wave layout only");
    static_assert(!RT1 || (TERMS != 6 && !SMALL), "RT1: the sixteen-wave layout's eight-step phase");
    constexpr int kPS = SMALL ? 4 : 8, kPSh = SMALL ? 2 : 3; // steps of a phase; step i: phase i >> kPSh, position i & (kPS - 1)
    auto quarter_of = [](int k) { return SMALL ? k : k >> 1; }; // position k of a phase -> channel quarter (the LDS parity is k & 1 either way)
    // separate objects: one per role of the data (hipcc orders LDS-DMA against every LDS access it cannot prove disjoint)
    constexpr int kPieces = TERMS == 6 ? 3 : 2;                             // 16-bit pieces of an operand
    constexpr bool F16 = TERMS == 2;                                        // ... fp16 pieces with a scale (vfa_split.h) instead of bf16 ones
    constexpr int kWinBytes = (TERMS == 6 ? kWinSlots3 : kWinSlots) * kQSlot;
    __shared__ __align__(16) unsigned char s_win[4 * kWinBytes];            // tap windows: [step parity][sub-tile of the set]
    __shared__ __align__(16) unsigned char s_planes[2 * kPieces * kPlaneBytes]; // A tiles: [step parity][piece]
    // box records of the group's sub-tiles, one layer: one OBJECT per set (sub-tiles 0, 1 / 2, 3) -- while the pooling waves read the
    // boxes of set 0 at the first step of a phase, the records of set 1 arrive by DMA, and the compiler drains every DMA in front of
    // an LDS read of an object the DMA may write
    __shared__ __align__(16) unsigned char s_rec0[2 * kTileBoxes * kRecBytes];
    __shared__ __align__(16) unsigned char s_rec1[2 * kTileBoxes * kRecBytes];
    __shared__ __align__(16) unsigned s_hdr[4][64];                         // headers of the phases in flight: [phase & 3][sub-tile][8]
    // per scale: integral, recs, wfrag, hdrs (lo, hi each), Hf, Wf | (ea + 128) << 16, nl * n_views, n_views; fp16 form: 2^(ea+ew), 2^-(ea+ew)
    __shared__ __align__(16) unsigned s_sc[kMaxScales][16];
    __shared__ unsigned s_amax[kMaxScales];
    // phase records (table wave): [phase & 3]{tile, views, w, -, then the 12 constants of the phase's scale (s_sc)}: whoever builds
    // tables from a phase record finds everything behind ONE LDS round trip (under the pooling waves' read traffic a dependent chain of
    // three -- record, scale constants, headers -- took make_desc 3 500 cycles)
    __shared__ __align__(16) unsigned s_phase[4][16];
    __shared__ __align__(16) unsigned s_groups[kGroupRing][4];              // group records (wave 0), a ring: {tile, views, scale | nj << 15 | more << 20, -}
    __shared__ __align__(16) unsigned s_desc[4][8][2][8];                   // fetch descriptors (wave 0): [phase & 3][step of the phase][sub-tile]
    const int tid = threadIdx.x, wave = uniform_i(tid >> 6), lane = tid & 63;

    const int nblk = gridDim.x;
    // The weight fragments in the workspace were split for ONE arithmetic (the geometry call's flags); a launch that asks for the other
    // would read fp16 pieces as bf16 ones: fail loudly -- a map of NaNs -- instead of returning plausible garbage.
    if (uniform_i(a.wexp[kMaxScales]) != (F16 ? 2 : 3)) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + tid; i < (size_t)a.L * a.W * kC; i += (size_t)nblk * blockDim.x)
            a.out[i] = __uint_as_float(0x7fc00000u);
        return;
    }
    const int lb = (int)xcd_contiguous(blockIdx.x, (nblk + 7) / 8);
    if (lb >= nblk) return;
    // the pieces of workgroup wg: the uniform split, or the bounds a balance call left for exactly this frame and launch size
    const long long t_start = __builtin_amdgcn_s_memtime();
    const bool balanced = uniform_i(a.balance[kBalTag]) == nblk && uniform_i(a.balance[kBalSig]) == uniform_i(a.chunk_start[kSigAt]) &&
                          uniform_i(a.balance[kBalSig + 1]) == uniform_i(a.chunk_start[kSigAt + 1]);
    auto range_of = [&](int wg, int &tb, int &kb, int &te, int &ke) {
        int c0 = (int)((long long)kChunks * wg / nblk), c1 = (int)((long long)kChunks * (wg + 1) / nblk);
        if (balanced) { c0 = uniform_i(a.balance[wg]); c1 = uniform_i(a.balance[wg + 1]); }
        tb = uniform_i(a.chunk_start[c0]); kb = uniform_i(a.chunk_rank[c0]);
        te = uniform_i(a.chunk_start[c1]); ke = uniform_i(a.chunk_rank[c1]);
    };
    const int rt = (RT1 || SMALL) ? 1 : (RTC ? RTC : a.rt); // tiles of a run: 1, 2 or 4 (a compile-time 1 in the tile-by-tile templates: their generator folds to one tile)
    int r_begin, k_begin, r_end, k_end; // runs of rt tiles, groups of a run (vfa_pipe_seq.h)
    range_of(lb, r_begin, k_begin, r_end, k_end);
    unsigned long long *wg_cycles = reinterpret_cast<unsigned long long *>(reinterpret_cast<unsigned char *>(a.balance) + kBalCyclesAt) + lb;
    if (r_begin > r_end || (r_begin == r_end && k_begin >= k_end)) {
        if (tid == 0) *wg_cycles = 0ull;
        return;
    }
    const unsigned view_mask = a.n_views >= 32 ? 0xffffffffu : ((1u << a.n_views) - 1u);
    DevMasks masks;
    masks.l0 = a.sc[0].live; masks.l1 = a.sc[1].live; masks.l2 = a.sc[2].live;
    masks.view_mask = view_mask;

    // a run is SHARED when another workgroup holds groups of it too
    auto shared_run = [&](int run) { return (run == r_begin && k_begin > 0) || (run == r_end && k_end > 0); };

    // the workgroups that hold groups of `run` beside this one: first, last, how many (this one included)
    auto share_of = [&](int run, int &first, int &last, int &parts) {
        first = lb; last = lb; parts = 1;
        for (int j = lb - 1; j >= 0; --j) {
            int tb, kb, te, ke;
            range_of(j, tb, kb, te, ke);
            if (te < run || (te == run && ke == 0)) break; // ends in front of the run
            if (tb > te || (tb == te && kb >= ke)) continue; // (empty range)
            first = j; ++parts;
        }
        for (int j = lb + 1; j < nblk; ++j) {
            int tb, kb, te, ke;
            range_of(j, tb, kb, te, ke);
            if (tb > run) break;
            if (tb > te || (tb == te && kb >= ke)) continue;
            last = j; ++parts;
        }
    };
    int sh_b_first = lb, sh_b_last = lb, sh_b_parts = 1, sh_e_first = lb, sh_e_last = lb, sh_e_parts = 1;
    if (k_begin > 0) share_of(r_begin, sh_b_first, sh_b_last, sh_b_parts);
    if (k_end > 0) share_of(r_end, sh_e_first, sh_e_last, sh_e_parts);

    // The loop below exists twice, once per role (`POOL`): a wave never changes its role, and inside ONE loop the registers of
    // both roles would be live at once.  Both copies take the same steps, hence the same barriers.
    // Per-scale constants in LDS: the kernel arguments live in memory (the scalar registers are full), and a table job that picks
    // one of three pointers by scale became a chain of dependent scalar loads -- make_desc alone took ~4 000 cycles on the wave
    // every other wave waits for.  One vector read of this table instead.
    if constexpr (F16) { // the largest |feature| of every scale (what the integral-image kernels saw): the scale 2^ea of its voxel features
        if (tid < kMaxScales) s_amax[tid] = 0u;
        __syncthreads();
        for (int s2 = 0; s2 < a.n_scales; ++s2) {
            unsigned m = 0u;
            for (int i = tid; i < a.sc[s2].amax_n; i += threads_of(TERMS)) m = max(m, a.sc[s2].amax[i]);
            m = wave_max_u32(m);
            if (lane == 0) atomicMax(&s_amax[s2], m);
        }
        __syncthreads();
    }
    if (tid < kMaxScales) {
        const int s3 = tid < a.n_scales ? tid : 0;
        unsigned *d = &s_sc[tid][0];
        const unsigned long long p0 = (unsigned long long)(size_t)a.sc[s3].integral, p1 = (unsigned long long)(size_t)a.sc[s3].recs,
                                 p2 = (unsigned long long)(size_t)a.sc[s3].wfrag, p3 = (unsigned long long)(size_t)a.sc[s3].hdrs;
        d[0] = (unsigned)p0; d[1] = (unsigned)(p0 >> 32); d[2] = (unsigned)p1; d[3] = (unsigned)(p1 >> 32);
        d[4] = (unsigned)p2; d[5] = (unsigned)(p2 >> 32); d[6] = (unsigned)p3; d[7] = (unsigned)(p3 >> 32);
        d[8] = (unsigned)a.sc[s3].Hf; d[9] = (unsigned)a.sc[s3].Wf; d[10] = (unsigned)(a.nl * a.n_views); d[11] = (unsigned)a.n_views;
        d[12] = d[13] = pow2_bits(0);
        if constexpr (F16) {
            const int ea = split_exponent(s_amax[s3], kExpA), ew = a.wexp[s3];
            d[9] |= (unsigned)(ea + 128) << 16; // (Wf <= 65533: vfa_pipe_boxes_f32)
            d[12] = pow2_bits(ea + ew); d[13] = pow2_bits(-(ea + ew));
        }
    }
    __syncthreads();
    auto run = [&](auto role_tag) {
        constexpr bool POOL = decltype(role_tag)::value;
        const int r = lane & 31, h = lane >> 5;           // matrix waves: row / column of the 32 x 32 block, k half
        const int pw = wave - kMatWaves;                  // pooling waves: 0..7
        constexpr bool W16 = pool_waves_of(TERMS) == 8;
        // ... sub-tile of the set, its upper / lower 16 boxes, and -- eight pooling waves: two waves per 16 boxes -- which of the
        // four 16-byte pieces: m = mpar, mpar + mstep, ...
        const int px = W16 ? pw >> 2 : pw >> 1, phalf = W16 ? (pw >> 1) & 1 : pw & 1, mpar = W16 ? pw & 1 : 0;
        constexpr int mstep = W16 ? 2 : 1, mcount = W16 ? 2 : 4;
        const bool dma_matrix = !W16 || kDmaOnMatrix;
        const int dw = dma_matrix ? wave : pw;            // the waves that fetch (step_dma): 0..7
        // The wave that runs the generator and fills the tables: matrix wave 0.  (On the last pooling wave -- they wait ~1 500 cycles
        // per step at the barrier -- the generator's scalar state no longer fits the scalar registers; what is spilled from them takes
        // vector registers of BOTH roles and the matrix loop spills: 23 scratch operations per step.)
        // (sixteen waves: a POOLING wave -- they wait 600-3 000 cycles at every step barrier --, and of those the fourth: the
        // older pooling wave of its SIMD wins the issue conflicts with the younger one (waves 12-15) and arrives 150-250 cycles
        // per step earlier; on the last wave, the latest of all, the tables delayed every step they ran in)
        const bool table_wave = W16 ? wave == kMatWaves + 3 : wave == 0;
        const int pb = lane >> 2, pi = lane & 3;          // ... box 0..15 of the wave's half sub-tile, 16-byte piece 0..3

        // ---------------------------------------------------------------- matrix-wave state
        f32x16 acc[4];
        Frag3 wq[4]; // (lo2: the three-piece variant only)
        float bc[kMaxScales];
        if constexpr (!POOL) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
#pragma unroll
            for (int s = 0; s < kMaxScales; ++s) {
                bc[s] = (s < a.n_scales && a.sc[s].bias) ? a.sc[s].bias[wave * 32 + r] : 0.0f;
                if constexpr (F16) bc[s] *= __uint_as_float(s_sc[s][12]); // the bias rides in the accumulator: in its units, 2^(ea+ew)
            }
        }
        // 2^-(ea+ew) of a scale (1 in the bf16 forms): what turns accumulator units back into the map's
        auto inv_of = [&](int s) { return F16 ? __uint_as_float((unsigned)uniform_i((int)s_sc[s][13])) : 1.0f; };
        // ---------------------------------------------------------------- pooling-wave state: the wave's 16 boxes of each set
        LaneBox boxA, boxB;
        bool globA = false, globB = false, liveA = false, liveB = false;

        // A phase record (LDS, written by the table wave): the run, the sub-tiles of the group (vfa_pipe_seq.h: sub_byte -- view and tile
        // offset inside the run, one byte each), w = scale | layer << 2 | nj << 15 | more << 20, the sliver shifts of the sub-tiles' (tile,
        // scale) (one byte each)
        struct PhaseRec {
            int run; unsigned subs, w, sh;
            __device__ __forceinline__ bool valid() const { return run >= 0; }
            __device__ __forceinline__ int shift(int j) const { return (int)((sh >> (8 * j)) & 0xffu); } // fp16 form only, else 0
            __device__ __forceinline__ int scale() const { return (int)(w & 3u); }
            __device__ __forceinline__ int layer() const { return (int)((w >> 2) & 1023u); }
            __device__ __forceinline__ int nj() const { return (int)((w >> 15) & 7u); }
            __device__ __forceinline__ bool more() const { return (w >> 20) & 1u; } // another group of this workgroup follows in the run
            __device__ __forceinline__ int ci() const { return (int)((w >> 21) & 15u); } // contribution index of the head tile (vfa_pipe_seq.h)
            __device__ __forceinline__ int view(int j) const { return sub_view(subs, j); }
            __device__ __forceinline__ int tile(int j, int rt_) const { return run * rt_ + sub_tile_off(subs, j); }
        };
        auto phase_rec = [&](int n) {
            const uint4 v = *reinterpret_cast<const uint4 *>(&s_phase[n & 3][0]);
            PhaseRec p;
            p.run = uniform_i((int)v.x); p.subs = (unsigned)uniform_i((int)v.y); p.w = (unsigned)uniform_i((int)v.z);
            p.sh = F16 ? (unsigned)uniform_i((int)v.w) : 0u;
            return p;
        };
        // output rows of a tile: register i of lane (r, h) is row (i & 3) + 8 (i >> 2) + 4 h of the 32 x 32 block, column r
        auto write_tile = [&](int tile, const f32x16 &v, bool have) __attribute__((always_inline)) {
            if (DIAG && (a.debug & kDbgDumpVox)) return; // (the output buffer holds the dumped voxel features)
            const int tl = tile / a.tiles_w, tw = tile - tl * a.tiles_w;
            float extra = 0.0f; // fully masked (view, scale) of this tile: vox = 0 -> relu(bias)
#pragma unroll
            for (int s = 0; s < kMaxScales; ++s)
                if (s < a.n_scales) extra += (float)(a.n_views - __popc(masks(s, tile))) * (F16 ? relu_t(bc[s]) * inv_of(s) : relu_t(bc[s]));
            int h2 = h, r2 = r;
            asm volatile("" : "+v"(h2), "+v"(r2)); // (keeps the 16 row offsets out of long-lived registers)
            float *ocol = a.out + wave * 32 + r2;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * h2;
                const int cl = tl * kTileL + (row >> 3), cw = tw * kTileW + (row & 7);
                if (cl < a.L && cw < a.W) {
                    float *o = ocol + (size_t)(cl * a.W + cw) * kC;
                    float x = (have ? v[i] : 0.0f) + extra;
                    if (a.accumulate) x += *o; // the workgroup owns these rows
                    *o = x;
                }
            }
        };
        auto empty_tiles = [&](int t0, int t1) __attribute__((always_inline)) { // tiles without a group inside this workgroup's range
            if constexpr (!POOL) {
                for (int t2 = t0; t2 < t1; ++t2) write_tile(t2, acc[0], false);
            }
        };

        unsigned long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
        int dbg_pos = 0; // position of the current step in its phase (i & 7)
        auto tick = [&](int k) {
            if (DIAG) {
                // (stamps only where they are asked for: a stamp is an s_memtime round trip, seven per step -- the ablation masks without
                // 0x80 then time the kernel, not the clock reads: until round 6 they carried ~450 us of them on the five-layer MultiviewC frame)
                if (!(a.debug & 0x80)) return;
                if (a.debug & 32) { // (diagnostic 32: only the wait at the step barrier, by position of the step in its phase)
                    if (k == 5) t_prev = __builtin_amdgcn_s_memtime();
                    else if (k == 6) {
                        const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_prev;
#pragma unroll
                        for (int q = 0; q < 8; ++q) stamp[q] += dbg_pos == q ? dt : 0ull; // (no dynamic index: the array stays in registers)
                    }
                    return;
                }
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                stamp[k] += now - t_prev;
                t_prev = now;
            }
        };
        // ---------------------------------------------------------------- tables and DMA
        // Control is organised so that NO wave does scalar work that the others repeat: twelve waves share the CU's one scalar
        // unit, and ~40 scalar instructions at the head of a step in every wave cost 600-1000 cycles per step.  Every phase has
        // exactly eight steps (step i: phase i >> 3, quarter (i & 7) >> 1, set i & 1), so a loop counter names the step; what a
        // step needs beyond that comes from three small LDS tables that wave 0 fills AHEAD, with the lanes as the loop:
        //   s_phase  the generator's record of the phase                                    (at step 4 of the phase before)
        //   s_hdr    the window headers of the group's sub-tiles at the phase's layer: DMA   (at step 5)
        //   s_desc   per (step, sub-tile): image / record / weight addresses, flags, slots   (at step 6; first used at step 7)
        // The generator (vfa_pipe_seq.h: tiles -> scales -> groups of <= 4 live views -> layers), in the kernel as VECTOR code: the
        // scalar state machine with its three dependent scalar loads per tile sat on matrix wave 0 for ~1 400 cycles in three
        // steps of every phase, and every wave waits for wave 0 at the barrier.  Now lane = tile: 64 tiles of the workgroup's range
        // at a time are expanded into a ring of group records in LDS (view masks by vector loads, a lane scan for the positions,
        // the cut at both ends of the range by rank); a phase is then a counter and one LDS read.
        int gen_t = r_begin, gen_filled = 0, gen_next = 0, gen_layer = 0; // next RUN to expand; groups written / handed out; layer
        const int gen_t_lim = k_end > 0 ? r_end + 1 : r_end;
        auto fill_groups = [&]() { // (table wave, all lanes; lane = run)
            const int t = gen_t + lane;
            const bool on = t < gen_t_lim;
            const int base_tile = t * rt;
            // live-view mask of (scale s2, tile `off` of the lane's run): vector loads (L2 hits: the geometry pass wrote them)
            auto mask_of = [&](int s2, int off) -> unsigned {
                const unsigned *lv = s2 == 0 ? a.sc[0].live : (s2 == 1 ? a.sc[1].live : a.sc[2].live);
                return (on && base_tile + off < a.n_tiles) ? (lv[base_tile + off] & view_mask) : 0u;
            };
            int gt = 0; // groups of the run: per scale, its live sub-tiles in fours
#pragma unroll 1
            for (int s2 = 0; s2 < a.n_scales; ++s2) {
                int cnt_s = 0;
#pragma unroll 1
                for (int off = 0; off < rt; ++off) cnt_s += __popc(mask_of(s2, off));
                gt += groups_of_count(cnt_s);
            }
            const int lo = t == r_begin ? k_begin : 0;
            const int hi = min(t == r_end ? k_end : gt, gt);
            const int cnt = on ? max(hi - lo, 0) : 0;
            int incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            const int cap = kGroupRing - 8 - (gen_filled - gen_next); // (the groups of the phases in flight stay untouched)
            const bool take = on && incl <= cap;
            const int tiles_taken = __popcll(__ballot(take)); // (a prefix of the lanes: incl does not decrease)
            const int groups_taken = tiles_taken > 0 ? __shfl(incl, tiles_taken - 1) : 0;
            if (take && cnt > 0) {
                const int base = gen_filled + incl - cnt - lo;
                walk_groups(a.n_scales, rt, lo, mask_of, [&](int r, int s2, unsigned subs, int nj, int ci) {
                    if (r >= lo && r < hi) {
                        unsigned shifts = 0u; // (fp16 split: the sliver shift of every sub-tile's (tile, scale))
                        if constexpr (F16) {
                            const unsigned *shp = s2 == 0 ? a.sc[0].shift : (s2 == 1 ? a.sc[1].shift : a.sc[2].shift);
                            for (int j = 0; j < nj; ++j) shifts |= (shp[base_tile + sub_tile_off(subs, j)] & 0xffu) << (8 * j);
                        }
                        const unsigned w = (unsigned)s2 | ((unsigned)nj << 15) | (r + 1 < hi ? 1u << 20 : 0u) | ((unsigned)ci << 21);
                        *reinterpret_cast<uint4 *>(&s_groups[(base + r) & (kGroupRing - 1)][0]) = make_uint4((unsigned)t, subs, w, shifts);
                    }
                });
            }
            gen_filled += uniform_i(groups_taken);
            gen_t += uniform_i(tiles_taken);
            if (gen_t >= gen_t_lim && lane == 0) // the end of the sequence
                *reinterpret_cast<uint4 *>(&s_groups[gen_filled & (kGroupRing - 1)][0]) = make_uint4(0xffffffffu, 0u, 0u, 0u);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        auto gen_phase = [&](int n) { // (table wave)
            // (cold: the register allocator must spill INSIDE the expansion, not take a box of the pooling loops out of its registers)
            while (__builtin_expect(gen_layer == 0 && gen_next == gen_filled && gen_t < gen_t_lim, 0)) fill_groups(); // (runs without a live view add none)
            const uint4 g = *reinterpret_cast<const uint4 *>(&s_groups[gen_next & (kGroupRing - 1)][0]);
            const bool end = uniform_i((int)g.x) < 0;
            {
                const unsigned sc_w = s_sc[g.z & 3u][lane >= 4 && lane < 16 ? lane - 4 : 0];
                const unsigned w0 = lane == 0 ? g.x : (lane == 1 ? g.y : (lane == 2 ? (g.z | ((unsigned)gen_layer << 2)) : g.w));
                if (lane < 16) s_phase[n & 3][lane] = lane < 4 ? w0 : sc_w;
            }
            if (!end && ++gen_layer == a.nl) { gen_layer = 0; ++gen_next; }
        };
        auto hdr_dma = [&](int n) { // (wave 0) the headers of the sub-tiles of phase n: 8 lanes each, per-lane addresses
            const uint4 v = *reinterpret_cast<const uint4 *>(&s_phase[n & 3][0]);
            if (uniform_i((int)v.x) < 0) return;
            const int layer = (int)((v.z >> 2) & 1023u), nj = (int)((v.z >> 15) & 7u);
            int j = lane >> 3;
            j = j < nj ? j : nj - 1;
            const int view = sub_view(v.y, j), tile = (int)v.x * rt + sub_tile_off(v.y, j);
            const uint4 c1 = *reinterpret_cast<const uint4 *>(&s_phase[n & 3][8]), c2 = *reinterpret_cast<const uint4 *>(&s_phase[n & 3][12]);
            const unsigned long long item = (unsigned long long)((unsigned)tile * c2.z + (unsigned)layer * c2.w + (unsigned)view);
            const unsigned long long p = ((unsigned long long)c1.w << 32 | c1.z) + item * kHdrBytes + (unsigned)(lane & 7) * 4u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                             (__attribute__((address_space(3))) void *)(&s_hdr[n & 3][0]), 4, 0, 0);
        };
        // entry (k, x): {image address of (view, quarter), record address, flags | slots << 8, weight slice address (lanes x = 0)}
        auto make_desc = [&](int n) { // (table wave) lanes 0..15 = (step k, sub-tile x) of phase n
            const int k = (lane >> 1) & (kPS - 1), x = lane & 1, q = quarter_of(k), j = SMALL ? x : 2 * (k & 1) + x;
            // the phase record and its scale constants in one round trip
            const uint4 v = *reinterpret_cast<const uint4 *>(&s_phase[n & 3][0]);
            const uint4 c0 = *reinterpret_cast<const uint4 *>(&s_phase[n & 3][4]), c1 = *reinterpret_cast<const uint4 *>(&s_phase[n & 3][8]),
                        c2 = *reinterpret_cast<const uint4 *>(&s_phase[n & 3][12]);
            if (uniform_i((int)v.x) < 0) return;
            const int layer = (int)((v.z >> 2) & 1023u), nj = (int)((v.z >> 15) & 7u);
            const int jj = j < nj ? j : 0;
            const int tile = (int)v.x * rt + sub_tile_off(v.y, jj);
            // (the header read stays BEHIND the validity branch and indexed by jj: reading slot j in front of it -- one round trip
            // less -- ended in memory faults on full-size frames in the optimised build only; not understood, not used)
            const uint2 hd = *reinterpret_cast<const uint2 *>(&s_hdr[n & 3][jj * 8]);
            const int view = sub_view(v.y, jj);
            const unsigned flags = hd.x, n_slots = hd.y;
            const int Hf = (int)c2.x, Wf = (int)(c2.y & 0xffffu);
            // (fp16 form: the exponent of the phase's voxel-feature factor 2^(ea - shift), + 128, in the upper half)
            const unsigned ea64 = (c2.y & 0xffff0000u) - (F16 ? ((v.w >> (8 * jj)) & 0xffu) << 16 : 0u); // (the shift of the sub-tile's (tile, scale))
            const unsigned long long item = (unsigned long long)((unsigned)tile * c2.z + (unsigned)layer * c2.w + (unsigned)view);
            const unsigned long long img = ((unsigned long long)c0.y << 32 | c0.x) +
                                           (unsigned long long)view * (unsigned)((Hf + 2) * (Wf + 2)) * kSlotBytes + (unsigned)(q * kQSlot);
            const unsigned long long rec = ((unsigned long long)c0.w << 32 | c0.z) + item * (kTileBoxes * kRecBytes);
            const unsigned long long wsl = ((unsigned long long)c1.y << 32 | c1.x) +
                                           (unsigned long long)(((unsigned)layer * 8u * kSteps + (unsigned)q * 4u) * (unsigned)kWPlanes * 64u) * 16u;
            const unsigned fw = j < nj ? ((flags & 0xffu) | (n_slots << 8)) : 0u;
            if (lane < 2 * kPS) {
                uint4 *d = reinterpret_cast<uint4 *>(&s_desc[n & 3][k][x][0]);
                d[0] = make_uint4((unsigned)img, (unsigned)(img >> 32), (unsigned)rec, (unsigned)(rec >> 32));
                d[1] = make_uint4(fw, (unsigned)wsl, (unsigned)(wsl >> 32), (unsigned)(Wf + 2) | ea64);
            }
        };
        // Tap window (and, at the first quarter of a layer, the box records) of ONE sub-tile of step i: pooling waves 0-3 fetch for
        // the first sub-tile of the set, 4-7 for the second (the matrix waves are the longer role at four waves per SIMD: the fetch
        // cost each of them 1 500-2 400 cycles per step).  Addresses stay in vector registers (the same value in every lane).
        auto step_dma = [&](auto set_tag, int i) { // (set_tag: i & 1, a fact of the caller's position in the unrolled loop)
            constexpr int DSET = SMALL ? 0 : decltype(set_tag)::value; // (the tag is the parity of i: the set, except in the four-step phase)
            const int n = i >> kPSh, k = i & (kPS - 1), x = dw >> 2, wq4 = dw & 3, j = 2 * DSET + x;
            const uint4 *dp = reinterpret_cast<const uint4 *>(&s_desc[n & 3][k][x][0]);
            const uint4 d0 = dp[0], d1 = dp[1];
            const uint4 *hp = reinterpret_cast<const uint4 *>(&s_hdr[n & 3][j * 8]);
            const uint4 h0 = hp[0], h1 = hp[1];
            const int fw = uniform_i((int)d1.x);
            if (!(fw & kTileLive)) return;
            if (quarter_of(k) == 0 && (wq4 == 0 || (wq4 == 1 && lane < 32))) { // (32 boxes x 48 bytes = a load and a half)
                const unsigned long long p = ((unsigned long long)d0.w << 32 | d0.z) + (unsigned)(wq4 * 1024 + lane * 16);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                                 (__attribute__((address_space(3))) void *)((DSET ? s_rec1 : s_rec0) + x * kTileBoxes * kRecBytes + wq4 * 1024), 16, 0, 0);
            }
            if ((fw & kTileDirect) || ((kAblate & 1) || (DIAG && (a.debug & 1)))) return; // (diagnostic 1: no window fills; the records still come)
            const int n_slots = fw >> 8;
            const int cw = (int)h0.z, inv = (int)h0.w, x0 = (int)h1.x, t0 = (int)h1.y, top = (int)h1.z, b0 = (int)h1.w;
            const int n_fill = (n_slots + 3) >> 2;
            const unsigned long long img = ((unsigned long long)d0.y << 32 | d0.x) + (unsigned)((lane & 15) * 16);
            const int wpad = (int)(d1.w & 0xffffu); // (Wf + 2 of the scale: see make_desc)
            unsigned char *dst = s_win + ((k & 1) * 2 + x) * kWinBytes;
            for (int f = wq4; f < n_fill; f += 4) { // four quarter slots per instruction, 16 lanes each
                const int slot = min(4 * f + (lane >> 4), n_slots - 1);
                // (24-bit multiplies, full rate -- a 32-bit one is a quarter-rate instruction: slot < 2^7, inv <= 2^16, the pixel
                // index < 2^24 (larger images are pooled directly: pipe_records_kernel))
                const int wr = (int)(__umul24((unsigned)slot, (unsigned)inv) >> 16), wc = slot - (int)__umul24((unsigned)wr, (unsigned)cw);
                const int y = wr < top ? t0 + wr : b0 + (wr - top), xx = x0 + wc;
                const unsigned long long src = img + (unsigned long long)(__umul24((unsigned)(y + 1), (unsigned)wpad) + (unsigned)(xx + 1)) * kSlotBytes;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(dst + f * 1024), 16, 0, 0);
            }
        };
        // the 64 k x 32 n slice of collapse.weight of step i for this wave: 8 coalesced 1 KiB loads
        // (the slice address of the step sits in scalar registers -- the same for every lane --, the lane's part is one add)
        unsigned w_lo = 0, w_hi = 0;
        auto w_addr = [&](int i) {
            const uint4 d1 = reinterpret_cast<const uint4 *>(&s_desc[(i >> kPSh) & 3][i & (kPS - 1)][0][0])[1];
            w_lo = (unsigned)uniform_i((int)d1.y); w_hi = (unsigned)uniform_i((int)d1.z);
        };
        auto w_load = [&](int ks) {
            const char *base = reinterpret_cast<const char *>((size_t)((unsigned long long)w_hi << 32 | w_lo));
            const uint4 *src = reinterpret_cast<const uint4 *>(base + (unsigned)(wave * kSteps * kWPlanes * 64 * 16 + lane * 16)) + ks * kWPlanes * 64;
            const uint4 uh = src[0], ul = src[64];
            wq[ks].hi = *reinterpret_cast<const bf16x8 *>(&uh);
            wq[ks].lo = *reinterpret_cast<const bf16x8 *>(&ul);
            if constexpr (TERMS == 6) {
                const uint4 u2 = src[128];
                wq[ks].lo2 = *reinterpret_cast<const bf16x8 *>(&u2);
            }
        };

        // ---------------------------------------------------------------- products of one step (matrix waves)
        // SET is the parity of the step's index, hence a compile-time fact of the loop body it is called from; the weight is
        // reloaded behind the k-steps of set 1 (the next step starts another slice), never behind those of set 0 -- a reload
        // decided at run time inside the k-loop cost register copies at every merge point.
        // one 32 x 32 x 16 product on the matrix pipe: fp16 pieces (default) or bf16 pieces -- the registers hold 8 x 16 bits either way
        auto mfma16 = [](const bf16x8 &av, const bf16x8 &bv, const f32x16 &cv) -> f32x16 {
            if constexpr ((kAblate & 1024) != 0) { // (ablation: everything but the matrix instruction itself -- its operands stay alive)
                f32x16 r = cv;
                asm volatile("" : "+v"(r) : "v"(av), "v"(bv));
                return r;
            }
            if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), cv, 0, 0, 0);
            else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, cv, 0, 0, 0);
        };
        auto multiply = [&](auto set_tag, const PhaseRec &ph, int k, int par, bool next_chunk) {
            constexpr int SET = decltype(set_tag)::value;
            constexpr bool reload = SET == 1 || SMALL; // (four-step phase: every step ends its chunk)
            const int q = quarter_of(k);
            const bool grp_first = ph.layer() == 0 && q == 0;
            if (q == 0) {
                // The bias rides in the accumulator: at the first step of a group both accumulators of the set restart from it.  A
                // SELECT (on the first quarter of every layer), not an assignment under `grp_first`: the assignment made the
                // allocator keep the old and the new accumulators in different registers and copy all 32 at the join.
                const float bsc = ph.scale() == 0 ? bc[0] : (ph.scale() == 1 ? bc[1] : bc[2]);
                // (in each sub-tile's units: the sub-tiles of a group may sit in different tiles, with different sliver shifts)
                const float b0 = bsc * pow2f(-ph.shift(2 * SET)), b1 = bsc * pow2f(-ph.shift(2 * SET + 1));
                int first = grp_first ? 1 : 0;
                asm volatile("" : "+v"(first)); // (opaque: keeps the compiler from turning the selects back into that assignment)
                f32x16 bv0, bv1;
#pragma unroll
                for (int i = 0; i < 16; ++i) { bv0[i] = b0; bv1[i] = b1; }
                acc[2 * SET] = first ? bv0 : acc[2 * SET];
                acc[2 * SET + 1] = first ? bv1 : acc[2 * SET + 1];
            }
            const bool work = 2 * SET < ph.nj() && !((kAblate & 4) || (DIAG && (a.debug & 4))); // (set 1 of a group of one or two views is empty)
            // A fragments: lane (r, h) of row block rb reads chunk 2 ks + h, row 32 rb + r (read_frags)
            const unsigned pa = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)s_planes +
                                (unsigned)(par * kPieces * kPlaneBytes + h * kChunkStride + r * 16);
            // ONE set of fragment registers: the reads of k-step ks + 1 are issued behind the MFMAs of ks (which latched their A
            // operands when they issued) and land under them and under the partner wave's MFMAs; a second set for reading ahead
            // does not fit beside four accumulators, the tile sums and the weight slice (168 registers at three waves per SIMD)
            bf16x8 h0 = {}, h1 = {}, l0 = {}, l1 = {}; // row blocks 0 / 1 of the hi and of the lo plane: one k-step ahead of the MFMAs
            // A sub-tile without a live box in this layer was ZEROED by its pooling wave and is multiplied like any other; the second
            // row block of a set that has only ONE sub-tile (the last set of a group with an odd number of views: one step in four
            // on a seven-camera rig) is left out -- BOTH = false: half the MFMAs of the step, its accumulator is never read.
            const bool BOTH = 2 * SET + 1 < ph.nj() || TERMS == 6; // (uniform; the six-product variant, bound by the matrix pipe of its own schedule, keeps both)
            auto kstep = [&](auto ks_tag) {
                constexpr int KS = decltype(ks_tag)::value;
                {
                    if constexpr (TERMS == 6) {
                        // three pieces per operand, the six products down to 2^-16 of the largest (x = p0 + p1 + p2 to 2^-25:
                        // p0 p0, p0 p1, p1 p0, p0 p2, p2 p0, p1 p1; what is dropped is <= 2^-23 of the product): sgemm-class
                        bf16x8 a0, a1, a2;
                        read_frags3<KS, 0, kPlaneBytes>(pa, a0, a1, a2);
                        __builtin_amdgcn_sched_barrier(0);
                        acc[2 * SET] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wq[KS].lo2, acc[2 * SET], 0, 0, 0);
                        acc[2 * SET] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, wq[KS].hi, acc[2 * SET], 0, 0, 0);
                        acc[2 * SET] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wq[KS].lo, acc[2 * SET], 0, 0, 0);
                        acc[2 * SET] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wq[KS].lo, acc[2 * SET], 0, 0, 0);
                        acc[2 * SET] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wq[KS].hi, acc[2 * SET], 0, 0, 0);
                        acc[2 * SET] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wq[KS].hi, acc[2 * SET], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        {
                            read_frags3<KS, 1, kPlaneBytes>(pa, a0, a1, a2);
                            __builtin_amdgcn_sched_barrier(0);
                            acc[2 * SET + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wq[KS].lo2, acc[2 * SET + 1], 0, 0, 0);
                            acc[2 * SET + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, wq[KS].hi, acc[2 * SET + 1], 0, 0, 0);
                            acc[2 * SET + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wq[KS].lo, acc[2 * SET + 1], 0, 0, 0);
                            acc[2 * SET + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wq[KS].lo, acc[2 * SET + 1], 0, 0, 0);
                            acc[2 * SET + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wq[KS].hi, acc[2 * SET + 1], 0, 0, 0);
                            acc[2 * SET + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wq[KS].hi, acc[2 * SET + 1], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {
                    // the hi pair of this k-step was requested behind the hi MFMAs of the last one, the lo pair behind its lo MFMAs
                    wait_pair<2>(h0, h1);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[2 * SET] = mfma16(h0, wq[KS].lo, acc[2 * SET]);
                    if (BOTH) acc[2 * SET + 1] = mfma16(h1, wq[KS].lo, acc[2 * SET + 1]);
                    acc[2 * SET] = mfma16(h0, wq[KS].hi, acc[2 * SET]);
                    if (BOTH) acc[2 * SET + 1] = mfma16(h1, wq[KS].hi, acc[2 * SET + 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (KS < 3) { read_pair<KS + 1, 0>(pa, h0, h1); wait_pair<2>(l0, l1); } // (the MFMAs latched h0, h1 at issue)
                    else wait_pair<0>(l0, l1);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[2 * SET] = mfma16(l0, wq[KS].hi, acc[2 * SET]);
                    if (BOTH) acc[2 * SET + 1] = mfma16(l1, wq[KS].hi, acc[2 * SET + 1]);
                    if (TERMS >= 4) {
                        acc[2 * SET] = mfma16(l0, wq[KS].lo, acc[2 * SET]);
                        if (BOTH) acc[2 * SET + 1] = mfma16(l1, wq[KS].lo, acc[2 * SET + 1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (KS < 3) read_pair<KS + 1, 1>(pa, l0, l1);
                    __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (reload) w_load(KS); // the next slice (w_addr), k-step by k-step, into the registers just used
            };
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            tick(2);
            if (work) { // (one branch around the whole step: the fragment registers live across the k-steps)
                if constexpr (TERMS != 6) { read_pair<0, 0>(pa, h0, h1); read_pair<0, 1>(pa, l0, l1); }
                kstep(std::integral_constant<int, 0>{});
                kstep(std::integral_constant<int, 1>{});
                kstep(std::integral_constant<int, 2>{});
                kstep(std::integral_constant<int, 3>{});
            }
            tick(3);
            // A group of one or two views has nothing in set 1: that step is a barrier and little else, too short to cover a weight
            // load.  The next slice is then requested HERE, behind set 0 (whole, not interleaved: its registers are free now).
            if (!SMALL && SET == 0 && ph.nj() <= 2 && next_chunk) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) w_load(ks);
            }
        };
        // A group ends: relu and the view sum, vfa_op.py:124; vfanet.py:79, 82.  The sub-tiles of a group may belong to different tiles of
        // the run, and a tile gets contributions from several groups (its views may straddle two or three groups; its three scales are
        // three separate passes over the run).  The sub-tiles of ONE tile inside the group are added in registers, views in index order
        // (sub-tiles 0, 1 in front of the group's last step -- set 1 of the last quarter; set 0 is complete --, 2, 3 behind it), and the
        // result times 2^-(ea+ew-shift) is STORED as one contribution to a buffer of the workgroup's own, [tile of the run][scale][index]
        // (vfa_pipe_seq.h: walk_groups hands out the index): plain stores, nothing to wait for.  finish_run adds the contributions of a
        // tile in (scale, index) order.  (First attempt of round 6: running sums in the workspace, every sub-tile added with no-return
        // float atomics -- 134 M atomic dwords per five-layer MultiviewC frame at the L2's one atomic per clock and channel: 375 us of a
        // 1 690 us launch.)
        unsigned long long cmask = 0ull; // contributions stored per (tile of the run, scale): 4 bits each, index (off * kMaxScales + s)
        // The lane's offset inside a 32-row x 256-column buffer of the workspace, formed AT the use: written as `lane` the compiler hoists
        // these per-lane addresses out of the step loop, keeps them in vector registers across it and spills something else -- the
        // address of the weight slice, reloaded from scratch inside every k-step behind a vmcnt(0) that also waits for the slice
        // loads in flight: +1 000 cycles per step in the matrix waves (stamps of the diagnostic build, round 6).
        auto lane_off = [&]() __attribute__((always_inline)) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            return wave * 16 * 64 + ln;
        };
        auto store_seg = [&](const f32x16 &seg, float inv, int off, int sc, int ci) __attribute__((always_inline)) {
            if (kAblate & 256) return;
            float *bp = a.slots + ((((size_t)lb * kRunTiles + off) * kMaxScales + sc) * a.n_contrib + ci) * (8 * 16 * 64) + lane_off();
#pragma unroll
            for (int i = 0; i < 16; ++i) bp[i * 64] = F16 ? seg[i] * inv : seg[i];
            const int sh = (off * kMaxScales + sc) * 4;
            cmask = (cmask & ~(15ull << sh)) | ((unsigned long long)(ci + 1) << sh);
        };
        auto relu16 = [&](f32x16 &v) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = relu_t(v[i]);
        };
        auto add16 = [&](f32x16 &v, const f32x16 &w) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] += w[i];
        };
        // 2^-(ea + ew - shift) of sub-tile j: back to the map's units (exact)
        auto inv_sub = [&](const PhaseRec &ph, int j) __attribute__((always_inline)) { return F16 ? inv_of(ph.scale()) * pow2f(ph.shift(j)) : 1.0f; };
        // Straight-line on purpose: one segment (= the sub-tiles of one tile) at a time, summed into ONE temporary from the untouched
        // accumulators with uniform selects (x + 0 = x), stored, forgotten.  A version that summed in place under uniform branches
        // (the head tile's sum carried from the first pair to the second) left the compiler with sixteen-register values merging
        // at every join: it spilled whole accumulators and reloaded all sixteen registers once per element stored.
        bool tile_open = false; // (four-step phase: the tile's sum so far sits in acc[2]; RT1: in the workspace slot)
        auto rt1_begin = [&](const PhaseRec &ph) __attribute__((always_inline)) { // in front of the group's last step
            const bool two = ph.nj() > 1;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float r0 = relu_t(acc[0][i]), r1 = relu_t(acc[1][i]);
                acc[0][i] = two ? r0 + r1 : r0;
            }
            const float *slot = a.slots + (size_t)lb * (8 * 16 * 64) + lane_off();
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[1][i] = slot[i * 64]; // (always: what an unopened tile's slot holds is never used)
        };
        auto group_finish = [&](const PhaseRec &ph) __attribute__((always_inline)) {
            const int nj = ph.nj(), sc = ph.scale();
            if constexpr (RT1) {
                // (sub-tiles 0, 1 summed into acc[0] and the tile's sum so far requested into acc[1] in front of the last step: rt1_begin)
                const float inv = inv_sub(ph, 0);
                const bool three = nj >= 3, four = nj >= 4, open = tile_open;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float g = acc[0][i];
                    const float r2 = relu_t(acc[2][i]), r3 = relu_t(acc[3][i]);
                    g = three ? g + r2 : g;
                    g = four ? g + r3 : g;
                    const float s0 = open ? acc[1][i] : 0.0f;
                    if constexpr (F16) acc[0][i] = fmaf(g, inv, s0); // (g 2^-(ea+ew-shift) is exact: the same bits as multiply, then add)
                    else acc[0][i] = s0 + g;
                }
                if (ph.more()) {
                    float *slot = a.slots + (size_t)lb * (8 * 16 * 64) + lane_off();
#pragma unroll
                    for (int i = 0; i < 16; ++i) slot[i * 64] = acc[0][i];
                }
                tile_open = ph.more();
                return;
            }
            if constexpr (SMALL) {
                // Four-step phase (one or two views, runs of one tile): the products use acc[0], acc[1] only, so the tile's sum stays in
                // acc[2] across its groups (one per scale) -- no contributions, nothing to read back.  Same values in the same order
                // as the contributions would be added in: ((s0 + s1) + s2), each the group's relu'd sum times 2^-(ea+ew-shift).
                relu16(acc[0]);
                if (nj >= 2) { relu16(acc[1]); add16(acc[0], acc[1]); }
                const float inv = inv_sub(ph, 0);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float c = F16 ? acc[0][i] * inv : acc[0][i];
                    acc[2][i] = tile_open ? acc[2][i] + c : c;
                }
                tile_open = true;
                return;
            }
            const int o0 = sub_tile_off(ph.subs, 0), o1 = sub_tile_off(ph.subs, 1), o2 = sub_tile_off(ph.subs, 2), o3 = sub_tile_off(ph.subs, 3);
            relu16(acc[0]);
            if (nj >= 2) relu16(acc[1]);
            if (nj >= 3) relu16(acc[2]);
            if (nj >= 4) relu16(acc[3]);
            // seg = first + [c1] a + [c2] b + [c3] c, views in index order
            auto segment = [&](const f32x16 &first, bool c1, const f32x16 &x1, bool c2, const f32x16 &x2, bool c3, const f32x16 &x3, float inv, int off, int ci)
                               __attribute__((always_inline)) {
                f32x16 seg;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float v = first[i];
                    v += c1 ? x1[i] : 0.0f;
                    v += c2 ? x2[i] : 0.0f;
                    v += c3 ? x3[i] : 0.0f;
                    seg[i] = v;
                }
                store_seg(seg, inv, off, sc, ci);
            };
            const bool h1 = nj >= 2, h2 = nj >= 3, h3 = nj >= 4;
            segment(acc[0], h1 && o1 == o0, acc[1], h2 && o2 == o0, acc[2], h3 && o3 == o0, acc[3], inv_sub(ph, 0), o0, ph.ci());
            if (h1 && o1 != o0) segment(acc[1], h2 && o2 == o1, acc[2], h3 && o3 == o1, acc[3], false, acc[3], inv_sub(ph, 1), o1, 0);
            if (h2 && o2 != o1) segment(acc[2], h3 && o3 == o2, acc[3], false, acc[3], false, acc[3], inv_sub(ph, 2), o2, 0);
            if (h3 && o3 != o2) segment(acc[3], false, acc[3], false, acc[3], false, acc[3], inv_sub(ph, 3), o3, 0);
        };
        auto group_begin = [&](const PhaseRec &) __attribute__((always_inline)) {}; // (twelve-wave layout: everything behind the last step)
        auto group_end = [&](const PhaseRec &ph) __attribute__((always_inline)) { group_finish(ph); };

        // ---------------------------------------------------------------- pooling of one step (pooling waves)
        // returns false when none of the wave's 16 boxes has anything to pool (all masked, none NaN): the wave then writes zeros
        auto unpack = [&](auto set_tag, LaneBox &bx, bool &glob, unsigned wp, int x, bool direct) -> bool {
            constexpr int USET = decltype(set_tag)::value;
            const uint4 *rp = reinterpret_cast<const uint4 *>((USET ? s_rec1 : s_rec0) + x * kTileBoxes * kRecBytes) + (phalf * 16 + pb) * (kRecBytes / 16);
            uint4 rv[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) rv[k] = rp[k];
            {   // the sixteen tap weights (vfa_geom.h: make_axis, bilinear_weights -- the same operations, so the same bits)
                const float xl1 = __uint_as_float(rv[0].x), xr1 = __uint_as_float(rv[0].y), yt1 = __uint_as_float(rv[0].z), yb1 = __uint_as_float(rv[0].w);
                const float xl0 = 1.0f - xl1, xr0 = 1.0f - xr1, yt0 = 1.0f - yt1, yb0 = 1.0f - yb1;
                bx.wt[0] = yt0 * xl0; bx.wt[1] = yt0 * xl1; bx.wt[2] = yt1 * xl0; bx.wt[3] = yt1 * xl1;     // lt
                bx.wt[4] = yb0 * xr0; bx.wt[5] = yb0 * xr1; bx.wt[6] = yb1 * xr0; bx.wt[7] = yb1 * xr1;     // rb
                bx.wt[8] = yt0 * xr0; bx.wt[9] = yt0 * xr1; bx.wt[10] = yt1 * xr0; bx.wt[11] = yt1 * xr1;   // rt
                bx.wt[12] = yb0 * xl0; bx.wt[13] = yb0 * xl1; bx.wt[14] = yb1 * xl0; bx.wt[15] = yb1 * xl1; // lb
            }
            const bool vis = (rv[1].y & (unsigned)kVis) != 0u;
            // a masked box reads slot / pixel 0 (finite) and multiplies by its masked value (0, or NaN for a NaN box)
            bx.scl = __uint_as_float(rv[1].x);
            bx.asc = __uint_as_float(rv[2].z);
            bx.back = 1.0f;
            if constexpr (F16) { // times 2^k, k = ea - shift: powers of two, the quotient comes out as RN(v / area) 2^k exactly
                const int k2 = (int)(wp >> 16) - 128;
                bx.scl *= pow2f(k2); bx.asc *= pow2f(-k2);
                if constexpr (DIAG) bx.back = pow2f(-k2);
            }
            wp &= 0xffffu;
            unsigned rw[4] = {rv[1].z & 0xffffu, rv[1].z >> 16, rv[1].w & 0xffffu, rv[1].w >> 16};
            unsigned cl[4] = {rv[2].x & 0xffffu, rv[2].x >> 16, rv[2].y & 0xffffu, rv[2].y >> 16};
            glob = direct;
            // pixel of the padded image (direct) / slot of the window = row part + column part; a masked box reads slot / pixel 0
            const unsigned unit = direct ? (unsigned)kSlotBytes : (unsigned)kQSlot;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bx.rowb[i] = (vis ? (direct ? rw[i] * wp : rw[i]) : 0u) * unit;
                bx.colb[i] = (vis ? cl[i] : 0u) * unit + (unsigned)(pi * 16);
            }
            return __ballot(vis || bx.scl != bx.scl) != 0ull;
        };
        // the wave's 16 boxes x the 64 channels of quarter q: lane (box pb, piece pi) takes the 16-byte pieces
        // ((pb + m) & 3) * 4 + pi, m = 0..3, of its taps' quarter slots -- the four boxes of an LDS cycle read different 64-byte
        // quarters of the banks whatever slots they hold
        // (m_first, m_count: the wave's 16-byte pieces of the quarter -- {mpar, mpar + mstep, ...}, or, in a set that has ONE sub-tile,
        // a single piece: all eight pooling waves then share that sub-tile, see pool_step)
        auto pool = [&](auto glob_tag, const LaneBox &bx, int i, int x, int tile, int m_first, int m_count) {
            constexpr bool GLOB = decltype(glob_tag)::value;
            const int k = i & (kPS - 1), set = k & 1; // (`set`: the parity of the step = its LDS buffers)
            const unsigned char *win = s_win + (set * 2 + x) * kWinBytes;
            const char *img = nullptr;
            if constexpr (GLOB) { // (image address of (view, quarter): descriptor of the step)
                const uint4 d0 = reinterpret_cast<const uint4 *>(&s_desc[(i >> kPSh) & 3][k][x][0])[0];
                img = reinterpret_cast<const char *>((size_t)((unsigned long long)(unsigned)uniform_i((int)d0.y) << 32 | (unsigned)uniform_i((int)d0.x)));
            }
            const int row = x * 32 + phalf * 16 + pb;
            unsigned char *planes = s_planes + set * kPieces * kPlaneBytes;
            // All sixteen taps of a 16-byte piece are requested at once and consumed as they arrive (counted waits).  A hand-made
            // software pipeline across the four pieces (the next piece's taps requested as soon as half of this piece's were
            // consumed) measured 35 % SLOWER (4 900 against 3 620 cycles per step): more registers, spills, and the
            // scheduling barriers it needs keep the compiler from interleaving arithmetic and reads.
#pragma unroll
            for (int mm = 0; mm < mcount; ++mm) {
                if (mm >= m_count) break; // (uniform)
                const int m = m_first + mstep * mm; // (eight pooling waves: the partner wave, same boxes, takes the other two)
                const unsigned piece = (unsigned)((pb + m) & 3);
                const unsigned rot = piece << 6;
                float4 lt, rb, rt, lb2;
                // taps: index 4 * row + col over {top, top + 1, bottom, bottom + 1} x {left, left + 1, right, right + 1}
                {
                    unsigned wbase = rot;
                    if constexpr (!GLOB) wbase += (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char *)win;
                    unsigned colr[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) colr[k] = bx.colb[k] + wbase;
                    auto read4 = [&](f32x4 &t0, f32x4 &t1, f32x4 &t2, f32x4 &t3, unsigned p0, unsigned p1, unsigned p2, unsigned p3) {
                        if constexpr (GLOB) glob_read4(t0, t1, t2, t3, img, p0, p1, p2, p3);
                        else lds_read4(t0, t1, t2, t3, p0, p1, p2, p3);
                    };
                    auto wait4 = [&](auto n_tag, f32x4 &t0, f32x4 &t1, f32x4 &t2, f32x4 &t3) {
                        constexpr int N = decltype(n_tag)::value;
                        if constexpr (GLOB) glob_wait4<N>(t0, t1, t2, t3);
                        else lds_wait4<N>(t0, t1, t2, t3);
                    };
                    using W4 = std::integral_constant<int, 4>;
                    using W0 = std::integral_constant<int, 0>;
                    f32x4 a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3, d0, d1, d2, d3;
                    read4(a0, a1, a2, a3, bx.rowb[0] + colr[0], bx.rowb[0] + colr[1], bx.rowb[1] + colr[0], bx.rowb[1] + colr[1]);
                    read4(b0, b1, b2, b3, bx.rowb[2] + colr[2], bx.rowb[2] + colr[3], bx.rowb[3] + colr[2], bx.rowb[3] + colr[3]);
                    wait4(W4{}, a0, a1, a2, a3);
                    lt = sample4(as_float4(a0), as_float4(a1), as_float4(a2), as_float4(a3), bx.wt[0], bx.wt[1], bx.wt[2], bx.wt[3]);
                    read4(c0, c1, c2, c3, bx.rowb[0] + colr[2], bx.rowb[0] + colr[3], bx.rowb[1] + colr[2], bx.rowb[1] + colr[3]);
                    wait4(W4{}, b0, b1, b2, b3);
                    rb = sample4(as_float4(b0), as_float4(b1), as_float4(b2), as_float4(b3), bx.wt[4], bx.wt[5], bx.wt[6], bx.wt[7]);
                    read4(d0, d1, d2, d3, bx.rowb[2] + colr[0], bx.rowb[2] + colr[1], bx.rowb[3] + colr[0], bx.rowb[3] + colr[1]);
                    wait4(W4{}, c0, c1, c2, c3);
                    rt = sample4(as_float4(c0), as_float4(c1), as_float4(c2), as_float4(c3), bx.wt[8], bx.wt[9], bx.wt[10], bx.wt[11]);
                    wait4(W0{}, d0, d1, d2, d3);
                    lb2 = sample4(as_float4(d0), as_float4(d1), as_float4(d2), as_float4(d3), bx.wt[12], bx.wt[13], bx.wt[14], bx.wt[15]);
                }
                // RN((((lt + rb) - rt) - lb) / area)  (times the phase's power of two)                  (A.6)
                float4 v = make_float4(lt.x + rb.x, lt.y + rb.y, lt.z + rb.z, lt.w + rb.w);
                v = make_float4(v.x - rt.x, v.y - rt.y, v.z - rt.z, v.w - rt.w);
                v = make_float4(v.x - lb2.x, v.y - lb2.y, v.z - lb2.z, v.w - lb2.w);
                const float xs[4] = {box_quotient_scaled(v.x, bx.asc, bx.scl), box_quotient_scaled(v.y, bx.asc, bx.scl),
                                     box_quotient_scaled(v.z, bx.asc, bx.scl), box_quotient_scaled(v.w, bx.asc, bx.scl)};
                if (DIAG && (a.debug & kDbgDumpVox)) { // (one view, one scale, one layer: every sub-tile is a tile of its own; the power-of-two factor of the fp16 split taken out again: exact)
                    const int tl = tile / a.tiles_w, tw = tile - tl * a.tiles_w, brow = phalf * 16 + pb;
                    const int cl = tl * kTileL + (brow >> 3), cw = tw * kTileW + (brow & 7);
                    const float back = bx.back;
                    if (cl < a.L && cw < a.W)
                        *reinterpret_cast<float4 *>(a.out + (size_t)(cl * a.W + cw) * kC + quarter_of(k) * 64 + (int)piece * 16 + pi * 4) =
                            make_float4(xs[0] * back, xs[1] * back, xs[2] * back, xs[3] * back);
                }
                // x = hi + lo + r, |r| <= 2^-17 |x|: hi = RNE bf16(x), lo = RNE bf16(x - hi)
                // (three-piece variant: lo2 = RNE bf16(x - hi - lo), |x - hi - lo - lo2| <= 2^-25 |x|)
                union { __bf16 b[4]; uint2 u; } hi, lo, lo2;
                if constexpr (F16) { // (fp16 form: two fp16 pieces of the scaled value, vfa_split.h)
                    split_f16x4(xs[0], xs[1], xs[2], xs[3], hi.u, lo.u);
                } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    hi.b[k] = (__bf16)xs[k];
                    const float r1 = xs[k] - (float)hi.b[k];
                    lo.b[k] = (__bf16)r1;
                    if constexpr (TERMS == 6) lo2.b[k] = (__bf16)(r1 - (float)lo.b[k]);
                }
                }
                // channels 16 piece + 4 pi .. + 3 of the quarter: chunk 2 piece + (pi >> 1), half pi & 1
                const int off = (int)(2 * piece + (pi >> 1)) * kChunkStride + row * 16 + (pi & 1) * 8;
                *reinterpret_cast<uint2 *>(planes + off) = hi.u;
                *reinterpret_cast<uint2 *>(planes + kPlaneBytes + off) = lo.u;
                if constexpr (TERMS == 6) *reinterpret_cast<uint2 *>(planes + 2 * kPlaneBytes + off) = lo2.u;
            }
        };
        // A set with ONE sub-tile (set 0 of a one-view group, set 1 of a three-view group: one step in five on a seven-camera rig): the
        // waves that would pool the missing sub-tile take half of the pieces of the one there is -- wave (px, phalf, mpar) pools boxes
        // phalf of sub-tile 0, piece mpar + 2 px --, so the step's pooling takes half the time instead of leaving four waves idle.
        auto pool_step = [&](auto set_tag, int i, const PhaseRec &ph) {
            constexpr int SET = SMALL ? 0 : decltype(set_tag)::value; // (the tag is the parity of i)
            const int nj = ph.nj();
            const bool single = W16 && nj == 2 * SET + 1;
            const int n = i >> kPSh, k = i & (kPS - 1), x = single ? 0 : px;
            const int tile = DIAG ? ph.tile(2 * SET + x, rt) : 0; // (the diagnostic dump of the voxel features addresses by cell)
            const int m_first = single ? mpar + 2 * px : mpar, m_count = single ? 1 : mcount;
            auto one = [&](LaneBox &bx, bool &glob, bool &live) { // (called with the registers of the step's set)
                if (quarter_of(k) == 0) { // first quarter of the layer: this wave's 16 boxes for the whole layer
                    const uint4 d1 = reinterpret_cast<const uint4 *>(&s_desc[n & 3][k][x][0])[1];
                    const int fw = uniform_i((int)d1.x);
                    live = (fw & kTileLive) != 0;
                    if (live) live = unpack(std::integral_constant<int, SET>{}, bx, glob, d1.w, x, (fw & kTileDirect) != 0);
                }
                if (!live) { // no live box in this layer (or no such sub-tile in the group): the matrix waves multiply zeros
                    const int row = x * 32 + phalf * 16 + pb;
                    unsigned char *planes = s_planes + (k & 1) * kPieces * kPlaneBytes;
#pragma unroll
                    for (int mm = 0; mm < mcount; ++mm) {
                        if (mm >= m_count) break;
                        const int m = m_first + mstep * mm;
                        const int off = (2 * m + (pi >> 1)) * kChunkStride + row * 16 + (pi & 1) * 8;
                        *reinterpret_cast<uint2 *>(planes + off) = make_uint2(0u, 0u);
                        *reinterpret_cast<uint2 *>(planes + kPlaneBytes + off) = make_uint2(0u, 0u);
                        if constexpr (TERMS == 6) *reinterpret_cast<uint2 *>(planes + 2 * kPlaneBytes + off) = make_uint2(0u, 0u);
                    }
                    return;
                }
                if ((kAblate & 2) || (DIAG && (a.debug & 2))) return;
                if (glob) pool(std::true_type{}, bx, i, x, tile, m_first, m_count);
                else pool(std::false_type{}, bx, i, x, tile, m_first, m_count);
            };
            if constexpr (SET == 0) one(boxA, globA, liveA);
            else one(boxB, globB, liveB);
        };

        // ---------------------------------------------------------------- a workgroup's part of a run is complete
        // Matrix waves: the sum of every tile of the run from its contributions, (scale, index) order, three loads in flight (acc[1..3]
        // are free between groups).  sc1 loads: the buffers are reused run after run, a plain load could hit a line of the CU's vector
        // cache from the run before.  A run nobody else holds groups of is written straight to the map; of a SHARED run every tile's sum
        // goes to the workspace and whoever arrives LAST adds the parts (in workgroup order: one fixed association) and stores the
        // tiles; the others go on.  sc1 stores and loads on both sides, every storing wave drained, then one ticket per workgroup
        // (guide: inter-workgroup visibility, valid forms).  The ticket is an acquire-release operation at agent scope: the memory model
        // then orders the parts of every earlier arriver in front of the last arriver's loads (the sc1 forms alone rest on how gfx950
        // happens to treat them); it runs at most twice per workgroup and launch.
        auto finish_run = [&](int run, int next_tile) __attribute__((always_inline)) {
            const int base = run * rt;
            if (kAblate & 512) { cmask = 0ull; return; }
            const bool shared = shared_run(run);
            const int which = (run == r_begin && k_begin > 0) ? 0 : 1;
            tile_open = false;
            auto load16 = [&](f32x16 &v, const float *p) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __hip_atomic_load(p + i * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            // a finished tile: to the map (nobody else holds groups of the run) or to this workgroup's part of a shared run
            auto emit_tile = [&](int off) __attribute__((always_inline)) {
                if (!shared) write_tile(base + off, acc[0], true);
                else {
                    float *pp = a.partial + (((size_t)lb * 2 + which) * kRunTiles + off) * (8 * 16 * 64) + lane_off();
#pragma unroll
                    for (int i = 0; i < 16; ++i) __hip_atomic_store(pp + i * 64, acc[0][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[0][i] = 0.0f;
            };
            if constexpr (!POOL && SMALL) { // (the sum of the run's one tile is in acc[2]: group_finish)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[0][i] = acc[2][i];
                emit_tile(0);
            }
            if constexpr (!POOL && RT1) emit_tile(0); // (the tile's sum is in acc[0]: group_finish)
            if constexpr (!POOL && !SMALL && !RT1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the stores of the last group
                // The contributions of the run as ONE flat sequence (tile, scale, index ascending), three loads in flight all the way
                // (a batch per (tile, scale) was twelve exposed round trips per run: 85 us per finish on the five-layer MultiviewC frame)
                const int n_off = min(rt, a.n_tiles - base);
                auto cnt_of = [&](int off, int sc) { return (int)((cmask >> ((off * kMaxScales + sc) * 4)) & 15ull); };
                auto seek = [&](int &off, int &sc, int &ci) { // the next contribution at or behind (off, sc, ci); off == n_off: none
                    while (off < n_off) {
                        if (ci < cnt_of(off, sc)) return;
                        ci = 0;
                        if (++sc == a.n_scales) { sc = 0; ++off; }
                    }
                };
                int p_off = 0, p_sc = 0, p_ci = 0, c_off = 0, c_sc = 0, c_ci = 0, cur = 0;
                auto issue = [&](f32x16 &buf) __attribute__((always_inline)) {
                    seek(p_off, p_sc, p_ci);
                    if (p_off >= n_off) return;
                    load16(buf, a.slots + ((((size_t)lb * kRunTiles + p_off) * kMaxScales + p_sc) * a.n_contrib + p_ci) * (8 * 16 * 64) + lane_off());
                    ++p_ci;
                };
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[0][i] = 0.0f;
                issue(acc[1]); issue(acc[2]); issue(acc[3]);
                bool going = true;
                auto consume = [&](f32x16 &buf) __attribute__((always_inline)) {
                    if (!going) return;
                    seek(c_off, c_sc, c_ci);
                    while (cur < min(c_off, n_off)) { emit_tile(cur); ++cur; } // every tile in front of the next contribution is complete
                    if (c_off >= n_off) { going = false; return; }
                    add16(acc[0], buf);
                    ++c_ci;
                    issue(buf);
                };
                while (going) { consume(acc[1]); consume(acc[2]); consume(acc[3]); }
                cmask = 0ull;
            }
            if (__builtin_expect(shared, 0)) {
                if constexpr (!POOL) {
                    // Every matrix wave owns 32 columns of every tile from the first product to the store: the hand-off is PER WAVE -- its
                    // part drained, then its own ticket (tickets[run][wave]) --, no barrier, and the pooling waves are not involved at all
                    // (they go on pooling the next step).  Until round 6 the whole workgroup met at two barriers here.
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    // the workgroups that hold groups of this run (found once, in front of the loop: share_of)
                    const int first = run == r_begin && k_begin > 0 ? sh_b_first : sh_e_first;
                    const int last = run == r_begin && k_begin > 0 ? sh_b_last : sh_e_last;
                    const int parts = run == r_begin && k_begin > 0 ? sh_b_parts : sh_e_parts;
                    unsigned *ticket = a.tickets + (size_t)run * 8 + wave;
                    unsigned old = 0u;
                    if (lane == 0) old = __hip_atomic_fetch_add(ticket, 1u, VFA_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);
                    const bool am_last = uniform_i((int)old) == parts - 1;
                    if (am_last) {
                        // (nobody else touches this ticket in this launch: clear it for the next call on the same workspace)
                        if (lane == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        // The workgroups that hold groups of the run, in workgroup order, found ONCE (their ranges are two dependent
                        // loads each: inside the loop over the tiles that was 4 x parts x 2 round trips for whoever arrived last):
                        // (j << 1 | which part of j) in ten bits each, twelve of them; more sharers (a frame of a handful of runs
                        // on 256 workgroups) are looked up again tile by tile.
                        unsigned long long plist0 = 0ull, plist1 = 0ull;
                        int n_parts = 0;
#pragma unroll 1
                        for (int j = first; j <= last; ++j) {
                            int tb, kb, te, ke;
                            range_of(j, tb, kb, te, ke);
                            if (tb > te || (tb == te && kb >= ke)) continue;
                            const unsigned long long e = ((unsigned long long)j << 1) | ((run == tb && kb > 0) ? 0ull : 1ull);
                            if (n_parts < 6) plist0 |= e << (10 * n_parts);
                            else if (n_parts < 12) plist1 |= e << (10 * (n_parts - 6));
                            ++n_parts;
                        }
#pragma unroll 1
                        for (int off = 0; off < rt; ++off) {
                            if (base + off >= a.n_tiles) break;
#pragma unroll
                            for (int i = 0; i < 16; ++i) acc[0][i] = 0.0f;
                            // (this workgroup's own part comes back from the workspace like the others'; three parts per round trip)
                            int k = 0, jslow = first;
                            auto next_part = [&]() -> const float * {
                                int j, wj;
                                if (n_parts <= 12) {
                                    if (k >= n_parts) return nullptr;
                                    const unsigned e = (unsigned)((k < 6 ? plist0 >> (10 * k) : plist1 >> (10 * (k - 6))) & 1023ull);
                                    ++k;
                                    j = (int)(e >> 1); wj = (int)(e & 1u);
                                } else {
                                    for (;; ++jslow) {
                                        if (jslow > last) return nullptr;
                                        int tb, kb, te, ke;
                                        range_of(jslow, tb, kb, te, ke);
                                        if (tb > te || (tb == te && kb >= ke)) continue;
                                        j = jslow++; wj = (run == tb && kb > 0) ? 0 : 1;
                                        break;
                                    }
                                }
                                return a.partial + (((size_t)j * 2 + wj) * kRunTiles + off) * (8 * 16 * 64) + lane_off();
                            };
#pragma unroll 1
                            for (;;) {
                                const float *q1 = next_part(), *q2 = next_part(), *q3 = next_part();
                                if (!q1) break;
                                load16(acc[1], q1);
                                if (q2) load16(acc[2], q2);
                                if (q3) load16(acc[3], q3);
                                add16(acc[0], acc[1]);
                                if (q2) add16(acc[0], acc[2]);
                                if (q3) add16(acc[0], acc[3]);
                                if (!q3) break;
                            }
                            write_tile(base + off, acc[0], true);
                        }
                    }
                }
            }
            empty_tiles(min(base + rt, a.n_tiles), next_tile);
        };
        const int end_tile = min(r_end * rt, a.n_tiles); // the first tile behind this workgroup's whole runs

        // ---------------------------------------------------------------- the loop
        auto lds_fence_barrier = [&]() {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        // prologue: phase 0's record, headers, descriptors; the windows / records and the weight slice of step 0
        if (table_wave) {
            gen_phase(0);
        }
        lds_fence_barrier();
        PhaseRec rec = phase_rec(0); // matrix waves: the phase of the step being multiplied; pooling waves: of the step being pooled
        empty_tiles(min((k_begin == 0 ? r_begin : r_begin + 1) * rt, a.n_tiles), rec.valid() ? rec.run * rt : end_tile);
        if (!rec.valid()) return;
        if (table_wave) hdr_dma(0);
        if (W16 && table_wave) { // (sixteen waves: the tables run a phase further ahead, see `body`)
            gen_phase(1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            hdr_dma(1);
        }
        lds_fence_barrier();
        if (table_wave) make_desc(0);
        lds_fence_barrier();
        if (POOL != dma_matrix) step_dma(std::integral_constant<int, 0>{}, 0);
        if constexpr (!POOL) {
            w_addr(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) w_load(ks);
        }
        lds_fence_barrier();
        if (DIAG) t_prev = __builtin_amdgcn_s_memtime();
        // Iteration i: the pooling waves pool step i (set i & 1), the matrix waves fetch for step i + 1 and multiply step i - 1.
        // `live` bit 0 / 1 / 2: step i - 1 / i / i + 1 exists (a phase record with tile < 0 ends the sequence).
        unsigned live = 2u | 4u;
        auto body = [&](auto pset_tag, int i) {
            constexpr int PSET = decltype(pset_tag)::value, MSET = PSET ^ 1;
            const int m = i & (kPS - 1);
            dbg_pos = m;
            tick(0);
            if constexpr (POOL) {
                if (m == 0 && i > 0) rec = phase_rec(i >> kPSh);
                // the next step's windows (and records) first, so that they land under this step's pooling
                const bool bare = (kAblate & 64) || (DIAG && (a.debug & 64)); // (diagnostic 64: the loop, the tables and the barrier only)
                // The tables of the next phase (see `tables and DMA`: steps 4, 5, 6 of this one), FIRST in the step: make_desc reads
                // the header buffer, a DMA target -- behind this step's window requests the compiler drains them in front of that
                // read, and the table wave was 3 000-4 500 cycles late at every sixth step.
                // (make_desc reads the header buffer, a DMA target: behind this step's window requests the compiler would drain them
                // in front of that read -- so it goes first, at a point where this wave has nothing in flight)
                if (W16 && table_wave && m == 0) {
                    unsigned long long t_tab = 0;
                    if (DIAG && (a.debug & 32) && (a.debug & 16)) t_tab = __builtin_amdgcn_s_memtime();
                    make_desc((i >> kPSh) + 1);
                    if (DIAG && (a.debug & 32) && (a.debug & 16)) { // (diagnostic 32 + 16: the time of make_desc, in the upper bits of slot 0)
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        stamp[0] += (__builtin_amdgcn_s_memtime() - t_tab) << 24;
                    }
                }
                if (W16 && table_wave && m == 2) hdr_dma((i >> kPSh) + 2); // (first too: its 256 bytes land under the pooling)
                if (!dma_matrix && (live & 4u) && !bare) step_dma(std::integral_constant<int, PSET ^ 1>{}, i + 1);
                if ((live & 2u) && !bare) pool_step(std::integral_constant<int, PSET>{}, i, rec);
                // The tables (see `tables and DMA`), a phase further ahead than on twelve waves and behind the pooling: the descriptors
                // of phase n + 1 (~1 000 cycles; its headers were requested a phase ago) in the FIRST step of phase n, where the
                // matrix waves end a group (relu, view sum, tile store) and the pooling waves wait longest at the barrier; the record
                // of phase n + 2 (~550) and the request for its headers (~300) in the two steps behind.  In one step (any) the three
                // together made every wave wait for the table wave.
                if (W16 && table_wave) {
                    if (m == 1) gen_phase((i >> kPSh) + 2);
                }
                tick(1); // (pooling waves: slot 1 = requests + pooling, slot 2 = waiting for the next step's windows to land)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                tick(2);
            } else {
                // the weight slice requested during the last step (and tile stores).  The BUILTIN, not assembly: it tells the
                // compiler's wait-count model that nothing is pending; otherwise it waits for those loads itself -- vmcnt(0) in front
                // of the first MFMA: a memory round trip per step
                __builtin_amdgcn_s_waitcnt(0x0f70);
                tick(1);
                if (m == 1 && i > 1) rec = phase_rec((i - 1) >> kPSh);
                if (!W16 && table_wave) { // the tables of the next phase (see `tables and DMA`)
                    if (m == 4) gen_phase((i >> 3) + 1);
                    else if (m == 5) hdr_dma((i >> 3) + 1);
                    else if (m == 6) make_desc((i >> 3) + 1);
                }
                // Twelve-wave layout: the matrix waves fetch.  The two matrix waves of a SIMD take their two jobs in opposite order
                // (waves 0-3 request the next step's window and then multiply, waves 4-7 multiply first).
                const bool dma_first = wave < 4;
                if (!W16 && dma_first && (live & 4u)) step_dma(std::integral_constant<int, PSET ^ 1>{}, i + 1);
                // (step i - 1 ends its group: the last quarter of the last layer, set 1 -- `rec` is still that step's phase)
                const bool group_ends = MSET == 1 && ((i - 1) & 7) == 7 && (live & 1u) && rec.layer() == a.nl - 1;
                if ((live & 1u) && !((kAblate & 64) || (DIAG && (a.debug & 64)))) {
                    if (MSET == 1 && group_ends) group_begin(rec);
                    // the slice of the next chunk: behind the k-steps of set 1 (steps i - 1 = set 1, i = set 0 of the next chunk), or,
                    // when set 1 of the group is empty, already behind set 0 (steps i - 1 = set 0, i = set 1, i + 1 = the next chunk)
                    bool next_chunk = false;
                    if (MSET == 1 && (live & 2u)) { w_addr(i); next_chunk = true; }
                    if (MSET == 0 && rec.nj() <= 2 && (live & 4u)) { w_addr(i + 1); next_chunk = true; }
                    multiply(std::integral_constant<int, MSET>{}, rec, (i - 1) & 7, MSET, next_chunk);
                    if (MSET == 1 && group_ends) group_end(rec);
                }
                if (!W16 && !dma_first && (live & 4u)) step_dma(std::integral_constant<int, PSET ^ 1>{}, i + 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (wave 0: the tables; the header DMA is waited for below)
                if (!W16 || (table_wave && m == 5)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (what this wave fetched has landed)
                tick(4);
            }
            if constexpr (MSET == 1) { // (step i - 1 was the last of its phase's quarter 3?)
                const PhaseRec &pr = rec; // matrix waves: the record of step i - 1; pooling waves hold the record of step i
                if constexpr (POOL) {
                    // (the pooling waves have no part in the end of a run: the matrix waves hand off wave by wave, finish_run)
                } else {
                    if (__builtin_expect(((i - 1) & 7) == 7 && (live & 1u) && pr.layer() == a.nl - 1, 0)) {
                        if (!pr.more()) {
                            const PhaseRec nxt = phase_rec(((i - 1) >> 3) + 1);
                            finish_run(pr.run, nxt.valid() ? nxt.run * rt : end_tile);
                        }
                    }
                }
            }
            tick(5);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            tick(6);
            if (DIAG && !(a.debug & 32)) stamp[7] += 1;
            // the step that enters at i + 2: a new phase when (i + 2) & 7 == 0 (its record was written at step 4 of this phase)
            unsigned next_live = (live >> 2) & 1u;
            if (((i + 2) & (kPS - 1)) == 0 && next_live) next_live = uniform_i((int)s_phase[((i + 2) >> kPSh) & 3][0]) >= 0 ? 1u : 0u;
            live = (live >> 1) | (next_live << 2);
        };
        if constexpr (W16 && !POOL) {
            // Matrix role, sixteen-wave layout: the loop by PHASE.  Iteration (n, j) multiplies step j of phase n while the pooling
            // waves are one step ahead (their loop is `body` below; both roles pass 8 P + 1 step barriers for P phases, and the
            // hand-off barriers of a shared tile in the same iteration).  With j static the set, the quarter, the table wave's job
            // and the end of the group are facts of the code position; the phase record is read once per phase, the weight-slice
            // address comes from it by scalar arithmetic instead of an LDS read per chunk.
            auto w_set = [&](int scale, int layer, int q) {
                if (kAblate & 4096) layer = 0; // (ablation: every layer reads the first layer's weights -- a weight set that fits L2)
                const uint2 wf = *reinterpret_cast<const uint2 *>(&s_sc[scale][4]);
                const unsigned long long p = ((unsigned long long)wf.y << 32 | wf.x) +
                                             (unsigned long long)(((unsigned)layer * 8u * kSteps + (unsigned)q * 4u) * (unsigned)kWPlanes * 64u) * 16u;
                w_lo = (unsigned)uniform_i((int)(unsigned)p); w_hi = (unsigned)uniform_i((int)(unsigned)(p >> 32));
            };
            __builtin_amdgcn_s_waitcnt(0x0f70);
            if (dma_matrix) { // (iteration 0: the pooling waves pool step 0; the windows of step 1)
                step_dma(std::integral_constant<int, 1>{}, 1);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // The steps of a GROUP are the inner loop; what follows a group -- relu, the tiles' contributions, and at the end of a run the
            // tiles' sums and the hand-off -- sits BEHIND it, outside: inside the loop body the temporaries of that code spilled the
            // loop's own values (25 scratch operations per phase in the matrix waves: +13 % on the five-layer MultiviewC frame).  The
            // accumulators of a finished group are untouched until the next group's first product, and nothing here meets a barrier.
            int n = 0;
            PhaseRec nx = rec;
            for (bool done = false; !done;) {
            bool group_ended = false;
            for (;;) {
                auto iter = [&](auto j_tag) {
                    constexpr int J = decltype(j_tag)::value, SET = SMALL ? 0 : J & 1, PAR = J & 1;
                    dbg_pos = (J + 1) & (kPS - 1);
                    tick(0);
                    // The weight slice requested during the last step (see `body`).  (Round 6: the loads as assembly with a counted vmcnt in
                    // front of every k-step -- the slice requested last lands under the first products instead of being waited for
                    // here -- left every workload where it was, within 1 %: the matrix waves wait at the step barrier anyway.)
                    __builtin_amdgcn_s_waitcnt(0x0f70);
                    tick(1);
                    // (the tables of the next phase are the last pooling wave's job in this layout: `body`)
                    if constexpr (J == kPS - 2) nx = phase_rec(n + 1); // (written a phase ago)
                    // the windows of step 8 n + J + 2 (the pooling waves are at 8 n + J + 1): the two matrix waves of a SIMD take
                    // their two jobs in opposite order (waves 0-3 request and then multiply, waves 4-7 multiply first)
                    const bool dma_now = dma_matrix && (J < kPS - 2 || nx.valid()) && !((kAblate & 64) || (DIAG && (a.debug & 64)));
                    if (dma_now && wave < 4) step_dma(std::integral_constant<int, J & 1>{}, kPS * n + J + 2);
                    if (!((kAblate & 64) || (DIAG && (a.debug & 64)))) {
                        // the slice of the next chunk: behind the k-steps of set 1, or, when set 1 of the group is empty, already
                        // behind set 0; the chunk after quarter 3 is the next phase's first
                        bool next_chunk = false;
                        if (SMALL || SET == 1 || rec.nj() <= 2) {
                            const int jn = (SMALL || SET == 1) ? J + 1 : J + 2; // first step of the next chunk
                            if (jn < kPS) { w_set(rec.scale(), rec.layer(), quarter_of(jn)); next_chunk = true; }
                            // (not across the end of a GROUP: the code behind it needs the registers of the weight slice -- with the slice
                            // of the next group in them the compiler spilled whole accumulators around the contributions' stores)
                            // (the four-step phase has registers to spare -- acc[2], acc[3] carry no products -- and a light end of group)
                            else if (nx.valid() && (SMALL || RT1 || rec.layer() != a.nl - 1)) { w_set(nx.scale(), nx.layer(), 0); next_chunk = true; }
                        }
                        if constexpr (RT1 && J == kPS - 1) {
                            if (rec.layer() == a.nl - 1) rt1_begin(rec);
                        }
                        multiply(std::integral_constant<int, SET>{}, rec, J, PAR, next_chunk);
                    }
                    if (dma_now && wave >= 4) step_dma(std::integral_constant<int, J & 1>{}, kPS * n + J + 2);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (dma_matrix) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (what this wave requested has landed)
                    tick(4);
                    tick(5);
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    tick(6);
                    if (DIAG && !(a.debug & 32)) stamp[7] += 1;
                };
                iter(std::integral_constant<int, 0>{});
                iter(std::integral_constant<int, 1>{});
                iter(std::integral_constant<int, 2>{});
                iter(std::integral_constant<int, 3>{});
                if constexpr (!SMALL) {
                    iter(std::integral_constant<int, 4>{});
                    iter(std::integral_constant<int, 5>{});
                    iter(std::integral_constant<int, 6>{});
                    iter(std::integral_constant<int, 7>{});
                }
                group_ended = rec.layer() == a.nl - 1;
                if (group_ended || !nx.valid()) break;
                rec = nx;
                ++n;
            }
            if (!SMALL && !RT1 && group_ended && nx.valid()) { // the first weight slice of the next group: requested here, it lands under the stores below
                w_set(nx.scale(), nx.layer(), 0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) w_load(ks);
            }
            if (group_ended && !((kAblate & 64) || (DIAG && (a.debug & 64)))) {
                group_finish(rec);
                if (!rec.more()) finish_run(rec.run, nx.valid() ? nx.run * rt : end_tile);
            }
            if (!nx.valid()) done = true;
            else { rec = nx; ++n; }
            }
        } else {
        for (int i = 0;; i += 2) {
            if (!(live & 3u)) break;
            body(std::integral_constant<int, 0>{}, i); // (steps 0, 2, ... are pooled here: set 0)
            if (!(live & 3u)) break;
            body(std::integral_constant<int, 1>{}, i + 1);
        }
        }
        if (DIAG && a.diag && (a.debug & 0x80) && tid == ((a.debug >> 8) & 15) * 64)
            for (int k = 0; k < 8; ++k) a.diag[(size_t)blockIdx.x * 8 + k] = stamp[k];
    };
    if (wave >= kMatWaves) {
        // the pooling wave is the busiest third of its SIMD and its step the longest: it goes first (its step head 150 against 500
        // cycles, pooling 3 800 against 4 000 without the priority)
        if (!(DIAG && (a.debug & 8))) __builtin_amdgcn_s_setprio(VFA_PIPE_PRIO_POOL);
        if constexpr (F16) fp16_saturate_mode(true); // (the pooling waves convert and never multiply; the matrix waves keep the default mode: vfa_split.h)
#ifndef VFA_PIPE_NO_POOL
        run(std::true_type{});
#endif
    } else {
        if (DIAG && (a.debug & 16)) __builtin_amdgcn_s_setprio(2);
        else if (VFA_PIPE_PRIO_MAT) __builtin_amdgcn_s_setprio(VFA_PIPE_PRIO_MAT);
#ifndef VFA_PIPE_NO_MAT
        run(std::false_type{});
#endif
        // what this workgroup took (matrix wave 0 leaves the loop with the last step): vfa_pipe_balance_f32 moves the bounds by it
        if (tid == 0) *wg_cycles = (unsigned long long)(__builtin_amdgcn_s_memtime() - t_start);
    }
}

// Bounds of the workgroups' shares that minimise the HEAVIEST share (estimated cost), for a launch of nblk workgroups.  The uniform
// split puts a bound at every (kChunks / nblk)-th piece and each bound then snaps to the nearest group -- a group is indivisible: up
// to four views x all layers, 5 % of a workgroup's load on the shipped five-layer MultiviewC grid --, so a share can be a whole group
// above the mean (slowest / mean workgroup 1.09 measured there).  Here: the smallest B for which the pieces split into at most nblk
// contiguous shares of cost <= B (512 candidates per round, each thread sweeps its own; cost of a share = difference of the cumulative
// costs the cuts kernel left per piece), then the greedy split for that B.  One workgroup; mode 0 clears the state.
__global__ __launch_bounds__(kMaxBlocks) void pipe_balance_kernel(int *bal, const int *chunk_start, const unsigned long long *cost, int nblk, int mode)
{
    __shared__ unsigned long long s_lo, s_hi;
    __shared__ int s_ok[kMaxBlocks];
    const int tid = threadIdx.x;
    if (mode == 0) {
        if (tid == 0) { bal[kBalTag] = 0; bal[kBalSig] = 0; bal[kBalSig + 1] = 0; }
        return;
    }
    const unsigned long long total = (unsigned long long)(unsigned)chunk_start[kSigAt] | ((unsigned long long)(unsigned)chunk_start[kSigAt + 1] << 32);
    auto G = [&](int c) { const unsigned long long v = cost[c]; return v == ~0ull ? total : v; };
    // last piece boundary e in [s, kChunks] with G(e) <= G(s) + B
    auto reach = [&](int s0, unsigned long long B) {
        const unsigned long long lim = G(s0) + B;
        int lo = s0, hi = kChunks;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (G(mid) <= lim) lo = mid; else hi = mid - 1;
        }
        return lo;
    };
    auto feasible = [&](unsigned long long B) {
        int s0 = 0;
        for (int parts = 0; parts < nblk; ++parts) {
            const int e = reach(s0, B);
            if (e >= kChunks) return true;
            if (e == s0) return false; // (one group heavier than B)
            s0 = e;
        }
        return false;
    };
    if (tid == 0) { s_lo = total / (unsigned long long)nblk; s_hi = total; if (s_lo > 0) s_lo -= 1; }
    __syncthreads();
    for (int round = 0; round < 5; ++round) {
        const unsigned long long lo = s_lo, hi = s_hi, span = hi - lo;
        const unsigned long long B = lo + (span * (unsigned long long)(tid + 1)) / kMaxBlocks; // (tid = 511: hi, known feasible)
        s_ok[tid] = feasible(B) ? 1 : 0;
        __syncthreads();
        if (tid == 0) {
            int j = 0;
            while (j < kMaxBlocks - 1 && !s_ok[j]) ++j;
            s_hi = lo + (span * (unsigned long long)(j + 1)) / kMaxBlocks;
            s_lo = j == 0 ? lo : lo + (span * (unsigned long long)j) / kMaxBlocks;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const unsigned long long B = s_hi;
        int s0 = 0;
        bal[0] = 0;
        for (int wg = 0; wg < nblk; ++wg) {
            int e = s0 < kChunks ? reach(s0, B) : kChunks;
            if (wg == nblk - 1) e = kChunks;
            bal[wg + 1] = e;
            s0 = e;
        }
        bal[kBalTag] = nblk; bal[kBalSig] = chunk_start[kSigAt]; bal[kBalSig + 1] = chunk_start[kSigAt + 1];
    }
}

// workgroups of a launch of the frame kernel: one per CU (less the reserved ones), a multiple of eight
inline int pipe_blocks(int n_tiles, int reserved_cus)
{
    int n_cu = 256;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            cus > 0)
            n_cu = cus;
    }
    if (reserved_cus > 0 && n_cu - reserved_cus >= 8) n_cu -= reserved_cus;
    int nblk = n_tiles < n_cu ? n_tiles : n_cu;
    nblk = (nblk + 7) / 8 * 8; // xcd_contiguous deals whole eighths; surplus workgroups find an empty range and leave
    return nblk > kMaxBlocks ? kMaxBlocks : nblk;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct PipeLayout {
    size_t live[kMaxScales], shifts[kMaxScales], subcost[kMaxScales], tickets, globs, masks_bytes, hdrs[kMaxScales], recs[kMaxScales], wfrag[kMaxScales], chunks, ranks, costs, partial, slots, diag, balance, wmax, wexp, amax, total;
    int tiles_l, tiles_w, n_tiles;
};
inline PipeLayout layout_of(int n_views, int L, int W, int nl, int n_scales)
{
    PipeLayout w;
    w.tiles_l = (L + kTileL - 1) / kTileL;
    w.tiles_w = (W + kTileW - 1) / kTileW;
    w.n_tiles = w.tiles_l * w.tiles_w;
    size_t off = 0;
    for (int s = 0; s < kMaxScales; ++s) { // view masks and tickets first, contiguous: zeroed by ONE memset
        w.live[s] = off;
        off = align_up(off + (s < n_scales ? (size_t)w.n_tiles * 4 : 0), 256);
    }
    for (int s = 0; s < kMaxScales; ++s) {
        w.shifts[s] = off;
        off = align_up(off + (s < n_scales ? (size_t)w.n_tiles * 4 : 0), 256);
    }
    w.tickets = off; // (one per (run, matrix wave); sized for runs of one tile)
    off = align_up(off + (size_t)w.n_tiles * 8 * 4, 256);
    w.globs = off;
    off = align_up(off + (size_t)w.n_tiles * 4, 256);
    for (int s = 0; s < kMaxScales; ++s) { // (zeroed with the masks: the geometry pass adds the sub-tiles' costs up)
        w.subcost[s] = off;
        off = align_up(off + (s < n_scales ? (size_t)w.n_tiles * n_views * 4 : 0), 256);
    }
    w.masks_bytes = off;
    const size_t items = (size_t)w.n_tiles * nl * n_views;
    for (int s = 0; s < kMaxScales; ++s) {
        const bool on = s < n_scales;
        w.hdrs[s] = off;  off = align_up(off + (on ? items * kHdrBytes : 0), 256);
        w.recs[s] = off;  off = align_up(off + (on ? items * kTileBoxes * kRecBytes : 0), 256);
        w.wfrag[s] = off; off = align_up(off + (on ? (size_t)nl * 8 * kSteps * kWPlanes * 64 * 16 : 0), 256);
    }
    w.chunks = off;  off = align_up(off + (kChunks + 1) * sizeof(int), 256);
    w.ranks = off;   off = align_up(off + (kChunks + 1) * sizeof(int), 256);
    w.costs = off;   off = align_up(off + (kChunks + 1) * sizeof(unsigned long long), 256);
    w.partial = off; off = align_up(off + (size_t)kMaxBlocks * 2 * kRunTiles * 8 * 16 * 64 * sizeof(float), 256); // hand-off parts of the (first, last) run of a workgroup
    w.slots = off;   off = align_up(off + (size_t)pipe_blocks(w.n_tiles, 0) * kRunTiles * kMaxScales * contributions_of(n_views) * 8 * 16 * 64 * sizeof(float), 256); // contributions to the tiles of the run a workgroup is in
    w.diag = off;    off = align_up(off + (size_t)kMaxBlocks * 8 * sizeof(unsigned long long), 256);
    w.balance = off; off = align_up(off + kBalanceBytes, 256); // work-cut bounds per workgroup + the last launch's times (vfa_pipe_balance_f32)
    w.wmax = off;    off = align_up(off + (size_t)kMaxScales * kWmaxParts * sizeof(unsigned), 256); // fp16 split: partial maxima of |W| per scale,
    w.wexp = off;    off = align_up(off + (kMaxScales + 1) * sizeof(int), 256);                            // ... the weight exponents,
    w.amax = off;    off = align_up(off + (size_t)kMaxScales * kFallbackStats * sizeof(unsigned), 256); // ... feature statistics made here for callers that pass none
    w.total = off;
    return w;
}

// run length of a frame (vfa_pipe_seq.h: run_tiles_of); VFA_AMD_PIPE_RT = 1 | 2 | 4 overrides it for A/B measurements (tools/)
inline int frame_run_tiles(int n_views, int n_tiles, int n_scales, int nl)
{
    if (const char *e = getenv("VFA_AMD_PIPE_RT")) {
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4) return v;
    }
    return run_tiles_of(n_views, n_tiles, n_scales, nl, pipe_blocks(n_tiles, 0));
}

inline bool dims_ok(int n_views, int L, int W, int nl, int n_scales)
{
    return n_views >= 0 && L >= 0 && W >= 0 && nl >= 1 && n_scales >= 1 && n_scales <= kMaxScales;
}

} // namespace

extern "C" {
#ifdef VFA_CUTS_STAMPS
int vfa_debug_cut_stamps(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cut_stamps), sizeof(unsigned long long) * 16); }
#endif

size_t vfa_pipe_workspace_bytes(int n_views, int L, int W, int n_layers, int n_scales)
{
    if (!dims_ok(n_views, L, W, n_layers, n_scales)) return 0;
    return layout_of(n_views, L, W, n_layers, n_scales).total;
}

int vfa_pipe_workspace_layout(int n_views, int L, int W, int n_layers, int n_scales, size_t *offsets, int *tiles)
{
    if (!dims_ok(n_views, L, W, n_layers, n_scales) || !offsets || !tiles) return VFA_ERR_BAD_ARGUMENT;
    const PipeLayout lay = layout_of(n_views, L, W, n_layers, n_scales);
    for (int k = 0; k < kMaxScales; ++k) {
        offsets[4 * k + 0] = lay.live[k];
        offsets[4 * k + 1] = lay.hdrs[k];
        offsets[4 * k + 2] = lay.recs[k];
        offsets[4 * k + 3] = lay.wfrag[k];
    }
    offsets[12] = lay.tickets;
    offsets[17] = lay.globs;
    offsets[13] = lay.chunks;
    offsets[14] = lay.ranks;
    offsets[15] = lay.diag;
    offsets[16] = lay.total;
    offsets[18] = lay.balance;
    for (int k = 0; k < kMaxScales; ++k) offsets[19 + k] = lay.shifts[k]; // (ABI v8: 22 entries)
    tiles[0] = lay.tiles_l;
    tiles[1] = lay.tiles_w;
    tiles[2] = kWinSlots;
    tiles[4] = kWinSlots3;
    tiles[3] = kChunks;
    return 0;
}

int vfa_pipe_boxes_f32(const float *calibs, const float *grid, const float *z_layers, int n_layers, const float *corner_off, int n_views,
                       int L, int W, int conv_kind, float img_w, float img_h, float cmin, float cmax, int n_scales, const int *feat_hw,
                       int flags, void *workspace, size_t workspace_bytes, void *stream)
{
    const int terms = flags & VFA_FLAG_TERMS_MASK;
    if ((flags & ~VFA_FLAG_TERMS_MASK) || (terms != 0 && terms != 2 && terms != 3 && terms != 4 && terms != 6)) return VFA_ERR_BAD_ARGUMENT;
    if (!dims_ok(n_views, L, W, n_layers, n_scales) || conv_kind < 0 || conv_kind > 2 || !feat_hw) return VFA_ERR_BAD_ARGUMENT;
    if (n_views > 32) return VFA_ERR_UNSUPPORTED; // live-view masks are 32 bits wide
    const PipeLayout lay = layout_of(n_views, L, W, n_layers, n_scales);
    if (lay.n_tiles == 0 || n_views == 0) return 0;
    if ((long long)n_views * lay.n_tiles >= (1ll << 31) - 2) return VFA_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < lay.total) return VFA_ERR_BAD_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    RecordArgs a;
    a.g = BoxGeom{calibs, grid, z_layers, corner_off, conv_kind, img_w, img_h, cmin, cmax};
    a.n_views = n_views; a.L = L; a.W = W; a.tiles_w = lay.tiles_w; a.n_tiles = lay.n_tiles; a.n_scales = n_scales; a.nl = n_layers;
    a.win_slots = terms == 6 ? kWinSlots3 : kWinSlots; // (the kernel variant that will read these records)
    for (int k = 0; k < kMaxScales; ++k) {
        a.dims[k].Hf = k < n_scales ? feat_hw[2 * k] : 1;
        a.dims[k].Wf = k < n_scales ? feat_hw[2 * k + 1] : 1;
        if (a.dims[k].Hf <= 0 || a.dims[k].Wf <= 0 || a.dims[k].Hf > 65533 || a.dims[k].Wf > 65533) return VFA_ERR_BAD_ARGUMENT;
        // (tap positions of a direct item are byte offsets into one view's padded image, 32 bits)
        if ((unsigned long long)(a.dims[k].Hf + 2) * (a.dims[k].Wf + 2) * kSlotBytes >= (1ull << 32)) return VFA_ERR_UNSUPPORTED;
        a.live[k] = reinterpret_cast<unsigned *>(ws + lay.live[k]);
        a.shift[k] = reinterpret_cast<unsigned *>(ws + lay.shifts[k]);
        a.subcost[k] = reinterpret_cast<unsigned *>(ws + lay.subcost[k]);
        a.hdrs[k] = ws + lay.hdrs[k];
        a.recs[k] = ws + lay.recs[k];
    }
    a.globs = reinterpret_cast<unsigned *>(ws + lay.globs);
    const hipError_t e = zero_fill(ws, lay.masks_bytes, s); // view masks, tile tickets, counters (a kernel, not hipMemsetAsync: see vfa_geom.h)
    if (e != hipSuccess) return (int)e;
    const long long units = (long long)n_views * lay.n_tiles * n_layers; // (view, tile, layer)
    if ((units + 1) / 2 >= (1ll << 31)) return VFA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pipe_records_kernel, dim3((unsigned)((units + 1) / 2)), dim3(kWave), 0, s, a);
    return (int)hipGetLastError();
}

int vfa_pipe_cuts_f32(int n_views, int L, int W, int n_layers, int n_scales, const float *const *weights, int flags, void *workspace,
                      size_t workspace_bytes, void *stream)
{
    const int terms = flags & VFA_FLAG_TERMS_MASK;
    if ((flags & ~VFA_FLAG_TERMS_MASK) || (terms != 0 && terms != 2 && terms != 3 && terms != 4 && terms != 6)) return VFA_ERR_BAD_ARGUMENT;
    if (!dims_ok(n_views, L, W, n_layers, n_scales)) return VFA_ERR_BAD_ARGUMENT;
    if (n_views > 32) return VFA_ERR_UNSUPPORTED;
    const PipeLayout lay = layout_of(n_views, L, W, n_layers, n_scales);
    if (lay.n_tiles == 0 || n_views == 0) return 0;
    if (!workspace || workspace_bytes < lay.total) return VFA_ERR_BAD_ARGUMENT;
    hipStream_t s = (hipStream_t)stream;
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    CutArgs ca;
    for (int k = 0; k < kMaxScales; ++k) ca.live[k] = reinterpret_cast<const unsigned *>(ws + lay.live[k < n_scales ? k : 0]);
    ca.globs = reinterpret_cast<const unsigned *>(ws + lay.globs);
    for (int k = 0; k < kMaxScales; ++k) ca.subcost[k] = reinterpret_cast<const unsigned *>(ws + lay.subcost[k < n_scales ? k : 0]);
    ca.n_scales = n_scales; ca.n_tiles = lay.n_tiles; ca.n_views = n_views; ca.nl = n_layers;
    ca.rt = frame_run_tiles(n_views, lay.n_tiles, n_scales, n_layers);
    ca.chunk_start = reinterpret_cast<int *>(ws + lay.chunks);
    ca.chunk_rank = reinterpret_cast<int *>(ws + lay.ranks);
    ca.chunk_cost = reinterpret_cast<unsigned long long *>(ws + lay.costs);
    SplitArgs sa = {};
    ca.wmax_count = 0;
    if (weights) {
        sa.nl = n_layers;
        for (int k = 0; k < kMaxScales; ++k) {
            sa.w[k] = weights[k < n_scales ? k : 0];
            sa.out[k] = reinterpret_cast<uint4 *>(ws + lay.wfrag[k < n_scales ? k : 0]);
            if (!sa.w[k]) return VFA_ERR_BAD_ARGUMENT;
        }
        sa.wmax = reinterpret_cast<unsigned *>(ws + lay.wmax);
        sa.wexp = reinterpret_cast<int *>(ws + lay.wexp);
        sa.f16 = (terms == 0 || terms == 2) ? 1 : 0;
        if (sa.f16) ca.wmax_count = (long long)kC * kC * n_layers; // the partial maxima: spare blocks of the cuts launch
    }
    ca.split = sa;
    // the frame's masks and sub-tile costs in LDS where they fit beside the scan's 8 KB (static)
    const size_t stage_bytes = (size_t)n_scales * lay.n_tiles * (1 + (size_t)n_views) * 4;
    ca.staged = stage_bytes <= 144 * 1024 ? 1 : 0;
    if (ca.staged) {
        const hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(pipe_cuts_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        if (e0 != hipSuccess) return (int)e0;
    }
    hipLaunchKernelGGL(pipe_cuts_kernel, dim3(1 + (ca.wmax_count ? kWmaxParts * n_scales : 0)), dim3(1024), ca.staged ? stage_bytes : 0, s, ca);
    int st = (int)hipGetLastError();
    if (st) return st;
    if (weights) {
        hipLaunchKernelGGL(pipe_split_weight_kernel, dim3(8 * kSteps * 64 / 256, n_scales * n_layers), dim3(256), 0, s, sa);
        st = (int)hipGetLastError();
    }
    return st;
}

int vfa_pipe_balance_f32(int n_views, int L, int W, int n_layers, int n_scales, int reserved_cus, int mode, void *workspace,
                         size_t workspace_bytes, void *stream)
{
    if (!dims_ok(n_views, L, W, n_layers, n_scales) || (mode != 0 && mode != 1)) return VFA_ERR_BAD_ARGUMENT;
    const PipeLayout lay = layout_of(n_views, L, W, n_layers, n_scales);
    if (lay.n_tiles == 0 || n_views == 0) return 0;
    if (!workspace || workspace_bytes < lay.total) return VFA_ERR_BAD_ARGUMENT;
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    hipLaunchKernelGGL(pipe_balance_kernel, dim3(1), dim3(kMaxBlocks), 0, (hipStream_t)stream, reinterpret_cast<int *>(ws + lay.balance),
                       reinterpret_cast<const int *>(ws + lay.chunks), reinterpret_cast<const unsigned long long *>(ws + lay.costs),
                       pipe_blocks(lay.n_tiles, reserved_cus), mode);
    return (int)hipGetLastError();
}

int vfa_pipe_records_f32(const float *calibs, const float *grid, const float *z_layers, int n_layers, const float *corner_off,
                         int n_views, int L, int W, int conv_kind, float img_w, float img_h, float cmin, float cmax, int n_scales,
                         const int *feat_hw, const float *const *weights, int flags, void *workspace, size_t workspace_bytes, void *stream)
{
    const int st = vfa_pipe_boxes_f32(calibs, grid, z_layers, n_layers, corner_off, n_views, L, W, conv_kind, img_w, img_h, cmin, cmax,
                                      n_scales, feat_hw, flags, workspace, workspace_bytes, stream);
    if (st) return st;
    return vfa_pipe_cuts_f32(n_views, L, W, n_layers, n_scales, weights, flags, workspace, workspace_bytes, stream);
}

int vfa_pipe_collapse_relu_sum_f32(const float *const *integrals, const unsigned *const *feat_absmax, const float *const *biases,
                                   void *workspace, size_t workspace_bytes, float *out, int n_views, int L, int W, int n_layers,
                                   int n_scales, const int *feat_hw, int accumulate, int flags, void *stream)
{
    const int terms = flags & VFA_FLAG_TERMS_MASK, reserved_cus = (flags >> 8) & 0xff;
    const int debug = ((flags >> 16) & 0xfff) | ((flags & VFA_FLAG_DUMP_VOX) ? kDbgDumpVox : 0);
    if (debug && terms != 0 && terms != 2) return VFA_ERR_BAD_ARGUMENT; // (the diagnostic build exists for the default arithmetic only)
    if (flags & ~(VFA_FLAG_TERMS_MASK | 0xfffff00 | VFA_FLAG_DUMP_VOX)) return VFA_ERR_BAD_ARGUMENT;
    if (!dims_ok(n_views, L, W, n_layers, n_scales) || !feat_hw || !integrals ||
        (terms != 0 && terms != 2 && terms != 3 && terms != 4 && terms != 6))
        return VFA_ERR_BAD_ARGUMENT;
    const bool f16 = terms == 0 || terms == 2;
    if (n_views > 32) return VFA_ERR_UNSUPPORTED;
    const PipeLayout lay = layout_of(n_views, L, W, n_layers, n_scales);
    if (lay.n_tiles == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (n_views == 0) {
        if (!accumulate) return (int)zero_fill(out, (size_t)L * W * kC * sizeof(float), s);
        return 0;
    }
    if (!workspace || workspace_bytes < lay.total || !out) return VFA_ERR_BAD_ARGUMENT;
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    PipeArgs a;
    for (int k = 0; k < kMaxScales; ++k) {
        const int q = k < n_scales ? k : 0;
        a.sc[k].integral = integrals[q];
        a.sc[k].bias = biases ? biases[q] : nullptr;
        a.sc[k].wfrag = reinterpret_cast<const uint4 *>(ws + lay.wfrag[q]);
        a.sc[k].live = reinterpret_cast<const unsigned *>(ws + lay.live[q]);
        a.sc[k].shift = reinterpret_cast<const unsigned *>(ws + lay.shifts[q]);
        a.sc[k].hdrs = ws + lay.hdrs[q];
        a.sc[k].recs = ws + lay.recs[q];
        a.sc[k].Hf = feat_hw[2 * q];
        a.sc[k].Wf = feat_hw[2 * q + 1];
        if (!a.sc[k].integral) return VFA_ERR_BAD_ARGUMENT;
        a.sc[k].amax = nullptr; a.sc[k].amax_n = 0;
    }
    if (f16) {
        // the scale of the fp16 split: what the integral-image call left (feat_absmax), or one pass over the integral images here
        for (int k = 0; k < n_scales; ++k) {
            if (feat_absmax && feat_absmax[k]) {
                a.sc[k].amax = feat_absmax[k];
                a.sc[k].amax_n = (int)feature_stats_count(n_views, kC, a.sc[k].Hf);
            } else {
                unsigned *dst = reinterpret_cast<unsigned *>(ws + lay.amax) + (size_t)k * kFallbackStats;
                const int st = integral_absmax_folded(a.sc[k].integral, dst, n_views, kC, a.sc[k].Hf, a.sc[k].Wf, kFallbackStats, &a.sc[k].amax_n, stream);
                if (st) return st;
                a.sc[k].amax = dst;
            }
        }
        for (int k = n_scales; k < kMaxScales; ++k) { a.sc[k].amax = a.sc[0].amax; a.sc[k].amax_n = a.sc[0].amax_n; }
    }
    a.wexp = reinterpret_cast<const int *>(ws + lay.wexp);
    a.n_scales = n_scales; a.n_views = n_views; a.nl = n_layers; a.L = L; a.W = W; a.tiles_w = lay.tiles_w; a.n_tiles = lay.n_tiles;
    a.rt = frame_run_tiles(n_views, lay.n_tiles, n_scales, n_layers); // (the same for every launch size: the cuts were made once)
    a.out = out; a.accumulate = accumulate;
    a.chunk_start = reinterpret_cast<const int *>(ws + lay.chunks);
    a.chunk_rank = reinterpret_cast<const int *>(ws + lay.ranks);
    a.partial = reinterpret_cast<float *>(ws + lay.partial);
    a.slots = reinterpret_cast<float *>(ws + lay.slots);
    a.n_contrib = contributions_of(n_views);
    a.tickets = reinterpret_cast<unsigned *>(ws + lay.tickets);
    a.diag = reinterpret_cast<unsigned long long *>(ws + lay.diag);
    a.debug = debug;
    a.balance = reinterpret_cast<int *>(ws + lay.balance);
    const int nblk = pipe_blocks(lay.n_tiles, reserved_cus);
    if (debug & kDbgDumpVox) { // (one view, one scale, one layer: `out` has room for exactly one set of voxel features)
        if (n_views != 1 || n_scales != 1 || n_layers != 1 || accumulate) return VFA_ERR_BAD_ARGUMENT;
        const hipError_t e0 = hipMemsetAsync(out, 0, (size_t)L * W * kC * sizeof(float), s);
        if (e0 != hipSuccess) return (int)e0;
    }
    // (the tickets are clear: zeroed with the masks by the geometry call, and put back by the last arriver of every earlier launch)
    const bool small = n_views <= 2 && a.rt == 1; // (groups of at most two sub-tiles: the four-step phase)
    const bool rt1 = n_views > 2 && a.rt == 1;    // (tile by tile: the tile's sum waits in a slot of the workspace between its groups)
    if (debug && small && !(debug & kDbgDumpVox)) // (diagnostic build of the default arithmetic)
        hipLaunchKernelGGL((pipe_kernel<2, true, true>), dim3(nblk), dim3(threads_of(2)), 0, s, a);
    else if (debug && rt1 && !(debug & kDbgDumpVox))
        hipLaunchKernelGGL((pipe_kernel<2, true, false, true>), dim3(nblk), dim3(threads_of(2)), 0, s, a);
    else if (debug)
        hipLaunchKernelGGL((pipe_kernel<2, true>), dim3(nblk), dim3(threads_of(2)), 0, s, a);
    else if (terms == 4)
        hipLaunchKernelGGL((pipe_kernel<4, false>), dim3(nblk), dim3(threads_of(4)), 0, s, a);
    else if (terms == 6)
        hipLaunchKernelGGL((pipe_kernel<6, false>), dim3(nblk), dim3(threads_of(6)), 0, s, a);
    else if (terms == 3 && small)
        hipLaunchKernelGGL((pipe_kernel<3, false, true>), dim3(nblk), dim3(threads_of(3)), 0, s, a);
    else if (terms == 3)
        hipLaunchKernelGGL((pipe_kernel<3, false>), dim3(nblk), dim3(threads_of(3)), 0, s, a);
    else if (small)
        hipLaunchKernelGGL((pipe_kernel<2, false, true>), dim3(nblk), dim3(threads_of(2)), 0, s, a);
    else if (rt1)
        hipLaunchKernelGGL((pipe_kernel<2, false, false, true>), dim3(nblk), dim3(threads_of(2)), 0, s, a);
    else if (a.rt == 2)
        hipLaunchKernelGGL((pipe_kernel<2, false, false, false, 2>), dim3(nblk), dim3(threads_of(2)), 0, s, a);
    else if (a.rt == 4)
        hipLaunchKernelGGL((pipe_kernel<2, false, false, false, 4>), dim3(nblk), dim3(threads_of(2)), 0, s, a);
    else
        hipLaunchKernelGGL((pipe_kernel<2, false>), dim3(nblk), dim3(threads_of(2)), 0, s, a);
    return (int)hipGetLastError();
}

} // extern "C"
