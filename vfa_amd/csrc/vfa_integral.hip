// vfa_integral.hip -- integral images of all the feature scales of a frame in one launch pair.
//
// Reference: vfa/model/vfa_op.py:110, 172-173 (features.cumsum(-1).cumsum(-2), once per scale and camera) and, with the
// affine variant, vfa/model/vfanet.py:72-74 (GroupNorm affine + ReLU of the lateral branch) -- SURVEY.md section 8 a1 / f3.
//
// Arithmetic is ATen's CPU cumsum, operation for operation: a double accumulator per (row, channel) running LEFT TO RIGHT,
// rounded to fp32 at every element; then a double accumulator per (column, channel) running TOP TO BOTTOM over those fp32
// values, rounded at every element.  No partial sum is ever re-associated (a tree / look-back scan would round differently
// whenever a double sum is inexact), so the result is bit-identical to the reference for any input.
//
// Why one launch pair per FRAME: the stride-16 and stride-32 maps are too small to fill 256 CUs (630 + 161 wave-rows); in
// their own launches they cost ~30 us per scale of mostly launch latency and tail.  Batched, they ride along with the
// stride-8 map.  Layout and borders as in vfa_kernels.hip (channels-last, one zero pixel all round).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vfa_hip.h"
#include "vfa_geom.h"
#include "vfa_split.h"

namespace {
using namespace vfa_dev;

constexpr int kMaxMaps = 4;
constexpr int kChunk = 32;    // columns staged per step

struct MapDesc {
    const float *feat;   // (n_views, C, H, W)
    float *out;          // (n_views, H + 2, W + 2, C)
    const float *scale;  // (n_views, C) or NULL
    const float *shift;
    int H, W;
    unsigned row_blocks; // H * (C / 64) * n_views
    unsigned long long col_vecs; // n_views * (W + 2) * C / 4
    unsigned *absmax;    // (row_blocks) or NULL: max |feature| (fp32 bits, sign cleared) over the wave-row of every block of pass 1
};
struct MapArgs {
    MapDesc m[kMaxMaps];
    int n_maps, n_views, C;
};

// pass 1: one wave = 64 channels of one image row, 32-column chunks (the scheme of integral_rows_kernel in vfa_kernels.hip, which
// measured faster than a variant with register scans and loads one chunk ahead: 8.4 KiB of LDS keep every wave-row of the
// frame resident at once).  NCHW loads: 8 lanes x float4 = one 128-byte row piece of a channel; the tile is scanned in place
// with lane = channel (pitch 33: conflict-free); channels-last stores: 16 lanes x float4 = the 256 B of a pixel.
template <bool AFFINE>
__global__ __launch_bounds__(kWave) void rows_batched_kernel(MapArgs a)
{
    __shared__ __align__(16) float tile[kWave][kChunk + 1];
    const int lane = threadIdx.x;
    unsigned b = blockIdx.x;
    int mi = 0;
    while (mi + 1 < a.n_maps && b >= a.m[mi].row_blocks) { b -= a.m[mi].row_blocks; ++mi; }
    const MapDesc &m = a.m[mi];
    const int H = m.H, W = m.W, C = a.C, cblocks = C / kWave;
    const int y = (int)(b % (unsigned)H);
    const unsigned rest = b / (unsigned)H;
    const int c0 = (int)(rest % (unsigned)cblocks) * kWave, v = (int)(rest / (unsigned)cblocks);
    const size_t plane = (size_t)H * W;
    const float *src = m.feat + ((size_t)v * C + c0) * plane + (size_t)y * W;
    float *dst = m.out + (((size_t)v * (H + 2) + (y + 1)) * (W + 2)) * C + c0; // padded row y + 1, padded column 0
    dst[lane] = 0.0f;                       // left border
    dst[(size_t)(W + 1) * C + lane] = 0.0f; // right border
    float sa = 1.0f, sb = 0.0f;
    if (AFFINE) {
        sa = m.scale[(size_t)v * C + c0 + lane];
        sb = m.shift[(size_t)v * C + c0 + lane];
    }
    auto act = [&](float x) {
        if (!AFFINE) return x;
        float t = x * sa;
        t = t + sb;
        return (t < 0.0f) ? 0.0f : t; // NaN stays NaN
    };
    const int q = lane & 7, cq = lane & 15;
    double acc = 0.0;
    unsigned amax = 0u; // largest |feature| of this lane's channel row (what the fp16 split of the fused kernels is scaled by: vfa_split.h)
    for (int x0 = 0; x0 < W; x0 += kChunk) {
        const int nx = min(kChunk, W - x0); // a multiple of 4
        if (4 * q < nx) {
            for (int r = lane >> 3; r < kWave; r += 8) {
                const float4 t = *reinterpret_cast<const float4 *>(src + (size_t)r * plane + x0 + 4 * q);
                tile[r][4 * q + 0] = t.x; tile[r][4 * q + 1] = t.y; tile[r][4 * q + 2] = t.z; tile[r][4 * q + 3] = t.w;
            }
        }
        __syncthreads();
        for (int k = 0; k < nx; ++k) {
            const float f = act(tile[lane][k]);
            amax = max(amax, __float_as_uint(f) & 0x7fffffffu);
            acc += (double)f;
            tile[lane][k] = (float)acc;
        }
        __syncthreads();
        for (int k = lane >> 4; k < nx; k += 4) {
            const float4 t = make_float4(tile[4 * cq + 0][k], tile[4 * cq + 1][k], tile[4 * cq + 2][k], tile[4 * cq + 3][k]);
            *reinterpret_cast<float4 *>(dst + (size_t)(x0 + k + 1) * C + 4 * cq) = t;
        }
        __syncthreads();
    }
    if (m.absmax) {
        amax = wave_max_u32(amax);
        if (lane == 0) m.absmax[b] = amax;
    }
}

// pass 1 for a CHANNELS-LAST input (n_views, H, W, C) -- the lateral convolution of vfa_lateral.hip writes that --: lane = channel,
// the row is walked left to right with the loads eight pixels ahead; every load and store of the wave is one 256-byte piece of a
// pixel.  No LDS, no transpose.  Same arithmetic as above, element for element.
template <bool AFFINE>
__global__ __launch_bounds__(kWave) void rows_hwc_kernel(MapArgs a)
{
    const int lane = threadIdx.x;
    unsigned b = blockIdx.x;
    int mi = 0;
    while (mi + 1 < a.n_maps && b >= a.m[mi].row_blocks) { b -= a.m[mi].row_blocks; ++mi; }
    const MapDesc &m = a.m[mi];
    const int H = m.H, W = m.W, C = a.C, cblocks = C / kWave;
    const int y = (int)(b % (unsigned)H);
    const unsigned rest = b / (unsigned)H;
    const int c0 = (int)(rest % (unsigned)cblocks) * kWave, v = (int)(rest / (unsigned)cblocks);
    const float *src = m.feat + (((size_t)v * H + y) * W) * C + c0 + lane;
    float *dst = m.out + (((size_t)v * (H + 2) + (y + 1)) * (W + 2)) * C + c0 + lane;
    dst[0] = 0.0f;
    dst[(size_t)(W + 1) * C] = 0.0f;
    float sa = 1.0f, sb = 0.0f;
    if (AFFINE) {
        sa = m.scale[(size_t)v * C + c0 + lane];
        sb = m.shift[(size_t)v * C + c0 + lane];
    }
    auto act = [&](float x) {
        if (!AFFINE) return x;
        float t = x * sa;
        t = t + sb;
        return (t < 0.0f) ? 0.0f : t; // NaN stays NaN
    };
    constexpr int U = 8;
    double acc = 0.0;
    unsigned amax = 0u;
    int x = 0;
    for (; x + U <= W; x += U) {
        float t[U];
#pragma unroll
        for (int k = 0; k < U; ++k) t[k] = src[(size_t)(x + k) * C];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const float f = act(t[k]);
            amax = max(amax, __float_as_uint(f) & 0x7fffffffu);
            acc += (double)f;
            dst[(size_t)(x + k + 1) * C] = (float)acc;
        }
    }
    for (; x < W; ++x) {
        const float f = act(src[(size_t)x * C]);
        amax = max(amax, __float_as_uint(f) & 0x7fffffffu);
        acc += (double)f;
        dst[(size_t)(x + 1) * C] = (float)acc;
    }
    if (m.absmax) {
        amax = wave_max_u32(amax);
        if (lane == 0) m.absmax[b] = amax;
    }
}

// pass 2: cumsum along H, in place, plus the zero top and bottom border rows.  One thread owns four channels of one padded
// column of one view of one map; its loads do not depend on the running sum and are issued eight rows ahead.
__global__ __launch_bounds__(256) void cols_batched_kernel(MapArgs a, unsigned long long total)
{
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    // columns are taken in the REVERSE of the order the row pass wrote them: what it wrote last is still in the memory-side cache
    i = total - 1 - i;
    int mi = 0;
    while (mi + 1 < a.n_maps && i >= a.m[mi].col_vecs) { i -= a.m[mi].col_vecs; ++mi; }
    const MapDesc &m = a.m[mi];
    const int H = m.H;
    const size_t row_vecs = (size_t)(m.W + 2) * a.C / 4;
    const size_t v = i / row_vecs, r = i % row_vecs;
    float4 *p = reinterpret_cast<float4 *>(m.out) + v * (size_t)(H + 2) * row_vecs + r;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    p[0] = zero;
    p[(size_t)(H + 1) * row_vecs] = zero;
    p += row_vecs; // first interior row
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    auto step = [&](float4 &t) {
        a0 += (double)t.x; t.x = (float)a0;
        a1 += (double)t.y; t.y = (float)a1;
        a2 += (double)t.z; t.z = (float)a2;
        a3 += (double)t.w; t.w = (float)a3;
    };
    constexpr int U = 8;
    int y = 0;
    for (; y + U <= H; y += U) {
        float4 t[U];
#pragma unroll
        for (int k = 0; k < U; ++k) t[k] = p[(size_t)(y + k) * row_vecs];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            step(t[k]);
            p[(size_t)(y + k) * row_vecs] = t[k];
        }
    }
    for (; y < H; ++y) {
        float4 t = p[(size_t)y * row_vecs];
        step(t);
        p[(size_t)y * row_vecs] = t;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// ONE pass: both cumsums of a (map, view, 16-channel block) inside one 512-thread workgroup, the map read once and the integral
// image written once (the two-pass form above moves every byte twice: 544 MB per bench frame against 272 MB).
//
// Both scans keep ATen's order -- a double accumulator running left to right along every row, rounded to fp32 at every element,
// then a double accumulator running top to bottom over those fp32 values --, so the only freedom is WHICH chains run side by side:
//   phase R  thread (row r of 32, channel c of 16) runs the row scan of its (row, channel) over a strip of 32 columns -- 512
//            independent chains -- and leaves the rounded fp32 values in LDS;
//   phase C  thread (column x of 32, channel c of 16) picks its column of that tile up: the column accumulator (a double per
//            (column, channel), kept in LDS between the row batches) takes the 32 rows in order, each sum is rounded in place;
//   phase S  the finished tile goes to the channels-last image 16 bytes per lane (4 lanes = the 64 contiguous bytes of a
//            pixel's channel block).
// The loads of the next strip are issued (into registers) in front of phase R of this one and land under the phases; rows are
// taken 32 at a time (a batch), the row carry of a thread lives in its registers across the strips of a batch.  Bit-identical to
// the two-pass kernels by construction (the same operations in the same order per chain), checked by the same tests.
// Statistics (absmax): the workgroup of channel block cb leaves its maximum in entry (view, row cb & 3, block cb >> 2) of the
// two-pass layout and zeroes the entries of rows cb & 3 + 4 k: the consumers take the maximum over all entries of a map.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int kOpRows = 32, kOpCols = 32, kOpCh = 16, kOpThreads = kOpRows * kOpCh;
constexpr int kOpRowPitch = kOpCols * kOpCh + 16; // floats per tile row: the padding puts rows r and r + 1 into different halves of the banks
struct OnePassArgs {
    MapDesc m[kMaxMaps];
    unsigned unit_end[kMaxMaps]; // cumulative units (view, channel block, column part) per map
    int split[kMaxMaps];         // 0: one unit per (view, channel block); s > 0: two -- part 0 takes the strips [0, s), part 1 the rest
    int n_maps, n_views, C;
};
// A workgroup barrier that orders LDS accesses only: `__syncthreads()` also drains the vector-memory queue (vmcnt(0)) -- here the
// loads of the next strip and the stores of the last one, a full memory round trip at every one of the four barriers of a strip.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// (launch bounds: FOUR waves per SIMD = two of these 512-thread workgroups per CU.  The LDS footprint was sized for two (79.9 KB on the
// bench frame), but at 136 registers only one fitted: round 5's second session capped the kernel at 128 -- 3 spilled registers -- and
// the three maps of the bench frame take 87 instead of 97 us: a unit is a chain of dependent fp64 scans, the second workgroup fills its
// stalls.)
template <bool AFFINE>
__global__ __launch_bounds__(kOpThreads, 4) void integral_onepass_kernel(OnePassArgs a)
{
    extern __shared__ __align__(16) unsigned char op_lds[];
    float *tile = reinterpret_cast<float *>(op_lds);                                  // [32 rows][kOpRowPitch]
    double *colacc = reinterpret_cast<double *>(op_lds + kOpRows * kOpRowPitch * 4);  // [W][16]
    const int tid = threadIdx.x;
    unsigned u = blockIdx.x;
    int mi = 0;
    while (mi + 1 < a.n_maps && u >= a.unit_end[mi]) ++mi;
    if (mi > 0) u -= a.unit_end[mi - 1];
    const MapDesc &m = a.m[mi];
    const int H = m.H, W = m.W, C = a.C, cblocks = C / kOpCh;
    // A wide map is cut into TWO units by columns (the element costs four f32 <-> f64 conversions and two fp64 adds, ~80 cycles of
    // a wave: one workgroup per (view, channel block) keeps 112 CUs busy for 80 us on the stride-8 maps of the bench frame while the
    // rest of the chip idles).  The row scan of the right part needs the row sums of the left one: it runs the bare accumulation --
    // conversion + add, no rounding, no LDS, a third of the work -- over the left part's strips first (those reads hit L2: the left
    // unit streams the same lines).  No workgroup waits for another.
    const int split = a.split[mi], part = split ? (int)(u & 1u) : 0;
    if (split) u >>= 1;
    const int v = (int)(u / (unsigned)cblocks), cb = (int)(u % (unsigned)cblocks), c0 = cb * kOpCh;
    const int r = tid >> 4, c = tid & 15;     // phase R: row of the batch, channel; phase C: column of the strip (r), channel
    const size_t plane = (size_t)H * W;
    float *out_v = m.out + (size_t)v * (H + 2) * (W + 2) * C + c0; // padded pixel (0, 0) of the view, first channel of the block
    const int n_strips = (W + kOpCols - 1) / kOpCols;
    const int s_first = part ? split : 0, s_last = (split && !part) ? split : n_strips; // this unit's strips [s_first, s_last)
    const int xa = s_first * kOpCols, xb = min(W, s_last * kOpCols);                     // ... = its columns [xa, xb)
    {   // zero border: rows 0 and H + 1 over the unit's padded columns, column 0 / W + 1 of the rows in between
        float *o = out_v + c;
        const int pa = part ? xa + 1 : 0, pb = (split && !part) ? xb + 1 : W + 2;
        for (int x = pa + r; x < pb; x += kOpRows) { o[(size_t)x * C] = 0.0f; o[((size_t)(H + 1) * (W + 2) + x) * C] = 0.0f; }
        for (int y = r; y < H; y += kOpRows) {
            if (!part) o[(size_t)(y + 1) * (W + 2) * C] = 0.0f;
            if (!split || part) o[((size_t)(y + 1) * (W + 2) + W + 1) * C] = 0.0f;
        }
    }
    for (int i = tid; i < (xb - xa) * kOpCh; i += kOpThreads) colacc[i] = 0.0;
    float sa = 1.0f, sb = 0.0f;
    if (AFFINE) { sa = m.scale[(size_t)v * C + c0 + c]; sb = m.shift[(size_t)v * C + c0 + c]; }
    auto act = [&](float x) {
        if (!AFFINE) return x;
        float t = x * sa;
        t = t + sb;
        return (t < 0.0f) ? 0.0f : t; // NaN stays NaN
    };
    unsigned amax = 0u;
    // Every global access of the loop is 16 bytes per lane (a 4-byte-per-lane store instruction moves 256 bytes and costs the
    // issuing wave as much as a 1 KiB one: with them phase C alone took 70 us for the stride-8 map).
    // The raw strip of a batch in registers: 8 x float4 along x.  Rows and columns beyond the map read clamped and are never used.
    // (Row-contiguous NCHW maps only: on the channels-last output of vfa_lateral_conv_f32 this kernel measured 154 us per bench
    // frame against 97 us for rows_hwc_kernel + cols_batched_kernel -- there every access of the two-pass kernels is a full
    // 256-byte piece already -- so vfa_integral_images_hwc_f32 keeps the two passes.)
    auto load_strip = [&](float4 (&buf)[kOpCols / 4], int y0, int x0) {
        {
            // eight lanes = the 128 bytes of one (row, channel) of the strip, eight such lines per wave instruction: with a whole
            // line per LANE (64 lines per instruction, each touched by eight instructions) the address path of the CU, not its
            // arithmetic, set the pace of the kernel.  The 64 lines of a wave's eight loads are the lines of the WAVE ITSELF (line =
            // thread of phase R): see the transpose below
            const int x4 = tid & 7;
#pragma unroll
            for (int j = 0; j < kOpCols / 4; ++j) {
                const int line = (tid & ~63) + 8 * j + ((tid & 63) >> 3), row = line >> 4, ch = line & 15;
                buf[j] = *reinterpret_cast<const float4 *>(m.feat + ((size_t)v * C + c0 + ch) * plane + (size_t)min(y0 + row, H - 1) * W +
                                                           min(x0 + 4 * x4, W - 4)); // (W is a multiple of 4 on this path)
            }
        }
    };
    // Order inside a strip: phase R consumes the raw strip, THEN the next strip is requested into the same registers (it lands under
    // phases C and S), then C, then S.  The wait for a strip at the head of phase R is written by hand: vmcnt counts in order, behind
    // the strip's eight loads sit exactly the eight stores of the last phase S (unconditional: rows and columns beyond the tile
    // repeat the last valid ones), so `vmcnt(8)` leaves those stores draining under this strip.  Left to itself the compiler
    // waited for nearly everything in flight in front of every element of phase R: a memory round trip per strip.
    float4 cur[kOpCols / 4];
    load_strip(cur, 0, 0);
    __syncthreads(); // colacc is cleared
    bool stores_behind = false; // the last thing this thread issued were the eight stores of a phase S (behind the loads of `cur`)
    for (int y0 = 0; y0 < H; y0 += kOpRows) {
        const int rows = min(kOpRows, H - y0);
        double racc = 0.0;
        for (int s = 0; s < s_last; ++s) {
            const int x0 = s * kOpCols, nx = min(kOpCols, W - x0);
            const bool carry_only = s < s_first; // (the right part, over the left part's strips: the row sums only)
            const bool last_strip = s + 1 == s_last;
            const bool more = !last_strip || y0 + kOpRows < H;
            if (stores_behind) __builtin_amdgcn_s_waitcnt(0x0f78); // vmcnt(8)
            else __builtin_amdgcn_s_waitcnt(0x0f70);               // vmcnt(0)
            float *trow = tile + r * kOpRowPitch + c;
            {
                // row-contiguous map: the strip as it was loaded -- thread (line, x quad) -- goes through LDS (512 lines of 33 floats:
                // conflict-free both ways, the size of the tile) and comes back as this thread's own line (r, c) = line tid.  A wave
                // loads ITS OWN 64 lines and their 64 x 33 floats are the four tile rows it writes in phase R: the transpose is private
                // to the wave -- no barrier in it (two barriers per strip fewer: 90.6 -> 87.3 us per bench frame in one process, round
                // 6); the barrier at the end of a strip keeps the other waves' phase S of the last tile in front of it.
                float *raw = tile;
#pragma unroll
                for (int j = 0; j < kOpCols / 4; ++j) {
                    float *p = raw + ((tid & ~63) + 8 * j + ((tid & 63) >> 3)) * 33 + 4 * (tid & 7);
                    p[0] = cur[j].x; p[1] = cur[j].y; p[2] = cur[j].z; p[3] = cur[j].w;
                }
#pragma unroll
                for (int j = 0; j < kOpCols / 4; ++j)
                    cur[j] = make_float4(raw[tid * 33 + 4 * j], raw[tid * 33 + 4 * j + 1], raw[tid * 33 + 4 * j + 2], raw[tid * 33 + 4 * j + 3]);
            }
            auto raw_of = [&](int k) {
                const float4 q = cur[k / 4];
                return (k & 3) == 0 ? q.x : (k & 3) == 1 ? q.y : (k & 3) == 2 ? q.z : q.w;
            };
            if (carry_only) {
                if (r < rows) {
#pragma unroll
                    for (int k = 0; k < kOpCols; ++k) racc += (double)act(raw_of(k)); // (a left strip is always full)
                }
                if (more) load_strip(cur, last_strip ? y0 + kOpRows : y0, last_strip ? 0 : x0 + kOpCols);
                stores_behind = false;
                continue;
            }
            // ---- phase R: row scan of (row r, channel c) over the strip, in place (full strips: no per-element branch)
            if (r < rows) {
                if (nx == kOpCols) {
#pragma unroll
                    for (int k = 0; k < kOpCols; ++k) {
                        const float f = act(raw_of(k));
                        amax = max(amax, __float_as_uint(f) & 0x7fffffffu);
                        racc += (double)f;
                        trow[k * kOpCh] = (float)racc;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < kOpCols; ++k) {
                        if (k < nx) {
                            const float f = act(raw_of(k));
                            amax = max(amax, __float_as_uint(f) & 0x7fffffffu);
                            racc += (double)f;
                            trow[k * kOpCh] = (float)racc;
                        }
                    }
                }
            }
            // the next strip (of this batch, or the first of the next one): requested now, into the registers just consumed
            if (more) load_strip(cur, last_strip ? y0 + kOpRows : y0, last_strip ? 0 : x0 + kOpCols);
            lds_barrier();
            // ---- phase C: column scan of (column x0 + r, channel c) over the rows of the batch, in place
            if (r < nx) {
                double acc = colacc[(x0 - xa + r) * kOpCh + c];
                float *tcol = tile + r * kOpCh + c;
                if (rows == kOpRows) {
#pragma unroll
                    for (int k = 0; k < kOpRows; ++k) {
                        acc += (double)tcol[k * kOpRowPitch];
                        tcol[k * kOpRowPitch] = (float)acc;
                    }
                } else {
                    for (int k = 0; k < rows; ++k) {
                        acc += (double)tcol[k * kOpRowPitch];
                        tcol[k * kOpRowPitch] = (float)acc;
                    }
                }
                colacc[(x0 - xa + r) * kOpCh + c] = acc;
            }
            lds_barrier();
            // ---- phase S: the finished tile to the channels-last image, 16 bytes per lane, always eight stores per thread
#pragma unroll
            for (int j = 0; j < kOpCols / 4; ++j) {
                const int idx = tid + kOpThreads * j, row = min(idx >> 7, rows - 1), x = min((idx >> 2) & 31, nx - 1), c4 = idx & 3;
                *reinterpret_cast<float4 *>(out_v + ((size_t)(y0 + row + 1) * (W + 2) + (x0 + x + 1)) * C + 4 * c4) =
                    *reinterpret_cast<const float4 *>(tile + row * kOpRowPitch + x * kOpCh + 4 * c4);
            }
            stores_behind = more; // (without a next strip the stores are the only thing in flight; nobody waits for them)
            lds_barrier();
        }
    }
    if (m.absmax) {
        __shared__ unsigned s_max[kOpThreads / kWave];
        amax = wave_max_u32(amax);
        if ((tid & 63) == 0) s_max[tid >> 6] = amax;
        __syncthreads();
        if (tid == 0) {
            unsigned mx = 0u;
            for (int k = 0; k < kOpThreads / kWave; ++k) mx = max(mx, s_max[k]);
            // entry (view, row y, 64-channel block) of the two-pass layout: the units of a channel block own the rows q, q + 8, ... with
            // q = 2 (cb & 3) + part (the maximum of a right part covers its own columns only: the left ones are its partner's)
            unsigned *e = m.absmax + ((size_t)v * (C / kWave) + (cb >> 2)) * H;
            const int q = 2 * (cb & 3) + part;
            for (int y = q; y < H; y += 8) e[y] = y == q ? mx : 0u;
            if (!split) for (int y = q + 1; y < H; y += 8) e[y] = 0u;
        }
    }
}

// The same statistic from a finished integral image (callers that have no feature map at hand, odd shapes): the feature value
// of a pixel is the second difference of its four integral-image neighbours -- exact up to the rounding of the integral image
// (~6e-8 of its largest value), which is all a power-of-two scale with a factor 32 of headroom needs.  One wave per entry =
// (view, row, 64-channel block), the layout of pass 1 above.
__global__ __launch_bounds__(kWave) void integral_absmax_kernel(const float *integral, unsigned *absmax, int n_views, int C, int H, int W,
                                                                unsigned units)
{
    const int lane = threadIdx.x, cblocks = (C + kWave - 1) / kWave;
    unsigned amax = 0u;
    for (unsigned b = blockIdx.x; b < units; b += gridDim.x) { // (one entry per BLOCK: a grid smaller than `units` folds them)
        const int y = (int)(b % (unsigned)H);
        const unsigned rest = b / (unsigned)H;
        const int c = (int)(rest % (unsigned)cblocks) * kWave + lane, v = (int)(rest / (unsigned)cblocks);
        if (c < C) {
            const float *up = integral + (((size_t)v * (H + 2) + y) * (W + 2)) * C + c, *dn = up + (size_t)(W + 2) * C;
            float u0 = up[0], d0 = dn[0];
            for (int x = 1; x <= W; ++x) {
                const float u1 = up[(size_t)x * C], d1 = dn[(size_t)x * C];
                const float f = ((d1 - u1) - d0) + u0;
                amax = max(amax, __float_as_uint(f) & 0x7fffffffu);
                u0 = u1; d0 = d1;
            }
        }
    }
    amax = wave_max_u32(amax);
    if (lane == 0) absmax[blockIdx.x] = amax;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

} // namespace

namespace vfa_dev {
size_t feature_stats_count(int n_views, int C, int Hf)
{
    if (n_views < 0 || C <= 0 || Hf <= 0) return 0;
    return (size_t)n_views * Hf * ((C + kWave - 1) / kWave);
}
// the same folded into at most `max_entries` entries (the frame entry points, for callers without statistics: their workspaces
// hold VFA_FALLBACK_STATS entries per scale); returns the number of entries written through *n_entries
int integral_absmax_folded(const float *integral, unsigned *absmax, int n_views, int C, int Hf, int Wf, int max_entries, int *n_entries,
                               void *stream)
{
    const size_t n = feature_stats_count(n_views, C, Hf);
    if (!integral || !absmax || max_entries <= 0 || n == 0 || n >= (1ull << 31)) return VFA_ERR_BAD_ARGUMENT;
    const unsigned grid = n < (size_t)max_entries ? (unsigned)n : (unsigned)max_entries;
    hipLaunchKernelGGL(integral_absmax_kernel, dim3(grid), dim3(kWave), 0, (hipStream_t)stream, integral, absmax, n_views, C, Hf, Wf, (unsigned)n);
    *n_entries = (int)grid;
    return (int)hipGetLastError();
}

} // namespace vfa_dev

extern "C" {

static int integral_images_launch(const float *const *features, const float *const *scales, const float *const *shifts,
                                  float *const *integrals, unsigned *const *absmax, int n_views, int C, int n_maps, const int *feat_hw,
                                  bool hwc, void *stream)
{
    MapArgs a = {};
    a.n_maps = n_maps; a.n_views = n_views; a.C = C;
    unsigned long long row_blocks = 0, col_vecs = 0;
    for (int s = 0; s < n_maps; ++s) {
        MapDesc &m = a.m[s];
        m.feat = features[s]; m.out = integrals[s];
        m.scale = scales ? scales[s] : nullptr; m.shift = shifts ? shifts[s] : nullptr;
        m.absmax = absmax ? absmax[s] : nullptr;
        m.H = feat_hw[2 * s]; m.W = feat_hw[2 * s + 1];
        const unsigned long long rb = (unsigned long long)m.H * (C / kWave) * n_views;
        if (rb >= (1ull << 31)) return VFA_ERR_UNSUPPORTED;
        m.row_blocks = (unsigned)rb;
        m.col_vecs = (unsigned long long)n_views * (m.W + 2) * C / 4;
        row_blocks += rb; col_vecs += m.col_vecs;
    }
    if (row_blocks >= (1ull << 31) || (col_vecs + 255) / 256 >= (1ull << 31)) return VFA_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    // One pass (integral_onepass_kernel) wherever its tiling fits: 16-channel blocks, the column accumulators of a map in LDS
    // beside the tile, statistics entries for four row classes.  Otherwise the two-pass kernels (same bits).
    // (... and where there are units to fill the chip: a unit takes ~55 us on a 90 x 160 map however few there are -- one camera of
    // the bench rig: 32 + 16 + 16 units -- while the two-pass kernels scale down with the bytes)
    bool onepass = !hwc && C % kOpCh == 0 && C % kWave == 0 && (long long)n_views * (C / kOpCh) >= 64;
    int w_max = 0;
    for (int s = 0; s < n_maps && onepass; ++s) {
        onepass = a.m[s].H >= 8 && a.m[s].W >= 4 && (size_t)kOpRows * kOpRowPitch * 4 + (size_t)a.m[s].W * kOpCh * 8 <= 160 * 1024 - 1024;
        w_max = a.m[s].W > w_max ? a.m[s].W : w_max;
    }
    if (onepass) {
        OnePassArgs oa = {};
        oa.n_maps = n_maps; oa.n_views = n_views; oa.C = C;
        unsigned long long units = 0;
        for (int s = 0; s < n_maps; ++s) {
            oa.m[s] = a.m[s];
            const int n_strips = (a.m[s].W + kOpCols - 1) / kOpCols;
            // two units by columns from four strips up; the left one takes 0.6 of the strips: s = 0.6 n balances s against
            // s / 3 (the right unit's bare accumulation over the left strips) + n - s
            oa.split[s] = n_strips >= 4 ? (6 * n_strips + 5) / 10 : 0;
            units += (unsigned long long)n_views * (C / kOpCh) * (oa.split[s] ? 2 : 1);
            oa.unit_end[s] = (unsigned)units;
        }
        if (units >= (1ull << 31)) return VFA_ERR_UNSUPPORTED;
        // (the tile + the column accumulators of the widest unit: 79.9 KB on the bench frame, so that two workgroups share a CU --
        // a wave issues one of these fp64 instructions every 6-8 cycles, a SIMD takes one every 2: tools/micro/dp_rates.hip)
        int part_max = 0;
        for (int s = 0; s < n_maps; ++s) {
            const int left = oa.split[s] * kOpCols, wide = oa.split[s] ? (left > a.m[s].W - left ? left : a.m[s].W - left) : a.m[s].W;
            part_max = wide > part_max ? wide : part_max;
        }
        const size_t lds = (size_t)kOpRows * kOpRowPitch * 4 + (size_t)part_max * kOpCh * 8;
        auto launch = [&](auto kern) {
            hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e0 != hipSuccess) return (int)e0;
            hipLaunchKernelGGL(kern, dim3((unsigned)units), dim3(kOpThreads), lds, st, oa);
            return (int)hipGetLastError();
        };
        return scales ? launch(integral_onepass_kernel<true>) : launch(integral_onepass_kernel<false>);
    }
    if (hwc) {
        if (scales) hipLaunchKernelGGL((rows_hwc_kernel<true>), dim3((unsigned)row_blocks), dim3(kWave), 0, st, a);
        else hipLaunchKernelGGL((rows_hwc_kernel<false>), dim3((unsigned)row_blocks), dim3(kWave), 0, st, a);
    } else {
        if (scales) hipLaunchKernelGGL((rows_batched_kernel<true>), dim3((unsigned)row_blocks), dim3(kWave), 0, st, a);
        else hipLaunchKernelGGL((rows_batched_kernel<false>), dim3((unsigned)row_blocks), dim3(kWave), 0, st, a);
    }
    int e = (int)hipGetLastError();
    if (e) return e;
    hipLaunchKernelGGL(cols_batched_kernel, dim3((unsigned)((col_vecs + 255) / 256)), dim3(256), 0, st, a, col_vecs);
    return (int)hipGetLastError();
}

size_t vfa_feature_stats_count(int n_views, int C, int Hf) { return vfa_dev::feature_stats_count(n_views, C, Hf); }

int vfa_integral_absmax_f32(const float *integral, unsigned *absmax, int n_views, int C, int Hf, int Wf, void *stream)
{
    if (!integral || !absmax || n_views < 0 || C <= 0 || Hf <= 0 || Wf <= 0) return VFA_ERR_BAD_ARGUMENT;
    const size_t n = vfa_feature_stats_count(n_views, C, Hf);
    if (n == 0) return 0;
    if (n >= (1ull << 31)) return VFA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(integral_absmax_kernel, dim3((unsigned)n), dim3(kWave), 0, (hipStream_t)stream, integral, absmax, n_views, C, Hf, Wf,
                       (unsigned)n);
    return (int)hipGetLastError();
}

int vfa_integral_images_f32(const float *const *features, const float *const *scales, const float *const *shifts,
                            float *const *integrals, unsigned *const *absmax, int n_views, int C, int n_maps, const int *feat_hw,
                            void *stream)
{
    if (!features || !integrals || !feat_hw || n_views < 0 || C <= 0 || n_maps < 0) return VFA_ERR_BAD_ARGUMENT;
    if ((scales == nullptr) != (shifts == nullptr)) return VFA_ERR_BAD_ARGUMENT;
    for (int s = 0; s < n_maps; ++s) {
        if (feat_hw[2 * s] <= 0 || feat_hw[2 * s + 1] <= 0 || !features[s] || !integrals[s]) return VFA_ERR_BAD_ARGUMENT;
        if (scales && (!scales[s] || !shifts[s])) return VFA_ERR_BAD_ARGUMENT;
    }
    if (n_views == 0 || n_maps == 0) return 0;
    bool fast = (C % kWave == 0) && n_maps <= kMaxMaps;
    for (int s = 0; s < n_maps && fast; ++s)
        fast = (feat_hw[2 * s + 1] % 4 == 0) && aligned16(features[s]) && aligned16(integrals[s]);
    if (!fast) { // odd shapes: the per-map kernels of vfa_kernels.hip (same arithmetic)
        for (int s = 0; s < n_maps; ++s) {
            const int st = scales ? vfa_affine_relu_integral_image_f32(features[s], scales[s], shifts[s], integrals[s], n_views, C,
                                                                        feat_hw[2 * s], feat_hw[2 * s + 1], stream)
                                  : vfa_integral_image_f32(features[s], integrals[s], n_views, C, feat_hw[2 * s], feat_hw[2 * s + 1], stream);
            if (st) return st;
            if (absmax && absmax[s]) { // (the per-map kernels keep no statistics: one more pass, over the result)
                const int st2 = vfa_integral_absmax_f32(integrals[s], absmax[s], n_views, C, feat_hw[2 * s], feat_hw[2 * s + 1], stream);
                if (st2) return st2;
            }
        }
        return 0;
    }
    return integral_images_launch(features, scales, shifts, integrals, absmax, n_views, C, n_maps, feat_hw, false, stream);
}

int vfa_integral_images_hwc_f32(const float *const *features_hwc, const float *const *scales, const float *const *shifts,
                                float *const *integrals, unsigned *const *absmax, int n_views, int C, int n_maps, const int *feat_hw,
                                void *stream)
{
    if (!features_hwc || !integrals || !feat_hw || n_views < 0 || C <= 0 || n_maps < 0) return VFA_ERR_BAD_ARGUMENT;
    if ((scales == nullptr) != (shifts == nullptr)) return VFA_ERR_BAD_ARGUMENT;
    for (int s = 0; s < n_maps; ++s) {
        if (feat_hw[2 * s] <= 0 || feat_hw[2 * s + 1] <= 0 || !features_hwc[s] || !integrals[s]) return VFA_ERR_BAD_ARGUMENT;
        if (scales && (!scales[s] || !shifts[s])) return VFA_ERR_BAD_ARGUMENT;
    }
    if (n_views == 0 || n_maps == 0) return 0;
    if (C % kWave != 0 || n_maps > kMaxMaps) return VFA_ERR_UNSUPPORTED;
    for (int s = 0; s < n_maps; ++s)
        if (!aligned16(integrals[s])) return VFA_ERR_UNSUPPORTED; // (the column pass works on float4)
    return integral_images_launch(features_hwc, scales, shifts, integrals, absmax, n_views, C, n_maps, feat_hw, true, stream);
}

} // extern "C"
