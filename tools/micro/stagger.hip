// stagger.hip -- schedule study for the single-layer frame kernel (vfa_fused.hip: pool_collapse_kernel), round 5.
// Eight waves (two per SIMD), W resident (128 registers per wave: fp16 hi + lo of 32 output columns), the reference's pooling arithmetic,
// three fp16 MFMA products.  Per 32-row item and wave: 4 boxes x 256 channels pooled (4 quarter passes), 48 MFMAs.
//   MODE 0  serial (what ships): all waves pool the item | barrier | all waves multiply, the next window arrives by LDS-DMA | barrier
//   MODE 1  anti-phase halves: the item in two 128-channel halves; waves 0-3 (X) and 4-7 (Y) alternate roles every half-step --
//           X pools its boxes of half j while Y multiplies half j - 1, then Y pools half j while X multiplies half j - 1.  Same
//           instruction counts, four barriers per item instead of two, tap windows as 512-byte half slots in two buffers.
//   MODE 2  serial, but the lo fragments of k-steps 8-15 are NOT resident (requested behind the pooling pass, 8 KiB per wave and item
//           out of L2, used from k-step 8 on): 32 registers for all sixteen taps of a quarter pass in flight -- one LDS round trip per
//           quarter instead of two.
//   hipcc --offload-arch=gfx950 -O3 -o stagger tools/micro/stagger.hip && ./stagger
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSlots = 120;
constexpr int kRowBytes = 512, kPlane = 32 * kRowBytes; // A tile: 32 rows x 256 k fp16, 16-byte chunks XOR-swizzled with (row & 15); hi plane, lo plane
constexpr int kRecBytes = 96;
constexpr int kQStride = 32 * 16 + 32, kQPlane = 8 * kQStride, kQBuf = 2 * kQPlane; // MODE 1: planes of one 64-channel quarter, chunk-major

__device__ __forceinline__ f32x4 fma4(f32x4 a, float w, f32x4 c) { return f32x4{fmaf(a[0], w, c[0]), fmaf(a[1], w, c[1]), fmaf(a[2], w, c[2]), fmaf(a[3], w, c[3])}; }
__device__ __forceinline__ f32x4 mul4(f32x4 a, float w) { return f32x4{a[0] * w, a[1] * w, a[2] * w, a[3] * w}; }
__device__ __forceinline__ f32x4 sample4(f32x4 a, f32x4 b, f32x4 c, f32x4 d, float w0, float w1, float w2, float w3)
{
    return fma4(d, w3, fma4(c, w2, fma4(b, w1, mul4(a, w0))));
}
__device__ __forceinline__ float quot(float v, float as, float rs)
{
    const float q0 = v * rs;
    const float q1 = fmaf(fmaf(-as, q0, v), rs, q0);
    return fmaf(fmaf(-as, q1, v), rs, q1);
}
template <int OFF>
__device__ __forceinline__ void lds_read4(f32x4 &t0, f32x4 &t1, f32x4 &t2, f32x4 &t3, unsigned p0, unsigned p1, unsigned p2, unsigned p3)
{
    asm volatile("ds_read_b128 %0, %4 offset:%8\n\tds_read_b128 %1, %5 offset:%8\n\tds_read_b128 %2, %6 offset:%8\n\tds_read_b128 %3, %7 offset:%8"
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait4(f32x4 &t0, f32x4 &t1, f32x4 &t2, f32x4 &t3)
{
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "n"(N));
}

struct BoxState { float rs, as; unsigned tb[16]; unsigned plane0; }; // (the sixteen tap weights are re-read from the record in LDS by every pooling step: not live across a multiply step)

template <int MODE>
__global__ __launch_bounds__(512) void k(const uint4 *__restrict__ wsrc, const uint4 *__restrict__ recs, const float *__restrict__ image, float *out, int items,
                                         unsigned long long *cyc)
{
    // MODE 0: one window of 1 KiB slots; MODE 1: two buffers of 512-byte half slots
    __shared__ __align__(16) unsigned char s_win[MODE != 1 ? kSlots * 1024 : 2 * kSlots * 256];
    __shared__ __align__(16) unsigned char s_planes[2 * kPlane];
    __shared__ __align__(16) unsigned char s_rec[2][32 * kRecBytes];
    __shared__ float s_sum[MODE == 1 ? 8 * 16 * 64 : 64];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int grp = lane >> 4, cq = lane & 15;
    for (int i = tid; i < (MODE != 1 ? kSlots * 256 : 2 * kSlots * 64); i += 512) reinterpret_cast<float *>(s_win)[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
    for (int i = tid; i < (MODE == 1 ? 8 * 16 * 64 : 64); i += 512) s_sum[i] = 0.0f;
    for (int i = tid; i < 2 * kPlane / 4; i += 512) reinterpret_cast<unsigned *>(s_planes)[i] = 0x3c003c00u;
    for (int i = tid; i < 2 * 32 * kRecBytes / 16; i += 512) reinterpret_cast<uint4 *>(s_rec)[i] = recs[(size_t)blockIdx.x * 2 * 32 * 6 + i];
    f16x8 wh[16], wl[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        wh[ks] = __builtin_bit_cast(f16x8, wsrc[((wave * 16 + ks) * 2 + 0) * 64 + lane]);
        wl[ks] = __builtin_bit_cast(f16x8, wsrc[((wave * 16 + ks) * 2 + 1) * 64 + lane]);
        asm volatile("" : "+v"(wh[ks]), "+v"(wl[ks]));
    }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned win = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)s_win;
    const unsigned pla = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)s_planes;
    const int row = 4 * wave + grp;
    constexpr unsigned kUnit = MODE != 1 ? 1024u : 256u; // bytes per window slot

    BoxState bs;
    auto unpack = [&](int buf) {
        const unsigned ra = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)s_rec[buf] + (unsigned)(row * kRecBytes);
        uint4 v[6];
        asm volatile("ds_read_b128 %0, %2 offset:64\n\tds_read_b128 %1, %2 offset:80\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v[4]), "=&v"(v[5]) : "v"(ra) : "memory");
        const float rcp = __uint_as_float(v[4].x), masked = __uint_as_float(v[5].z), area = __uint_as_float(v[5].w);
        const bool vis = (v[4].y & 1u) != 0u;
        unsigned rw[4] = {v[4].z & 0xffffu, v[4].z >> 16, v[4].w & 0xffffu, v[4].w >> 16};
        unsigned cl[4] = {v[5].x & 0xffffu, v[5].x >> 16, v[5].y & 0xffffu, v[5].y >> 16};
#pragma unroll
        for (int i = 0; i < 4; ++i) { rw[i] = (vis ? rw[i] : 0u) * kUnit + (unsigned)cq * 16u + win; cl[i] = (vis ? cl[i] : 0u) * kUnit; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) bs.tb[i * 4 + j] = rw[i] + cl[j];
        bs.rs = (vis ? rcp : masked) * 0x1p3f; bs.as = area * 0x1p-3f;
        if constexpr (MODE != 1) bs.plane0 = pla + (unsigned)(row * kRowBytes + (((((cq >> 1) ^ (row & 15)) << 4)) | ((cq & 1) << 3)));
        else bs.plane0 = pla + (unsigned)((cq >> 1) * kQStride + row * 16 + (cq & 1) * 8); // chunk-major quarter buffer: [chunk of 8 k][row][16 B], padded
    };
    // one 64-channel quarter pass of this lane's box: tap reads at byte offset OFF from the box's tap addresses, planes chunk flip PX
    float wt[16];
    auto weights = [&](int buf) {
        const unsigned ra = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)s_rec[buf] + (unsigned)(row * kRecBytes);
        f32x4 q0, q1, q2, q3;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(ra) : "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) { wt[i] = q0[i]; wt[4 + i] = q1[i]; wt[8 + i] = q2[i]; wt[12 + i] = q3[i]; }
    };
    auto pass = [&](auto off_tag, auto px_tag) {
        constexpr int OFF = decltype(off_tag)::value, PX = decltype(px_tag)::value;
        f32x4 a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3, d0, d1, d2, d3;
        f32x4 lt, rb, rt, lb;
        if constexpr (MODE == 2) {
            lds_read4<OFF>(a0, a1, a2, a3, bs.tb[0], bs.tb[1], bs.tb[4], bs.tb[5]);
            lds_read4<OFF>(b0, b1, b2, b3, bs.tb[10], bs.tb[11], bs.tb[14], bs.tb[15]);
            lds_read4<OFF>(c0, c1, c2, c3, bs.tb[2], bs.tb[3], bs.tb[6], bs.tb[7]);
            lds_read4<OFF>(d0, d1, d2, d3, bs.tb[8], bs.tb[9], bs.tb[12], bs.tb[13]);
            lds_wait4<12>(a0, a1, a2, a3);
            lt = sample4(a0, a1, a2, a3, wt[0], wt[1], wt[2], wt[3]);
            lds_wait4<8>(b0, b1, b2, b3);
            rb = sample4(b0, b1, b2, b3, wt[4], wt[5], wt[6], wt[7]);
            lds_wait4<4>(c0, c1, c2, c3);
            rt = sample4(c0, c1, c2, c3, wt[8], wt[9], wt[10], wt[11]);
            lds_wait4<0>(d0, d1, d2, d3);
            lb = sample4(d0, d1, d2, d3, wt[12], wt[13], wt[14], wt[15]);
        } else {
        lds_read4<OFF>(a0, a1, a2, a3, bs.tb[0], bs.tb[1], bs.tb[4], bs.tb[5]);
        lds_read4<OFF>(b0, b1, b2, b3, bs.tb[10], bs.tb[11], bs.tb[14], bs.tb[15]);
        lds_wait4<4>(a0, a1, a2, a3);
        lt = sample4(a0, a1, a2, a3, wt[0], wt[1], wt[2], wt[3]);
        lds_read4<OFF>(c0, c1, c2, c3, bs.tb[2], bs.tb[3], bs.tb[6], bs.tb[7]);
        lds_wait4<4>(b0, b1, b2, b3);
        rb = sample4(b0, b1, b2, b3, wt[4], wt[5], wt[6], wt[7]);
        lds_read4<OFF>(d0, d1, d2, d3, bs.tb[8], bs.tb[9], bs.tb[12], bs.tb[13]);
        lds_wait4<4>(c0, c1, c2, c3);
        rt = sample4(c0, c1, c2, c3, wt[8], wt[9], wt[10], wt[11]);
        lds_wait4<0>(d0, d1, d2, d3);
        lb = sample4(d0, d1, d2, d3, wt[12], wt[13], wt[14], wt[15]);
        }
        f32x4 v = ((lt + rb) - rt) - lb;
        v = f32x4{quot(v[0], bs.as, bs.rs), quot(v[1], bs.as, bs.rs), quot(v[2], bs.as, bs.rs), quot(v[3], bs.as, bs.rs)};
        const f16x4 hi = __builtin_convertvector(v, f16x4);
        const f32x4 back = __builtin_convertvector(hi, f32x4);
        const f16x4 lo = __builtin_convertvector(v - back, f16x4);
        if constexpr (MODE != 1) {
            const unsigned at = bs.plane0 ^ (unsigned)(PX << 7);
            asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:%3" :: "v"(at), "v"(hi), "v"(lo), "n"(kPlane) : "memory");
        } else {
            asm volatile("ds_write_b64 %0, %1 offset:%3\n\tds_write_b64 %0, %2 offset:%4" :: "v"(bs.plane0), "v"(hi), "v"(lo), "n"(PX * kQBuf), "n"(PX * kQBuf + kQPlane) : "memory");
        }
    };
    // window DMA: `n` pieces of 1 KiB by this wave into byte offset `dst` of the window area
    auto fill = [&](int item, int j, unsigned dst) {
        const char *src = reinterpret_cast<const char *>(image) + ((size_t)((blockIdx.x * 61 + item * 7 + j * 13 + wave * 3) & 4095)) * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)(s_win + dst), 16, 0, 0);
    };

    f32x16 acc, sum;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.5f; sum[i] = 0.0f; }
    const int key = r & 15;
    const unsigned pa = pla + (unsigned)(r * kRowBytes + ((h ^ (key & 1)) << 4));
    auto frags = [&](int kk, f16x8 &hh, f16x8 &ll) {
        const unsigned addr = pa + (unsigned)((kk ^ (key >> 1)) << 5);
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3" : "=&v"(hh), "=&v"(ll) : "v"(addr), "n"(kPlane) : "memory");
    };
    unsigned qa = pla + (unsigned)(h * kQStride + r * 16);
    // MODE 0: the MFMAs of k-steps [K0, K0 + 16); MODE 1: k-steps [K0, K0 + 4) out of quarter buffer B.  `nf` window pieces issued in between.
    auto multiply = [&](auto k0_tag, auto b_tag, int item, int nf, unsigned dst) {
        constexpr int K0 = decltype(k0_tag)::value, B = decltype(b_tag)::value, NK = MODE != 1 ? 16 : 4;
        f16x8 fh[3], fl[3];
        unsigned qb = qa;
        auto rd = [&](auto kk_tag, f16x8 &hh, f16x8 &ll, unsigned qaddr) {
            constexpr int KK = decltype(kk_tag)::value;
            if constexpr (MODE != 1) frags(K0 + KK, hh, ll);
            else asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(hh), "=&v"(ll)
                              : "v"(qaddr), "n"(B * kQBuf + 2 * KK * kQStride), "n"(B * kQBuf + 2 * KK * kQStride + kQPlane) : "memory");
        };
        rd(std::integral_constant<int, 0>{}, fh[0], fl[0], qb);
        rd(std::integral_constant<int, 1>{}, fh[1], fl[1], qb);
        auto step = [&](auto kk_tag) {
            constexpr int kk = decltype(kk_tag)::value;
            if (kk < nf) fill(item, kk, dst + (unsigned)((wave & 3) + 4 * kk) * 1024u);
            if constexpr (kk + 2 < NK) {
                rd(std::integral_constant<int, kk + 2>{}, fh[(kk + 2) % 3], fl[(kk + 2) % 3], qb);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fh[kk % 3]), "+v"(fl[kk % 3]));
            } else if constexpr (kk + 1 < NK) {
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fh[kk % 3]), "+v"(fl[kk % 3]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fh[kk % 3]), "+v"(fl[kk % 3]));
            }
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[kk % 3], wl[K0 + kk], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[kk % 3], wh[K0 + kk], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[kk % 3], wh[K0 + kk], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
        if constexpr (MODE != 1) {
            step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{}); step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
            step(std::integral_constant<int, 8>{}); step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
            step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{}); step(std::integral_constant<int, 14>{}); step(std::integral_constant<int, 15>{});
        }
    };
    auto epilogue = [&]() {
#pragma unroll
        for (int i = 0; i < 16; ++i) { sum[i] = fmaf(acc[i] > 0.0f ? acc[i] : 0.0f, 0x1p-20f, sum[i]); acc[i] = 0.5f; }
    };
    using I0 = std::integral_constant<int, 0>;
    using I8 = std::integral_constant<int, 8>;
    using I16 = std::integral_constant<int, 16>;

    if constexpr (MODE == 2) {
        const uint4 *wlo = wsrc + (size_t)(wave * 16 + 8) * 2 * 64 + 64 + lane; // lo fragment of k-step 8 of this wave
        for (int item = 0; item < items; ++item) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unpack(item & 1);
            weights(item & 1);
            pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            pass(std::integral_constant<int, 256>{}, std::integral_constant<int, 1>{});
            pass(std::integral_constant<int, 512>{}, std::integral_constant<int, 2>{});
            pass(std::integral_constant<int, 768>{}, std::integral_constant<int, 3>{});
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int ks = 8; ks < 16; ++ks) wl[ks] = __builtin_bit_cast(f16x8, wlo[(size_t)((ks - 8) * 2 + (item & 1) * 0) * 64]); // (lands under k-steps 0-7)
            __syncthreads();
            multiply(I0{}, I0{}, item, wave < 4 ? 9 : 0, 0u);
            epilogue();
#pragma unroll
            for (int ks = 8; ks < 16; ++ks) asm volatile("" : "=v"(wl[ks])); // (dead until the next item requests them again)
        }
    } else if constexpr (MODE == 0) {
        for (int item = 0; item < items; ++item) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unpack(item & 1);
            weights(item & 1);
            pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            pass(std::integral_constant<int, 256>{}, std::integral_constant<int, 1>{});
            pass(std::integral_constant<int, 512>{}, std::integral_constant<int, 2>{});
            pass(std::integral_constant<int, 768>{}, std::integral_constant<int, 3>{});
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            multiply(I0{}, I0{}, item, wave < 4 ? 9 : 0, 0u); // ~36 slots per item, issued by the four older waves
            epilogue();
        }
    } else {
        // quarter n = 4 item + q: window buffer n & 1 (quarter slots of 256 B), planes buffer n & 1 (chunk-major, padded), the tile's sums in LDS.
        //   X: pool(n) | BAR | multiply(n - 1) | BAR          Y: DMA(n + 1), multiply(n - 1) | BAR | pool(n), vmcnt(0) | BAR
        const bool isx = wave < 4;
        float *ssum = s_sum + wave * 16 * 64 + lane;
        auto qstep = [&](auto q_tag, int item) {
            constexpr int Q = decltype(q_tag)::value, WB = Q & 1, PB = Q & 1;
            auto do_pool = [&]() {
                if (Q == 0) unpack(item & 1);
                weights(item & 1);
                pass(std::integral_constant<int, WB * kSlots * 256>{}, std::integral_constant<int, PB>{});
            };
            auto do_mul = [&](int nf) {
                // quarter Q - 1 (mod 4): k-steps 4 (Q - 1) ..; planes buffer (Q - 1) & 1
                constexpr int QM = (Q + 3) & 3;
                multiply(std::integral_constant<int, 4 * QM>{}, std::integral_constant<int, (QM & 1)>{}, item, nf, (unsigned)((WB ^ 1) * kSlots * 256));
                if (QM == 3) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) { ssum[i * 64] = fmaf(acc[i] > 0.0f ? acc[i] : 0.0f, 0x1p-20f, ssum[i * 64]); acc[i] = 0.5f; }
                }
            };
            if (isx) {
                do_pool();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                do_mul(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            } else {
                do_mul(2); // (the next quarter's window: ~8 KiB = 32 quarter slots by the four Y waves)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                do_pool();
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        };
        for (int item = 0; item < items; ++item) {
            qstep(std::integral_constant<int, 0>{}, item);
            qstep(std::integral_constant<int, 1>{}, item);
            qstep(std::integral_constant<int, 2>{}, item);
            qstep(std::integral_constant<int, 3>{}, item);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) sum[i] = ssum[i * 64];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) out[((size_t)blockIdx.x * 512 + tid) * 16 + i] = sum[i] + acc[i];
    if (tid == 0) cyc[blockIdx.x] = (unsigned long long)(__builtin_amdgcn_s_memtime() - t0);
}

template <int MODE> void run(const char *name, const uint4 *w, const uint4 *recs, const float *image, float *out, unsigned long long *cyc)
{
    const int items = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, w, recs, image, out, items, cyc);
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, w, recs, image, out, items, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc[256];
    hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < 256; ++i) mean += (double)hc[i] / 256;
    const hipError_t err = hipGetLastError();
    printf("%-28s %8.1f us per launch, %6.3f us per item, s_memtime ticks per item %7.1f  (%s)\n", name, ms * 1e3 / 10, ms * 1e3 / 10 / items, mean / items,
           hipGetErrorString(err));
}

int main()
{
    uint4 *w, *recs; float *out, *image; unsigned long long *cyc;
    const size_t wbytes = (size_t)8 * 16 * 2 * 64 * 16;
    hipMalloc(&w, wbytes);
    hipMalloc(&recs, (size_t)256 * 2 * 32 * kRecBytes);
    hipMalloc(&out, (size_t)256 * 512 * 16 * 4);
    hipMalloc(&image, (size_t)4096 * 1024 + 4096);
    hipMalloc(&cyc, 256 * 8);
    hipMemset(image, 0, (size_t)4096 * 1024 + 4096);
    unsigned *hw = (unsigned *)malloc(wbytes);
    for (size_t i = 0; i < wbytes / 4; ++i) hw[i] = 0x2c002c00u + (rand() & 0x03ff03ff);
    hipMemcpy(w, hw, wbytes, hipMemcpyHostToDevice);
    // records: 16 weights in (0, 1), rcp, flags (visible), 4 row parts + 4 column parts as slot indices (row part + column part < 40), masked 0, area
    const size_t nrec = (size_t)256 * 2 * 32;
    unsigned *hr = (unsigned *)malloc(nrec * kRecBytes);
    for (size_t b = 0; b < nrec; ++b) {
        unsigned *p = hr + b * 24;
        for (int i = 0; i < 16; ++i) { float f = 0.05f + 0.9f * (float)(rand() & 1023) / 1024.f; p[i] = *reinterpret_cast<unsigned *>(&f); }
        float rcp = 0.25f, area = 4.0f, masked = 0.0f;
        const unsigned r0 = rand() % 3, r1 = r0 + 1, r2 = r0 + 1 + rand() % 2, r3 = r2 + 1; // window rows 0..5, 8 columns wide
        const unsigned c0 = rand() % 3, c1 = c0 + 1, c2 = c0 + 2 + rand() % 2, c3 = c2 + 1; // columns 0..6
        p[16] = *reinterpret_cast<unsigned *>(&rcp); p[17] = 1u;
        p[18] = (r0 * 8) | ((r1 * 8) << 16); p[19] = (r2 * 8) | ((r3 * 8) << 16);
        p[20] = c0 | (c1 << 16); p[21] = c2 | (c3 << 16);
        p[22] = *reinterpret_cast<unsigned *>(&masked); p[23] = *reinterpret_cast<unsigned *>(&area);
    }
    hipMemcpy(recs, hr, nrec * kRecBytes, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("serial (pool | multiply)", w, recs, image, out, cyc);
        run<1>("anti-phase quarters", w, recs, image, out, cyc);
        run<2>("serial, 16 taps in flight", w, recs, image, out, cyc);
    }
    return 0;
}
