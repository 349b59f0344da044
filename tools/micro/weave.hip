// weave.hip -- ONE wave per SIMD (256 threads, 512 registers per lane: the fp16 hi + lo weight of the wave's 64 output columns in 256 of
// them) with the pooling of quarter n woven BY HAND into the MFMA shadows of quarter n - 2 (stream: tools/gen_weave.py ->
// weave_body.inc).  Same work per 32-row item as the shipped serial kernel (vfa_fused.hip): 32 boxes x 256 channels pooled with the
// reference's arithmetic, 3 fp16 MFMA products, 96 MFMAs per SIMD.  Quarter slots (256 B) in a ring of three windows, chunk-major
// quarter planes in a ring of three, one barrier per quarter-step.
//   python tools/gen_weave.py tools/micro/weave_body.inc && hipcc --offload-arch=gfx950 -O3 -o weave tools/micro/weave.hip && ./weave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSlots = 116;
constexpr int kWinBytes = kSlots * 256;                                   // one quarter window
constexpr int kQStride = 32 * 16 + 32, kQPlane = 8 * kQStride, kQBuf = 2 * kQPlane; // planes of one quarter: [chunk of 8 k][row][16 B], padded
constexpr int kRecBytes = 96;
constexpr int kWinAt = 0, kPlanesAt = 3 * kWinBytes, kRecAt = kPlanesAt + 3 * kQBuf, kSumAt = kRecAt + 2 * 32 * kRecBytes, kLdsBytes = kSumAt + 256 * 32 * 4;
// (the tile's running sums live in LDS: 32 floats per lane, touched once per item)

__device__ __forceinline__ float quot(float v, float as, float rs)
{
    const float q0 = v * rs;
    const float q1 = fmaf(fmaf(-as, q0, v), rs, q0);
    return fmaf(fmaf(-as, q1, v), rs, q1);
}

struct Box { float rs, as; unsigned tb[16]; }; // (the tap weights are read from the box record in LDS corner by corner)

template <int DMA, int ABL> // ABL: ablation bits -- 1: no MFMAs, 2: no pooling arithmetic (reads stay), 4: no tap reads either
__global__ __launch_bounds__(256) void k(const uint4 *__restrict__ wsrc, const uint4 *__restrict__ recs, const float *__restrict__ image, float *out, int items,
                                         unsigned long long *cyc)
{
    __shared__ __align__(16) unsigned char lds[kLdsBytes];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int grp = lane >> 4, cq = lane & 15;
    for (int i = tid; i < 3 * kWinBytes / 4; i += 256) reinterpret_cast<float *>(lds + kWinAt)[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
    for (int i = tid; i < 3 * kQBuf / 4; i += 256) reinterpret_cast<unsigned *>(lds + kPlanesAt)[i] = 0x3c003c00u;
    for (int i = tid; i < 2 * 32 * kRecBytes / 16; i += 256) reinterpret_cast<uint4 *>(lds + kRecAt)[i] = recs[(size_t)blockIdx.x * 2 * 32 * 6 + i];
    f16x8 wreg[16][2][2]; // [k-step][column block][hi, lo]
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int p = 0; p < 2; ++p) wreg[ks][cb][p] = __builtin_bit_cast(f16x8, wsrc[(((wave * 16 + ks) * 2 + cb) * 2 + p) * 64 + lane]);
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;

    Box bx[2];
    unsigned rec_at[2] = {0u, 0u}; // LDS address of the box record of pass p
    f32x4 wq[3];                   // the four tap weights of a corner (ring of three, like the taps)
    f32x4 tap[3][4];          // ring of three corner buffers: reads run two corners ahead
    f32x4 sm[4];              // lt, rb, rt, lb of the pass in work
    f32x4 vq[3];              // box sum / quotient of pass 0, 1; [2]: pass 1 of the previous step (carried)
    float rs_c = 0.0f, as_c = 1.0f;
    unsigned pl[2], pl_c = 0; // plane store address of the pass's box (quarter buffer 0)
    f16x4 shi[3], slo[3];
    f16x8 fh[2], fl[2];       // A fragments of k-step k in fh[k % 2]: one k-step (six MFMAs) ahead
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.5f;
    f32x4 *ssum = reinterpret_cast<f32x4 *>(lds + kSumAt) + tid; // [8][256 lanes] float4: conflict-free 16-byte accesses
#pragma unroll
    for (int i = 0; i < 8; ++i) ssum[i * 256] = f32x4{0.f, 0.f, 0.f, 0.f};
    vq[2] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned fa = base + kPlanesAt + (unsigned)(h * kQStride + r * 16);

    auto unpack = [&](int p, int buf) {
        const int row = 8 * wave + 4 * p + grp;
        const unsigned ra = base + kRecAt + (unsigned)(buf * 32 * kRecBytes + row * kRecBytes);
        uint4 v[6];
        asm volatile("ds_read_b128 %0, %2 offset:64\n\tds_read_b128 %1, %2 offset:80\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v[4]), "=&v"(v[5]) : "v"(ra) : "memory");
        rec_at[p] = ra;
        const float rcp = __uint_as_float(v[4].x), masked = __uint_as_float(v[5].z), area = __uint_as_float(v[5].w);
        const bool vis = (v[4].y & 1u) != 0u;
        unsigned rw[4] = {v[4].z & 0xffffu, v[4].z >> 16, v[4].w & 0xffffu, v[4].w >> 16};
        unsigned cl[4] = {v[5].x & 0xffffu, v[5].x >> 16, v[5].y & 0xffffu, v[5].y >> 16};
#pragma unroll
        for (int i = 0; i < 4; ++i) { rw[i] = (vis ? rw[i] : 0u) * 256u + (unsigned)cq * 16u + base + kWinAt; cl[i] = (vis ? cl[i] : 0u) * 256u; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) bx[p].tb[i * 4 + j] = rw[i] + cl[j];
        bx[p].rs = (vis ? rcp : masked) * 0x1p3f; bx[p].as = area * 0x1p-3f;
        pl[p] = base + kPlanesAt + (unsigned)((cq >> 1) * kQStride + row * 16 + (cq & 1) * 8);
    };
    auto fill = [&](int n, int j, unsigned dst) {
        const char *src = reinterpret_cast<const char *>(image) + ((size_t)((blockIdx.x * 61 + n * 7 + j * 13 + wave * 3) & 4095)) * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)(lds + dst), 16, 0, 0);
    };

    // one quarter-step; N12 = n mod 12: Q = quarter of the item, window / planes buffers by n mod 3
    auto step = [&](auto n_tag, int item) {
        constexpr int N12 = decltype(n_tag)::value, Q = N12 & 3, WB = N12 % 3, PW = N12 % 3, PR = (N12 + 1) % 3, PC = (N12 + 2) % 3, QM = (Q + 2) & 3;
        if constexpr (Q == 0) { unpack(0, item & 1); unpack(1, item & 1); }
        // corner c of a pass: taps (row pair, column pair): lt (0,0) rb (2,2) rt (0,2) lb (2,0)
#define TBI(c, i) ((((c) == 0 || (c) == 2) ? 0 : 8) + (((c) == 0 || (c) == 3) ? 0 : 2) + ((i) >> 1) * 4 + ((i) & 1))
#define R_TAPS(p, c)                                                                                                                             \
    if (!(ABL & 4)) asm volatile("ds_read_b128 %0, %5 offset:%10\n\tds_read_b128 %1, %6 offset:%10\n\tds_read_b128 %2, %7 offset:%10\n\tds_read_b128 %3, %8 offset:%10\n\t" \
                 "ds_read_b128 %4, %9 offset:%11"                                                                                                \
                 : "=&v"(tap[((p) * 4 + (c)) % 3][0]), "=&v"(tap[((p) * 4 + (c)) % 3][1]), "=&v"(tap[((p) * 4 + (c)) % 3][2]), "=&v"(tap[((p) * 4 + (c)) % 3][3]), \
                   "=&v"(wq[((p) * 4 + (c)) % 3])                                                                                                            \
                 : "v"(bx[p].tb[TBI(c, 0)]), "v"(bx[p].tb[TBI(c, 1)]), "v"(bx[p].tb[TBI(c, 2)]), "v"(bx[p].tb[TBI(c, 3)]), "v"(rec_at[p]),        \
                   "n"(WB * kWinBytes), "n"(16 * (c)) : "memory")
#define WAIT_TAPS(p, c, cnt)                                                                                                                     \
    if (!(ABL & 4)) asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(tap[((p) * 4 + (c)) % 3][0]), "+v"(tap[((p) * 4 + (c)) % 3][1]), "+v"(tap[((p) * 4 + (c)) % 3][2]), "+v"(tap[((p) * 4 + (c)) % 3][3]), "+v"(wq[((p) * 4 + (c)) % 3]) : "n"(cnt))
#define S_TAP(p, c, i)                                                                                                                           \
    do {                                                                                                                                         \
        if (ABL & 2) break;                                                                                                                                         \
        const f32x4 t_ = tap[((p) * 4 + (c)) % 3][i];                                                                                            \
        const float w_ = wq[((p) * 4 + (c)) % 3][i];                                                                                                \
        if ((i) == 0) sm[c] = f32x4{t_[0] * w_, t_[1] * w_, t_[2] * w_, t_[3] * w_};                                                          \
        else sm[c] = f32x4{fmaf(t_[0], w_, sm[c][0]), fmaf(t_[1], w_, sm[c][1]), fmaf(t_[2], w_, sm[c][2]), fmaf(t_[3], w_, sm[c][3])}; \
    } while (0)
#define F_SUM(p) if (!(ABL & 2)) vq[p] = ((sm[0] + sm[1]) - sm[2]) - sm[3]
#define F_QUOT(p, j) if (!(ABL & 2)) vq[p][j] = quot(vq[p][j], (p) == 2 ? as_c : bx[(p) & 1].as, (p) == 2 ? rs_c : bx[(p) & 1].rs)
#define F_SPLIT(p)                                                                                                                               \
    do {                                                                                                                                         \
        if (ABL & 2) break;                                                                                                                                         \
        shi[p] = __builtin_convertvector(vq[p], f16x4);                                                                                          \
        slo[p] = __builtin_convertvector(vq[p] - __builtin_convertvector(shi[p], f32x4), f16x4);                                                 \
    } while (0)
#define F_STORE(p)                                                                                                                               \
    asm volatile("ds_write_b64 %0, %1 offset:%3\n\tds_write_b64 %0, %2 offset:%4" ::"v"((p) == 2 ? pl_c : pl[(p) & 1]), "v"(shi[p]), "v"(slo[p]),     \
                 "n"(((p) == 2 ? PC : PW) * kQBuf), "n"(((p) == 2 ? PC : PW) * kQBuf + kQPlane) : "memory")
#define R_FRAG(kk)                                                                                                                               \
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(fh[(kk) % 2]), "=&v"(fl[(kk) % 2])                       \
                 : "v"(fa), "n"(PR * kQBuf + 2 * (kk) * kQStride), "n"(PR * kQBuf + 2 * (kk) * kQStride + kQPlane) : "memory")
#define WAIT_FRAG(kk, cnt) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fh[(kk) % 2]), "+v"(fl[(kk) % 2]) : "n"(cnt))
#define MFMA(m)                                                                                                                                  \
    do {                                                                                                                                         \
        constexpr int k_ = (m) / 6, cb_ = ((m) / 3) & 1, pr_ = (m) % 3;                                                                          \
        if (ABL & 1) break;                                                                                                                      \
        if (DMA && pr_ == 0 && cb_ == 0 && k_ < 2) fill(N12 + 12 * item, k_, (unsigned)(kWinAt + ((WB + 2) % 3) * kWinBytes + (wave + 4 * k_) * 1024)); \
        acc[cb_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr_ == 2 ? fl[k_ % 2] : fh[k_ % 2], wreg[4 * QM + k_][cb_][pr_ == 0 ? 1 : 0], acc[cb_], 0, 0, 0); \
    } while (0)
#include "weave_body.inc"
        // carry pass 1 into the next step
        vq[2] = vq[1]; rs_c = bx[1].rs; as_c = bx[1].as; pl_c = pl[1];
        if constexpr (QM == 3) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    f32x4 sv = ssum[(cb * 4 + i4) * 256];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float a_ = acc[cb][4 * i4 + j]; sv[j] = fmaf(a_ > 0.0f ? a_ : 0.0f, 0x1p-20f, sv[j]); acc[cb][4 * i4 + j] = 0.5f; }
                    ssum[(cb * 4 + i4) * 256] = sv;
                }
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); // (the fills of the PREVIOUS step have landed: two per step)
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    for (int item = 0; item + 2 < items; item += 3) {
        step(std::integral_constant<int, 0>{}, item); step(std::integral_constant<int, 1>{}, item);
        step(std::integral_constant<int, 2>{}, item); step(std::integral_constant<int, 3>{}, item);
        step(std::integral_constant<int, 4>{}, item + 1); step(std::integral_constant<int, 5>{}, item + 1);
        step(std::integral_constant<int, 6>{}, item + 1); step(std::integral_constant<int, 7>{}, item + 1);
        step(std::integral_constant<int, 8>{}, item + 2); step(std::integral_constant<int, 9>{}, item + 2);
        step(std::integral_constant<int, 10>{}, item + 2); step(std::integral_constant<int, 11>{}, item + 2);
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) out[((size_t)blockIdx.x * 256 + tid) * 32 + cb * 16 + i] = ssum[(cb * 4 + (i >> 2)) * 256][i & 3] + acc[cb][i] + vq[2][i & 3];
    if (tid == 0) cyc[blockIdx.x] = (unsigned long long)(__builtin_amdgcn_s_memtime() - t0);
}

template <int DMA, int ABL> void run(const char *name, const uint4 *w, const uint4 *recs, const float *image, float *out, unsigned long long *cyc)
{
    const int items = 201;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<DMA, ABL>), dim3(256), dim3(256), 0, 0, w, recs, image, out, items, cyc);
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL((k<DMA, ABL>), dim3(256), dim3(256), 0, 0, w, recs, image, out, items, cyc);
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
    const size_t wbytes = (size_t)4 * 16 * 2 * 2 * 64 * 16;
    hipMalloc(&w, wbytes);
    hipMalloc(&recs, (size_t)256 * 2 * 32 * kRecBytes);
    hipMalloc(&out, (size_t)256 * 256 * 32 * 4);
    hipMalloc(&image, (size_t)4096 * 1024 + 4096);
    hipMalloc(&cyc, 256 * 8);
    hipMemset(image, 0, (size_t)4096 * 1024 + 4096);
    unsigned *hw = (unsigned *)malloc(wbytes);
    for (size_t i = 0; i < wbytes / 4; ++i) hw[i] = 0x2c002c00u + (rand() & 0x03ff03ff);
    hipMemcpy(w, hw, wbytes, hipMemcpyHostToDevice);
    const size_t nrec = (size_t)256 * 2 * 32;
    unsigned *hr = (unsigned *)malloc(nrec * kRecBytes);
    for (size_t b = 0; b < nrec; ++b) {
        unsigned *p = hr + b * 24;
        for (int i = 0; i < 16; ++i) { float f = 0.05f + 0.9f * (float)(rand() & 1023) / 1024.f; p[i] = *reinterpret_cast<unsigned *>(&f); }
        float rcp = 0.25f, area = 4.0f, masked = 0.0f;
        const unsigned r0 = rand() % 3, r1 = r0 + 1, r2 = r0 + 1 + rand() % 2, r3 = r2 + 1;
        const unsigned c0 = rand() % 3, c1 = c0 + 1, c2 = c0 + 2 + rand() % 2, c3 = c2 + 1;
        p[16] = *reinterpret_cast<unsigned *>(&rcp); p[17] = 1u;
        p[18] = (r0 * 8) | ((r1 * 8) << 16); p[19] = (r2 * 8) | ((r3 * 8) << 16);
        p[20] = c0 | (c1 << 16); p[21] = c2 | (c3 << 16);
        p[22] = *reinterpret_cast<unsigned *>(&masked); p[23] = *reinterpret_cast<unsigned *>(&area);
    }
    hipMemcpy(recs, hr, nrec * kRecBytes, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 1; ++rep) {
        run<0, 0>("woven, no DMA", w, recs, image, out, cyc);
        run<1, 0>("woven + window DMA", w, recs, image, out, cyc);
        run<0, 1>("no MFMAs", w, recs, image, out, cyc);
        run<0, 2>("no pooling arithmetic", w, recs, image, out, cyc);
        run<0, 6>("no pooling, no tap reads", w, recs, image, out, cyc);
        run<0, 7>("skeleton", w, recs, image, out, cyc);
    }
    return 0;
}
