// vfa_split.h -- the reference-width `collapse` product of the fused kernels (reference vfa/model/vfa_op.py:59, :123: an fp32
// nn.Linear) on the 16-bit matrix pipe: both operands are scaled by a power of two and split EXACTLY into two fp16 pieces,
//
//     x * 2^e = hi + lo + r,   hi = RN_f16(x 2^e),  lo = RN_f16(x 2^e - hi),   |r| <= 2^-23 |x 2^e|  (or 2^-25: the fp16 subnormal step)
//
// and a product is the three MFMA products  hi.hi + hi.lo + lo.hi  (v_mfma_f32_32x32x16_f16, fp32 accumulation; the dropped lo.lo
// is <= 2^-22 of the product).  Two 11-bit mantissas cover 22 of fp32's 24 bits: against float64 the result has the error of an
// sgemm (2e-7 normwise at K = 256; tests/test_split_arithmetic.py), at HALF the matrix work of the three-piece bf16 form (six
// products) and the SAME work as the two-piece bf16 form (16 bits, 4e-6) it replaces as the default.
//
// fp16 has five exponent bits, so the pieces need a scale.  One power of two per operand and feature scale:
//   A (voxel features):  2^ea with  absmax(feature map) 2^ea in [2^13, 2^14).  In exact arithmetic a box mean is at most absmax / 4
//                        (the box area of vfa_op.py:104 is four times the pixel area): an honest voxel feature sits below 2^12, a
//                        factor 16 under fp16's 65504.  The fp32 rounding noise of the integral image does NOT obey that bound: the
//                        reference keeps boxes down to area 1e-6 (vfa_op.py:106) and divides the noise by that area -- the shipped
//                        Wildtrack grid holds visible boxes of 1e-5 pixels whose voxel features reach 2.8 absmax (round-4 verdict).
//                        What is bounded is  |vox| / absmax <= 1/4 + 2^-18 Hf Wf / area  (vfa_geom.h: sliver_shift, with its derivation):
//                        the geometry pass turns that bound into a SHIFT per (tile, view, scale) item (serial kernel; per (tile, scale) in
//                        the pipelined kernel, whose accumulators run over views and layers) -- 0 for every honest box, floor(log2
//                        bound) + 1 for a sliver -- and the frame kernels scale that item by 2^(ea - shift) instead: its voxel features
//                        stay below 2^14 WHATEVER the noise does, its bias enters the accumulator times 2^(ea + ew - shift) and its
//                        epilogue multiplies by 2^-(ea + ew - shift): powers of two, exact.  The absolute error of a piece is at most
//                        2^-25 (half a subnormal step), i.e. 2^(shift - 37) of the largest honest voxel feature: with the largest
//                        possible shift (area -> 1e-6 on a 270 x 480 map: 19) still 2^-18 of it, far inside the 1e-4 of the path; the
//                        rows of an unshifted item keep the full 22 bits down to 2^-13 of the largest.  absmax comes from the
//                        integral-image kernels, which see every feature value (vfa_integral.hip), or from a pass over the integral
//                        images when the caller has none (integral_absmax_kernel).
//   W (collapse.weight): 2^ew with  absmax(W) 2^ew in [2^14, 2^15).
// The accumulator starts at bias 2^(ea+ew) and the epilogue multiplies relu(acc) by 2^-(ea+ew): powers of two, exact.
// A value beyond fp16's range (an infinite or absurd feature; no finite map reaches it past the shift above) must not turn into
// Inf - Inf: the CONVERSIONS run under MODE.FP16_OVFL, where a result that overflows is +-65504 instead of infinity (true infinities and
// NaNs stay what they are: measured, tools/micro/fp16_modes.hip).  The MFMAs must NOT: under that mode v_mfma_f32_32x32x16_f16
// reads a NaN operand as a number and an infinite one as FLT_MAX (same tool) -- a NaN box (vfa_op.py:118-119: NaN * 0 stays NaN)
// or a NaN feature would vanish from the map.  The mode is per wave: the pooling waves of vfa_pipe.hip set it for good, the
// serial kernel of vfa_fused.hip switches it on around its pooling phase.
#ifndef VFA_SPLIT_H
#define VFA_SPLIT_H
#include <hip/hip_runtime.h>

namespace vfa_dev {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int kExpA = 13, kExpW = 14, kExpLim = 40;

// power-of-two scale exponent that brings a maximum with the fp32 bit pattern `absmax_bits` (sign cleared) into
// [2^target, 2^(target+1)); 0 for an all-zero or non-finite operand
__host__ __device__ __forceinline__ int split_exponent(unsigned absmax_bits, int target)
{
    if (absmax_bits == 0u || absmax_bits >= 0x7f800000u) return 0;
    const int e = (int)(absmax_bits >> 23) - 127; // (a subnormal maximum reads -127: the clamp below takes it)
    int s = target - e;
    s = s > kExpLim ? kExpLim : s;
    s = s < -kExpLim ? -kExpLim : s;
    return s;
}
__host__ __device__ __forceinline__ unsigned pow2_bits(int e) { return (unsigned)(127 + e) << 23; }
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float(pow2_bits(e)); }

// MODE.FP16_OVFL: fp16 results that overflow are clamped to +-MAX_FP16 instead of becoming infinities.  The empty asm statements keep
// the loads in front of the conversions and the stores behind them on their side of the switch (the conversions themselves read
// MODE: the backend orders them against s_setreg).
__device__ __forceinline__ void fp16_saturate_mode(bool on)
{
    asm volatile("" ::: "memory");
    if (on) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
    else __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 0);
    asm volatile("" ::: "memory");
}

// four fp32 values (already scaled) -> hi / lo fp16 quads: 2 x v_cvt_pk_f16_f32 for the hi pieces, then ONE v_fma_mix{lo,hi}_f16 per lo
// piece -- fma(float(hi), -1, x) rounded to fp16 in the instruction: x - float(hi) is exact in fp32 (hi is x rounded to 11 bits), so the
// one rounding is the rounding of the three-instruction form (v_cvt_f32_f16, v_sub_f32, v_cvt_pk_f16_f32: 12 instructions per quad, 6
// now).  Bit-identical on 1.7e7 values of every kind -- NaN, infinities, denormals, the fp16 limit -- with MODE.FP16_OVFL on and off
// (round 5, tools/micro/mix_split.hip).  The statements are `volatile`: their rounding and saturation read MODE.FP16_OVFL, which the
// compiler does not know of an asm statement -- volatile keeps them in program order against the s_setreg of fp16_saturate_mode (a
// side-effecting intrinsic) and its fences instead of leaving that to luck (round-5 advisor finding).
__device__ __forceinline__ void split_f16x4(float x0, float x1, float x2, float x3, uint2 &hi, uint2 &lo)
{
    const f32x4v x = {x0, x1, x2, x3};
    const f16x4 h = __builtin_convertvector(x, f16x4);
    hi = __builtin_bit_cast(uint2, h);
    unsigned l0, l1;
    asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi.x), "v"(x0));
    asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l0) : "v"(hi.x), "v"(x1));
    asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi.y), "v"(x2));
    asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l1) : "v"(hi.y), "v"(x3));
    lo = make_uint2(l0, l1);
}

// wave-wide maximum of an unsigned value (all lanes get it)
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)v, m, 64);
        v = o > v ? o : v;
    }
    return v;
}

// host side, defined in vfa_integral.hip: the statistic of one scale from a finished integral image, folded into at most
// `max_entries` entries (the frame entry points call it for callers that pass no statistics)
constexpr int kFallbackStats = 1024;
size_t feature_stats_count(int n_views, int C, int Hf);
int integral_absmax_folded(const float *integral, unsigned *absmax, int n_views, int C, int Hf, int Wf, int max_entries, int *n_entries,
                           void *stream);

} // namespace vfa_dev
#endif // VFA_SPLIT_H
