// Shared pieces of the split-precision kernels (pointmlp_split.hip, gemm_split.hip).
//
// A float is handled as hi + lo 16-bit pieces (round-to-nearest-even, lo = round(x - hi)) and a product block as three MFMAs
// with f32 accumulation, x.w ~= x_lo.w_hi + x_hi.w_lo + x_hi.w_hi.  Fragments travel as raw 128-bit values; the element type
// (F16 = true: IEEE half, 11 + 11 significant bits, logits within ~2e-6 of the float64 evaluation = float32's own distance;
// F16 = false: bf16, 8 + 8 bits, ~2e-5) only matters where a float is split and where the MFMA is issued.
#pragma once
#include "cg_common.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 frag;

// IEEE-half range.  A value at or beyond HALF_MAX has an infinite hi piece and a meaningless result (which a NaN-ignoring
// max-pool can even make look finite); a tensor whose values all lie below HALF_LOW has its lo pieces in the half
// subnormals, where they carry an absolute error (2^-25) instead of a relative one.  The half kernels track the largest
// magnitude they split per layer and OR these bits into the caller's status word (a device int passed per call -- no
// process-global state); the engine then re-runs the batch with bf16 pieces, which have float32's exponent range.
constexpr float HALF_MAX = 65504.f;
constexpr float HALF_LOW = 0.015625f;      // 2^-6: worst relative error of a split value >= 2^-6 is 2^-19 of the tensor's scale
constexpr int CG_HALF_OVERFLOW = 1;
constexpr int CG_HALF_UNDERFLOW = 2;

template <bool F16>
__device__ __forceinline__ f32x16 mfma_x(frag a, frag b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// two floats -> packed pair of the high parts and packed pair of the residuals (v_cvt_pk_{bf16,f16}_f32 x2);
// amax accumulates the largest magnitude seen (half only: one v_max3_f32 with |.| source modifiers)
template <bool F16>
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo, float& amax) {
  if constexpr (F16) {
    amax = fmaxf(fmaxf(amax, fabsf(a)), fabsf(b));
    const f16x2 h = {(_Float16)a, (_Float16)b};
    hi = __builtin_bit_cast(unsigned, h);
    // a - float(hi) as ONE mixed-precision FMA each: v_fma_mix_f32 reads the half operand straight out of the packed register
    // (op_sel picks the low / high half), so the two v_cvt_f32_f16 per pair disappear -- a quarter of the split's VALU work.  The
    // product float(hi) * -1 is exact, so this is the same correctly rounded difference.  (hipcc does not form the mix op itself.)
    float la, lb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(la) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hi), "v"(b));
    const f16x2 l = {(_Float16)la, (_Float16)lb};
    lo = __builtin_bit_cast(unsigned, l);
  } else {
    const bf16x2 h = {(__bf16)a, (__bf16)b};
    hi = __builtin_bit_cast(unsigned, h);
    const float ha = __uint_as_float(hi << 16), hb = __uint_as_float(hi & 0xffff0000u);
    const bf16x2 l = {(__bf16)(a - ha), (__bf16)(b - hb)};
    lo = __builtin_bit_cast(unsigned, l);
  }
}

// f16fp8x2 mode: four floats of one MX unit -> the two packed half hi pairs (for the f16 image) and, returned / through d_lo, one dword
// each of the e4m3 images of the hi pieces and of the residuals: fp8(a / sc_hi) and fp8((a - half(a)) / sc_lo), sc_* powers of two
// (v_cvt_scalef32_pk_fp8_f32 divides by its scale operand and rounds to nearest even; measured, scripts/probe/mx_probe.hip).
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned split_mx(float a0, float a1, float a2, float a3, float sc_hi, float sc_lo, unsigned& h01, unsigned& h23,
                                             unsigned& d_lo) {
  const f16x2 p01 = {(_Float16)a0, (_Float16)a1}, p23 = {(_Float16)a2, (_Float16)a3};
  h01 = __builtin_bit_cast(unsigned, p01); h23 = __builtin_bit_cast(unsigned, p23);
  float l0, l1, l2, l3;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h01), "v"(a0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h01), "v"(a1));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l2) : "v"(h23), "v"(a2));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l3) : "v"(h23), "v"(a3));
  s16x2 dh = {0, 0}, dl = {0, 0};
  dh = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(dh, a0, a1, sc_hi, false);
  dh = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(dh, a2, a3, sc_hi, true);
  dl = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(dl, l0, l1, sc_lo, false);
  dl = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(dl, l2, l3, sc_lo, true);
  d_lo = __builtin_bit_cast(unsigned, dl);
  return __builtin_bit_cast(unsigned, dh);
}
