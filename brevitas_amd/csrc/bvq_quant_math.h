// bvq_quant_math.h -- per-element arithmetic of the affine integer quantizer, with the reference's
// rounding points.
//
// The reference (B/core/quant/int_base.py:63-97) issues one torch op per line; on a tensor of
// dtype CT every op computes in float32 and rounds its result to CT.  rnd<CT>() marks each of
// those rounding points, so the fused kernel reproduces the chain bit for bit:
//     y = x / scale            -> rnd
//     y = y + zero_point       -> rnd
//     y = float_to_int_impl(y) -> exact (integers are representable)
//     y = tensor_clamp(y, min_int, max_int)
//     y = y - zero_point       -> rnd
//     y = y * scale            -> rnd
// The division is a true IEEE division (the file is built with -ffp-contract=off and hipcc's
// default correctly-rounded fp32 division): x * (1/scale) would flip codes at .5 ties.
#pragma once

#include "bvq_common.h"

namespace bvq {

// float_to_int_impl on a value already rounded to CT
template <typename CT, int RM>
__device__ __forceinline__ float round_op(float t) {
  if constexpr (RM == BVQ_ROUND) {
    return __builtin_rintf(t);  // torch.round: half to even
  } else if constexpr (RM == BVQ_FLOOR) {
    return __builtin_floorf(t);
  } else if constexpr (RM == BVQ_CEIL) {
    return __builtin_ceilf(t);
  } else if constexpr (RM == BVQ_ROUND_TO_ZERO) {
    // torch.sign(x) * torch.floor(torch.abs(x))   (B/function/ops.py:52); sign(NaN) = 0
    float sg = (float)(0.f < t) - (float)(t < 0.f);
    return sg * __builtin_floorf(__builtin_fabsf(t));
  } else {
    // torch.where((x < 0.) & (x - torch.floor(x) == 0.5), torch.ceil(x), torch.round(x))
    // (B/function/ops.py:71)
    float fr = rnd<CT>(t - __builtin_floorf(t));
    return (t < 0.f && fr == 0.5f) ? __builtin_ceilf(t) : __builtin_rintf(t);
  }
}

// runtime-selected rounding mode (the rarely used variants share one kernel instantiation)
constexpr int kAnyRM = -1;
template <typename CT>
__device__ __forceinline__ float round_any(float t, int mode) {
  switch (mode) {
    case BVQ_ROUND:
      return round_op<CT, BVQ_ROUND>(t);
    case BVQ_FLOOR:
      return round_op<CT, BVQ_FLOOR>(t);
    case BVQ_CEIL:
      return round_op<CT, BVQ_CEIL>(t);
    case BVQ_ROUND_TO_ZERO:
      return round_op<CT, BVQ_ROUND_TO_ZERO>(t);
    default:
      return round_op<CT, BVQ_DPU_ROUND>(t);
  }
}

// tensor_clamp: where(x > max, max, x) then where(out < min, min, out)  (B/function/ops.py:98-100)
// NaN fails both comparisons and passes through, as in the reference.
__device__ __forceinline__ float clamp_where(float t, float qmin, float qmax) {
  t = t > qmax ? qmax : t;
  t = t < qmin ? qmin : t;
  return t;
}

// ------------------------------------------------------------------------------------------------
// The same arithmetic on PAIRS of elements.  16-bit tensors are VALU-bound on MI355X if every op is
// issued per element (the backward needs ~40 of them against 6 bytes of traffic): float2 values let
// the compiler use the packed-fp32 instructions (v_pk_mul_f32 / v_pk_add_f32: two lanes of work per
// issue) and ONE v_cvt_pk_bf16_f32 per rounding point of a pair instead of one per element.  Results
// are the same bits: every component goes through the same IEEE operations as the scalar code.
// ------------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int b2 __attribute__((ext_vector_type(2)));  // comparison result: -1 / 0 per component

template <typename T>
__device__ __forceinline__ f2 rnd2(f2 v) {
  f2 r = {rnd<T>(v.x), rnd<T>(v.y)};
  return r;
}
template <>
__device__ __forceinline__ f2 rnd2<float>(f2 v) {
  return v;
}
template <>
__device__ __forceinline__ f2 rnd2<bf16_t>(f2 v) {
  typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
  const uint32_t bits = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));  // v_cvt_pk_bf16_f32
  f2 r = {__builtin_bit_cast(float, bits << 16), __builtin_bit_cast(float, bits & 0xffff0000u)};
  return r;
}

// float16: one v_cvt_pk_f16_f32 for the pair.  The empty asm keeps hipcc from folding a preceding multiply
// into v_fma_mix*_f16 with a +0 addend (which would turn a -0 product into +0; see rnd<f16_t>).
template <>
__device__ __forceinline__ f2 rnd2<f16_t>(f2 v) {
  typedef f16_t f16x2 __attribute__((ext_vector_type(2)));
  float a = v.x, b = v.y;
  asm volatile("" : "+v"(a), "+v"(b));
  f2 w = {a, b};
  return __builtin_convertvector(__builtin_convertvector(w, f16x2), f2);
}

// two floats -> two T (round to nearest even), stored to a and b
template <typename T>
__device__ __forceinline__ void pack2(f2 v, T& a, T& b) {
  a = from_f<T>(v.x);
  b = from_f<T>(v.y);
}
template <>
__device__ __forceinline__ void pack2<bf16_t>(f2 v, bf16_t& a, bf16_t& b) {
  typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  a = h.x;
  b = h.y;
}

template <>
__device__ __forceinline__ void pack2<f16_t>(f2 v, f16_t& a, f16_t& b) {
  typedef f16_t f16x2 __attribute__((ext_vector_type(2)));
  float x = v.x, y = v.y;
  asm volatile("" : "+v"(x), "+v"(y));
  f2 w = {x, y};
  const f16x2 h = __builtin_convertvector(w, f16x2);
  a = h.x;
  b = h.y;
}

template <typename T>
__device__ __forceinline__ f2 widen2(T a, T b) {
  f2 r = {to_f<T>(a), to_f<T>(b)};
  return r;
}

__device__ __forceinline__ f2 splat2(float v) {
  f2 r = {v, v};
  return r;
}

template <typename CT, int RM>
__device__ __forceinline__ f2 round_op2(f2 t) {
  if constexpr (RM == BVQ_ROUND) {
    return __builtin_elementwise_roundeven(t);
  } else if constexpr (RM == BVQ_FLOOR) {
    return __builtin_elementwise_floor(t);
  } else if constexpr (RM == BVQ_CEIL) {
    return __builtin_elementwise_ceil(t);
  } else {
    f2 r = {round_op<CT, RM>(t.x), round_op<CT, RM>(t.y)};
    return r;
  }
}

template <typename CT>
__device__ __forceinline__ f2 round_any2(f2 t, int mode) {
  switch (mode) {
    case BVQ_ROUND:
      return round_op2<CT, BVQ_ROUND>(t);
    case BVQ_FLOOR:
      return round_op2<CT, BVQ_FLOOR>(t);
    case BVQ_CEIL:
      return round_op2<CT, BVQ_CEIL>(t);
    case BVQ_ROUND_TO_ZERO:
      return round_op2<CT, BVQ_ROUND_TO_ZERO>(t);
    default:
      return round_op2<CT, BVQ_DPU_ROUND>(t);
  }
}

__device__ __forceinline__ f2 clamp_where2(f2 t, float qmin, float qmax) {
  const f2 hi = splat2(qmax), lo = splat2(qmin);
  t = t > hi ? hi : t;
  t = t < lo ? lo : t;
  return t;
}

}  // namespace bvq
