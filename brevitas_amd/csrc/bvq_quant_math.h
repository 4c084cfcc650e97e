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

}  // namespace bvq
