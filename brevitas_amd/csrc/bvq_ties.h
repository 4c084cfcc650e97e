// bvq_ties.h -- bookkeeping of the elements that attain a max/min statistic ("ties"), shared by the
// statistics backward (bvq_stats.hip) and the fused quantizer backward (bvq_fakequant_bwd.h).
//
// tie_info layout (unsigned 64-bit words, device memory, bvq_tie_info_bytes(channels) bytes):
//   channels > 1 : first[c]  = smallest (outer*inner + i) position matching stat[c]   (init: ~0)
//   channels == 1: [0] = number of ties (also the list cursor), [1] = unused,
//                  [2 .. 2+kTieCap) = flat element indices of the first kTieCap ties found
#pragma once

#include "bvq_common.h"

namespace bvq {

constexpr int kTieCap = 1024;
// batch-sharded tensors: a shard's claim on a channel's deposit is its rank, or this when it holds no arg-max
constexpr double kShardNoOwner = 1073741824.0;  // 2^30, above any rank

static inline int64_t tie_info_words(int64_t channels) { return channels > 1 ? channels : 2 + kTieCap; }

__device__ __forceinline__ void record_tie(unsigned long long* info, bool per_channel, int32_t channel,
                                           unsigned long long pos) {
  if (per_channel) {
    atomicMin(&info[channel], pos);
  } else {
    const unsigned long long slot = atomicAdd(&info[0], 1ull);
    if (slot < (unsigned long long)kTieCap) info[2 + slot] = pos;
  }
}

// |v| as an order-preserving integer key (float32 pattern for f32/bf16, 16-bit pattern for f16)
template <typename T>
__device__ __forceinline__ uint32_t abs_bits(T v);
template <>
__device__ __forceinline__ uint32_t abs_bits<float>(float v) {
  return __builtin_bit_cast(uint32_t, v) & 0x7fffffffu;
}
template <>
__device__ __forceinline__ uint32_t abs_bits<bf16_t>(bf16_t v) {
  return ((uint32_t)(__builtin_bit_cast(uint16_t, v) & 0x7fffu)) << 16;
}
template <>
__device__ __forceinline__ uint32_t abs_bits<f16_t>(f16_t v) {
  return (uint32_t)(__builtin_bit_cast(uint16_t, v) & 0x7fffu);
}

// torch.relu: x < 0 ? 0 : x  (NaN and -0.0 pass through unchanged)
__device__ __forceinline__ float relu_f(float v) { return v < 0.f ? 0.f : v; }

// |pre_op(v)| as the same key: negative (non-NaN) values become 0 under RELU
template <typename T, bool RELU>
__device__ __forceinline__ uint32_t pre_abs_bits(T v) {
  if constexpr (RELU) {
    return to_f<T>(v) < 0.f ? 0u : abs_bits<T>(v);
  } else {
    return abs_bits<T>(v);
  }
}

#ifdef __HIPCC__
// torch.sign
__device__ __forceinline__ float sgn_f(float v) { return (float)(0.f < v) - (float)(v < 0.f); }

// Where the statistic's gradient comes from: an array of the statistic's dtype, or the float32
// scale-gradient sums of the fused quantizer backward, taken through
//   dscale.to(scale dtype)  ->  / int_threshold (in the dtype torch computes that quotient in)  ->  .to(T)
// i.e. the backward of  scale = clamp_min_ste(stat) / int_threshold  (B/core/quant/int.py:160).
struct GstatSrc {
  const void* p;
  int32_t from_dscale;  // 0: p holds T values; 1: p holds float32 dscale sums
  int32_t scale_dtype;
  int32_t quot_dtype;
  float int_threshold;
  int32_t pre_relu;  // the statistic was taken of relu(x): the deposit's sign is sgn(relu(x))
};

__device__ __forceinline__ float round_rt(float v, int dt) {
  return dt == BVQ_F32 ? v : (dt == BVQ_BF16 ? rnd<bf16_t>(v) : rnd<f16_t>(v));
}

template <typename T>
__device__ __forceinline__ float gstat_value(const GstatSrc& g, int64_t c) {
  if (!g.from_dscale) return to_f<T>(reinterpret_cast<const T*>(g.p)[c]);
  float v = round_rt(reinterpret_cast<const float*>(g.p)[c], g.scale_dtype);
  v = round_rt(v / g.int_threshold, g.quot_dtype);
  return rnd<T>(v);
}

template <typename T, int MATCH>
__device__ __forceinline__ float deposit(float g, T xv, bool pre_relu = false) {
  if constexpr (MATCH == BVQ_MATCH_ABS) {
    return rnd<T>(g * sgn_f(pre_relu ? relu_f(to_f<T>(xv)) : to_f<T>(xv)));
  } else {
    return g;
  }
}

#endif

void launch_tie_init(unsigned long long* info, int64_t channels, hipStream_t st, int first_only = 0);

}  // namespace bvq
