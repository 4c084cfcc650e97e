// bvq_ties.h -- bookkeeping of the elements that attain a max/min statistic ("ties"), shared by the
// statistics backward (bvq_stats.hip) and the fused quantizer backward (bvq_fakequant.hip).
//
// tie_info layout (unsigned 64-bit words, device memory, bvq_tie_info_bytes(channels) bytes):
//   channels > 1 : first[c]  = smallest (outer*inner + i) position matching stat[c]   (init: ~0)
//   channels == 1: [0] = number of ties (also the list cursor), [1] = unused,
//                  [2 .. 2+kTieCap) = flat element indices of the first kTieCap ties found
#pragma once

#include "bvq_common.h"

namespace bvq {

constexpr int kTieCap = 1024;

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

void launch_tie_init(unsigned long long* info, int64_t channels, hipStream_t st, int first_only = 0);

}  // namespace bvq
