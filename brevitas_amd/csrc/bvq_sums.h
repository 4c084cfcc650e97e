// bvq_sums.h -- fixed-order reduction of per-unit partial sums, shared by the quantizer backward
// (dscale / dzp) and the moment statistics.
#pragma once

#include "bvq_common.h"

namespace bvq {

// Combine per-unit partial sums of one channel in a fixed order (double accumulation):
// channel c owns units (ob*channels + c)*ppr + p for ob in [0, nob), p in [0, ppr).
// grid = (channels, splits).  splits == 1: the workgroup sums everything and writes the float result.
// A per-tensor quantizer of a large activation has ~10^5 partials in its one channel; there the range is
// cut into slices whose double sums go to mid0/mid1[c * splits + s], and a second launch (PT = double,
// nob = 1, ppr = splits) finishes.  The order of additions is fixed either way: same bits on every run.
constexpr int64_t kSumSlice = 4096;

static inline int32_t sum_splits(int64_t partials_per_channel) {
  const int64_t s = (partials_per_channel + kSumSlice - 1) / kSumSlice;
  return (int32_t)(s < 1 ? 1 : s);
}

template <typename PT>
__global__ __launch_bounds__(kBlock) void channel_sum_kernel(const PT* __restrict__ part0,
                                                             const PT* __restrict__ part1,
                                                             float* __restrict__ out0,
                                                             float* __restrict__ out1, int64_t nob,
                                                             int32_t channels, int64_t ppr,
                                                             double* __restrict__ mid0,
                                                             double* __restrict__ mid1) {
  __shared__ double sh[2][kBlock];
  const int32_t c = blockIdx.x;
  const int64_t n = nob * ppr;
  const int64_t slice = (n + gridDim.y - 1) / gridDim.y;
  const int64_t k0 = (int64_t)blockIdx.y * slice;
  const int64_t k1 = k0 + slice < n ? k0 + slice : n;
  double acc0 = 0.0, acc1 = 0.0;
  for (int64_t k = k0 + threadIdx.x; k < k1; k += kBlock) {
    int64_t unit;
    if (nob == 1) {
      unit = (int64_t)c * ppr + k;
    } else {
      const int64_t o = k / ppr, p = k - o * ppr;
      unit = (o * channels + c) * ppr + p;
    }
    if (part0) acc0 += (double)part0[unit];
    if (part1) acc1 += (double)part1[unit];
  }
  sh[0][threadIdx.x] = acc0;
  sh[1][threadIdx.x] = acc1;
  __syncthreads();
  for (int st = kBlock / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + st];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (mid0) {
      const int64_t slot = (int64_t)c * gridDim.y + blockIdx.y;
      mid0[slot] = sh[0][0];
      mid1[slot] = sh[1][0];
    } else {
      if (out0) out0[c] = (float)sh[0][0];
      if (out1) out1[c] = (float)sh[1][0];
    }
  }
}


// bytes of the double-precision middle stage for `partials_per_channel` partials (0 if one stage does)
static inline int64_t channel_sums_mid_bytes(int64_t partials_per_channel, int64_t channels) {
  const int32_t splits = sum_splits(partials_per_channel);
  return splits > 1 ? 2 * channels * (int64_t)splits * (int64_t)sizeof(double) : 0;
}

// out0[c] = sum of part0 over channel c's units (same for part1/out1; either pair may be null).
// mid: 8-byte aligned scratch of channel_sums_mid_bytes() bytes (unused if that is 0).
static inline void launch_channel_sums(const float* part0, const float* part1, float* out0, float* out1,
                                       int64_t nob, int32_t channels, int64_t ppr, void* mid,
                                       hipStream_t st) {
  const int32_t splits = sum_splits(nob * ppr);
  if (splits > 1) {
    double* mid0 = reinterpret_cast<double*>(mid);
    double* mid1 = mid0 + (int64_t)channels * splits;
    channel_sum_kernel<float><<<dim3((unsigned)channels, (unsigned)splits), dim3(kBlock), 0, st>>>(
        part0, part1, nullptr, nullptr, nob, channels, ppr, mid0, mid1);
    channel_sum_kernel<double><<<dim3((unsigned)channels), dim3(kBlock), 0, st>>>(
        part0 ? mid0 : nullptr, part1 ? mid1 : nullptr, out0, out1, 1, channels, splits, nullptr, nullptr);
  } else {
    channel_sum_kernel<float><<<dim3((unsigned)channels), dim3(kBlock), 0, st>>>(
        part0, part1, out0, out1, nob, channels, ppr, nullptr, nullptr);
  }
}

}  // namespace bvq
