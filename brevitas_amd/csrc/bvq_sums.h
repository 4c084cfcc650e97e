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
// (1024 = four partials per thread, loaded together: with 4096 a thread walked 16 dependent-latency loads one after
//  the other and the per-tensor backward paid 30-90 us for its two finishing launches, profiles/r02_per_tensor_pieces.txt)
constexpr int64_t kSumSlice = 1024;

static inline int32_t sum_splits(int64_t partials_per_channel) {
  const int64_t s = (partials_per_channel + kSumSlice - 1) / kSumSlice;
  return (int32_t)(s < 1 ? 1 : s);
}

// Optional epilogue of the LAST stage: the scale gradient out0[c] carried on, in the same launch, through the
// backward of a learned scale   scale = abs_binary_sign_grad(clamp_min_ste(value, min_val)) / int_threshold
// (ParameterScaling / ParameterFromRuntimeStatsScaling after collection, B/core/scaling/standalone.py:75-152,
// 155-298; division B/core/quant/int.py:160) with torch's rounding points:
//   ds  = dscale.to(scale_dtype) [+ gscale]                  gradient of `scale` (gscale: other users of scale)
//   dt  = (ds / int_threshold  in scale_dtype).to(value dtype)   backward of the division
//   dv  = binary_sign(clamp_min(value)) * dt                     backward of |.| ; clamp_min_ste passes it on
// -- instead of a cast, a division, a sign-multiply and their launches.
struct LearnedScaleEpilogue {
  const void* value;   // [channels] the learned parameter; null: no epilogue
  void* dvalue;        // [channels] its gradient, dtype of value
  const void* gscale;  // nullable [channels], scale_dtype
  float min_val;       // already rounded to the value's dtype
  float int_threshold; // already rounded to the dtype the division runs in
  int32_t value_dtype, scale_dtype, use_min;
};

__device__ __forceinline__ float round_to_dtype(float v, int dt) {
  return dt == BVQ_F32 ? v : (dt == BVQ_BF16 ? rnd<bf16_t>(v) : rnd<f16_t>(v));
}

__device__ __forceinline__ void learned_scale_bwd_elem(const LearnedScaleEpilogue& ep, int32_t c, float dscale) {
  float ds = round_to_dtype(dscale, ep.scale_dtype);
  if (ep.gscale) ds = round_to_dtype(ds + load_scalar_as_f(ep.gscale, ep.scale_dtype, c), ep.scale_dtype);
  const float dt = round_to_dtype(round_to_dtype(ds / ep.int_threshold, ep.scale_dtype), ep.value_dtype);
  float v = load_scalar_as_f(ep.value, ep.value_dtype, c);
  if (ep.use_min && v < ep.min_val) v = ep.min_val;  // NaN passes, like torch.clamp_min
  const float sign = (float)(v >= 0.f) - (float)(v < 0.f);  // binary_sign: +1 at 0, 0 for NaN (B/function/ops.py:31-34)
  const float dv = round_to_dtype(sign * dt, ep.value_dtype);
  if (ep.value_dtype == BVQ_F32)
    reinterpret_cast<float*>(ep.dvalue)[c] = dv;
  else if (ep.value_dtype == BVQ_BF16)
    reinterpret_cast<bf16_t*>(ep.dvalue)[c] = (bf16_t)dv;
  else
    reinterpret_cast<f16_t*>(ep.dvalue)[c] = (f16_t)dv;
}

template <typename PT>
__global__ __launch_bounds__(kBlock) void channel_sum_kernel(const PT* __restrict__ part0,
                                                             const PT* __restrict__ part1,
                                                             float* __restrict__ out0,
                                                             float* __restrict__ out1, int64_t nob,
                                                             int32_t channels, int64_t ppr,
                                                             double* __restrict__ mid0,
                                                             double* __restrict__ mid1,
                                                             LearnedScaleEpilogue ep) {
  __shared__ double sh[2][kBlock];
  const int32_t c = blockIdx.x;
  const int64_t n = nob * ppr;
  const int64_t slice = (n + gridDim.y - 1) / gridDim.y;
  const int64_t k0 = (int64_t)blockIdx.y * slice;
  const int64_t k1 = k0 + slice < n ? k0 + slice : n;
  double acc0 = 0.0, acc1 = 0.0;
  constexpr int kU = 4;  // loads in flight per thread; the additions keep their order
  for (int64_t kb = k0 + threadIdx.x; kb < k1; kb += (int64_t)kU * kBlock) {
    PT v0[kU], v1[kU];
#pragma unroll
    for (int j = 0; j < kU; ++j) {
      const int64_t k = kb + (int64_t)j * kBlock;
      int64_t unit = 0;
      const bool in = k < k1;
      if (in) {
        if (nob == 1) {
          unit = (int64_t)c * ppr + k;
        } else {
          const int64_t o = k / ppr, p = k - o * ppr;
          unit = (o * channels + c) * ppr + p;
        }
      }
      v0[j] = (in && part0) ? part0[unit] : (PT)0;
      v1[j] = (in && part1) ? part1[unit] : (PT)0;
    }
#pragma unroll
    for (int j = 0; j < kU; ++j) {
      acc0 += (double)v0[j];
      acc1 += (double)v1[j];
    }
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
      if (ep.value) learned_scale_bwd_elem(ep, c, (float)sh[0][0]);
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
                                       hipStream_t st, const LearnedScaleEpilogue* epilogue = nullptr) {
  const int32_t splits = sum_splits(nob * ppr);
  const LearnedScaleEpilogue none = {};
  const LearnedScaleEpilogue ep = epilogue ? *epilogue : none;
  if (splits > 1) {
    double* mid0 = reinterpret_cast<double*>(mid);
    double* mid1 = mid0 + (int64_t)channels * splits;
    channel_sum_kernel<float><<<dim3((unsigned)channels, (unsigned)splits), dim3(kBlock), 0, st>>>(
        part0, part1, nullptr, nullptr, nob, channels, ppr, mid0, mid1, none);
    channel_sum_kernel<double><<<dim3((unsigned)channels), dim3(kBlock), 0, st>>>(
        part0 ? mid0 : nullptr, part1 ? mid1 : nullptr, out0, out1, 1, channels, splits, nullptr, nullptr, ep);
  } else {
    channel_sum_kernel<float><<<dim3((unsigned)channels), dim3(kBlock), 0, st>>>(
        part0, part1, out0, out1, nob, channels, ppr, nullptr, nullptr, ep);
  }
}

// Column-mapped partials [prows][L] -> one row [L].  Thread (j, slice) folds column j over the kFoldRows partial
// rows of its slice (coalesced across j, eight independent loads in flight); stages of that shrink prows by
// kFoldRows each until at most kFoldLast rows are left for one last pass.  (Round 1 folded prows / 64 rows per
// thread with one load in flight: at 8192 partial rows of 512 columns that pass alone took a third of the
// abs-max's time.)
// (templates only so that the header can be included by several translation units)
constexpr int kFoldRows = 32;
constexpr int kFoldLast = 64;

// rows of L entries the fold needs behind the partials
static inline int64_t cols_fold_scratch_rows(int64_t prows) {
  int64_t total = 1, r = prows;
  while (r > kFoldLast) {
    r = (r + kFoldRows - 1) / kFoldRows;
    total += r;
  }
  return total;
}

template <int UNUSED>
__global__ __launch_bounds__(kBlock) void cols_fold_max_kernel(const uint32_t* __restrict__ part, uint32_t* __restrict__ out,
                                                               int64_t prows, int64_t L, int64_t rows_per_slice) {
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= L) return;
  const int64_t o0 = (int64_t)blockIdx.y * rows_per_slice;
  const int64_t o1 = o0 + rows_per_slice < prows ? o0 + rows_per_slice : prows;
  uint32_t m = 0;
  for (int64_t o = o0; o < o1; o += 8) {
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = part[(o + k < o1 ? o + k : o0) * L + j];  // past the end: a row already seen
#pragma unroll
    for (int k = 0; k < 8; ++k) m = v[k] > m ? v[k] : m;
  }
  out[(int64_t)blockIdx.y * L + j] = m;
}

template <int UNUSED>
__global__ __launch_bounds__(kBlock) void cols_fold_sum_min_kernel(const float* __restrict__ ds_part,
                                                                   const unsigned long long* __restrict__ pos_part,
                                                                   float* __restrict__ ds_out,
                                                                   unsigned long long* __restrict__ pos_out, int64_t prows,
                                                                   int64_t L, int64_t rows_per_slice) {
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= L) return;
  const int64_t o0 = (int64_t)blockIdx.y * rows_per_slice;
  const int64_t o1 = o0 + rows_per_slice < prows ? o0 + rows_per_slice : prows;
  double acc = 0.0;
  unsigned long long pmin = ~0ull;
  for (int64_t o = o0; o < o1; o += 8) {
    float v[8];
    unsigned long long q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t oo = o + k < o1 ? o + k : o0;
      if (ds_part) v[k] = ds_part[oo * L + j];
      if (pos_part) q[k] = pos_part[oo * L + j];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (ds_part) acc += o + k < o1 ? (double)v[k] : 0.0;  // rows in order: the sum does not depend on the launch shape
      if (pos_part) pmin = q[k] < pmin ? q[k] : pmin;
    }
  }
  if (ds_part) ds_out[(int64_t)blockIdx.y * L + j] = (float)acc;
  if (pos_part) pos_out[(int64_t)blockIdx.y * L + j] = pmin;
}

// scratch: cols_fold_scratch_rows(prows) rows of L entries; returns the final row inside it
static inline uint32_t* launch_cols_fold_max(const uint32_t* part, int64_t prows, int64_t L, uint32_t* scratch,
                                             hipStream_t st) {
  const unsigned gx = (unsigned)((L + kBlock - 1) / kBlock);
  const uint32_t* src = part;
  uint32_t* dst = scratch;
  int64_t rows = prows;
  while (rows > kFoldLast) {
    const int64_t slices = (rows + kFoldRows - 1) / kFoldRows;
    cols_fold_max_kernel<0><<<dim3(gx, (unsigned)slices), dim3(kBlock), 0, st>>>(src, dst, rows, L, kFoldRows);
    src = dst;
    dst += slices * L;
    rows = slices;
  }
  cols_fold_max_kernel<0><<<dim3(gx, 1), dim3(kBlock), 0, st>>>(src, dst, rows, L, rows);
  return dst;
}

static inline void launch_cols_fold_sum_min(const float* ds_part, const unsigned long long* pos_part, int64_t prows,
                                            int64_t L, float* ds_scratch, unsigned long long* pos_scratch,
                                            float** ds_final, unsigned long long** pos_final, hipStream_t st) {
  const unsigned gx = (unsigned)((L + kBlock - 1) / kBlock);
  const float* dsrc = ds_part;
  const unsigned long long* psrc = pos_part;
  float* ddst = ds_part ? ds_scratch : nullptr;
  unsigned long long* pdst = pos_part ? pos_scratch : nullptr;
  int64_t rows = prows;
  while (rows > kFoldLast) {
    const int64_t slices = (rows + kFoldRows - 1) / kFoldRows;
    cols_fold_sum_min_kernel<0><<<dim3(gx, (unsigned)slices), dim3(kBlock), 0, st>>>(dsrc, psrc, ddst, pdst, rows, L,
                                                                                     kFoldRows);
    dsrc = ddst;
    psrc = pdst;
    if (ddst) ddst += slices * L;
    if (pdst) pdst += slices * L;
    rows = slices;
  }
  cols_fold_sum_min_kernel<0><<<dim3(gx, 1), dim3(kBlock), 0, st>>>(dsrc, psrc, ddst, pdst, rows, L, rows);
  *ds_final = ddst;
  if (pos_final) *pos_final = pdst;
}

}  // namespace bvq
