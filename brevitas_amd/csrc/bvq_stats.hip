// bvq_stats.hip -- scale statistics: AbsMax / AbsMinMax reductions and the AbsMax backward.
//
// Replaces torch.max(torch.abs(x)[, dim]) (B/core/stats/stats_op.py:129-141), which materialises
// |x| (read + write) and then reduces it (read), and for per-channel activations first makes a
// permuted contiguous copy (B/core/function_wrapper/shape.py:19-27), with ONE streaming read of x
// in its native [outer, channels, inner] layout.  Algorithmic bytes per element: sizeof(x).
//
// abs-max works on the raw bit patterns: for |x| the IEEE order equals the unsigned-integer order
// of (bits & ~sign), and every NaN pattern is larger than +inf, so an unsigned max IS
// torch.max(torch.abs(x)) including its NaN propagation.  The result is exact (a max never rounds).
#include "bvq_common.h"
#include "bvq_ties.h"

namespace bvq {

constexpr int kStatUnroll = 8;

struct StatArgs {
  Tiling t;
  const void* x;
  uint32_t* part_a;  // ABSMAX: abs bits (as a float32 pattern) ; MINMAX: max as float bits
  uint32_t* part_b;  // MINMAX: min as float bits
};

__device__ __forceinline__ bool locate(const Tiling& t, int64_t& unit, int64_t& start, int64_t& len,
                                       int32_t& channel, int64_t& row_off) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= t.units) return false;
  const int64_t row = unit / t.ppr;
  const int64_t piece = unit - row * t.ppr;
  row_off = piece * t.piece_len;
  start = row * t.row_len + row_off;
  const int64_t rest = t.row_len - row_off;
  len = rest < t.piece_len ? rest : t.piece_len;
  channel = (int32_t)(row % t.channels);
  return true;
}

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void absmax_kernel(StatArgs a) {
  int64_t unit, start, len, row_off;
  int32_t channel;
  if (!locate(a.t, unit, start, len, channel, row_off)) return;
  const int lane = threadIdx.x & 63;
  const T* __restrict__ xp = reinterpret_cast<const T*>(a.x) + start;
  uint32_t m = 0;
  const int64_t nvec = len / VEC;
  for (int64_t base = 0; base < nvec; base += (int64_t)kWave * kStatUnroll) {
    vec_t<T, VEC> xv[kStatUnroll];
#pragma unroll
    for (int j = 0; j < kStatUnroll; ++j) {
      const int64_t i = base + (int64_t)j * kWave + lane;
      if (i < nvec) xv[j] = load_vec<T, VEC>(xp + i * VEC);
    }
#pragma unroll
    for (int j = 0; j < kStatUnroll; ++j) {
      const int64_t i = base + (int64_t)j * kWave + lane;
      if (i < nvec) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const uint32_t b = abs_bits<T>(xv[j].v[k]);
          m = b > m ? b : m;
        }
      }
    }
  }
  const int64_t i = nvec * VEC + lane;
  if (i < len) {
    const uint32_t b = abs_bits<T>(xp[i]);
    m = b > m ? b : m;
  }
  m = wave_max_u32(m);
  if (lane == 0) a.part_a[unit] = m;
}

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void minmax_kernel(StatArgs a) {
  int64_t unit, start, len, row_off;
  int32_t channel;
  if (!locate(a.t, unit, start, len, channel, row_off)) return;
  const int lane = threadIdx.x & 63;
  const T* __restrict__ xp = reinterpret_cast<const T*>(a.x) + start;
  float mx = -__builtin_inff(), mn = __builtin_inff();
  uint32_t nan = 0;
  const int64_t nvec = len / VEC;
  for (int64_t base = 0; base < nvec; base += (int64_t)kWave * kStatUnroll) {
    vec_t<T, VEC> xv[kStatUnroll];
#pragma unroll
    for (int j = 0; j < kStatUnroll; ++j) {
      const int64_t i = base + (int64_t)j * kWave + lane;
      if (i < nvec) xv[j] = load_vec<T, VEC>(xp + i * VEC);
    }
#pragma unroll
    for (int j = 0; j < kStatUnroll; ++j) {
      const int64_t i = base + (int64_t)j * kWave + lane;
      if (i < nvec) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const float f = to_f<T>(xv[j].v[k]);
          nan |= (f != f) ? 1u : 0u;
          mx = fmaxf(mx, f);
          mn = fminf(mn, f);
        }
      }
    }
  }
  const int64_t i = nvec * VEC + lane;
  if (i < len) {
    const float f = to_f<T>(xp[i]);
    nan |= (f != f) ? 1u : 0u;
    mx = fmaxf(mx, f);
    mn = fminf(mn, f);
  }
  mx = wave_max(mx);
  mn = wave_min(mn);
  nan = wave_or_u32(nan);
  if (lane == 0) {
    // torch.max / torch.min propagate NaN
    a.part_a[unit] = nan ? 0x7fc00000u : __builtin_bit_cast(uint32_t, mx);
    a.part_b[unit] = nan ? 0x7fc00000u : __builtin_bit_cast(uint32_t, mn);
  }
}

__device__ __forceinline__ void store_stat(void* out, int out_dtype, int64_t idx, float v) {
  if (out_dtype == BVQ_F32)
    reinterpret_cast<float*>(out)[idx] = v;
  else if (out_dtype == BVQ_BF16)
    reinterpret_cast<bf16_t*>(out)[idx] = (bf16_t)v;  // exact: v is a bf16 value
  else
    reinterpret_cast<f16_t*>(out)[idx] = (f16_t)v;
}

// one workgroup per channel; combines the per-unit partials of that channel
template <int KIND>
__global__ __launch_bounds__(kBlock) void stat_finish_kernel(const uint32_t* __restrict__ part_a,
                                                             const uint32_t* __restrict__ part_b,
                                                             void* out, int out_dtype, int in_dtype,
                                                             int64_t outer, int32_t channels,
                                                             int64_t ppr) {
  __shared__ uint32_t sha[kBlock];
  __shared__ float shx[kBlock], shn[kBlock];
  __shared__ uint32_t shnan[kBlock];
  const int32_t c = blockIdx.x;
  const int64_t n = outer * ppr;
  uint32_t m = 0;
  float mx = -__builtin_inff(), mn = __builtin_inff();
  uint32_t nan = 0;
  for (int64_t k = threadIdx.x; k < n; k += kBlock) {
    const int64_t o = k / ppr, p = k - o * ppr;
    const int64_t unit = (o * channels + c) * ppr + p;
    if (KIND == BVQ_STAT_ABSMAX) {
      const uint32_t b = part_a[unit];
      m = b > m ? b : m;
    } else {
      const float a = __builtin_bit_cast(float, part_a[unit]);
      const float b = __builtin_bit_cast(float, part_b[unit]);
      nan |= (a != a) ? 1u : 0u;
      mx = fmaxf(mx, a);
      mn = fminf(mn, b);
    }
  }
  sha[threadIdx.x] = m;
  shx[threadIdx.x] = mx;
  shn[threadIdx.x] = mn;
  shnan[threadIdx.x] = nan;
  __syncthreads();
  for (int st = kBlock / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      const uint32_t o = sha[threadIdx.x + st];
      if (o > sha[threadIdx.x]) sha[threadIdx.x] = o;
      shx[threadIdx.x] = fmaxf(shx[threadIdx.x], shx[threadIdx.x + st]);
      shn[threadIdx.x] = fminf(shn[threadIdx.x], shn[threadIdx.x + st]);
      shnan[threadIdx.x] |= shnan[threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (KIND == BVQ_STAT_ABSMAX) {
      float v;
      if (in_dtype == BVQ_F16) {
        v = (float)__builtin_bit_cast(f16_t, (uint16_t)sha[0]);
      } else {
        v = __builtin_bit_cast(float, sha[0]);
      }
      store_stat(out, out_dtype, c, v);
    } else {
      const float qn = __builtin_nanf("");
      store_stat(out, out_dtype, c, shnan[0] ? qn : shx[0]);
      store_stat(out, out_dtype, (int64_t)channels + c, shnan[0] ? qn : shn[0]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Backward of the statistics: locate the elements that attain the extremum ("ties") and deposit the
// gradient there.  MATCH_ABS: |x| == stat, deposit scaled by sgn(x) (torch.abs backward);
// MATCH_VALUE: x == stat (torch.max / torch.min of x itself).
// ------------------------------------------------------------------------------------------------
// torch.abs backward uses sgn(x): 0 at 0
__device__ __forceinline__ float sgn_f(float v) { return (float)(0.f < v) - (float)(v < 0.f); }

template <typename T, int MATCH>
__device__ __forceinline__ bool is_tie(T v, T stat) {
  if constexpr (MATCH == BVQ_MATCH_ABS) {
    return abs_bits<T>(v) == abs_bits<T>(stat);
  } else {
    return to_f<T>(v) == to_f<T>(stat);  // -0 == +0, NaN never matches (torch: input == value)
  }
}
// what a non-tie element receives in the reference: 0 * sgn(x) for AbsMax (a signed zero), +0 else
template <typename T, int MATCH>
__device__ __forceinline__ T zero_like(T v) {
  if constexpr (MATCH == BVQ_MATCH_ABS) {
    return from_f<T>(0.f * sgn_f(to_f<T>(v)));
  } else {
    return from_f<T>(0.f);
  }
}

template <typename T, int VEC, int MATCH, bool WRITE_ZERO>
__global__ __launch_bounds__(kBlock) void tie_scan_kernel(Tiling t, const void* x, const void* stat,
                                                          unsigned long long* info, void* dx,
                                                          int64_t inner) {
  int64_t unit, start, len, row_off;
  int32_t channel;
  if (!locate(t, unit, start, len, channel, row_off)) return;
  const int lane = threadIdx.x & 63;
  const T* __restrict__ xp = reinterpret_cast<const T*>(x) + start;
  T* __restrict__ dp = reinterpret_cast<T*>(dx) + start;
  const T sv = reinterpret_cast<const T*>(stat)[channel];
  const bool per_channel = t.channels > 1;
  // position of this unit's first element in the reference's reduction order for its channel:
  // (outer index) * inner + offset inside the row
  const int64_t row = unit / t.ppr;
  const int64_t pos0 = per_channel ? (row / t.channels) * inner + row_off : start;
  const int64_t nvec = len / VEC;
  for (int64_t base = 0; base < nvec + 1; base += kWave) {
    const int64_t i = base + lane;
    const int64_t e0 = i * VEC;
    if (i < nvec) {
      const vec_t<T, VEC> xv = load_vec<T, VEC>(xp + e0);
      vec_t<T, VEC> zv;
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        zv.v[k] = zero_like<T, MATCH>(xv.v[k]);
        if (is_tie<T, MATCH>(xv.v[k], sv)) {
          record_tie(info, per_channel, channel, (unsigned long long)(pos0 + e0 + k));
        }
      }
      if (WRITE_ZERO) store_vec<T, VEC>(dp + e0, zv);
    } else if (i == nvec) {
      // ragged end: lane `nvec` walks the (< VEC) leftover elements
      for (int64_t e = nvec * VEC; e < len; ++e) {
        if (WRITE_ZERO) dp[e] = zero_like<T, MATCH>(xp[e]);
        if (is_tie<T, MATCH>(xp[e], sv)) {
          record_tie(info, per_channel, channel, (unsigned long long)(pos0 + e));
        }
      }
    }
  }
}

__global__ void tie_init_kernel(unsigned long long* info, int32_t channels) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (channels > 1) {
    if (i < channels) info[i] = ~0ull;
  } else {
    if (i < 2) info[i] = 0ull;
  }
}

void launch_tie_init(unsigned long long* info, int64_t channels, hipStream_t st) {
  tie_init_kernel<<<dim3((unsigned)((channels + 255) / 256)), dim3(256), 0, st>>>(info, (int32_t)channels);
}

template <typename T, int MATCH>
__device__ __forceinline__ float deposit(float g, T xv) {
  if constexpr (MATCH == BVQ_MATCH_ABS) {
    return rnd<T>(g * sgn_f(to_f<T>(xv)));
  } else {
    return g;
  }
}

// channels > 1: one thread per channel deposits the gradient at first[c]
template <typename T, int MATCH>
__global__ void tie_apply_first_kernel(const void* x, const void* gstat, const unsigned long long* info,
                                       void* dx, int64_t outer, int32_t channels, int64_t inner,
                                       int mode_add) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= channels) return;
  const unsigned long long pos = info[c];
  if (pos == ~0ull) return;  // no element equals the statistic (e.g. NaN)
  const int64_t o = (int64_t)(pos / (unsigned long long)inner);
  const int64_t i = (int64_t)(pos - (unsigned long long)o * inner);
  const int64_t flat = (o * channels + c) * inner + i;
  const T* xp = reinterpret_cast<const T*>(x);
  T* dp = reinterpret_cast<T*>(dx);
  const float term = deposit<T, MATCH>(to_f<T>(reinterpret_cast<const T*>(gstat)[c]), xp[flat]);
  dp[flat] = mode_add ? from_f<T>(to_f<T>(dp[flat]) + term) : from_f<T>(term);
}

// channels == 1, ties fit the list: each tie receives (gstat / count)
template <typename T, int MATCH>
__global__ void tie_apply_list_kernel(const void* x, const void* gstat, const unsigned long long* info,
                                      const unsigned long long* total, void* dx, int mode_add) {
  const unsigned long long local = info[0];
  if (local == 0 || local > (unsigned long long)kTieCap) return;
  const unsigned long long cnt = total ? total[0] : local;  // ties over all shards of the tensor
  const T* xp = reinterpret_cast<const T*>(x);
  T* dp = reinterpret_cast<T*>(dx);
  // grad / mask.sum(): the count is an integer tensor, the quotient has the gradient's dtype
  // (the integer count is converted to the gradient's dtype first, as torch's type promotion does)
  const float share = rnd<T>(to_f<T>(reinterpret_cast<const T*>(gstat)[0]) / rnd<T>((float)cnt));
  for (unsigned long long k = blockIdx.x * blockDim.x + threadIdx.x; k < local;
       k += (unsigned long long)gridDim.x * blockDim.x) {
    const int64_t flat = (int64_t)info[2 + k];
    const float term = deposit<T, MATCH>(share, xp[flat]);
    dp[flat] = mode_add ? from_f<T>(to_f<T>(dp[flat]) + term) : from_f<T>(term);
  }
}

// channels == 1, more ties than the list holds (constant tensors, binarised weights): full pass
template <typename T, int MATCH>
__global__ __launch_bounds__(kBlock) void tie_apply_full_kernel(const void* x, const void* stat,
                                                                const void* gstat,
                                                                const unsigned long long* info,
                                                                const unsigned long long* total,
                                                                void* dx, int64_t n, int mode_add) {
  const unsigned long long local = info[0];
  if (local <= (unsigned long long)kTieCap) return;
  const unsigned long long cnt = total ? total[0] : local;
  const T* xp = reinterpret_cast<const T*>(x);
  T* dp = reinterpret_cast<T*>(dx);
  const T sv = reinterpret_cast<const T*>(stat)[0];
  const float share = rnd<T>(to_f<T>(reinterpret_cast<const T*>(gstat)[0]) / rnd<T>((float)cnt));
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const T xv = xp[i];
    if (is_tie<T, MATCH>(xv, sv)) {
      const float term = deposit<T, MATCH>(share, xv);
      dp[i] = mode_add ? from_f<T>(to_f<T>(dp[i]) + term) : from_f<T>(term);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static Tiling stat_tiling(int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner,
                          int& vec) {
  int64_t rows, row_len;
  if (channels > 1) {
    rows = outer * channels;
    row_len = inner;
  } else {
    rows = 1;
    row_len = outer * inner;
  }
  const int full = 16 / dtype_size(dtype);
  const void* ptrs[1] = {x};
  const int els[1] = {dtype_size(dtype)};
  vec = pick_vec(full, rows, row_len, ptrs, els, 1);
  vec = vec == full ? full : (vec >= 2 && full > 2 ? 2 : 1);
  return make_tiling(rows, row_len, (int32_t)channels, vec);
}

static int64_t worst_units(int dtype, int64_t outer, int64_t channels, int64_t inner) {
  int64_t rows = channels > 1 ? outer * channels : 1;
  int64_t row_len = channels > 1 ? inner : outer * inner;
  int64_t worst = 0;
  for (int v = 1; v <= 16 / dtype_size(dtype); v <<= 1) {
    Tiling t = make_tiling(rows, row_len, (int32_t)channels, v);
    if (t.units > worst) worst = t.units;
  }
  return worst;
}

template <typename T>
static void launch_stat(int kind, const StatArgs& a, int vec, hipStream_t st) {
  constexpr int V = elem<T>::vec;
  const dim3 grid(grid_for_units(a.t.units)), block(kBlock);
  if (kind == BVQ_STAT_ABSMAX) {
    if (vec == V)
      absmax_kernel<T, V><<<grid, block, 0, st>>>(a);
    else if (vec == 2)
      absmax_kernel<T, 2><<<grid, block, 0, st>>>(a);
    else
      absmax_kernel<T, 1><<<grid, block, 0, st>>>(a);
  } else {
    if (vec == V)
      minmax_kernel<T, V><<<grid, block, 0, st>>>(a);
    else if (vec == 2)
      minmax_kernel<T, 2><<<grid, block, 0, st>>>(a);
    else
      minmax_kernel<T, 1><<<grid, block, 0, st>>>(a);
  }
}

template <typename T, int MATCH, bool WZ>
static void launch_tie_scan_v(const Tiling& t, int vec, const void* x, const void* stat,
                              unsigned long long* info, void* dx, int64_t inner, hipStream_t st) {
  constexpr int V = elem<T>::vec;
  const dim3 grid(grid_for_units(t.units)), block(kBlock);
  if (vec == V)
    tie_scan_kernel<T, V, MATCH, WZ><<<grid, block, 0, st>>>(t, x, stat, info, dx, inner);
  else if (vec == 2)
    tie_scan_kernel<T, 2, MATCH, WZ><<<grid, block, 0, st>>>(t, x, stat, info, dx, inner);
  else
    tie_scan_kernel<T, 1, MATCH, WZ><<<grid, block, 0, st>>>(t, x, stat, info, dx, inner);
}

template <typename T, int MATCH>
static void run_tie_scan(const Tiling& t, int vec, const void* x, const void* stat,
                         unsigned long long* info, void* dx, int64_t inner, int write_zeros,
                         hipStream_t st) {
  if (write_zeros)
    launch_tie_scan_v<T, MATCH, true>(t, vec, x, stat, info, dx, inner, st);
  else
    launch_tie_scan_v<T, MATCH, false>(t, vec, x, stat, info, dx, inner, st);
}

template <typename T, int MATCH>
static void run_tie_apply(const void* x, const void* stat, const void* gstat,
                          const unsigned long long* info, const unsigned long long* total, void* dx,
                          int64_t outer, int64_t channels, int64_t inner, int mode_add, hipStream_t st) {
  if (channels > 1) {
    const unsigned nb = (unsigned)((channels + 255) / 256);
    tie_apply_first_kernel<T, MATCH><<<dim3(nb), dim3(256), 0, st>>>(x, gstat, info, dx, outer,
                                                                     (int32_t)channels, inner, mode_add);
  } else {
    tie_apply_list_kernel<T, MATCH><<<dim3(4), dim3(256), 0, st>>>(x, gstat, info, total, dx, mode_add);
    const int64_t n = outer * inner;
    int64_t nb = (n + kBlock - 1) / kBlock;
    if (nb > 2048) nb = 2048;
    // exits immediately unless the tie list overflowed
    tie_apply_full_kernel<T, MATCH><<<dim3((unsigned)nb), dim3(kBlock), 0, st>>>(x, stat, gstat, info,
                                                                                 total, dx, n, mode_add);
  }
}

}  // namespace bvq

using namespace bvq;

static int bad_dtype(int dt) { return dt < BVQ_F32 || dt > BVQ_F16; }

extern "C" int64_t bvq_stats_workspace_bytes(int kind, int dtype, int64_t outer, int64_t channels,
                                             int64_t inner) {
  if (bad_dtype(dtype) || outer < 0 || channels < 1 || inner < 0) return -1;
  (void)kind;
  const int64_t units = worst_units(dtype, outer, channels, inner);
  const int64_t partials = 2 * units * (int64_t)sizeof(uint32_t);
  const int64_t tie = (channels > 1 ? channels : 2 + kTieCap) * (int64_t)sizeof(int64_t);
  return partials + tie + 256;
}

extern "C" int bvq_stats(int kind, int dtype, const void* x, int64_t outer, int64_t channels,
                         int64_t inner, int out_dtype, void* out, void* workspace,
                         int64_t workspace_bytes, bvq_stream_t stream) {
  if (bad_dtype(dtype) || bad_dtype(out_dtype) || outer < 0 || channels < 1 || inner < 0 ||
      (kind != BVQ_STAT_ABSMAX && kind != BVQ_STAT_MINMAX)) {
    set_error("bvq_stats: bad argument");
    return BVQ_ERR_INVALID;
  }
  if (out_dtype != BVQ_F32 && out_dtype != dtype) {
    set_error("bvq_stats: out_dtype must be f32 or the dtype of x");
    return BVQ_ERR_UNSUPPORTED;
  }
  const int64_t n = outer * channels * inner;
  if (n == 0) {
    set_error("bvq_stats: empty input has no maximum");  // torch.max raises on empty tensors too
    return BVQ_ERR_INVALID;
  }
  if (!x || !out || !workspace) {
    set_error("bvq_stats: null pointer");
    return BVQ_ERR_INVALID;
  }
  int vec;
  StatArgs a;
  a.t = stat_tiling(dtype, x, outer, channels, inner, vec);
  const int64_t need = 2 * a.t.units * (int64_t)sizeof(uint32_t);
  if (workspace_bytes < need) {
    set_error("bvq_stats: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
    return BVQ_ERR_WORKSPACE;
  }
  a.x = x;
  a.part_a = reinterpret_cast<uint32_t*>(workspace);
  a.part_b = a.part_a + a.t.units;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == BVQ_F32)
    launch_stat<float>(kind, a, vec, st);
  else if (dtype == BVQ_BF16)
    launch_stat<bf16_t>(kind, a, vec, st);
  else
    launch_stat<f16_t>(kind, a, vec, st);
  int rc = check_launch("bvq_stats");
  if (rc) return rc;
  const int64_t outer_rows = channels > 1 ? outer : 1;
  if (kind == BVQ_STAT_ABSMAX)
    stat_finish_kernel<BVQ_STAT_ABSMAX><<<dim3((unsigned)channels), dim3(kBlock), 0, st>>>(
        a.part_a, a.part_b, out, out_dtype, dtype, outer_rows, (int32_t)channels, a.t.ppr);
  else
    stat_finish_kernel<BVQ_STAT_MINMAX><<<dim3((unsigned)channels), dim3(kBlock), 0, st>>>(
        a.part_a, a.part_b, out, out_dtype, dtype, outer_rows, (int32_t)channels, a.t.ppr);
  return check_launch("bvq_stats/finish");
}

#define BVQ_DISPATCH_T_MATCH(dtype, match, FN, ...)              \
  do {                                                             \
    if ((dtype) == BVQ_F32) {                                      \
      if ((match) == BVQ_MATCH_ABS)                                \
        FN<float, BVQ_MATCH_ABS>(__VA_ARGS__);                     \
      else                                                         \
        FN<float, BVQ_MATCH_VALUE>(__VA_ARGS__);                   \
    } else if ((dtype) == BVQ_BF16) {                              \
      if ((match) == BVQ_MATCH_ABS)                                \
        FN<bf16_t, BVQ_MATCH_ABS>(__VA_ARGS__);                    \
      else                                                         \
        FN<bf16_t, BVQ_MATCH_VALUE>(__VA_ARGS__);                  \
    } else {                                                       \
      if ((match) == BVQ_MATCH_ABS)                                \
        FN<f16_t, BVQ_MATCH_ABS>(__VA_ARGS__);                     \
      else                                                         \
        FN<f16_t, BVQ_MATCH_VALUE>(__VA_ARGS__);                   \
    }                                                              \
  } while (0)

static int check_stat_args(const char* fn, int match, int dtype, int64_t outer, int64_t channels,
                           int64_t inner) {
  if (bad_dtype(dtype) || outer < 0 || channels < 1 || inner < 0 ||
      (match != BVQ_MATCH_ABS && match != BVQ_MATCH_VALUE)) {
    set_error("%s: bad argument", fn);
    return BVQ_ERR_INVALID;
  }
  return BVQ_OK;
}

extern "C" int64_t bvq_tie_info_bytes(int64_t channels) {
  if (channels < 1) return -1;
  return tie_info_words(channels) * (int64_t)sizeof(int64_t);
}

extern "C" int bvq_stat_tie_scan(int match, int dtype, const void* x, const void* stat, int64_t outer,
                                 int64_t channels, int64_t inner, void* dx_zero_fill, int64_t* tie_info,
                                 bvq_stream_t stream) {
  int rc = check_stat_args("bvq_stat_tie_scan", match, dtype, outer, channels, inner);
  if (rc) return rc;
  if (!tie_info) {
    set_error("bvq_stat_tie_scan: null tie_info");
    return BVQ_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* info = reinterpret_cast<unsigned long long*>(tie_info);
  launch_tie_init(info, channels, st);
  if (outer * channels * inner == 0) return check_launch("bvq_stat_tie_scan");
  if (!x || !stat) {
    set_error("bvq_stat_tie_scan: null pointer");
    return BVQ_ERR_INVALID;
  }
  int vec;
  Tiling t = stat_tiling(dtype, x, outer, channels, inner, vec);
  if (dx_zero_fill && reinterpret_cast<uintptr_t>(dx_zero_fill) % 16 != 0 && vec > 1) {
    vec = 1;
    t = make_tiling(t.rows, t.row_len, t.channels, 1);
  }
  BVQ_DISPATCH_T_MATCH(dtype, match, run_tie_scan, t, vec, x, stat, info, dx_zero_fill, inner,
                       dx_zero_fill != nullptr, st);
  return check_launch("bvq_stat_tie_scan");
}

extern "C" int bvq_stat_tie_apply(int match, int dtype, const void* x, const void* stat, const void* gstat,
                                  const int64_t* tie_info, const int64_t* total_ties, void* dx,
                                  int64_t outer, int64_t channels, int64_t inner, int mode_add,
                                  bvq_stream_t stream) {
  int rc = check_stat_args("bvq_stat_tie_apply", match, dtype, outer, channels, inner);
  if (rc) return rc;
  if (outer * channels * inner == 0) return BVQ_OK;
  if (!x || !stat || !gstat || !tie_info || !dx) {
    set_error("bvq_stat_tie_apply: null pointer");
    return BVQ_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  BVQ_DISPATCH_T_MATCH(dtype, match, run_tie_apply, x, stat, gstat,
                       reinterpret_cast<const unsigned long long*>(tie_info),
                       reinterpret_cast<const unsigned long long*>(total_ties), dx, outer, channels, inner,
                       mode_add, st);
  return check_launch("bvq_stat_tie_apply");
}

extern "C" int bvq_stat_bwd(int match, int dtype, const void* x, const void* stat, const void* gstat,
                            void* dx, int64_t outer, int64_t channels, int64_t inner, int mode_add,
                            void* workspace, int64_t workspace_bytes, bvq_stream_t stream) {
  int rc = check_stat_args("bvq_stat_bwd", match, dtype, outer, channels, inner);
  if (rc) return rc;
  if (outer * channels * inner == 0) return BVQ_OK;
  if (!workspace || workspace_bytes < bvq_tie_info_bytes(channels)) {
    set_error("bvq_stat_bwd: workspace too small");
    return BVQ_ERR_WORKSPACE;
  }
  int64_t* info = reinterpret_cast<int64_t*>(workspace);
  rc = bvq_stat_tie_scan(match, dtype, x, stat, outer, channels, inner, mode_add ? nullptr : dx, info,
                         stream);
  if (rc) return rc;
  return bvq_stat_tie_apply(match, dtype, x, stat, gstat, info, nullptr, dx, outer, channels, inner,
                            mode_add, stream);
}
