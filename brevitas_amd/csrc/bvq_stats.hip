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
#include <math.h>
#include <string.h>

#include "bvq_common.h"
#include "bvq_ties.h"
#include "bvq_sums.h"

namespace bvq {

constexpr int kStatUnroll = 8;

struct StatArgs {
  Tiling t;
  const void* x;
  uint32_t* part_a;  // ABSMAX: abs bits (as a float32 pattern) ; MINMAX: max as float bits
  uint32_t* part_b;  // MINMAX: min as float bits
  float* pivot;      // moments: [channels] the value the sums are shifted by (written by the kernel)
};

template <typename T, int VEC, bool NT, bool RELU>
__global__ __launch_bounds__(kBlock) void absmax_kernel(StatArgs a) {
  const Unit u = locate_unit(a.t);
  if (!u.valid) return;
  const int lane = threadIdx.x & 63;
  const T* __restrict__ xp = reinterpret_cast<const T*>(a.x) + u.base;
  uint32_t m = 0;
  ChunkWalker cur;
  cur.init(u, VEC, lane);
  const int64_t total = (int64_t)u.nrows * cur.cpr;
  for (int64_t done = 0; done < total; done += (int64_t)kWave * kStatUnroll) {
    vec_t<T, VEC> xv[kStatUnroll];
    bool ok[kStatUnroll];
#pragma unroll
    for (int j = 0; j < kStatUnroll; ++j) {
      ok[j] = cur.valid();
      if (ok[j]) xv[j] = load_vec<T, VEC, NT>(xp + cur.offset(u.row_stride, VEC));
      cur.next();
    }
#pragma unroll
    for (int j = 0; j < kStatUnroll; ++j) {
      if (ok[j]) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const uint32_t b = pre_abs_bits<T, RELU>(xv[j].v[k]);
          m = b > m ? b : m;
        }
      }
    }
  }
  // ragged ends: the (< VEC) elements after the last full chunk of every row of the unit
  const int32_t tail = (int32_t)(u.len - (int64_t)cur.cpr * VEC);
  for (int32_t e = lane; e < u.nrows * tail; e += kWave) {
    const int32_t r = e / tail, k = e - r * tail;
    const uint32_t b = pre_abs_bits<T, RELU>(xp[(int64_t)r * u.row_stride + (int64_t)cur.cpr * VEC + k]);
    m = b > m ? b : m;
  }
  m = wave_max_u32(m);
  if (lane == 0) a.part_a[u.id] = m;
}

template <typename T, int VEC, bool NT, bool RELU>
__global__ __launch_bounds__(kBlock) void minmax_kernel(StatArgs a) {
  const Unit u = locate_unit(a.t);
  if (!u.valid) return;
  const int lane = threadIdx.x & 63;
  const T* __restrict__ xp = reinterpret_cast<const T*>(a.x) + u.base;
  float mx = -__builtin_inff(), mn = __builtin_inff();
  uint32_t nan = 0;
  ChunkWalker cur;
  cur.init(u, VEC, lane);
  const int64_t total = (int64_t)u.nrows * cur.cpr;
  for (int64_t done = 0; done < total; done += (int64_t)kWave * kStatUnroll) {
    vec_t<T, VEC> xv[kStatUnroll];
    bool ok[kStatUnroll];
#pragma unroll
    for (int j = 0; j < kStatUnroll; ++j) {
      ok[j] = cur.valid();
      if (ok[j]) xv[j] = load_vec<T, VEC, NT>(xp + cur.offset(u.row_stride, VEC));
      cur.next();
    }
#pragma unroll
    for (int j = 0; j < kStatUnroll; ++j) {
      if (ok[j]) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const float f = RELU ? relu_f(to_f<T>(xv[j].v[k])) : to_f<T>(xv[j].v[k]);
          nan |= (f != f) ? 1u : 0u;
          mx = fmaxf(mx, f);
          mn = fminf(mn, f);
        }
      }
    }
  }
  const int32_t tail = (int32_t)(u.len - (int64_t)cur.cpr * VEC);  // ragged ends of every row
  for (int32_t e = lane; e < u.nrows * tail; e += kWave) {
    const int32_t r = e / tail, k = e - r * tail;
    const T xe = xp[(int64_t)r * u.row_stride + (int64_t)cur.cpr * VEC + k];
    const float f = RELU ? relu_f(to_f<T>(xe)) : to_f<T>(xe);
    nan |= (f != f) ? 1u : 0u;
    mx = fmaxf(mx, f);
    mn = fminf(mn, f);
  }
  mx = wave_max(mx);
  mn = wave_min(mn);
  nan = wave_or_u32(nan);
  if (lane == 0) {
    // torch.max / torch.min propagate NaN
    a.part_a[u.id] = nan ? 0x7fc00000u : __builtin_bit_cast(uint32_t, mx);
    a.part_b[u.id] = nan ? 0x7fc00000u : __builtin_bit_cast(uint32_t, mn);
  }
}

// ---- first and second moment of |x| (AbsAve, MeanSigmaStd, B/core/stats/stats_op.py:186-262) -------
// One streaming read: per-unit float32 partial sums of d = |x| - p and d^2, combined by the fixed-order
// channel sums (double).  p is the channel's PIVOT, |x| of its first element (0 if that is not finite):
// the variance (sum d^2 - (sum d)^2 / n) / (n - 1) of the shifted values does not cancel when the mean of
// |x| is far larger than its spread (|x| = 100 +- 0.01), where sum x^2 - (sum |x|)^2 / n of float32 sums
// is garbage.  torch.var itself is a two-pass / Welford computation.
template <typename T>
__device__ __forceinline__ float moments_pivot(const void* x, int64_t first) {
  const float p = fabsf(to_f<T>(reinterpret_cast<const T*>(x)[first]));
  return p <= 3.4028234663852886e38f ? p : 0.f;  // inf / NaN: no shift
}

template <typename T, int VEC, bool NT>
__global__ __launch_bounds__(kBlock) void absmoments_kernel(StatArgs a) {
  const Unit u = locate_unit(a.t);
  if (!u.valid) return;
  const int lane = threadIdx.x & 63;
  const T* __restrict__ xp = reinterpret_cast<const T*>(a.x) + u.base;
  const float p = moments_pivot<T>(a.x, (int64_t)u.channel * a.t.row_len);
  float s1 = 0.f, s2 = 0.f;
  ChunkWalker cur;
  cur.init(u, VEC, lane);
  const int64_t total = (int64_t)u.nrows * cur.cpr;
  for (int64_t done = 0; done < total; done += (int64_t)kWave * kStatUnroll) {
    vec_t<T, VEC> xv[kStatUnroll];
    bool ok[kStatUnroll];
#pragma unroll
    for (int j = 0; j < kStatUnroll; ++j) {
      ok[j] = cur.valid();
      if (ok[j]) xv[j] = load_vec<T, VEC, NT>(xp + cur.offset(u.row_stride, VEC));
      cur.next();
    }
#pragma unroll
    for (int j = 0; j < kStatUnroll; ++j) {
      if (ok[j]) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const float d = fabsf(to_f<T>(xv[j].v[k])) - p;
          s1 += d;
          s2 += d * d;
        }
      }
    }
  }
  const int64_t i = (int64_t)cur.cpr * VEC + lane;
  if (u.nrows == 1 && i < u.len) {
    const float d = fabsf(to_f<T>(xp[i])) - p;
    s1 += d;
    s2 += d * d;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) {
    a.part_a[u.id] = __builtin_bit_cast(uint32_t, s1);
    a.part_b[u.id] = __builtin_bit_cast(uint32_t, s2);
    if (u.base == (int64_t)u.channel * a.t.row_len) a.pivot[u.channel] = p;  // the channel's first unit
  }
}

// dx = sgn(x) * (a[c] + b[c] * |x|): the backward of any statistic that is a function of the mean and the
// variance of |x| (sgn(0) = 0, torch.abs's subgradient)
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void abs_affine_bwd_kernel(Tiling t, const void* x, const float* ca,
                                                                const float* cb, void* dx) {
  const Unit u = locate_unit(t);
  if (!u.valid) return;
  const int lane = threadIdx.x & 63;
  const T* __restrict__ xp = reinterpret_cast<const T*>(x) + u.base;
  T* __restrict__ dp = reinterpret_cast<T*>(dx) + u.base;
  const float a = ca[u.channel], b = cb[u.channel];
  auto f = [a, b](T v) -> T {
    const float xf = to_f<T>(v);
    const float m = a + b * fabsf(xf);
    return from_f<T>(xf > 0.f ? m : (xf < 0.f ? -m : (xf == 0.f ? 0.f : xf)));  // NaN in, NaN out
  };
  ChunkWalker cur;
  cur.init(u, VEC, lane);
  const int64_t total = (int64_t)u.nrows * cur.cpr;
  for (int64_t done = 0; done < total; done += kWave) {
    if (cur.valid()) {
      const int64_t off = cur.offset(u.row_stride, VEC);
      const vec_t<T, VEC> xv = load_vec<T, VEC>(xp + off);
      vec_t<T, VEC> dv;
#pragma unroll
      for (int k = 0; k < VEC; ++k) dv.v[k] = f(xv.v[k]);
      store_vec<T, VEC>(dp + off, dv);
    }
    cur.next();
  }
  if (u.nrows == 1 && lane == 0) {
    for (int64_t e = (int64_t)cur.cpr * VEC; e < u.len; ++e) dp[e] = f(xp[e]);
  }
}

// the same for channel-last layouts (short `inner`): the tensor is rows of L = channels * inner elements; a thread owns
// one 16-byte column chunk -- its 8 (4) channels and their coefficients never change -- and walks down the rows with a
// grid stride in y (consecutive threads read consecutive chunks of a row: coalesced)
template <typename T>
__global__ __launch_bounds__(kBlock) void abs_affine_bwd_cols_kernel(const void* x, const float* ca, const float* cb,
                                                                     void* dx, int64_t rows, int64_t L, int64_t inner) {
  constexpr int VEC = elem<T>::vec;
  const int64_t cc = (int64_t)blockIdx.x * kBlock + threadIdx.x;  // column chunk
  if (cc * VEC >= L) return;
  float a[VEC], b[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    const int64_t c = (cc * VEC + k) / inner;
    a[k] = ca[c];
    b[k] = cb[c];
  }
  const T* __restrict__ xp = reinterpret_cast<const T*>(x) + cc * VEC;
  T* __restrict__ dp = reinterpret_cast<T*>(dx) + cc * VEC;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const vec_t<T, VEC> xv = load_vec<T, VEC>(xp + r * L);
    vec_t<T, VEC> dv;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const float xf = to_f<T>(xv.v[k]);
      const float m = a[k] + b[k] * fabsf(xf);
      dv.v[k] = from_f<T>(xf > 0.f ? m : (xf < 0.f ? -m : (xf == 0.f ? 0.f : xf)));  // NaN in, NaN out
    }
    store_vec<T, VEC>(dp + r * L, dv);
  }
}

// ---- column-mapped abs-max (ColsPlan in bvq_common.h) -------------------------------------------------
// a lane keeps one running maximum per column of its chunk and writes them as partial row
// (row block * rpp + sub row) of the [partial rows][L] array the finishing kernel reduces
struct ColsStatArgs {
  ColsPlan p;
  const void* x;
  uint32_t* part;  // [prows][L]
};

template <typename T, bool NT, bool RELU>
__global__ __launch_bounds__(kBlock) void absmax_cols_kernel(ColsStatArgs a) {
  constexpr int VEC = elem<T>::vec;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.p.units) return;
  const int64_t rblk = unit / a.p.strips;
  const int32_t strip = (int32_t)(unit - rblk * a.p.strips);
  const int32_t sub = lane / a.p.lpr;                       // which of the rpp rows of a pass
  const int32_t chunk = strip * kWave + (lane - sub * a.p.lpr);  // column chunk of this lane
  const bool active = sub < a.p.rpp && chunk < a.p.cps;
  const int64_t row0 = rblk * a.p.rb + sub;
  const int64_t row_end = (rblk + 1) * a.p.rb < a.p.rows ? (rblk + 1) * a.p.rb : a.p.rows;
  const T* __restrict__ xp = reinterpret_cast<const T*>(a.x) + (int64_t)chunk * VEC;
  uint32_t m[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) m[k] = 0;
  if (active) {
    constexpr int kU = 4;
    for (int64_t r = row0; r < row_end; r += (int64_t)kU * a.p.rpp) {
      vec_t<T, VEC> xv[kU];
      bool ok[kU];
#pragma unroll
      for (int j = 0; j < kU; ++j) {
        const int64_t rr = r + (int64_t)j * a.p.rpp;
        ok[j] = rr < row_end;
        xv[j] = load_vec<T, VEC, NT>(xp + (ok[j] ? rr : row0) * a.p.L);
      }
#pragma unroll
      for (int j = 0; j < kU; ++j) {
        if (ok[j]) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            const uint32_t b = pre_abs_bits<T, RELU>(xv[j].v[k]);
            m[k] = b > m[k] ? b : m[k];
          }
        }
      }
    }
    uint32_t* out = a.part + (rblk * a.p.rpp + sub) * a.p.L + (int64_t)chunk * VEC;
#pragma unroll
    for (int k = 0; k < VEC; ++k) out[k] = m[k];
  }
}

// ---- column-mapped moments (AbsAve / MeanSigmaStd on channel-last layouts) ---------------------------------------
// the same decomposition: a lane keeps sum(|x| - p) and sum((|x| - p)^2) per column of its chunk (p: the pivot of the
// column's channel, moments_pivot above) and writes them as its partial row of two [partial rows][L] float arrays;
// the fold of the scale-gradient sums (double accumulation, rows in order) and channel_sum_kernel finish them.
struct ColsMomentArgs {
  ColsPlan p;
  const void* x;
  float* part1;  // [prows][L] sum of d
  float* part2;  // [prows][L] sum of d * d
  float* pivot;  // [channels], written by row block 0
  int64_t inner;
};

template <typename T, bool NT>
__global__ __launch_bounds__(kBlock) void absmoments_cols_kernel(ColsMomentArgs a) {
  constexpr int VEC = elem<T>::vec;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.p.units) return;
  const int64_t rblk = unit / a.p.strips;
  const int32_t strip = (int32_t)(unit - rblk * a.p.strips);
  const int32_t sub = lane / a.p.lpr;
  const int32_t chunk = strip * kWave + (lane - sub * a.p.lpr);
  const bool active = sub < a.p.rpp && chunk < a.p.cps;
  if (!active) return;
  const int64_t row0 = rblk * a.p.rb + sub;
  const int64_t row_end = (rblk + 1) * a.p.rb < a.p.rows ? (rblk + 1) * a.p.rb : a.p.rows;
  const T* __restrict__ xp = reinterpret_cast<const T*>(a.x) + (int64_t)chunk * VEC;
  float pv[VEC], s1[VEC], s2[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    const int64_t col = (int64_t)chunk * VEC + k;
    const int64_t c = col / a.inner;
    pv[k] = moments_pivot<T>(a.x, c * a.inner);  // |x| of the channel's first element (row 0)
    s1[k] = s2[k] = 0.f;
    if (rblk == 0 && sub == 0 && col == c * a.inner) a.pivot[c] = pv[k];
  }
  constexpr int kU = 4;
  for (int64_t r = row0; r < row_end; r += (int64_t)kU * a.p.rpp) {
    vec_t<T, VEC> xv[kU];
    bool ok[kU];
#pragma unroll
    for (int j = 0; j < kU; ++j) {
      const int64_t rr = r + (int64_t)j * a.p.rpp;
      ok[j] = rr < row_end;
      xv[j] = load_vec<T, VEC, NT>(xp + (ok[j] ? rr : row0) * a.p.L);
    }
#pragma unroll
    for (int j = 0; j < kU; ++j) {
      if (ok[j]) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const float d = fabsf(to_f<T>(xv[j].v[k])) - pv[k];
          s1[k] += d;
          s2[k] += d * d;
        }
      }
    }
  }
  const int64_t base = (rblk * a.p.rpp + sub) * a.p.L + (int64_t)chunk * VEC;
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    a.part1[base + k] = s1[k];
    a.part2[base + k] = s2[k];
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

// min/max over the same mapping.  Partials are order-preserving unsigned keys so that the abs-max fold
// (an unsigned max) serves both: columns [0, L) hold key(max), columns [L, 2L) hold key(-min); a NaN
// anywhere in a column turns both of its keys into 0xffffffff (torch.max / torch.min propagate NaN).
__device__ __forceinline__ uint32_t order_key(float f) {
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u >> 31) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float order_key_value(uint32_t k) {
  return __builtin_bit_cast(float, (k >> 31) ? (k ^ 0x80000000u) : ~k);  // 0xffffffff -> a NaN pattern
}

template <typename T, bool NT, bool RELU>
__global__ __launch_bounds__(kBlock) void minmax_cols_kernel(ColsStatArgs a) {
  constexpr int VEC = elem<T>::vec;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.p.units) return;
  const int64_t rblk = unit / a.p.strips;
  const int32_t strip = (int32_t)(unit - rblk * a.p.strips);
  const int32_t sub = lane / a.p.lpr;
  const int32_t chunk = strip * kWave + (lane - sub * a.p.lpr);
  const bool active = sub < a.p.rpp && chunk < a.p.cps;
  const int64_t row0 = rblk * a.p.rb + sub;
  const int64_t row_end = (rblk + 1) * a.p.rb < a.p.rows ? (rblk + 1) * a.p.rb : a.p.rows;
  const T* __restrict__ xp = reinterpret_cast<const T*>(a.x) + (int64_t)chunk * VEC;
  float mx[VEC], mn[VEC];
  uint32_t nan = 0;  // bit k: column k saw a NaN
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    mx[k] = -__builtin_inff();
    mn[k] = __builtin_inff();
  }
  if (active) {
    constexpr int kU = 4;
    for (int64_t r = row0; r < row_end; r += (int64_t)kU * a.p.rpp) {
      vec_t<T, VEC> xv[kU];
      bool ok[kU];
#pragma unroll
      for (int j = 0; j < kU; ++j) {
        const int64_t rr = r + (int64_t)j * a.p.rpp;
        ok[j] = rr < row_end;
        xv[j] = load_vec<T, VEC, NT>(xp + (ok[j] ? rr : row0) * a.p.L);
      }
#pragma unroll
      for (int j = 0; j < kU; ++j) {
        if (ok[j]) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            const float f = RELU ? relu_f(to_f<T>(xv[j].v[k])) : to_f<T>(xv[j].v[k]);
            nan |= (f != f) ? (1u << k) : 0u;
            mx[k] = fmaxf(mx[k], f);
            mn[k] = fminf(mn[k], f);
          }
        }
      }
    }
    uint32_t* out = a.part + (rblk * a.p.rpp + sub) * (2 * a.p.L) + (int64_t)chunk * VEC;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const bool bad = (nan >> k) & 1u;
      out[k] = bad ? 0xffffffffu : order_key(mx[k]);
      out[a.p.L + k] = bad ? 0xffffffffu : order_key(-mn[k]);
    }
  }
}

// folded keys [2][L] -> max[channels], min[channels]: channel c owns columns [c * inner, (c + 1) * inner)
__global__ __launch_bounds__(kWave) void minmax_cols_finish_kernel(const uint32_t* __restrict__ folded, void* out,
                                                                   int out_dtype, int32_t channels, int64_t inner) {
  const int32_t c = blockIdx.x;
  const int64_t L = (int64_t)channels * inner;
  uint32_t kx = 0, kn = 0;
  for (int64_t i = threadIdx.x; i < inner; i += kWave) {
    const uint32_t a = folded[(int64_t)c * inner + i], b = folded[L + (int64_t)c * inner + i];
    kx = a > kx ? a : kx;
    kn = b > kn ? b : kn;
  }
  kx = wave_max_u32(kx);
  kn = wave_max_u32(kn);
  if (threadIdx.x == 0) {
    const bool bad = kx == 0xffffffffu || kn == 0xffffffffu;
    const float qn = __builtin_nanf("");
    store_stat(out, out_dtype, c, bad ? qn : order_key_value(kx));
    store_stat(out, out_dtype, (int64_t)channels + c, bad ? qn : -order_key_value(kn));
  }
}

// optional epilogue of the abs-max finisher: statistic -> scale in the same launch
//   thr   = scalar_clamp_min_ste(stat, min_val)      (B/core/restrict_val.py:22-42)
//   scale = thr / int_threshold                       (B/core/quant/int.py:160)
// min_val is already rounded to the statistic's dtype and int_threshold to the dtype the division
// runs in, so the kernel only has to round the quotient to scale_dtype.
struct ScaleEpilogue {
  void* scale_out;  // null: no epilogue
  int32_t scale_dtype;
  int32_t use_min;
  float min_val;
  float int_threshold;
  // optionally, in the same launch: _RuntimeStats' running average of the statistic (B/core/stats/stats_wrapper.py:61-66)
  void* running;    // null: none
  int32_t run_dtype, first_batch;
  float momentum, one_minus_m;
};

// running *= out (first batch)  |  running *= (1 - momentum); running += momentum * out -- every torch op rounds to
// its result dtype: running's for the in-place ops, out's for momentum * out
__device__ __forceinline__ float running_update(float r, float o, int run_dtype, int stat_dtype, float one_minus_m,
                                                float m, int first) {
  auto round_to = [](float v, int dt) {
    return dt == BVQ_F32 ? v : (dt == BVQ_BF16 ? rnd<bf16_t>(v) : rnd<f16_t>(v));
  };
  if (first) return round_to(r * o, run_dtype);
  r = round_to(r * one_minus_m, run_dtype);
  const float u = round_to(o * m, stat_dtype);
  return round_to(r + u, run_dtype);
}

// Combines the per-unit partials of one channel.  grid = (channels, splits): with splits == 1 the
// workgroup of channel c reduces all of them and writes the statistic (and the scale epilogue); a
// per-tensor statistic of a large activation has ~10^5 partials in its single channel, so there the
// range is cut into `splits` slices whose results go to mid_a/mid_b[c * splits + s] and a second launch
// of this kernel (nob = 1, ppr = splits) finishes.  max/min are exact, so the split changes nothing.
constexpr int64_t kFinishSlice = 4096;  // partials per workgroup of the first stage

static inline int32_t finish_splits(int64_t partials_per_channel) {
  const int64_t s = (partials_per_channel + kFinishSlice - 1) / kFinishSlice;
  return (int32_t)(s < 1 ? 1 : s);
}

template <int KIND>
__global__ __launch_bounds__(kBlock) void stat_finish_kernel(const uint32_t* __restrict__ part_a,
                                                             const uint32_t* __restrict__ part_b,
                                                             void* out, int out_dtype, int in_dtype,
                                                             int64_t nob, int32_t channels,
                                                             int64_t ppr, ScaleEpilogue ep,
                                                             uint32_t* __restrict__ mid_a,
                                                             uint32_t* __restrict__ mid_b) {
  __shared__ uint32_t sha[kBlock];
  __shared__ float shx[kBlock], shn[kBlock];
  __shared__ uint32_t shnan[kBlock];
  const int32_t c = blockIdx.x;
  const int64_t n = nob * ppr;  // units of this channel: (outer_block * channels + c) * ppr + piece
  const int64_t slice = (n + gridDim.y - 1) / gridDim.y;
  const int64_t k0 = (int64_t)blockIdx.y * slice;
  const int64_t k1 = k0 + slice < n ? k0 + slice : n;
  uint32_t m = 0;
  float mx = -__builtin_inff(), mn = __builtin_inff();
  uint32_t nan = 0;
  for (int64_t k = k0 + threadIdx.x; k < k1; k += kBlock) {
    int64_t unit;
    if (nob == 1) {
      unit = (int64_t)c * ppr + k;
    } else {
      const int64_t o = k / ppr, p = k - o * ppr;
      unit = (o * channels + c) * ppr + p;
    }
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
  if (threadIdx.x == 0 && mid_a) {  // first stage of a split reduction
    const int64_t slot = (int64_t)c * gridDim.y + blockIdx.y;
    if (KIND == BVQ_STAT_ABSMAX) {
      mid_a[slot] = sha[0];
    } else {
      mid_a[slot] = shnan[0] ? 0x7fc00000u : __builtin_bit_cast(uint32_t, shx[0]);
      mid_b[slot] = __builtin_bit_cast(uint32_t, shn[0]);
    }
  } else if (threadIdx.x == 0) {
    if (KIND == BVQ_STAT_ABSMAX) {
      float v;
      if (in_dtype == BVQ_F16) {
        v = (float)__builtin_bit_cast(f16_t, (uint16_t)sha[0]);
      } else {
        v = __builtin_bit_cast(float, sha[0]);
      }
      store_stat(out, out_dtype, c, v);
      if (ep.scale_out) {
        const float thr = (ep.use_min && v < ep.min_val) ? ep.min_val : v;  // NaN passes, like torch.clamp_min
        store_stat(ep.scale_out, ep.scale_dtype, c, thr / ep.int_threshold);
      }
      if (ep.running) {
        const float r = load_scalar_as_f(ep.running, ep.run_dtype, c);
        // (v as the statistic tensor holds it: out_dtype is x's dtype on this route)
        store_stat(ep.running, ep.run_dtype, c,
                   running_update(r, v, ep.run_dtype, out_dtype, ep.one_minus_m, ep.momentum, ep.first_batch));
      }
    } else {
      const float qn = __builtin_nanf("");
      store_stat(out, out_dtype, c, shnan[0] ? qn : shx[0]);
      store_stat(out, out_dtype, (int64_t)channels + c, shnan[0] ? qn : shn[0]);
    }
  }
}

// ---- abs-max in ONE launch (per-channel layouts) ----------------------------------------------------------------
// Persistent waves walk the units with a grid stride; a wave keeps one running maximum for as long as its units
// belong to one channel (the grid is sized so that they always do when the layout allows), then ARRIVES: an atomic
// max into the channel's key word, and -- once that has returned -- an atomic add of the units it covered into the
// channel's counter.  The wave whose add completes the channel reads the key back and finishes the channel
// (statistic, scale epilogue, running statistic: what stat_finish_kernel does in a second launch).  Nobody waits:
// only the last arriver does the extra work.  A max is exact and order-independent, so the result is the same bits
// whoever arrives last.  Both words are handed back as zeros (exchange / store by the finishing lane), so the
// caller's arrival buffer needs no clearing between launches on one stream.
// (MI355X_MICROARCH.md, inter-workgroup visibility: agent-scope atomics both sides; no plain load of another
//  workgroup's stores anywhere.)
struct ArriveArgs {
  uint32_t* key;         // [channels] zero on entry and on exit
  uint32_t* cnt;         // [channels] zero on entry and on exit
  uint32_t per_channel;  // units of one channel
  void* stat_out;
  int32_t stat_dtype, in_dtype;
  uint32_t* part;        // non-null: no arrival, the unit's maximum goes to part[unit] (a finishing launch follows)
};

// one arrival: fold m into *key, then count n in; true (with the group's maximum in `out`, both words handed back as
// zeros) for the arrival that completes `expected`
__device__ __forceinline__ bool arrive_max(uint32_t* key, uint32_t* cnt, uint32_t m, uint32_t n, uint32_t expected,
                                           uint32_t& out) {
  const uint32_t seen = __hip_atomic_fetch_max(key, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t add = n;
  asm volatile("" : "+v"(add) : "v"(seen));  // the count is added only after the max has been performed (returned)
  const uint32_t before = __hip_atomic_fetch_add(cnt, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (before + add != expected) return false;
  // last arriver: every other arrival's max was performed before its add, and all adds before this one
  out = __hip_atomic_exchange(key, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

// statistic, scale epilogue, running statistic of channel c from the key `bits` (what stat_finish_kernel does)
__device__ __forceinline__ void absmax_finish(const ArriveArgs& r, const ScaleEpilogue& ep, int32_t c, uint32_t bits) {
  const float v = r.in_dtype == BVQ_F16 ? (float)__builtin_bit_cast(f16_t, (uint16_t)bits)
                                         : __builtin_bit_cast(float, bits);
  store_stat(r.stat_out, r.stat_dtype, c, v);
  if (ep.scale_out) {
    const float thr = (ep.use_min && v < ep.min_val) ? ep.min_val : v;  // NaN passes, like torch.clamp_min
    store_stat(ep.scale_out, ep.scale_dtype, c, thr / ep.int_threshold);
  }
  if (ep.running) {
    const float run = load_scalar_as_f(ep.running, ep.run_dtype, c);
    store_stat(ep.running, ep.run_dtype, c,
               running_update(run, v, ep.run_dtype, r.stat_dtype, ep.one_minus_m, ep.momentum, ep.first_batch));
  }
}

// per-channel layouts: every wave arrives for itself (m: the lanes' maxima of the wave's unit)
__device__ __forceinline__ void absmax_arrive(const ArriveArgs& r, const ScaleEpilogue& ep, int32_t c, uint32_t m,
                                              uint32_t n, int lane) {
  m = wave_max_u32(m);
  if (lane != 0) return;
  uint32_t bits;
  if (arrive_max(r.key + c, r.cnt + c, m, n, r.per_channel, bits)) absmax_finish(r, ep, c, bits);
}

// One LONG unit per wave (onepass_tiling: ~8192 waves per launch, each walking tens of rows of its channel), walked as a
// software pipeline: kOnepassDepth 16-byte chunks per lane are always in flight, the load of chunk i + depth is issued
// as chunk i is folded into the running maximum.  Buffer loads with an out-of-range offset for the lanes past the
// unit's end return zeros without touching memory (|0| never raises a maximum), so the loop has no branch and no
// predicated load.  (The first form of this kernel -- persistent waves over the two-launch kernel's short units, each
// a load-all / wait / fold batch behind a predicate -- streamed at 5.2 TB/s where the two-launch kernel reaches 6.6:
// profiles/r03_onepass.txt.)
// (depth 4 / 8 and 6144 / 8192 / 12288 units all stream within 1.5 us of each other on [256,512,56,56] bf16, depth 16
//  spills and crawls: profiles/r03_onepass.txt)
constexpr int kOnepassDepth = 4;

// the lanes' maxima of |x| (as abs_bits<> keys) over one unit
template <typename T, int VEC, bool NT, bool RELU>
__device__ __forceinline__ uint32_t onepass_unit_max(const StatArgs& a, const Unit& u, int lane) {
  const T* __restrict__ xp = reinterpret_cast<const T*>(a.x) + u.base;
  const int64_t extent = (int64_t)(u.nrows - 1) * u.row_stride + u.len;
  const buf_t bx = make_buf(xp, (uint32_t)(extent * (int64_t)sizeof(T)));
  constexpr uint32_t kSkip = 0x60000000u;  // element offset whose byte offset is >= 2^31 for 2- and 4-byte elements
  constexpr int kD = kOnepassDepth;
  ChunkCursor cur;
  cur.init(u, VEC, lane);
  const int64_t total = (int64_t)u.nrows * cur.cpr;
  const int32_t steps = (int32_t)((total + kWave - 1) / kWave);
  const uint32_t rs = (uint32_t)u.row_stride;
  uint32_t m = 0;
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  u16x2 m2 = {0, 0};
  auto fold = [&](const vec_t<T, VEC>& xv) {
    if constexpr (sizeof(T) == 2 && VEC % 2 == 0 && !RELU) {
      // two 16-bit keys per word: clear both sign bits, packed unsigned max
      const vec_t<uint32_t, VEC / 2> w = __builtin_bit_cast(vec_t<uint32_t, VEC / 2>, xv);
#pragma unroll
      for (int k = 0; k < VEC / 2; ++k)
        m2 = __builtin_elementwise_max(m2, __builtin_bit_cast(u16x2, w.v[k] & 0x7fff7fffu));
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const uint32_t b = pre_abs_bits<T, RELU>(xv.v[k]);
        m = b > m ? b : m;
      }
    }
  };
  vec_t<T, VEC> xb[kD];
#pragma unroll
  for (int j = 0; j < kD; ++j) {
    const uint32_t off = cur.valid() ? cur.offset32(rs, VEC) : kSkip;
    xb[j] = buf_load<T, VEC, NT>(bx, off * (uint32_t)sizeof(T));
    cur.next();
  }
  for (int32_t base = 0; base < steps; base += kD) {
#pragma unroll
    for (int j = 0; j < kD; ++j) {
      if (base + j >= steps) break;  // wave-uniform
      fold(xb[j]);
      const uint32_t off = cur.valid() ? cur.offset32(rs, VEC) : kSkip;
      xb[j] = buf_load<T, VEC, NT>(bx, off * (uint32_t)sizeof(T));
      cur.next();
    }
  }
  if constexpr (sizeof(T) == 2 && VEC % 2 == 0 && !RELU) {
    const uint32_t m16 = m2.x > m2.y ? m2.x : m2.y;
    m = elem<T>::id == BVQ_BF16 ? (m16 << 16) : m16;  // the abs_bits<> key space
  }
  // ragged ends: the (< VEC) elements after the last full chunk of every row of the unit
  const int32_t tail = (int32_t)(u.len - (int64_t)cur.cpr * VEC);
  for (int32_t e = lane; e < u.nrows * tail; e += kWave) {
    const int32_t tr = e / tail, k = e - tr * tail;
    const uint32_t b = pre_abs_bits<T, RELU>(xp[(int64_t)tr * u.row_stride + (int64_t)cur.cpr * VEC + k]);
    m = b > m ? b : m;
  }
  return m;
}

template <typename T, int VEC, bool NT, bool RELU>
__global__ __launch_bounds__(kBlock) void absmax_onepass_kernel(StatArgs a, ArriveArgs r, ScaleEpilogue ep) {
  const Unit u = locate_unit(a.t);
  if (!u.valid) return;
  const int lane = threadIdx.x & 63;
  uint32_t m = onepass_unit_max<T, VEC, NT, RELU>(a, u, lane);
  if (r.part) {  // (wave-uniform) whole-tensor statistic: ONE finishing launch over <= 4096 long units follows
    m = wave_max_u32(m);
    if (lane == 0) r.part[u.id] = m;
    return;
  }
  absmax_arrive(r, ep, u.channel, m, 1u, lane);
}

// ---- batch-sharded tensors: the bookkeeping around the two collectives, one launch each ---------------
// (brevitas_amd/distributed.py holds the same logic as torch ops for CPU tensors: the gloo protocol test)
//
// pack: this shard's message for the backward all-gather, float64 [2][channels]:
//   row 0 = the shard's dscale partial sums, row 1 = its claim on each channel's deposit:
//   per-channel / first-only layouts: `rank` if the shard holds an element attaining the statistic, else
//   NO_OWNER; whole-tensor layouts: its number of ties.

__global__ void shard_pack_kernel(const float* __restrict__ ds, const long long* __restrict__ tie_info,
                                  double* __restrict__ out, int32_t channels, int32_t rank, int per_channel) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= channels) return;
  out[c] = (double)ds[c];
  if (per_channel)
    out[channels + c] = tie_info[c] >= 0 ? (double)rank : kShardNoOwner;
  else
    out[channels + c] = (double)tie_info[0];
}

// unpack: from the gathered [world][2][channels] messages -- the total dscale (ranks added in rank order, in
// double: the same bits on every rank), and for per-channel layouts tie_info with every channel this shard does
// not own disabled (the owner is the lowest rank that claimed it); for whole-tensor layouts the total tie count.
__global__ void shard_unpack_kernel(const double* __restrict__ all, int32_t world, int32_t channels, int32_t rank,
                                    int per_channel, float* __restrict__ ds_total, long long* __restrict__ tie_info,
                                    long long* __restrict__ total_ties) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= channels) return;
  double sum = 0.0, owner = kShardNoOwner, count = 0.0;
  for (int r = 0; r < world; ++r) {
    const double* m = all + (size_t)r * 2 * channels;
    sum += m[c];
    const double k = m[channels + c];
    owner = k < owner ? k : owner;
    count += k;
  }
  ds_total[c] = (float)sum;
  if (per_channel) {
    if (owner != (double)rank) tie_info[c] = -1;
  } else if (c == 0) {
    total_ties[0] = (long long)count;
  }
}

// forward: the all-reduced float32 statistic -> statistic in x's dtype and the scale, with the rounding points of
// clamp_min_ste(stat, min_val) / int_threshold (ScaleEpilogue above) -- one launch instead of three tiny ops
__global__ void scale_from_stat_kernel(const float* __restrict__ stat32, void* stat_out, int stat_dtype, ScaleEpilogue ep,
                                       int32_t channels) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= channels) return;
  const float v = stat32[c];  // a value of stat_dtype: a max over shards of such values
  store_stat(stat_out, stat_dtype, c, v);
  const float thr = (ep.use_min && v < ep.min_val) ? ep.min_val : v;
  store_stat(ep.scale_out, ep.scale_dtype, c, thr / ep.int_threshold);
  if (ep.running) {
    const float r = load_scalar_as_f(ep.running, ep.run_dtype, c);
    store_stat(ep.running, ep.run_dtype, c,
               running_update(r, v, ep.run_dtype, stat_dtype, ep.one_minus_m, ep.momentum, ep.first_batch));
  }
}

// Running average of a statistic, as _RuntimeStats keeps it (B/core/stats/stats_wrapper.py:61-66):
//   first batch:  running *= out
//   afterwards :  running *= (1 - momentum) ; running += momentum * out
// every torch op rounds to its result dtype: running's for the in-place ops, out's for momentum * out.
__global__ void running_stats_kernel(void* running, int run_dtype, const void* stat, int stat_dtype,
                                     int64_t n, float one_minus_m, float m, int first) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float r = load_scalar_as_f(running, run_dtype, i);
  const float o = load_scalar_as_f(stat, stat_dtype, i);
  r = running_update(r, o, run_dtype, stat_dtype, one_minus_m, m, first);
  store_stat(running, run_dtype, i, r);
}

// ------------------------------------------------------------------------------------------------
// Backward of the statistics: locate the elements that attain the extremum ("ties") and deposit the
// gradient there.  MATCH_ABS: |x| == stat, deposit scaled by sgn(x) (torch.abs backward);
// MATCH_VALUE: x == stat (torch.max / torch.min of x itself).
// ------------------------------------------------------------------------------------------------
// torch.abs backward uses sgn(x): 0 at 0

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
                                                          int first_only) {
  const Unit u = locate_unit(t);
  if (!u.valid) return;
  const int lane = threadIdx.x & 63;
  const T* __restrict__ xp = reinterpret_cast<const T*>(x) + u.base;
  T* __restrict__ dp = reinterpret_cast<T*>(dx) + u.base;
  const T sv = reinterpret_cast<const T*>(stat)[u.channel];
  const bool per_channel = t.channels > 1 || first_only;  // record the first position only
  ChunkWalker cur;
  cur.init(u, VEC, lane);
  const int64_t total = (int64_t)u.nrows * cur.cpr;
  for (int64_t done = 0; done < total; done += kWave) {
    if (cur.valid()) {
      const int64_t off = cur.offset(u.row_stride, VEC);
      const int64_t pos = u.pos0 + cur.pos(t.row_len, VEC);
      const vec_t<T, VEC> xv = load_vec<T, VEC>(xp + off);
      vec_t<T, VEC> zv;
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        zv.v[k] = zero_like<T, MATCH>(xv.v[k]);
        if (is_tie<T, MATCH>(xv.v[k], sv))
          record_tie(info, per_channel, u.channel, (unsigned long long)(pos + k));
      }
      if (WRITE_ZERO) store_vec<T, VEC>(dp + off, zv);
    }
    cur.next();
  }
  // ragged end (single-row units only): lane 0 walks the (< VEC) leftover elements
  if (u.nrows == 1 && lane == 0) {
    for (int64_t e = (int64_t)cur.cpr * VEC; e < u.len; ++e) {
      if (WRITE_ZERO) dp[e] = zero_like<T, MATCH>(xp[e]);
      if (is_tie<T, MATCH>(xp[e], sv)) record_tie(info, per_channel, u.channel, (unsigned long long)(u.pos0 + e));
    }
  }
}

// the same scan on channel-last layouts (ColsPlan units: a lane owns the VEC columns of its chunk for a block of rows):
// per column the FIRST row whose element attains the statistic of the column's channel -- per-channel layouts record the
// first position only -- folded into info[channel] with one atomicMin per hit (rare: a handful per channel); optionally
// the zero fill of dx on the same read.  (The row-mapped scan degenerates to one-element rows on these layouts.)
template <typename T, int MATCH, bool WRITE_ZERO>
__global__ __launch_bounds__(kBlock) void tie_scan_cols_kernel(ColsPlan p, const void* x, const void* stat,
                                                               unsigned long long* info, void* dx, int64_t inner) {
  constexpr int VEC = elem<T>::vec;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= p.units) return;
  const int64_t rblk = unit / p.strips;
  const int32_t strip = (int32_t)(unit - rblk * p.strips);
  const int32_t sub = lane / p.lpr;
  const int32_t chunk = strip * kWave + (lane - sub * p.lpr);
  if (!(sub < p.rpp && chunk < p.cps)) return;
  const int64_t row0 = rblk * p.rb + sub;
  const int64_t row_end = (rblk + 1) * p.rb < p.rows ? (rblk + 1) * p.rb : p.rows;
  const T* __restrict__ xp = reinterpret_cast<const T*>(x) + (int64_t)chunk * VEC;
  T* __restrict__ dp = reinterpret_cast<T*>(dx) + (int64_t)chunk * VEC;
  T sv[VEC];
  int64_t first[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    sv[k] = reinterpret_cast<const T*>(stat)[((int64_t)chunk * VEC + k) / inner];
    first[k] = -1;
  }
  for (int64_t r = row0; r < row_end; r += p.rpp) {
    const vec_t<T, VEC> xv = load_vec<T, VEC>(xp + r * p.L);
    vec_t<T, VEC> zv;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      zv.v[k] = zero_like<T, MATCH>(xv.v[k]);
      if (first[k] < 0 && is_tie<T, MATCH>(xv.v[k], sv[k])) first[k] = r;
    }
    if (WRITE_ZERO) store_vec<T, VEC>(dp + r * p.L, zv);
  }
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    if (first[k] >= 0) {
      const int64_t col = (int64_t)chunk * VEC + k;
      record_tie(info, true, (int32_t)(col / inner), (unsigned long long)(first[k] * inner + col % inner));
    }
  }
}

__global__ void tie_init_kernel(unsigned long long* info, int32_t channels, int first_only) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (channels > 1 || first_only) {
    if (i < channels) info[i] = ~0ull;
  } else {
    if (i < 2) info[i] = 0ull;
  }
}

void launch_tie_init(unsigned long long* info, int64_t channels, hipStream_t st, int first_only) {
  tie_init_kernel<<<dim3((unsigned)((channels + 255) / 256)), dim3(256), 0, st>>>(info, (int32_t)channels,
                                                                                  first_only);
}

// channels > 1: one thread per channel deposits the gradient at first[c]
template <typename T, int MATCH>
__global__ void tie_apply_first_kernel(const void* x, GstatSrc gstat, const unsigned long long* info,
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
  const float term = deposit<T, MATCH>(gstat_value<T>(gstat, c), xp[flat], gstat.pre_relu != 0);
  dp[flat] = mode_add ? from_f<T>(to_f<T>(dp[flat]) + term) : from_f<T>(term);
}

// channels == 1, ties fit the list: each tie receives (gstat / count)
template <typename T, int MATCH>
__global__ void tie_apply_list_kernel(const void* x, GstatSrc gstat, const unsigned long long* info,
                                      const unsigned long long* total, void* dx, int mode_add) {
  const unsigned long long local = info[0];
  if (local == 0 || local > (unsigned long long)kTieCap) return;
  const unsigned long long cnt = total ? total[0] : local;  // ties over all shards of the tensor
  const T* xp = reinterpret_cast<const T*>(x);
  T* dp = reinterpret_cast<T*>(dx);
  // grad / mask.sum(): the count is an integer tensor, the quotient has the gradient's dtype
  // (the integer count is converted to the gradient's dtype first, as torch's type promotion does)
  const float share = rnd<T>(gstat_value<T>(gstat, 0) / rnd<T>((float)cnt));
  for (unsigned long long k = blockIdx.x * blockDim.x + threadIdx.x; k < local;
       k += (unsigned long long)gridDim.x * blockDim.x) {
    const int64_t flat = (int64_t)info[2 + k];
    const float term = deposit<T, MATCH>(share, xp[flat], gstat.pre_relu != 0);
    dp[flat] = mode_add ? from_f<T>(to_f<T>(dp[flat]) + term) : from_f<T>(term);
  }
}

// channels == 1, more ties than the list holds (constant tensors, binarised weights): full pass
template <typename T, int MATCH>
__global__ __launch_bounds__(kBlock) void tie_apply_full_kernel(const void* x, const void* stat,
                                                                GstatSrc gstat,
                                                                const unsigned long long* info,
                                                                const unsigned long long* total,
                                                                void* dx, int64_t n, int mode_add) {
  const unsigned long long local = info[0];
  if (local <= (unsigned long long)kTieCap) return;
  const unsigned long long cnt = total ? total[0] : local;
  const T* xp = reinterpret_cast<const T*>(x);
  T* dp = reinterpret_cast<T*>(dx);
  const T sv = reinterpret_cast<const T*>(stat)[0];
  const float share = rnd<T>(gstat_value<T>(gstat, 0) / rnd<T>((float)cnt));
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const T xv = xp[i];
    const bool tie = (gstat.pre_relu && MATCH == BVQ_MATCH_ABS)
                         ? pre_abs_bits<T, true>(xv) == abs_bits<T>(sv)
                         : is_tie<T, MATCH>(xv, sv);
    if (tie) {
      const float term = deposit<T, MATCH>(share, xv, gstat.pre_relu != 0);
      dp[i] = mode_add ? from_f<T>(to_f<T>(dp[i]) + term) : from_f<T>(term);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static Tiling stat_tiling(int dtype, const void* x, const void* dx, int64_t outer, int64_t channels,
                          int64_t inner, int& vec, bool ragged_ok = false) {
  // per-tensor: one row holding everything
  const int64_t t_outer = channels > 1 ? outer : 1;
  const int64_t row_len = channels > 1 ? inner : outer * inner;
  const int full = 16 / dtype_size(dtype);
  const void* ptrs[2] = {x, dx};
  const int els[2] = {dtype_size(dtype), dtype_size(dtype)};
  vec = pick_vec(full, t_outer * channels, row_len, ptrs, els, 2, ragged_ok);
  vec = vec == full ? full : 1;
  return make_tiling(t_outer, (int32_t)channels, row_len, vec);
}

static int64_t worst_units(int dtype, int64_t outer, int64_t channels, int64_t inner) {
  const int64_t t_outer = channels > 1 ? outer : 1;
  const int64_t row_len = channels > 1 ? inner : outer * inner;
  const int64_t a = make_tiling(t_outer, (int32_t)channels, row_len, 16 / dtype_size(dtype)).units;
  const int64_t b = make_tiling(t_outer, (int32_t)channels, row_len, 1).units;
  return a > b ? a : b;
}

template <typename T, bool RELU>
static void launch_stat_pre(int kind, const StatArgs& a, int vec, bool nt, hipStream_t st) {
  constexpr int V = elem<T>::vec;
  const dim3 grid(grid_for_units(a.t.units)), block(kBlock);
  if (kind == BVQ_STAT_ABSMAX) {
    if (vec == V && nt)
      absmax_kernel<T, V, true, RELU><<<grid, block, 0, st>>>(a);
    else if (vec == V)
      absmax_kernel<T, V, false, RELU><<<grid, block, 0, st>>>(a);
    else
      absmax_kernel<T, 1, false, RELU><<<grid, block, 0, st>>>(a);
  } else {
    if (vec == V && nt)
      minmax_kernel<T, V, true, RELU><<<grid, block, 0, st>>>(a);
    else if (vec == V)
      minmax_kernel<T, V, false, RELU><<<grid, block, 0, st>>>(a);
    else
      minmax_kernel<T, 1, false, RELU><<<grid, block, 0, st>>>(a);
  }
}

template <typename T>
static void launch_stat(int kind, int pre_op, const StatArgs& a, int vec, bool nt, hipStream_t st) {
  if (pre_op == BVQ_PRE_RELU)
    launch_stat_pre<T, true>(kind, a, vec, nt, st);
  else
    launch_stat_pre<T, false>(kind, a, vec, nt, st);
}

template <typename T, int MATCH, bool WZ>
static void launch_tie_scan_v(const Tiling& t, int vec, const void* x, const void* stat,
                              unsigned long long* info, void* dx, int first_only, hipStream_t st) {
  constexpr int V = elem<T>::vec;
  const dim3 grid(grid_for_units(t.units)), block(kBlock);
  if (vec == V)
    tie_scan_kernel<T, V, MATCH, WZ><<<grid, block, 0, st>>>(t, x, stat, info, dx, first_only);
  else
    tie_scan_kernel<T, 1, MATCH, WZ><<<grid, block, 0, st>>>(t, x, stat, info, dx, first_only);
}

template <typename T, int MATCH>
static void run_tie_scan(const Tiling& t, int vec, const void* x, const void* stat,
                         unsigned long long* info, void* dx, int write_zeros, int first_only,
                         hipStream_t st) {
  if (write_zeros)
    launch_tie_scan_v<T, MATCH, true>(t, vec, x, stat, info, dx, first_only, st);
  else
    launch_tie_scan_v<T, MATCH, false>(t, vec, x, stat, info, dx, first_only, st);
}

template <typename T, int MATCH>
static void run_tie_apply(const void* x, const void* stat, GstatSrc gstat,
                          const unsigned long long* info, const unsigned long long* total, void* dx,
                          int64_t outer, int64_t channels, int64_t inner, int mode_add, int first_only,
                          hipStream_t st) {
  if (channels > 1 || first_only) {
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
  const int64_t mid = channels * (int64_t)finish_splits(units / channels + 1);
  int64_t partials = 2 * (units + mid) * (int64_t)sizeof(uint32_t);
  const ColsPlan cp = cols_plan(dtype, outer, channels, inner);
  // (min/max keeps two key rows per partial row)
  if (cp.ok && (cp.prows + cols_fold_scratch_rows(cp.prows)) * 2 * cp.L * (int64_t)sizeof(uint32_t) > partials)
    partials = (cp.prows + cols_fold_scratch_rows(cp.prows)) * 2 * cp.L * (int64_t)sizeof(uint32_t);
  const int64_t tie = (channels > 1 ? channels : 2 + kTieCap) * (int64_t)sizeof(int64_t);
  return partials + tie + 256;
}

static int stats_impl(int kind, int pre_op, int dtype, const void* x, int64_t outer, int64_t channels,
                      int64_t inner, int out_dtype, void* out, const ScaleEpilogue& ep, void* workspace,
                      int64_t workspace_bytes, bvq_stream_t stream) {
  if (pre_op != BVQ_PRE_NONE && pre_op != BVQ_PRE_RELU) {
    set_error("bvq_stats: bad pre_op %d", pre_op);
    return BVQ_ERR_INVALID;
  }
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
  hipStream_t st = (hipStream_t)stream;
  bool nt = n * (int64_t)dtype_size(dtype) >= nt_threshold_bytes();
  // channel axis last (or nearly): column-mapped units, same finishing kernel
  const ColsPlan cp =
      (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? cols_plan(dtype, outer, channels, inner) : ColsPlan{};
  if (cp.ok) {
    const int64_t width = (kind == BVQ_STAT_MINMAX ? 2 : 1) * cp.L;  // entries per partial row
    if (workspace_bytes < (cp.prows + cols_fold_scratch_rows(cp.prows)) * width * (int64_t)sizeof(uint32_t)) {
      set_error("bvq_stats: workspace too small");
      return BVQ_ERR_WORKSPACE;
    }
    ColsStatArgs ca;
    ca.p = cp;
    ca.x = x;
    ca.part = reinterpret_cast<uint32_t*>(workspace);
    const dim3 grid(grid_for_units(cp.units)), block(kBlock);
    const bool relu = pre_op == BVQ_PRE_RELU;
#define BVQ_COLS_STAT(KERNEL, T)                              \
  do {                                                        \
    if (relu)                                                 \
      KERNEL<T, false, true><<<grid, block, 0, st>>>(ca);     \
    else if (nt)                                              \
      KERNEL<T, true, false><<<grid, block, 0, st>>>(ca);     \
    else                                                      \
      KERNEL<T, false, false><<<grid, block, 0, st>>>(ca);    \
  } while (0)
#define BVQ_COLS_STAT_DT(KERNEL)          \
  do {                                    \
    if (dtype == BVQ_F32)                 \
      BVQ_COLS_STAT(KERNEL, float);       \
    else if (dtype == BVQ_BF16)           \
      BVQ_COLS_STAT(KERNEL, bf16_t);      \
    else                                  \
      BVQ_COLS_STAT(KERNEL, f16_t);       \
  } while (0)
    if (kind == BVQ_STAT_MINMAX)
      BVQ_COLS_STAT_DT(minmax_cols_kernel);
    else
      BVQ_COLS_STAT_DT(absmax_cols_kernel);
#undef BVQ_COLS_STAT_DT
#undef BVQ_COLS_STAT
    int rc0 = check_launch("bvq_stats/cols");
    if (rc0) return rc0;
    uint32_t* folded = launch_cols_fold_max(ca.part, cp.prows, width, ca.part + cp.prows * width, st);  // [width]
    if (kind == BVQ_STAT_MINMAX) {
      minmax_cols_finish_kernel<<<dim3((unsigned)channels), dim3(kWave), 0, st>>>(folded, out, out_dtype,
                                                                                  (int32_t)channels, inner);
      return check_launch("bvq_stats/finish");
    }
    stat_finish_kernel<BVQ_STAT_ABSMAX><<<dim3((unsigned)channels), dim3(kBlock), 0, st>>>(
        folded, folded, out, out_dtype, dtype, 1, (int32_t)channels, inner, ep, nullptr, nullptr);
    return check_launch("bvq_stats/finish");
  }
  int vec;
  StatArgs a;
  a.t = stat_tiling(dtype, x, nullptr, outer, channels, inner, vec, true);
  // A whole-tensor abs-max as long units: at most kFinishSlice pieces, each walked by one wave of the software-pipelined
  // kernel of the one-launch route -- the partials then fit ONE finishing launch (the short-unit tiling leaves ~10^4-10^5
  // partials and needs two): [8192,8192] bf16 34 -> 26 us, profiles/r03_onepass.txt section 5.
  const bool long_units = kind == BVQ_STAT_ABSMAX && channels == 1 && vec == 16 / dtype_size(dtype) &&
                          a.t.units > kFinishSlice;
  if (long_units) {
    const int64_t quantum = (int64_t)kWave * vec;
    int64_t piece = (a.t.row_len + kFinishSlice - 1) / kFinishSlice;
    piece = ((piece + quantum - 1) / quantum) * quantum;
    a.t.piece_len = piece;
    a.t.ppr = (a.t.row_len + piece - 1) / piece;
    a.t.units = a.t.nob * channels * a.t.ppr;
  }
  const int32_t splits = finish_splits(a.t.nob * a.t.ppr);
  const int64_t mid_words = splits > 1 ? channels * (int64_t)splits : 0;
  const int64_t need = 2 * (a.t.units + mid_words) * (int64_t)sizeof(uint32_t);
  if (workspace_bytes < need) {
    set_error("bvq_stats: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
    return BVQ_ERR_WORKSPACE;
  }
  a.x = x;
  a.part_a = reinterpret_cast<uint32_t*>(workspace);
  a.part_b = a.part_a + a.t.units;
  if (long_units && cap_unit_extent(a.t, dtype_size(dtype))) {
    ArriveArgs r = {};
    r.part = a.part_a;
    const ScaleEpilogue no_ep = {};
    const dim3 grid(grid_for_units(a.t.units)), block(kBlock);
    const bool relu = pre_op == BVQ_PRE_RELU;
#define BVQ_LONG(T)                                                                           \
  do {                                                                                        \
    constexpr int V = elem<T>::vec;                                                           \
    if (relu)                                                                                 \
      absmax_onepass_kernel<T, V, false, true><<<grid, block, 0, st>>>(a, r, no_ep);          \
    else if (nt)                                                                              \
      absmax_onepass_kernel<T, V, true, false><<<grid, block, 0, st>>>(a, r, no_ep);          \
    else                                                                                      \
      absmax_onepass_kernel<T, V, false, false><<<grid, block, 0, st>>>(a, r, no_ep);         \
  } while (0)
    if (dtype == BVQ_F32)
      BVQ_LONG(float);
    else if (dtype == BVQ_BF16)
      BVQ_LONG(bf16_t);
    else
      BVQ_LONG(f16_t);
#undef BVQ_LONG
  } else if (dtype == BVQ_F32)
    launch_stat<float>(kind, pre_op, a, vec, nt, st);
  else if (dtype == BVQ_BF16)
    launch_stat<bf16_t>(kind, pre_op, a, vec, nt, st);
  else
    launch_stat<f16_t>(kind, pre_op, a, vec, nt, st);
  int rc = check_launch("bvq_stats");
  if (rc) return rc;
  uint32_t* mid_a = splits > 1 ? a.part_b + a.t.units : nullptr;
  uint32_t* mid_b = splits > 1 ? mid_a + mid_words : nullptr;
  const ScaleEpilogue none = {};
#define BVQ_FINISH(KIND)                                                                             \
  do {                                                                                               \
    if (splits > 1) {                                                                                \
      stat_finish_kernel<KIND><<<dim3((unsigned)channels, (unsigned)splits), dim3(kBlock), 0, st>>>( \
          a.part_a, a.part_b, out, out_dtype, dtype, a.t.nob, (int32_t)channels, a.t.ppr, none, mid_a, \
          mid_b);                                                                                    \
      stat_finish_kernel<KIND><<<dim3((unsigned)channels), dim3(kBlock), 0, st>>>(                   \
          mid_a, mid_b, out, out_dtype, dtype, 1, (int32_t)channels, splits, ep, nullptr, nullptr);  \
    } else {                                                                                         \
      stat_finish_kernel<KIND><<<dim3((unsigned)channels), dim3(kBlock), 0, st>>>(                   \
          a.part_a, a.part_b, out, out_dtype, dtype, a.t.nob, (int32_t)channels, a.t.ppr, ep, nullptr, \
          nullptr);                                                                                  \
    }                                                                                                \
  } while (0)
  if (kind == BVQ_STAT_ABSMAX)
    BVQ_FINISH(BVQ_STAT_ABSMAX);
  else
    BVQ_FINISH(BVQ_STAT_MINMAX);
#undef BVQ_FINISH
  return check_launch("bvq_stats/finish");
}

extern "C" int bvq_stats(int kind, int dtype, const void* x, int64_t outer, int64_t channels,
                         int64_t inner, int out_dtype, void* out, void* workspace,
                         int64_t workspace_bytes, bvq_stream_t stream) {
  ScaleEpilogue ep = {};
  return stats_impl(kind, BVQ_PRE_NONE, dtype, x, outer, channels, inner, out_dtype, out, ep, workspace,
                    workspace_bytes, stream);
}

extern "C" int bvq_stats_pre(int kind, int pre_op, int dtype, const void* x, int64_t outer, int64_t channels,
                             int64_t inner, int out_dtype, void* out, void* workspace,
                             int64_t workspace_bytes, bvq_stream_t stream) {
  ScaleEpilogue ep = {};
  return stats_impl(kind, pre_op, dtype, x, outer, channels, inner, out_dtype, out, ep, workspace,
                    workspace_bytes, stream);
}

extern "C" int bvq_absmax_scale(int pre_op, int dtype, const void* x, int64_t outer, int64_t channels,
                                int64_t inner,
                                void* stat_out, double min_val, int use_min, double int_threshold,
                                int scale_dtype, void* scale_out, void* workspace,
                                int64_t workspace_bytes, bvq_stream_t stream) {
  if (bad_dtype(scale_dtype) || !scale_out || !(int_threshold == int_threshold)) {
    set_error("bvq_absmax_scale: bad argument");
    return BVQ_ERR_INVALID;
  }
  ScaleEpilogue ep = {};
  ep.scale_out = scale_out;
  ep.scale_dtype = scale_dtype;
  ep.use_min = use_min;
  ep.min_val = round_host((float)min_val, dtype);  // python scalar -> the statistic's dtype
  ep.int_threshold = (float)int_threshold;
  return stats_impl(BVQ_STAT_ABSMAX, pre_op, dtype, x, outer, channels, inner, dtype, stat_out, ep, workspace,
                    workspace_bytes, stream);
}

// Units of the one-launch abs-max: long ones.  Short rows (NCHW activations): as many rows of one channel per wave as
// leave ~kOnepassUnits waves in the launch (a wave's arrival -- two atomic round trips -- then costs a percent of its
// life, not 7 %), taking the row count near that which wastes the fewest lanes of the 64-wide loads.  Long rows keep
// their pieces.  A unit's extent stays below 2^31 bytes (32-bit buffer offsets).
constexpr int64_t kOnepassUnits = 8192;

static bool onepass_tiling(int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner, Tiling& t,
                           int& vec) {
  t = stat_tiling(dtype, x, nullptr, outer, channels, inner, vec, true);
  if (t.ppr == 1 && t.outer > 1) {
    const int64_t cpr = t.row_len / vec;
    static const int64_t target = env_flag("BVQ_ONEPASS_UNITS", (int)kOnepassUnits);  // developer knob (tools/onepass_ab.py)
    int64_t r0 = t.outer * channels / target;
    r0 = r0 < 1 ? 1 : (r0 > t.outer ? t.outer : r0);
    int64_t best = r0;
    double best_eff = -1.0;
    for (int64_t rr = r0; rr >= 1 && 2 * rr > r0; --rr) {
      const int64_t loads = (rr * cpr + kWave - 1) / kWave;
      const double eff = loads > 0 ? (double)(rr * cpr) / (double)(loads * kWave) : 1.0;
      if (eff > best_eff + 1e-9) {
        best_eff = eff;
        best = rr;
      }
      if (eff >= 0.97) break;
    }
    t.rpu = (int32_t)best;
    t.nob = (t.outer + t.rpu - 1) / t.rpu;
    t.units = t.nob * channels * t.ppr;
  }
  return cap_unit_extent(t, dtype_size(dtype));
}

static bool onepass_layout(int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner) {
  // (a whole-tensor statistic keeps the two-launch route: thousands of arrivals at one word -- ~80 ns each when they
  //  come together -- cost more than its two finishing launches, profiles/r03_onepass.txt section 4)
  if (channels < 2 || outer < 1 || inner < 1) return false;
  if ((reinterpret_cast<uintptr_t>(x) & 15) == 0 && cols_plan(dtype, outer, channels, inner).ok) return false;
  return true;
}
// arrivals at one channel's words: agent-scope atomics on ONE address serialise at ~0.25 us each across the XCDs --
// [1024,16,32,32] with 512 arrivals per channel ran 126 us against the two-launch route's 18, [256,64,56,56] (128) 57
// against 25, [256,128,56,56] (64) 41 against 39, [64,256,56,56] (32) level, 16 and fewer ahead
// (profiles/r03_onepass.txt, section 6)
constexpr int64_t kMaxArrivalsPerChannel = 32;
static inline int64_t onepass_arrive_words(int64_t channels) { return 2 * channels; }

extern "C" int bvq_absmax_onepass_supported(int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner) {
  if (bad_dtype(dtype)) return 0;
  if (!onepass_layout(dtype, x, outer, channels, inner)) return 0;
  int vec;
  Tiling t;
  if (!onepass_tiling(dtype, x, outer, channels, inner, t, vec)) return 0;
  if (channels > 1 && t.nob * t.ppr > kMaxArrivalsPerChannel) return 0;  // few channels, many units each: two launches
  return t.nob * t.ppr < ((int64_t)1 << 31) ? 1 : 0;
}

extern "C" int bvq_absmax_scale_onepass(int pre_op, int dtype, const void* x, int64_t outer, int64_t channels,
                                        int64_t inner, int stat_dtype, void* stat_out, double min_val, int use_min,
                                        double int_threshold, int scale_dtype, void* scale_out, int run_dtype,
                                        void* running, double momentum, int first_batch, uint32_t* arrive,
                                        int64_t arrive_words, bvq_stream_t stream) {
  if (pre_op != BVQ_PRE_NONE && pre_op != BVQ_PRE_RELU) {
    set_error("bvq_absmax_scale_onepass: bad pre_op %d", pre_op);
    return BVQ_ERR_INVALID;
  }
  if (bad_dtype(dtype) || bad_dtype(stat_dtype) || (scale_out && bad_dtype(scale_dtype)) ||
      (running && bad_dtype(run_dtype)) || (scale_out && !(int_threshold == int_threshold))) {
    set_error("bvq_absmax_scale_onepass: bad argument");
    return BVQ_ERR_INVALID;
  }
  if (stat_dtype != BVQ_F32 && stat_dtype != dtype) {
    set_error("bvq_absmax_scale_onepass: stat_dtype must be f32 or the dtype of x");
    return BVQ_ERR_UNSUPPORTED;
  }
  if (!x || !stat_out || !arrive) {
    set_error("bvq_absmax_scale_onepass: null pointer");
    return BVQ_ERR_INVALID;
  }
  if (!bvq_absmax_onepass_supported(dtype, x, outer, channels, inner)) {
    set_error("bvq_absmax_scale_onepass: layout not covered (per-tensor or column-mapped): use bvq_absmax_scale");
    return BVQ_ERR_UNSUPPORTED;
  }
  if (arrive_words < onepass_arrive_words(channels)) {
    set_error("bvq_absmax_scale_onepass: arrival buffer of %lld words, %lld needed", (long long)arrive_words,
              (long long)onepass_arrive_words(channels));
    return BVQ_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = outer * channels * inner;
  const bool nt = n * (int64_t)dtype_size(dtype) >= nt_threshold_bytes();
  int vec;
  StatArgs a = {};
  onepass_tiling(dtype, x, outer, channels, inner, a.t, vec);
  a.x = x;
  ArriveArgs r;
  r.key = arrive;
  r.cnt = arrive + channels;
  r.per_channel = (uint32_t)(a.t.nob * a.t.ppr);
  r.stat_out = stat_out;
  r.stat_dtype = stat_dtype;
  r.in_dtype = dtype;
  r.part = nullptr;
  const unsigned blocks = grid_for_units(a.t.units);
  ScaleEpilogue ep = {};
  if (scale_out) {
    ep.scale_out = scale_out;
    ep.scale_dtype = scale_dtype;
    ep.use_min = use_min;
    ep.min_val = round_host((float)min_val, dtype);  // python scalar -> the statistic's dtype
    ep.int_threshold = (float)int_threshold;
  }
  if (running) {
    ep.running = running;
    ep.run_dtype = run_dtype;
    ep.first_batch = first_batch;
    // torch turns the python scalars (1 - momentum) and momentum into float32 for these dtypes (bvq_running_stats_update)
    ep.one_minus_m = (float)(1.0 - momentum);
    ep.momentum = (float)momentum;
  }
  const dim3 grid(blocks), block(kBlock);
  const bool relu = pre_op == BVQ_PRE_RELU;
#define BVQ_ONEPASS(T)                                                              \
  do {                                                                              \
    constexpr int V = elem<T>::vec;                                                 \
    if (relu && vec == V)                                                           \
      absmax_onepass_kernel<T, V, false, true><<<grid, block, 0, st>>>(a, r, ep);   \
    else if (relu)                                                                  \
      absmax_onepass_kernel<T, 1, false, true><<<grid, block, 0, st>>>(a, r, ep);   \
    else if (vec == V && nt)                                                        \
      absmax_onepass_kernel<T, V, true, false><<<grid, block, 0, st>>>(a, r, ep);   \
    else if (vec == V)                                                              \
      absmax_onepass_kernel<T, V, false, false><<<grid, block, 0, st>>>(a, r, ep);  \
    else                                                                            \
      absmax_onepass_kernel<T, 1, false, false><<<grid, block, 0, st>>>(a, r, ep);  \
  } while (0)
  if (dtype == BVQ_F32)
    BVQ_ONEPASS(float);
  else if (dtype == BVQ_BF16)
    BVQ_ONEPASS(bf16_t);
  else
    BVQ_ONEPASS(f16_t);
#undef BVQ_ONEPASS
  return check_launch("bvq_absmax_scale_onepass");
}

// ---- abs-max of a LIST of tensors sharing the channel axis --------------------------------------------------------
// _ParameterListStats with several tracked parameters (B/core/stats/stats_wrapper.py:83-114: a weight quantizer shared by
// several layers takes its statistic of the concatenation of their weights' views).  The reference materialises the
// concatenation (torch.cat: one read and one write of every parameter) and reduces that; here ONE launch walks every
// tensor where it lies: the dispatch's waves are dealt to the tensors in order (start[]), a wave finds its tensor by
// its slot, walks its unit like the one-launch abs-max does and arrives at the channel's words -- the last arriver of
// a channel, whichever tensor its unit came from, finishes statistic and scale.  A max is exact and order-independent:
// the bits equal those of the reference's reduction over the concatenation.
constexpr int kMaxListTensors = 8;
struct ListStatArgs {
  StatArgs a[kMaxListTensors];
  int64_t start[kMaxListTensors + 1];  // first dispatch slot of tensor i; start[n] = all slots
  int32_t vec[kMaxListTensors];        // elements per load of tensor i: 16 bytes' worth, or 1 (short / misaligned rows)
  int32_t n;
};

template <typename T>
__global__ __launch_bounds__(kBlock) void absmax_list_kernel(ListStatArgs la, ArriveArgs r, ScaleEpilogue ep) {
  constexpr int V = elem<T>::vec;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int64_t slot = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (slot >= la.start[la.n]) return;
  int p = 0;
  while (p + 1 < la.n && slot >= la.start[p + 1]) ++p;  // (wave-uniform)
  const Unit u = locate_unit_slot(la.a[p].t, slot - la.start[p]);  // valid: start[] holds the tilings' own unit counts
  const uint32_t m = la.vec[p] == V ? onepass_unit_max<T, V, false, false>(la.a[p], u, lane)
                                    : onepass_unit_max<T, 1, false, false>(la.a[p], u, lane);  // (wave-uniform)
  if (r.part) {  // (wave-uniform) whole-tensor statistic: one partial per unit, a finishing launch follows
    const uint32_t mw = wave_max_u32(m);
    if (lane == 0) r.part[slot] = mw;
    return;
  }
  absmax_arrive(r, ep, u.channel, m, 1u, lane);
}

// tilings of the list's tensors, units per channel over the whole list
static bool list_tilings(int dtype, int n, const void* const* xs, const int64_t* outers, int64_t channels,
                         const int64_t* inners, ListStatArgs& la, int64_t& per_channel) {
  if (n < 1 || n > kMaxListTensors || channels < 1) return false;
  per_channel = 0;
  la.n = n;
  la.start[0] = 0;
  for (int i = 0; i < n; ++i) {
    if (outers[i] < 1 || inners[i] < 1 || !xs[i]) return false;
    int v;
    Tiling& t = la.a[i].t;
    if (!onepass_tiling(dtype, xs[i], outers[i], channels, inners[i], t, v)) return false;
    if (channels == 1) {
      // a whole-tensor statistic: few, long pieces -- the units leave one partial each and ONE finishing launch
      // (a workgroup over <= kFinishSlice partials) follows; arrivals at a single pair of words would serialise
      const int64_t cap = kFinishSlice / n;
      if (t.nob * t.ppr > cap && t.nob == 1) {
        const int64_t quantum = (int64_t)kWave * v;
        int64_t piece = (t.row_len + cap - 1) / cap;
        piece = ((piece + quantum - 1) / quantum) * quantum;
        t.piece_len = piece;
        t.ppr = (t.row_len + piece - 1) / piece;
        t.units = t.nob * t.ppr;
        if (!cap_unit_extent(t, dtype_size(dtype))) return false;
      }
    }
    la.vec[i] = v;
    la.a[i].x = xs[i];
    la.start[i + 1] = la.start[i] + t.units;
    per_channel += t.nob * t.ppr;
  }
  if (channels == 1) return la.start[n] <= kFinishSlice;
  return per_channel <= kMaxArrivalsPerChannel && la.start[n] < ((int64_t)1 << 31);
}

extern "C" int bvq_absmax_list_supported(int dtype, int n, const void* const* xs, const int64_t* outers,
                                         int64_t channels, const int64_t* inners) {
  if (bad_dtype(dtype) || !xs || !outers || !inners) return 0;
  ListStatArgs la = {};
  int64_t per_channel;
  return list_tilings(dtype, n, xs, outers, channels, inners, la, per_channel) ? 1 : 0;
}

extern "C" int bvq_absmax_scale_list(int dtype, int n, const void* const* xs, const int64_t* outers, int64_t channels,
                                     const int64_t* inners, void* stat_out, double min_val, int use_min,
                                     double int_threshold, int scale_dtype, void* scale_out, uint32_t* arrive,
                                     int64_t arrive_words, void* workspace, int64_t workspace_bytes,
                                     bvq_stream_t stream) {
  if (bad_dtype(dtype) || (scale_out && bad_dtype(scale_dtype)) || (scale_out && !(int_threshold == int_threshold))) {
    set_error("bvq_absmax_scale_list: bad argument");
    return BVQ_ERR_INVALID;
  }
  if (!xs || !outers || !inners || !stat_out || (channels > 1 && !arrive)) {
    set_error("bvq_absmax_scale_list: null pointer");
    return BVQ_ERR_INVALID;
  }
  ListStatArgs la = {};
  int64_t per_channel;
  if (!list_tilings(dtype, n, xs, outers, channels, inners, la, per_channel)) {
    set_error("bvq_absmax_scale_list: list not covered (1..%d tensors, <= %lld units per channel, one row per tensor)",
              kMaxListTensors, (long long)kMaxArrivalsPerChannel);
    return BVQ_ERR_UNSUPPORTED;
  }
  if (channels > 1 && arrive_words < onepass_arrive_words(channels)) {
    set_error("bvq_absmax_scale_list: arrival buffer of %lld words, %lld needed", (long long)arrive_words,
              (long long)onepass_arrive_words(channels));
    return BVQ_ERR_WORKSPACE;
  }
  if (channels == 1 && (!workspace || workspace_bytes < la.start[n] * (int64_t)sizeof(uint32_t))) {
    set_error("bvq_absmax_scale_list: workspace %lld < %lld bytes", (long long)workspace_bytes,
              (long long)(la.start[n] * (int64_t)sizeof(uint32_t)));
    return BVQ_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  ArriveArgs r;
  r.key = arrive;
  r.cnt = arrive + channels;
  r.per_channel = (uint32_t)per_channel;
  r.stat_out = stat_out;
  r.stat_dtype = dtype;
  r.in_dtype = dtype;
  r.part = channels == 1 ? reinterpret_cast<uint32_t*>(workspace) : nullptr;
  ScaleEpilogue ep = {};
  if (scale_out) {
    ep.scale_out = scale_out;
    ep.scale_dtype = scale_dtype;
    ep.use_min = use_min;
    ep.min_val = round_host((float)min_val, dtype);  // python scalar -> the statistic's dtype
    ep.int_threshold = (float)int_threshold;
  }
  const dim3 grid(grid_for_units(la.start[n])), block(kBlock);
  if (dtype == BVQ_F32)
    absmax_list_kernel<float><<<grid, block, 0, st>>>(la, r, ep);
  else if (dtype == BVQ_BF16)
    absmax_list_kernel<bf16_t><<<grid, block, 0, st>>>(la, r, ep);
  else
    absmax_list_kernel<f16_t><<<grid, block, 0, st>>>(la, r, ep);
  if (channels == 1) {
    int rc = check_launch("bvq_absmax_scale_list");
    if (rc) return rc;
    stat_finish_kernel<BVQ_STAT_ABSMAX><<<dim3(1), dim3(kBlock), 0, st>>>(r.part, r.part, stat_out, dtype, dtype, 1, 1,
                                                                          la.start[n], ep, nullptr, nullptr);
  }
  return check_launch("bvq_absmax_scale_list");
}

extern "C" int bvq_absmax_scale_running(int pre_op, int dtype, const void* x, int64_t outer, int64_t channels,
                                        int64_t inner, void* stat_out, double min_val, int use_min,
                                        double int_threshold, int scale_dtype, void* scale_out, int run_dtype,
                                        void* running, double momentum, int first_batch, void* workspace,
                                        int64_t workspace_bytes, bvq_stream_t stream) {
  if (bad_dtype(scale_dtype) || bad_dtype(run_dtype) || !scale_out || !running || !(int_threshold == int_threshold)) {
    set_error("bvq_absmax_scale_running: bad argument");
    return BVQ_ERR_INVALID;
  }
  ScaleEpilogue ep = {};
  ep.scale_out = scale_out;
  ep.scale_dtype = scale_dtype;
  ep.use_min = use_min;
  ep.min_val = round_host((float)min_val, dtype);
  ep.int_threshold = (float)int_threshold;
  ep.running = running;
  ep.run_dtype = run_dtype;
  ep.first_batch = first_batch;
  // torch turns the python scalars (1 - momentum) and momentum into float32 for these dtypes (bvq_running_stats_update)
  ep.one_minus_m = (float)(1.0 - momentum);
  ep.momentum = (float)momentum;
  return stats_impl(BVQ_STAT_ABSMAX, pre_op, dtype, x, outer, channels, inner, dtype, stat_out, ep, workspace,
                    workspace_bytes, stream);
}

extern "C" int64_t bvq_abs_moments_workspace_bytes(int dtype, int64_t outer, int64_t channels, int64_t inner) {
  if (bad_dtype(dtype) || outer < 0 || channels < 1 || inner < 0) return -1;
  const int64_t units = worst_units(dtype, outer, channels, inner);
  int64_t need = 2 * units * (int64_t)sizeof(float) + 8 + channel_sums_mid_bytes(units / channels + 1, channels) + 256;
  const ColsPlan cp = cols_plan(dtype, outer, channels, inner);
  if (cp.ok) {  // column-mapped: two [partial rows + fold scratch][L] float arrays
    const int64_t cols = 2 * (cp.prows + cols_fold_scratch_rows(cp.prows)) * cp.L * (int64_t)sizeof(float) + 8 +
                         channel_sums_mid_bytes(inner, channels) + 256;
    if (cols > need) need = cols;
  }
  return need;
}

extern "C" int bvq_abs_moments(int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner,
                               float* sums, void* workspace, int64_t workspace_bytes, bvq_stream_t stream) {
  if (bad_dtype(dtype) || outer < 0 || channels < 1 || inner < 0) {
    set_error("bvq_abs_moments: bad argument");
    return BVQ_ERR_INVALID;
  }
  if (!sums) {
    set_error("bvq_abs_moments: null pointer");
    return BVQ_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = outer * channels * inner;
  if (n == 0) {  // an empty sum
    (void)hipMemsetAsync(sums, 0, 3 * sizeof(float) * channels, st);
    return BVQ_OK;
  }
  if (!x || !workspace) {
    set_error("bvq_abs_moments: null pointer");
    return BVQ_ERR_INVALID;
  }
  const bool nt = n * (int64_t)dtype_size(dtype) >= nt_threshold_bytes();
  // channel axis last (or nearly): column-mapped units
  const ColsPlan cp =
      (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? cols_plan(dtype, outer, channels, inner) : ColsPlan{};
  if (cp.ok) {
    const int64_t rows_all = cp.prows + cols_fold_scratch_rows(cp.prows);
    const int64_t mid_off_c = ((2 * rows_all * cp.L * (int64_t)sizeof(float) + 7) / 8) * 8;
    if (workspace_bytes < mid_off_c + channel_sums_mid_bytes(inner, channels)) {
      set_error("bvq_abs_moments: workspace too small");
      return BVQ_ERR_WORKSPACE;
    }
    ColsMomentArgs ca;
    ca.p = cp;
    ca.x = x;
    ca.part1 = reinterpret_cast<float*>(workspace);
    ca.part2 = ca.part1 + rows_all * cp.L;
    ca.pivot = sums + 2 * channels;
    ca.inner = inner;
    const dim3 cgrid(grid_for_units(cp.units)), cblock(kBlock);
#define BVQ_MOMC(T)                                                   \
  do {                                                                \
    if (nt)                                                           \
      absmoments_cols_kernel<T, true><<<cgrid, cblock, 0, st>>>(ca);  \
    else                                                              \
      absmoments_cols_kernel<T, false><<<cgrid, cblock, 0, st>>>(ca); \
  } while (0)
    if (dtype == BVQ_F32)
      BVQ_MOMC(float);
    else if (dtype == BVQ_BF16)
      BVQ_MOMC(bf16_t);
    else
      BVQ_MOMC(f16_t);
#undef BVQ_MOMC
    int rc0 = check_launch("bvq_abs_moments/cols");
    if (rc0) return rc0;
    float *f1 = nullptr, *f2 = nullptr;
    launch_cols_fold_sum_min(ca.part1, nullptr, cp.prows, cp.L, ca.part1 + cp.prows * cp.L, nullptr, &f1, nullptr, st);
    launch_cols_fold_sum_min(ca.part2, nullptr, cp.prows, cp.L, ca.part2 + cp.prows * cp.L, nullptr, &f2, nullptr, st);
    // the folded rows are the (nob = 1, channels, ppr = inner) layout of the finishing sums
    launch_channel_sums(f1, f2, sums, sums + channels, 1, (int32_t)channels, inner,
                        reinterpret_cast<char*>(workspace) + mid_off_c, st);
    return check_launch("bvq_abs_moments/cols_sums");
  }
  int vec;
  StatArgs a;
  a.t = stat_tiling(dtype, x, nullptr, outer, channels, inner, vec);
  const int64_t mid_off = ((2 * a.t.units * (int64_t)sizeof(float) + 7) / 8) * 8;
  const int64_t need = mid_off + channel_sums_mid_bytes(a.t.nob * a.t.ppr, channels);
  if (workspace_bytes < need) {
    set_error("bvq_abs_moments: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
    return BVQ_ERR_WORKSPACE;
  }
  a.x = x;
  a.part_a = reinterpret_cast<uint32_t*>(workspace);
  a.part_b = a.part_a + a.t.units;
  a.pivot = sums + 2 * channels;
  const dim3 grid(grid_for_units(a.t.units)), block(kBlock);
#define BVQ_MOM(T)                                                     \
  do {                                                                 \
    constexpr int V = elem<T>::vec;                                    \
    if (vec == V && nt)                                                \
      absmoments_kernel<T, V, true><<<grid, block, 0, st>>>(a);        \
    else if (vec == V)                                                 \
      absmoments_kernel<T, V, false><<<grid, block, 0, st>>>(a);       \
    else                                                               \
      absmoments_kernel<T, 1, false><<<grid, block, 0, st>>>(a);       \
  } while (0)
  if (dtype == BVQ_F32)
    BVQ_MOM(float);
  else if (dtype == BVQ_BF16)
    BVQ_MOM(bf16_t);
  else
    BVQ_MOM(f16_t);
#undef BVQ_MOM
  int rc = check_launch("bvq_abs_moments");
  if (rc) return rc;
  launch_channel_sums(reinterpret_cast<const float*>(a.part_a), reinterpret_cast<const float*>(a.part_b), sums,
                      sums + channels, a.t.nob, (int32_t)channels, a.t.ppr,
                      reinterpret_cast<char*>(workspace) + mid_off, st);
  return check_launch("bvq_abs_moments/sums");
}

extern "C" int bvq_abs_affine_bwd(int dtype, const void* x, const float* a, const float* b, void* dx,
                                  int64_t outer, int64_t channels, int64_t inner, bvq_stream_t stream) {
  if (bad_dtype(dtype) || outer < 0 || channels < 1 || inner < 0) {
    set_error("bvq_abs_affine_bwd: bad argument");
    return BVQ_ERR_INVALID;
  }
  if (outer * channels * inner == 0) return BVQ_OK;
  if (!x || !a || !b || !dx) {
    set_error("bvq_abs_affine_bwd: null pointer");
    return BVQ_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0 &&
      cols_plan(dtype, outer, channels, inner).ok) {  // channel axis last (or nearly): flat chunks, channel per element
    const int64_t L = channels * inner, cps = L / (16 / dtype_size(dtype));
    const int64_t gx = (cps + kBlock - 1) / kBlock;
    int64_t gy = (16384 + gx - 1) / gx;  // ~16 k workgroups in all
    gy = gy > outer ? outer : gy;
    gy = gy > 65535 ? 65535 : (gy < 1 ? 1 : gy);
    const dim3 fgrid((unsigned)gx, (unsigned)gy), fblock(kBlock);
    if (dtype == BVQ_F32)
      abs_affine_bwd_cols_kernel<float><<<fgrid, fblock, 0, st>>>(x, a, b, dx, outer, L, inner);
    else if (dtype == BVQ_BF16)
      abs_affine_bwd_cols_kernel<bf16_t><<<fgrid, fblock, 0, st>>>(x, a, b, dx, outer, L, inner);
    else
      abs_affine_bwd_cols_kernel<f16_t><<<fgrid, fblock, 0, st>>>(x, a, b, dx, outer, L, inner);
    return check_launch("bvq_abs_affine_bwd/cols");
  }
  int vec;
  const Tiling t = stat_tiling(dtype, x, dx, outer, channels, inner, vec);
  const dim3 grid(grid_for_units(t.units)), block(kBlock);
#define BVQ_AFF(T)                                                               \
  do {                                                                           \
    constexpr int V = elem<T>::vec;                                              \
    if (vec == V)                                                                \
      abs_affine_bwd_kernel<T, V><<<grid, block, 0, st>>>(t, x, a, b, dx);       \
    else                                                                         \
      abs_affine_bwd_kernel<T, 1><<<grid, block, 0, st>>>(t, x, a, b, dx);       \
  } while (0)
  if (dtype == BVQ_F32)
    BVQ_AFF(float);
  else if (dtype == BVQ_BF16)
    BVQ_AFF(bf16_t);
  else
    BVQ_AFF(f16_t);
#undef BVQ_AFF
  return check_launch("bvq_abs_affine_bwd");
}

extern "C" int bvq_shard_pack(const float* dscale, const int64_t* tie_info, int64_t channels, int rank,
                              int per_channel, double* message, bvq_stream_t stream) {
  if (channels < 1 || rank < 0 || !dscale || !tie_info || !message) {
    set_error("bvq_shard_pack: bad argument");
    return BVQ_ERR_INVALID;
  }
  shard_pack_kernel<<<dim3((unsigned)((channels + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
      dscale, reinterpret_cast<const long long*>(tie_info), message, (int32_t)channels, rank, per_channel);
  return check_launch("bvq_shard_pack");
}

extern "C" int bvq_shard_unpack(const double* gathered, int world, int64_t channels, int rank, int per_channel,
                                float* dscale_total, int64_t* tie_info, int64_t* total_ties, bvq_stream_t stream) {
  if (channels < 1 || world < 1 || rank < 0 || rank >= world || !gathered || !dscale_total || !tie_info ||
      (!per_channel && !total_ties)) {
    set_error("bvq_shard_unpack: bad argument");
    return BVQ_ERR_INVALID;
  }
  shard_unpack_kernel<<<dim3((unsigned)((channels + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
      gathered, world, (int32_t)channels, rank, per_channel, dscale_total, reinterpret_cast<long long*>(tie_info),
      reinterpret_cast<long long*>(total_ties));
  return check_launch("bvq_shard_unpack");
}

extern "C" int bvq_scale_from_stat(const float* stat32, int64_t channels, int stat_dtype, void* stat_out,
                                   double min_val, int use_min, double int_threshold, int scale_dtype,
                                   void* scale_out, bvq_stream_t stream) {
  return bvq_scale_from_stat_running(stat32, channels, stat_dtype, stat_out, min_val, use_min, int_threshold,
                                     scale_dtype, scale_out, BVQ_F32, nullptr, 0.0, 0, stream);
}

extern "C" int bvq_scale_from_stat_running(const float* stat32, int64_t channels, int stat_dtype, void* stat_out,
                                           double min_val, int use_min, double int_threshold, int scale_dtype,
                                           void* scale_out, int run_dtype, void* running, double momentum,
                                           int first_batch, bvq_stream_t stream) {
  if (channels < 1 || bad_dtype(stat_dtype) || bad_dtype(scale_dtype) || !stat32 || !stat_out || !scale_out ||
      !(int_threshold == int_threshold) || (running && bad_dtype(run_dtype))) {
    set_error("bvq_scale_from_stat: bad argument");
    return BVQ_ERR_INVALID;
  }
  ScaleEpilogue ep = {};
  if (running) {
    ep.running = running;
    ep.run_dtype = run_dtype;
    ep.first_batch = first_batch;
    ep.one_minus_m = (float)(1.0 - momentum);
    ep.momentum = (float)momentum;
  }
  ep.scale_out = scale_out;
  ep.scale_dtype = scale_dtype;
  ep.use_min = use_min;
  ep.min_val = round_host((float)min_val, stat_dtype);
  ep.int_threshold = (float)int_threshold;
  scale_from_stat_kernel<<<dim3((unsigned)((channels + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
      stat32, stat_out, stat_dtype, ep, (int32_t)channels);
  return check_launch("bvq_scale_from_stat");
}

// ---- histogram over [-absmax, absmax] (KLMinimizerThreshold, B/core/stats/stats_op.py:311-312) ------------------
// torch.histc(x, bins, min=-absmax, max=absmax): equal-width bins, values outside the range ignored, the value `max`
// itself counted in the last bin; bin = (int)((v - min) * bins / (max - min)) in float32, the device kernel's formula
// (ATen/native/cuda/SummaryOps.cu, getBin).  One streaming read; per-workgroup bins in LDS, flushed with one
// atomic add per non-empty bin.  absmax is read from the device (dtype of x): no host sync.
constexpr int kHistMaxBins = 8192;

template <typename T>
__global__ __launch_bounds__(kBlock) void histc_kernel(const T* __restrict__ x, int64_t n, const void* absmax,
                                                       int32_t bins, int* __restrict__ out) {
  __shared__ int sh[kHistMaxBins];
  for (int i = threadIdx.x; i < bins; i += kBlock) sh[i] = 0;
  __syncthreads();
  const float hi = to_f<T>(*reinterpret_cast<const T*>(absmax)), lo = -hi;
  const float width = hi - lo;
  constexpr int VEC = elem<T>::vec;
  const int64_t nvec = n / VEC;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  auto count = [&](float v) {
    if (v >= lo && v <= hi) {
      int b = (int)((v - lo) * (float)bins / width);
      if (b == bins) b -= 1;
      if (b >= 0 && b < bins) atomicAdd(&sh[b], 1);  // (width == 0: NaN bin index, nothing counted by the guard)
    }
  };
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
    const vec_t<T, VEC> xv = load_vec<T, VEC>(x + i * VEC);
#pragma unroll
    for (int k = 0; k < VEC; ++k) count(to_f<T>(xv.v[k]));
  }
  const int64_t t = nvec * VEC + (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t < n) count(to_f<T>(x[t]));
  __syncthreads();
  for (int i = threadIdx.x; i < bins; i += kBlock)
    if (sh[i]) atomicAdd(&out[i], sh[i]);
}

__global__ void histc_zero_kernel(int* p, int32_t n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}

extern "C" int bvq_histc(int dtype, const void* x, int64_t n, const void* absmax, int bins, int32_t* counts,
                         bvq_stream_t stream) {
  if (bad_dtype(dtype) || n < 0 || bins < 1 || bins > kHistMaxBins || !counts || !absmax || (n > 0 && !x)) {
    set_error("bvq_histc: bad argument (1 <= bins <= %d)", kHistMaxBins);
    return BVQ_ERR_INVALID;
  }
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) {
    set_error("bvq_histc: x must be 16-byte aligned");
    return BVQ_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  histc_zero_kernel<<<dim3((unsigned)((bins + 255) / 256)), dim3(256), 0, st>>>(counts, bins);
  if (n > 0) {
    const int64_t per_block = (int64_t)kBlock * 16 * (16 / dtype_size(dtype));
    int64_t blocks = (n + per_block - 1) / per_block;
    if (blocks > 2048) blocks = 2048;
    const dim3 grid((unsigned)blocks), block(kBlock);
    if (dtype == BVQ_F32)
      histc_kernel<float><<<grid, block, 0, st>>>(reinterpret_cast<const float*>(x), n, absmax, bins, counts);
    else if (dtype == BVQ_BF16)
      histc_kernel<bf16_t><<<grid, block, 0, st>>>(reinterpret_cast<const bf16_t*>(x), n, absmax, bins, counts);
    else
      histc_kernel<f16_t><<<grid, block, 0, st>>>(reinterpret_cast<const f16_t*>(x), n, absmax, bins, counts);
  }
  return check_launch("bvq_histc");
}

// learned scale, forward:  scale = |clamp_min_ste(value, min_val)| / int_threshold  in ONE launch
// (ParameterScaling.forward, B/core/scaling/standalone.py:143-146, then the division of
// RescalingIntQuant.forward, B/core/quant/int.py:160): clamp and |.| are exact, the quotient is rounded to
// scale_dtype.  Its backward rides on the quantizer backward's last launch (bvq_fakequant_bwd_learned).
__global__ void learned_scale_kernel(const void* __restrict__ value, int value_dtype, ScaleEpilogue ep, int32_t n) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  float v = load_scalar_as_f(value, value_dtype, c);
  if (ep.use_min && v < ep.min_val) v = ep.min_val;  // NaN passes, like torch.clamp_min
  store_stat(ep.scale_out, ep.scale_dtype, c, fabsf(v) / ep.int_threshold);
}

extern "C" int bvq_learned_scale(int value_dtype, const void* value, int64_t n, double min_val, int use_min,
                                 double int_threshold, int scale_dtype, void* scale_out, bvq_stream_t stream) {
  if (n < 1 || n > (1ll << 30) || bad_dtype(value_dtype) || bad_dtype(scale_dtype) || !value || !scale_out ||
      !(int_threshold == int_threshold)) {
    set_error("bvq_learned_scale: bad argument");
    return BVQ_ERR_INVALID;
  }
  ScaleEpilogue ep = {};
  ep.scale_out = scale_out;
  ep.scale_dtype = scale_dtype;
  ep.use_min = use_min;
  ep.min_val = round_host((float)min_val, value_dtype);
  ep.int_threshold = (float)int_threshold;
  learned_scale_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(value, value_dtype, ep,
                                                                                                 (int32_t)n);
  return check_launch("bvq_learned_scale");
}

extern "C" int bvq_running_stats_update(int run_dtype, void* running, int stat_dtype, const void* stat,
                                        int64_t n, double momentum, int first_batch,
                                        bvq_stream_t stream) {
  if (bad_dtype(run_dtype) || bad_dtype(stat_dtype) || n < 0) {
    set_error("bvq_running_stats_update: bad argument");
    return BVQ_ERR_INVALID;
  }
  if (n == 0) return BVQ_OK;
  if (!running || !stat) {
    set_error("bvq_running_stats_update: null pointer");
    return BVQ_ERR_INVALID;
  }
  // torch turns the python scalars (1 - momentum) and momentum into float32 for these dtypes
  running_stats_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
      running, run_dtype, stat, stat_dtype, n, (float)(1.0 - momentum), (float)momentum, first_batch);
  return check_launch("bvq_running_stats_update");
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

static int check_stat_args(const char* fn, int match_flags, int dtype, int64_t outer, int64_t channels,
                           int64_t inner) {
  const int match = match_flags & ~BVQ_MATCH_FIRST;
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
  const int first_only = (match & BVQ_MATCH_FIRST) != 0;
  match &= ~BVQ_MATCH_FIRST;
  launch_tie_init(info, channels, st, first_only);
  if (outer * channels * inner == 0) return check_launch("bvq_stat_tie_scan");
  if (!x || !stat) {
    set_error("bvq_stat_tie_scan: null pointer");
    return BVQ_ERR_INVALID;
  }
  // channel-last (or nearly) per-channel layouts: column-mapped units
  const ColsPlan cp = (channels > 1 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx_zero_fill)) & 15) == 0)
                          ? cols_plan(dtype, outer, channels, inner)
                          : ColsPlan{};
  if (cp.ok) {
    const dim3 grid(grid_for_units(cp.units)), block(kBlock);
#define BVQ_TSC(T, M)                                                                                        \
  do {                                                                                                       \
    if (dx_zero_fill)                                                                                        \
      tie_scan_cols_kernel<T, M, true><<<grid, block, 0, st>>>(cp, x, stat, info, dx_zero_fill, inner);       \
    else                                                                                                     \
      tie_scan_cols_kernel<T, M, false><<<grid, block, 0, st>>>(cp, x, stat, info, nullptr, inner);           \
  } while (0)
#define BVQ_TSC_T(T)                      \
  do {                                    \
    if (match == BVQ_MATCH_ABS)           \
      BVQ_TSC(T, BVQ_MATCH_ABS);          \
    else                                  \
      BVQ_TSC(T, BVQ_MATCH_VALUE);        \
  } while (0)
    if (dtype == BVQ_F32)
      BVQ_TSC_T(float);
    else if (dtype == BVQ_BF16)
      BVQ_TSC_T(bf16_t);
    else
      BVQ_TSC_T(f16_t);
#undef BVQ_TSC_T
#undef BVQ_TSC
    return check_launch("bvq_stat_tie_scan/cols");
  }
  int vec;
  const Tiling t = stat_tiling(dtype, x, dx_zero_fill, outer, channels, inner, vec);
  BVQ_DISPATCH_T_MATCH(dtype, match, run_tie_scan, t, vec, x, stat, info, dx_zero_fill,
                       dx_zero_fill != nullptr, first_only, st);
  return check_launch("bvq_stat_tie_scan");
}

extern "C" int bvq_stat_tie_apply(int match, int pre_op, int dtype, const void* x, const void* stat,
                                  const void* gstat, const int64_t* tie_info, const int64_t* total_ties,
                                  void* dx, int64_t outer, int64_t channels, int64_t inner, int mode_add,
                                  bvq_stream_t stream) {
  int rc = check_stat_args("bvq_stat_tie_apply", match, dtype, outer, channels, inner);
  if (rc) return rc;
  if (outer * channels * inner == 0) return BVQ_OK;
  if (!x || !stat || !gstat || !tie_info || !dx) {
    set_error("bvq_stat_tie_apply: null pointer");
    return BVQ_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  GstatSrc src = {};
  src.p = gstat;
  src.pre_relu = pre_op == BVQ_PRE_RELU;
  const int first_only = (match & BVQ_MATCH_FIRST) != 0;
  match &= ~BVQ_MATCH_FIRST;
  BVQ_DISPATCH_T_MATCH(dtype, match, run_tie_apply, x, stat, src,
                       reinterpret_cast<const unsigned long long*>(tie_info),
                       reinterpret_cast<const unsigned long long*>(total_ties), dx, outer, channels, inner,
                       mode_add, first_only, st);
  return check_launch("bvq_stat_tie_apply");
}

extern "C" int bvq_stat_tie_apply_dscale(int pre_op, int dtype, const void* x, const void* stat,
                                         const float* dscale,
                                         int scale_dtype, double int_threshold, int quot_dtype,
                                         const int64_t* tie_info, const int64_t* total_ties, void* dx,
                                         int64_t outer, int64_t channels, int64_t inner,
                                         bvq_stream_t stream) {
  int rc = check_stat_args("bvq_stat_tie_apply_dscale", BVQ_MATCH_ABS, dtype, outer, channels, inner);
  if (rc) return rc;
  if (bad_dtype(scale_dtype) || bad_dtype(quot_dtype)) {
    set_error("bvq_stat_tie_apply_dscale: bad dtype");
    return BVQ_ERR_INVALID;
  }
  if (outer * channels * inner == 0) return BVQ_OK;
  if (!x || !stat || !dscale || !tie_info || !dx) {
    set_error("bvq_stat_tie_apply_dscale: null pointer");
    return BVQ_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  GstatSrc src = {};
  src.p = dscale;
  src.from_dscale = 1;
  src.scale_dtype = scale_dtype;
  src.quot_dtype = quot_dtype;
  src.int_threshold = (float)int_threshold;
  src.pre_relu = pre_op == BVQ_PRE_RELU;
  BVQ_DISPATCH_T_MATCH(dtype, BVQ_MATCH_ABS, run_tie_apply, x, stat, src,
                       reinterpret_cast<const unsigned long long*>(tie_info),
                       reinterpret_cast<const unsigned long long*>(total_ties), dx, outer, channels, inner, 1,
                       0, st);
  return check_launch("bvq_stat_tie_apply_dscale");
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
  return bvq_stat_tie_apply(match, BVQ_PRE_NONE, dtype, x, stat, gstat, info, nullptr, dx, outer, channels,
                            inner, mode_add, stream);
}
