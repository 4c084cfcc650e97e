// bvq_fakequant.hip -- fused affine quantize/dequantize forward and its autograd backward.
//
// Replaces the ~9 full-tensor ATen passes of IntQuant.forward (B/core/quant/int_base.py:63-97)
// with one read of x and one write of y, and the ~8 passes autograd runs for its backward with one
// read of g, one read of x and one write of dx (the per-channel scale / zero-point gradient sums,
// and the search for the elements that attain the abs-max statistic, ride on the same reads).
// HBM-bound: algorithmic bytes per element are
//   forward  sizeof(x) + sizeof(y)          backward  sizeof(g) + sizeof(x) + sizeof(dx).
#include "bvq_quant_math.h"
#include "bvq_ties.h"
#include "bvq_sums.h"

namespace bvq {

struct QuantArgs {
  Tiling t;
  const void* x;
  const void* scale;
  const void* zp;
  void* y;          // fwd: output; bwd: dx
  void* codes;      // fwd only, nullable; element type codes_dtype
  const void* g;    // bwd only
  float* ds_part;   // bwd only, per-unit partial of dscale
  float* dzp_part;  // bwd only, per-unit partial of dzp (kBwdDsBounds: of d(qmin))
  float* dq_part;   // bwd only, kBwdDsBounds: per-unit partial of d(qmax)
  const float* bounds;  // nullable: [qmin, qmax] as float32 ON THE DEVICE (a learned bit width) instead of qmin/qmax
  const void* tie_stat;          // bwd only: abs-max statistic (dtype of x) whose ties are recorded
  unsigned long long* tie_info;  // bwd only (bvq_ties.h)
  unsigned long long* pos_part;  // bwd only: per-unit first position attaining tie_stat (instead of tie_info)
  float qmin, qmax;
  int32_t scale_dtype, zp_dtype;
  int32_t scale_pc, zp_pc;
  int32_t scalar_cast;
  int32_t clamp_ste;
  int32_t out_int;
  int32_t round_mode;
  int32_t pre_relu;  // x is passed through torch.relu first (FusedActivationQuantProxy)
  int32_t codes_dtype;
  // bwd only, kBwdDsArrive (the stats-scaled backward in one launch): per-channel arrival counters, zero on entry
  // and on exit; the wave that completes a channel sums its partials, turns dscale into the statistic's gradient and
  // deposits it on the arg-max element of dx
  uint32_t* arrive;
  uint32_t arrive_per_channel;  // units of one channel
  float* dscale_out;            // [channels]
  int32_t gs_scale_dtype, gs_quot_dtype;  // GstatSrc of the deposit (bvq_ties.h)
  float gs_int_threshold;
  // batch-sharded tensors: instead of the deposit, the finishing wave writes this shard's message for the backward
  // all-gather -- float64 [2][channels]: the channel's dscale sum (NOT rounded to float32: the sums of all shards are
  // added in double and rounded once) and its claim on the deposit (shard_rank, or 2^30 with no arg-max here) -- and the
  // first arg-max position (-1: none) for bvq_shard_unpack_deposit
  double* shard_msg;
  long long* shard_pos;
  int32_t shard_rank;
};

#ifndef BVQ_FWD_UNROLL
#define BVQ_FWD_UNROLL 8
#endif
#ifndef BVQ_BWD_WAVES
#define BVQ_BWD_WAVES 4  // occupancy floor handed to the register allocator (waves per SIMD)
#endif
// developer switches of the float16 backward (tools/variant_bench.py): the division (0: guarded reciprocal DivF16,
// 1: refined reciprocal product DivF16R) and the walk (0: batches of two chunks, 1: the software-pipelined walk)
#ifndef BVQ_F16_BWD_DIV
#define BVQ_F16_BWD_DIV 1
#endif
#ifndef BVQ_F16_BWD_PIPE
#define BVQ_F16_BWD_PIPE 1
#endif
#ifndef BVQ_F16_COLS_DIV   // the column-mapped kernels' float16 division, same choice
#define BVQ_F16_COLS_DIV 1
#endif
#ifndef BVQ_BWD_DEPTH
#define BVQ_BWD_DEPTH 4
#endif
constexpr int kUnroll = BVQ_FWD_UNROLL;   // forward: 16-byte loads of x in flight per lane before arithmetic
constexpr int kBwdDepth = BVQ_BWD_DEPTH;  // backward: chunks of each input stream (x, g) prefetched ahead of the arithmetic

template <typename CT>
__device__ __forceinline__ void load_scale_zp(const QuantArgs& a, int32_t channel, float& s, float& z) {
  s = load_scalar_as_f(a.scale, a.scale_dtype, a.scale_pc ? channel : 0);
  z = load_scalar_as_f(a.zp, a.zp_dtype, a.zp_pc ? channel : 0);
  if (a.scalar_cast) {
    // device-torch semantics for a 0-dim operand wider than the compute dtype (see bvq.h)
    if (!a.scale_pc) s = rnd<CT>(s);
    if (!a.zp_pc) z = rnd<CT>(z);
  }
}

template <typename CT, int RM>
__device__ __forceinline__ float do_round(float t, int mode) {
  if constexpr (RM == kAnyRM) {
    return round_any<CT>(t, mode);
  } else {
    return round_op<CT, RM>(t);
  }
}
template <typename CT, int RM>
__device__ __forceinline__ f2 do_round2(f2 t, int mode) {
  if constexpr (RM == kAnyRM) {
    return round_any2<CT>(t, mode);
  } else {
    return round_op2<CT, RM>(t);
  }
}
__device__ __forceinline__ f2 relu2(f2 v) {
  const f2 zero = splat2(0.f);
  return v < zero ? zero : v;  // NaN and -0.0 pass through, like relu_f
}

// ------------------------------------------------------------------------------------------------
// division by the (wave-uniform) scale
// ------------------------------------------------------------------------------------------------
// DivExact: IEEE division, always right.
// DivBf16 : a * (1/s) for a bf16 quotient of bf16 operands (any zero-point).  The reference computes
//           RN_bf16(RN_f32(a / s)).  With a and s both bf16 values (8-bit significands) the exact
//           quotient is never closer than 2^-17 (relative) to a bf16 rounding boundary and never ON
//           one (a = m*s with m a 9-bit odd-ended midpoint needs >= 9 significant bits), while
//           a * RN_f32(1/s) is within 2^-23 of it: both round to the same bf16.  Used only when the
//           scale is a bf16 value in [2^-14, 2^14] (wave-uniform check); tests/test_fastdiv_exact.py
//           verifies the claim exhaustively over every bf16 numerator.
struct DivExact {
  float s;
  __device__ __forceinline__ float operator()(float a) const { return a / s; }
  __device__ __forceinline__ f2 operator()(f2 a) const {
    f2 r = {a.x / s, a.y / s};
    return r;
  }
};
struct DivBf16 {
  float r;
  __device__ __forceinline__ float operator()(float a) const { return a * r; }
  __device__ __forceinline__ f2 operator()(f2 a) const { return a * r; }
};

// the same with one scale per element of a pair (column-mapped kernels: a lane's columns differ in channel)
struct DivExactV {
  f2 s;
  __device__ __forceinline__ f2 operator()(f2 a) const {
    f2 r = {a.x / s.x, a.y / s.y};
    return r;
  }
};
struct DivBf16V {
  f2 r;
  __device__ __forceinline__ f2 operator()(f2 a) const { return a * r; }
};

// DivF16  : the same idea for float16 (11-bit significands): the exact quotient of two float16 values stays
//           >= 2^-22 (relative) away from every float16 rounding boundary of a NORMAL result, a * RN_f32(1/s)
//           is within 2^-23 of it.  Subnormal results round on an absolute grid where that argument fails
//           (254 wrong quotients in 10^8 random pairs, all subnormal), so a quotient with 0 < |q| < 2^-14
//           (taken with a margin: < 0x38810000) sends the whole wave through the IEEE division -- rare: it
//           needs |a| < 2^-14 s.  The backward has three divisions per element and was VALU-bound without
//           it.  Scale: a float16 value in [2^-14, 2^14].  tests/test_fastdiv_exact.py checks every
//           float16 numerator against 5 full binades of scales and a sample of the rest, the GPU test all.
__device__ __forceinline__ bool f16_quot_small(float q) {
  return (__builtin_bit_cast(uint32_t, q) & 0x7fffffffu) - 1u < 0x38810000u - 1u;
}
struct DivF16 {
  float s, r;
  __device__ __forceinline__ float operator()(float a) const {
    float q = a * r;
    if (__builtin_amdgcn_ballot_w64(f16_quot_small(q)) != 0) q = a / s;
    return q;
  }
  __device__ __forceinline__ f2 operator()(f2 a) const {
    f2 q = a * r;
    if (__builtin_amdgcn_ballot_w64(f16_quot_small(q.x) || f16_quot_small(q.y)) != 0) {
      q.x = a.x / s;
      q.y = a.y / s;
    }
    return q;
  }
};
struct DivF16V {
  f2 s, r;
  __device__ __forceinline__ f2 operator()(f2 a) const {
    f2 q = a * r;
    if (__builtin_amdgcn_ballot_w64(f16_quot_small(q.x) || f16_quot_small(q.y)) != 0) {
      q.x = a.x / s.x;
      q.y = a.y / s.y;
    }
    return q;
  }
};
// DivF16R : the correctly rounded float32 quotient from the correctly rounded reciprocal r = RN(1/s) (computed once
//           per wave by an IEEE division) in four full-rate instructions per element instead of the ~11 (one of them
//           the quarter-rate v_rcp) of a/s: q0 = a*r is within an ulp of a/s, rem = fma(-q0, s, a) is its exact
//           remainder, fma(rem, r, q0) rounds a/s correctly (Markstein), and v_div_fixup restores what the fmas
//           lose -- the sign of a zero numerator, an infinite numerator (rem would be NaN), NaN.  No wave-wide check,
//           no branch.  float16 operands keep every intermediate far from float32's overflow / underflow ranges
//           (|a/s| in [2^-38, 2^30]); equality with a/s is checked over EVERY float16 numerator x every float16
//           scale in [2^-14, 2^14] on the GPU (tests/test_gpu_fastdiv.py) and over a sample of scales with libm's
//           fmaf on the CPU (tests/test_fastdiv_exact.py).
__device__ __forceinline__ float div_refined(float a, float s, float r) {
  const float q0 = a * r;
  const float rem = __builtin_fmaf(-q0, s, a);
  return __builtin_amdgcn_div_fixupf(__builtin_fmaf(rem, r, q0), s, a);
}
struct DivF16R {
  float s, r;
  __device__ __forceinline__ float operator()(float a) const { return div_refined(a, s, r); }
  __device__ __forceinline__ f2 operator()(f2 a) const {
    f2 q = {div_refined(a.x, s, r), div_refined(a.y, s, r)};
    return q;
  }
};
#if BVQ_F16_COLS_DIV
#define BVQ_DIVF16V DivF16RV
#else
#define BVQ_DIVF16V DivF16V
#endif
struct DivF16RV {
  f2 s, r;
  __device__ __forceinline__ f2 operator()(f2 a) const {
    f2 q = {div_refined(a.x, s.x, r.x), div_refined(a.y, s.y, r.y)};
    return q;
  }
};
__device__ __forceinline__ bool f16_scale_ok(float s) {
  const uint32_t sb = __builtin_bit_cast(uint32_t, s);
  return (sb & 0x1fffu) == 0u && s >= 6.103515625e-05f && s <= 16384.f;
}

__device__ __forceinline__ bool bf16_scale_ok(float s) {
  const uint32_t sb = __builtin_bit_cast(uint32_t, s);
  return (sb & 0xffffu) == 0u && s >= 6.103515625e-05f && s <= 16384.f;
}
// the zero-point is exactly +0.0 (symmetric quantizers): see ZP0 below
__device__ __forceinline__ bool zp_is_pos_zero(float z) { return __builtin_bit_cast(uint32_t, z) == 0u; }

// ------------------------------------------------------------------------------------------------
// column-mapped quantizer kernels (ColsPlan, bvq_common.h): channel axis last or nearly last
// ------------------------------------------------------------------------------------------------
struct ColsQuantArgs {
  ColsPlan p;
  const void* x;
  const void* g;      // bwd
  void* y;            // fwd: y, bwd: dx
  const void* scale;  // [channels]
  const void* zp;     // [channels] or [1]
  float* ds_part;                // bwd: [prows][L] or null
  unsigned long long* pos_part;  // bwd: [prows][L] first position attaining tie_stat, or null
  const void* tie_stat;          // bwd: [channels] or null
  unsigned long long* tie_info;  // bwd: [channels] (atomic minimum) or null
  int64_t inner;
  int32_t channels;
  float qmin, qmax;
  int32_t scale_dtype, zp_dtype, zp_pc, scalar_cast, clamp_ste, round_mode, pre_relu;
};

// what a lane needs to know about its VEC columns, loaded once per unit
template <typename T>
struct ColsLane {
  static constexpr int VEC = elem<T>::vec;
  int64_t row0, row_end;
  int32_t chunk, sub;
  bool active;
  int32_t ch[VEC];
  f2 s2[VEC / 2], z2[VEC / 2];
  bool fast, zp0;  // wave-uniform: every column's scale suits the bf16 reciprocal / every zero-point is +0

  __device__ __forceinline__ bool init(const ColsQuantArgs& a) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    if (unit >= a.p.units) return false;
    const int64_t rblk = unit / a.p.strips;
    const int32_t strip = (int32_t)(unit - rblk * a.p.strips);
    sub = lane / a.p.lpr;
    chunk = strip * kWave + (lane - sub * a.p.lpr);
    active = sub < a.p.rpp && chunk < a.p.cps;
    row0 = rblk * a.p.rb + sub;
    row_end = (rblk + 1) * a.p.rb < a.p.rows ? (rblk + 1) * a.p.rb : a.p.rows;
    bool ok_fast = true, ok_zp0 = true;
    float sv[VEC], zv[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int64_t col = active ? (int64_t)chunk * VEC + k : 0;
      ch[k] = (int32_t)(col / a.inner);
      sv[k] = load_scalar_as_f(a.scale, a.scale_dtype, ch[k]);
      zv[k] = load_scalar_as_f(a.zp, a.zp_dtype, a.zp_pc ? ch[k] : 0);
      if (a.scalar_cast && !a.zp_pc) zv[k] = rnd<T>(zv[k]);
      ok_fast = ok_fast && (elem<T>::id == BVQ_F16 ? f16_scale_ok(sv[k]) : bf16_scale_ok(sv[k]));
      ok_zp0 = ok_zp0 && zp_is_pos_zero(zv[k]);
    }
#pragma unroll
    for (int k = 0; k < VEC / 2; ++k) {
      s2[k] = f2{sv[2 * k], sv[2 * k + 1]};
      z2[k] = f2{zv[2 * k], zv[2 * k + 1]};
    }
    fast = sizeof(T) == 2 && __builtin_amdgcn_ballot_w64(!ok_fast) == 0;
    zp0 = sizeof(T) == 2 && __builtin_amdgcn_ballot_w64(!ok_zp0) == 0;
    return true;
  }
};

// This file is compiled several times (brevitas_amd/csrc/build.py) so that its translation units build in parallel:
// BVQ_PART=1 holds the forward kernels and entry points, BVQ_PART=2 the backward entry points with the column-mapped
// and finishing kernels, BVQ_PART=21 / 22 / 23 the row-mapped backward kernel for bf16 / float16 / float32-arithmetic
// tensors (explicit instantiations of launch_bwd, the long pole of the build).
#ifndef BVQ_PART
#define BVQ_PART 0  // 0: everything in one translation unit
#endif
#define BVQ_BWD_CODE (BVQ_PART == 0 || BVQ_PART == 2 || BVQ_PART == 21 || BVQ_PART == 22 || BVQ_PART == 23)

#if BVQ_PART == 0 || BVQ_PART == 1
// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// ZP0: the zero-point is +0.0: "+ zp" only turns -0 into +0 and "- zp" is the identity, so their
// re-roundings are skipped (the values are already representable).
template <typename CT, int RM, bool ZP0, typename Div>
__device__ __forceinline__ float fwd_elem(float xf, const Div& div, float s, float z, float qmin,
                                          float qmax, bool out_int, int mode, float& q_out) {
  float t = rnd<CT>(div(xf));                  // y = x / scale            int_base.py:69
  t = ZP0 ? t + 0.f : rnd<CT>(t + z);          // y = y + zero_point       :70
  t = do_round<CT, RM>(t, mode);               // y = float_to_int_impl(y) :73
  const float q = clamp_where(t, qmin, qmax);  // y = tensor_clamp_impl(.) :74
  q_out = q;
  if (out_int) return q;
  return ZP0 ? rnd<CT>(q * s) : rnd<CT>(rnd<CT>(q - z) * s);  // (y_int - zero_point) * scale :93-94
}

// fwd_elem on a pair of elements (bvq_quant_math.h: packed fp32 / packed bf16 conversion)
template <typename CT, int RM, bool ZP0, typename Div, typename S>
__device__ __forceinline__ f2 fwd_elem2(f2 xf, const Div& div, S s, S z, float qmin, float qmax,
                                        bool out_int, int mode, f2& q_out) {
  f2 t = rnd2<CT>(div(xf));
  t = ZP0 ? t + 0.f : rnd2<CT>(t + z);
  t = do_round2<CT, RM>(t, mode);
  const f2 q = clamp_where2(t, qmin, qmax);
  q_out = q;
  if (out_int) return q;
  // the last rounding to CT is the caller's pack2<CT> (one v_cvt_pk_bf16_f32 for the pair)
  return ZP0 ? q * s : rnd2<CT>(q - z) * s;  // (y_int - zero_point) * scale :93-94
}

// store VEC integer codes (parity / export mode): int32, int8 or uint8
template <int VEC>
__device__ __forceinline__ void store_codes(void* base, int codes_dtype, int64_t off, const float* q) {
  if (codes_dtype == BVQ_CODES_I32) {
    vec_t<int32_t, VEC> cv;
#pragma unroll
    for (int k = 0; k < VEC; ++k) cv.v[k] = (int32_t)q[k];
    store_vec<int32_t, VEC>(reinterpret_cast<int32_t*>(base) + off, cv);
  } else if (codes_dtype == BVQ_CODES_I8) {
    vec_t<int8_t, VEC> cv;
#pragma unroll
    for (int k = 0; k < VEC; ++k) cv.v[k] = (int8_t)(int32_t)q[k];
    store_vec<int8_t, VEC>(reinterpret_cast<int8_t*>(base) + off, cv);
  } else {
    vec_t<uint8_t, VEC> cv;
#pragma unroll
    for (int k = 0; k < VEC; ++k) cv.v[k] = (uint8_t)(int32_t)q[k];
    store_vec<uint8_t, VEC>(reinterpret_cast<uint8_t*>(base) + off, cv);
  }
}

// NT: cache policy of the stores of y; NTL: of the loads of x (the same unless stated)
template <typename XT, typename CT, int VEC, int RM, bool NT, bool ZP0, bool PRE, bool NTL = NT, typename Div>
__device__ __forceinline__ void fwd_unit(const QuantArgs& a, const Unit& u, const Div& div, float s,
                                         float z, float qmin, float qmax) {
  const int lane = threadIdx.x & 63;
  const XT* __restrict__ xp = reinterpret_cast<const XT*>(a.x) + u.base;
  CT* __restrict__ yp = a.y ? reinterpret_cast<CT*>(a.y) + u.base : nullptr;
  void* const cp = a.codes;  // indexed from the tensor start: u.base + offset
  const bool out_int = a.out_int != 0;
  const int mode = a.round_mode;

  ChunkCursor cur;
  cur.init(u, VEC, lane);
  const int64_t total = (int64_t)u.nrows * cur.cpr;
  for (int64_t done = 0; done < total; done += (int64_t)kWave * kUnroll) {
    vec_t<XT, VEC> xv[kUnroll];
    int64_t off[kUnroll];
    bool ok[kUnroll];
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      ok[j] = cur.valid();
      off[j] = cur.offset(u.row_stride, VEC);
      xv[j] = load_vec<XT, VEC, NTL>(xp + (ok[j] ? off[j] : 0));  // past the end: re-read the unit's first chunk
      cur.next();
    }
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      if (ok[j]) {
        vec_t<CT, VEC> yv;
        float qv[VEC];
        if constexpr (VEC % 2 == 0) {
#pragma unroll
          for (int k = 0; k < VEC; k += 2) {
            f2 xf = widen2<XT>(xv[j].v[k], xv[j].v[k + 1]);
            if constexpr (PRE) xf = relu2(xf);
            f2 q2;
            const f2 r = fwd_elem2<CT, RM, ZP0>(xf, div, s, z, qmin, qmax, out_int, mode, q2);
            pack2<CT>(r, yv.v[k], yv.v[k + 1]);
            qv[k] = q2.x;
            qv[k + 1] = q2.y;
          }
        } else {
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            const float xf = PRE ? relu_f(to_f<XT>(xv[j].v[k])) : to_f<XT>(xv[j].v[k]);
            const float r = fwd_elem<CT, RM, ZP0>(xf, div, s, z, qmin, qmax, out_int, mode, qv[k]);
            yv.v[k] = from_f<CT>(r);
          }
        }
        if (yp) store_vec<CT, VEC, NT>(yp + off[j], yv);
        if (cp) store_codes<VEC>(cp, a.codes_dtype, u.base + off[j], qv);  // parity / export mode only
      }
    }
  }
  // ragged ends: the (< VEC) elements after the last full chunk of every row of the unit
  const int32_t tail = (int32_t)(u.len - (int64_t)cur.cpr * VEC);
  for (int32_t e = lane; e < u.nrows * tail; e += kWave) {
    const int32_t tr = e / tail, tk = e - tr * tail;
    const int64_t i = (int64_t)tr * u.row_stride + (int64_t)cur.cpr * VEC + tk;
    float q;
    const float xf = PRE ? relu_f(to_f<XT>(xp[i])) : to_f<XT>(xp[i]);
    const float r = fwd_elem<CT, RM, ZP0>(xf, div, s, z, qmin, qmax, out_int, mode, q);
    if (yp) yp[i] = from_f<CT>(r);
    if (cp) store_codes<1>(cp, a.codes_dtype, u.base + i, &q);
  }
}

template <typename XT, typename CT, int VEC, int RM, bool NT, bool NTL = NT>
__global__ __launch_bounds__(kBlock) void fakequant_fwd_kernel(QuantArgs a) {
  const Unit u = locate_unit(a.t);
  if (!u.valid) return;
  float s, z;
  load_scale_zp<CT>(a, u.channel, s, z);
  // the reference clamps against min_int/max_int converted to the tensor dtype (max_val.type_as(x))
  const float qmin = rnd<CT>(a.bounds ? a.bounds[0] : a.qmin), qmax = rnd<CT>(a.bounds ? a.bounds[1] : a.qmax);
  // wave-uniform choices: fused pre-activation, zero zero-point (16-bit compute types: saves two
  // re-roundings per element), and (bf16) the reciprocal fast path
  const bool zp0 = sizeof(CT) == 2 && zp_is_pos_zero(z);
#define BVQ_FWD_UNIT(ZP0, PRE, DIV) fwd_unit<XT, CT, VEC, RM, NT, ZP0, PRE, NTL>(a, u, DIV, s, z, qmin, qmax)
#define BVQ_FWD_PRE(ZP0, DIV)      \
  do {                             \
    if (a.pre_relu)                \
      BVQ_FWD_UNIT(ZP0, true, DIV); \
    else                           \
      BVQ_FWD_UNIT(ZP0, false, DIV); \
  } while (0)
  if constexpr (elem<CT>::id == BVQ_BF16) {
    if (bf16_scale_ok(s)) {
      const DivBf16 div{1.0f / s};
      if (zp0)
        BVQ_FWD_PRE(true, div);
      else
        BVQ_FWD_PRE(false, div);
      return;
    }
  }
  // float16: the refined reciprocal product (DivF16R: the exact float32 quotient in 4 instructions, no branch; the
  // guarded reciprocal's wave-wide check cost 15 % here: profiles/r01_f16_fastdiv.txt)
#ifndef BVQ_F16_FWD_EXACT
  if constexpr (elem<CT>::id == BVQ_F16) {
    if (f16_scale_ok(s)) {
      const DivF16R div{s, 1.0f / s};
      if (zp0)
        BVQ_FWD_PRE(true, div);
      else
        BVQ_FWD_PRE(false, div);
      return;
    }
  }
#endif
  const DivExact div{s};
  if constexpr (sizeof(CT) == 2) {
    if (zp0) {
      BVQ_FWD_PRE(true, div);
      return;
    }
  }
  BVQ_FWD_PRE(false, div);
#undef BVQ_FWD_PRE
#undef BVQ_FWD_UNIT
}


template <typename T, int RM, bool NT, bool ZP0, bool FAST>
__device__ __forceinline__ void cols_fwd_rows(const ColsQuantArgs& a, const ColsLane<T>& ln, float qmin, float qmax) {
  constexpr int VEC = elem<T>::vec;
#ifndef BVQ_COLS_FWD_UNROLL
#define BVQ_COLS_FWD_UNROLL 4  // rows in flight per lane
#endif
  constexpr int kU = BVQ_COLS_FWD_UNROLL;
  const T* __restrict__ xp = reinterpret_cast<const T*>(a.x) + (int64_t)ln.chunk * VEC;
  T* __restrict__ yp = reinterpret_cast<T*>(a.y) + (int64_t)ln.chunk * VEC;
  f2 r2[VEC / 2];
#pragma unroll
  for (int k = 0; k < VEC / 2; ++k) r2[k] = f2{1.0f / ln.s2[k].x, 1.0f / ln.s2[k].y};
  const int mode = a.round_mode;
  for (int64_t r = ln.row0; r < ln.row_end; r += (int64_t)kU * a.p.rpp) {
    vec_t<T, VEC> xv[kU];
    bool ok[kU];
#pragma unroll
    for (int j = 0; j < kU; ++j) {
      const int64_t rr = r + (int64_t)j * a.p.rpp;
      ok[j] = rr < ln.row_end;
      xv[j] = load_vec<T, VEC, NT>(xp + (ok[j] ? rr : ln.row0) * a.p.L);
    }
#pragma unroll
    for (int j = 0; j < kU; ++j) {
      if (ok[j]) {
        vec_t<T, VEC> yv;
#pragma unroll
        for (int k = 0; k < VEC; k += 2) {
          f2 xf = widen2<T>(xv[j].v[k], xv[j].v[k + 1]);
          if (a.pre_relu) xf = relu2(xf);
          f2 q2, res;
          if constexpr (FAST && elem<T>::id == BVQ_F16)
            res = fwd_elem2<T, RM, ZP0>(xf, BVQ_DIVF16V{ln.s2[k / 2], r2[k / 2]}, ln.s2[k / 2], ln.z2[k / 2], qmin, qmax, false,
                                        mode, q2);
          else if constexpr (FAST)
            res = fwd_elem2<T, RM, ZP0>(xf, DivBf16V{r2[k / 2]}, ln.s2[k / 2], ln.z2[k / 2], qmin, qmax, false, mode, q2);
          else
            res = fwd_elem2<T, RM, ZP0>(xf, DivExactV{ln.s2[k / 2]}, ln.s2[k / 2], ln.z2[k / 2], qmin, qmax, false, mode, q2);
          pack2<T>(res, yv.v[k], yv.v[k + 1]);
        }
        store_vec<T, VEC, NT>(yp + (r + (int64_t)j * a.p.rpp) * a.p.L, yv);
      }
    }
  }
}

template <typename T, int RM, bool NT>
__global__ __launch_bounds__(kBlock) void fakequant_fwd_cols_kernel(ColsQuantArgs a) {
  ColsLane<T> ln;
  if (!ln.init(a) || !ln.active) return;
  const float qmin = rnd<T>(a.qmin), qmax = rnd<T>(a.qmax);
  if constexpr (sizeof(T) == 2) {
    if (ln.fast) {
      if (ln.zp0)
        cols_fwd_rows<T, RM, NT, true, true>(a, ln, qmin, qmax);
      else
        cols_fwd_rows<T, RM, NT, false, true>(a, ln, qmin, qmax);
      return;
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (ln.zp0) {
      cols_fwd_rows<T, RM, NT, true, false>(a, ln, qmin, qmax);
      return;
    }
  }
  cols_fwd_rows<T, RM, NT, false, false>(a, ln, qmin, qmax);
}

// ------------------------------------------------------------------------------------------------
// statistic + quantizer in ONE kernel, small channels: the channel stays in registers between the two
// ------------------------------------------------------------------------------------------------
// AbsMax -> clamp_min -> / int_threshold -> IntQuant (zero zero-point): the stats-scaled graphs of
// SURVEY 8a.  The two-kernel form reads x twice (statistic, then quantize).  Here ONE workgroup owns a
// channel at a time: every wave loads one slice of the channel (<= 8 chunks of 16 bytes per lane: 8 KiB
// per wave) into registers, the workgroup agrees on the channel's maximum through LDS, and every wave
// quantizes what it still holds.  x is read ONCE, one launch instead of three.  Channels that do not fit
// one workgroup's registers take the two-kernel route: holding them across several workgroups (round 1) or
// pipelining slabs of channels through the Infinity Cache in one launch (round 2,
// profiles/r02_slab_pipeline_experiment.txt) both lost to it -- the hand-off between workgroups costs more
// than the saved read.
constexpr int kFusedSlots = 8;          // 16-byte chunks per lane held in registers
constexpr int kFusedSliceChunks = 512;  // kWave * kFusedSlots
constexpr int kFusedMaxWaves = 8;       // waves per workgroup

struct FusedArgs {
  const void* x;
  void* y;
  void* stat_out;   // [channels], dtype of x
  void* scale_out;  // [channels], scale_dtype
  int64_t outer, inner;
  int32_t channels;
  int32_t cpr;      // chunks per row
  int32_t spr;      // slices per row
  int32_t slices;   // slices per channel = outer * spr
  float qmin, qmax, min_val, int_threshold;
  int32_t use_min, scale_dtype, scale_pc, scalar_cast, round_mode, pre_relu;
};

// statistic (an |x| key) -> the statistic as a float and the scale, with the rounding points of
// clamp_min_ste(stat, min_val) / int_threshold (ScaleEpilogue of bvq_stats.hip)
template <typename T>
__device__ __forceinline__ float scale_from_key(uint32_t key, bool use_min, float min_val, float int_threshold,
                                                int scale_dtype, float& stat) {
  if constexpr (elem<T>::id == BVQ_F16)
    stat = (float)__builtin_bit_cast(f16_t, (uint16_t)key);
  else
    stat = __builtin_bit_cast(float, key);
  const float thr = (use_min && stat < min_val) ? min_val : stat;  // NaN passes, like torch.clamp_min
  float s = thr / int_threshold;
  // rounded to the scale's dtype as a tensor op would
  if (scale_dtype == BVQ_BF16)
    s = rnd<bf16_t>(s);
  else if (scale_dtype == BVQ_F16)
    s = rnd<f16_t>(s);
  return s;
}
template <typename T>
__device__ __forceinline__ void store_stat_scale(void* stat_out, void* scale_out, int scale_dtype, int32_t c,
                                                 float stat, float s) {
  if constexpr (elem<T>::id == BVQ_F32)
    reinterpret_cast<float*>(stat_out)[c] = stat;
  else
    reinterpret_cast<T*>(stat_out)[c] = (T)stat;  // exact: stat is a value of T
  if (scale_dtype == BVQ_F32)
    reinterpret_cast<float*>(scale_out)[c] = s;
  else if (scale_dtype == BVQ_BF16)
    reinterpret_cast<bf16_t*>(scale_out)[c] = (bf16_t)s;
  else
    reinterpret_cast<f16_t*>(scale_out)[c] = (f16_t)s;
}

template <typename T, int RM, bool PRE, typename Div>
__device__ __forceinline__ void fused_quantize(const vec_t<T, elem<T>::vec> (&xv)[kFusedSlots], const bool (&ok)[kFusedSlots],
                                               T* __restrict__ yp, int lane, const Div& div, float s,
                                               float qmin, float qmax, int mode) {
  constexpr int VEC = elem<T>::vec;
  constexpr bool ZP0 = sizeof(T) == 2;
#pragma unroll
  for (int j = 0; j < kFusedSlots; ++j) {
    if (ok[j]) {
      vec_t<T, VEC> yv;
#pragma unroll
      for (int k = 0; k < VEC; k += 2) {
        f2 xf = widen2<T>(xv[j].v[k], xv[j].v[k + 1]);
        if constexpr (PRE) xf = relu2(xf);
        f2 q2;
        const f2 r = fwd_elem2<T, RM, ZP0>(xf, div, s, 0.f, qmin, qmax, false, mode, q2);
        pack2<T>(r, yv.v[k], yv.v[k + 1]);
      }
      store_vec<T, VEC, true>(yp + (int64_t)(lane + kWave * j) * VEC, yv);
    }
  }
}

__global__ void fused_zero_kernel(uint32_t* p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}

template <typename T, int RM>
__global__ __launch_bounds__(kFusedMaxWaves * kWave) void fused_absmax_fakequant_kernel(FusedArgs a) {
  constexpr int VEC = elem<T>::vec;
  __shared__ uint32_t sh_max[kFusedMaxWaves];
  __shared__ uint32_t sh_stat;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nwaves = (int)(blockDim.x >> 6);
  const int q = wave;  // this wave's slice of every channel the workgroup visits
  const bool active = q < a.slices;
  const int r = active ? q / a.spr : 0;
  const int sl = active ? q - r * a.spr : 0;
  const int nch = active ? (a.cpr - sl * kFusedSliceChunks < kFusedSliceChunks ? a.cpr - sl * kFusedSliceChunks
                                                                              : kFusedSliceChunks)
                         : 0;
  bool ok[kFusedSlots];
#pragma unroll
  for (int j = 0; j < kFusedSlots; ++j) ok[j] = lane + kWave * j < nch;
  const float qmin = rnd<T>(a.qmin), qmax = rnd<T>(a.qmax);

  for (int32_t c = blockIdx.x; c < a.channels; c += gridDim.x) {
    const int64_t base = ((int64_t)r * a.channels + c) * a.inner + (int64_t)sl * kFusedSliceChunks * VEC;
    const T* __restrict__ xp = reinterpret_cast<const T*>(a.x) + base;
    T* __restrict__ yp = reinterpret_cast<T*>(a.y) + base;
    // phase 1: the slice into registers, its maximum |x| key
    vec_t<T, VEC> xv[kFusedSlots];
#pragma unroll
    for (int j = 0; j < kFusedSlots; ++j)
      xv[j] = load_vec<T, VEC, true>(ok[j] ? xp + (int64_t)(lane + kWave * j) * VEC : reinterpret_cast<const T*>(a.x));
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < kFusedSlots; ++j) {
      if (ok[j]) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const uint32_t b = a.pre_relu ? pre_abs_bits<T, true>(xv[j].v[k]) : pre_abs_bits<T, false>(xv[j].v[k]);
          m = b > m ? b : m;
        }
      }
    }
    m = wave_max_u32(m);
    if (lane == 0) sh_max[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t bm = 0;
      for (int w = 0; w < nwaves; ++w) bm = sh_max[w] > bm ? sh_max[w] : bm;
      sh_stat = bm;
    }
    __syncthreads();
    float stat;
    float s = scale_from_key<T>(sh_stat, a.use_min, a.min_val, a.int_threshold, a.scale_dtype, stat);
    if (threadIdx.x == 0) store_stat_scale<T>(a.stat_out, a.scale_out, a.scale_dtype, c, stat, s);
    // a 0-dim float32 scale next to a 16-bit tensor is rounded again by the device's scalar semantics
    if (a.scalar_cast && !a.scale_pc) s = rnd<T>(s);
    // phase 2: quantize what the registers still hold
    const int mode = a.round_mode;
    if constexpr (elem<T>::id == BVQ_BF16) {
      if (bf16_scale_ok(s)) {
        const DivBf16 div{1.0f / s};
        if (a.pre_relu)
          fused_quantize<T, RM, true>(xv, ok, yp, lane, div, s, qmin, qmax, mode);
        else
          fused_quantize<T, RM, false>(xv, ok, yp, lane, div, s, qmin, qmax, mode);
        continue;
      }
    }
    if constexpr (elem<T>::id == BVQ_F16) {
      if (f16_scale_ok(s)) {
        const DivF16R div{s, 1.0f / s};
        if (a.pre_relu)
          fused_quantize<T, RM, true>(xv, ok, yp, lane, div, s, qmin, qmax, mode);
        else
          fused_quantize<T, RM, false>(xv, ok, yp, lane, div, s, qmin, qmax, mode);
        continue;
      }
    }
    const DivExact div{s};
    if (a.pre_relu)
      fused_quantize<T, RM, true>(xv, ok, yp, lane, div, s, qmin, qmax, mode);
    else
      fused_quantize<T, RM, false>(xv, ok, yp, lane, div, s, qmin, qmax, mode);
  }
}

#endif  // forward part

#if BVQ_BWD_CODE
// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// Autograd of the forward chain:  round_ste passes the gradient; TensorClamp masks
// clipped positions (TensorClampSte passes them); x/scale and y*scale give
//   dx     = (pass ? g*scale : 0) / scale
//   dscale = sum g*(q - zp)  -  sum dt * ((x/scale)/scale)        (torch: -grad * ((a/b)/b))
//   dzp    = sum dt  -  sum g*scale
// MODE: 0 = dx only, 1 = + dscale, 2 = + dscale and dzp, 3 = + dscale and abs-max tie search,
// 4 = + dscale and the gradients of the clamp bounds (a learned bit width with a plain TensorClamp: the two
//     torch.where of tensor_clamp send the gradient of a replaced value to the bound that replaced it).
// 5 = mode 3 finished in the same launch (per-channel layouts): dx stores and the per-unit partials are written
//     through (sc1), every wave counts itself in on its channel's arrival counter, and the wave that completes the
//     channel does what bwd_stats_finish_kernel does in a second launch.  Nobody waits.
enum { kBwdDx = 0, kBwdDs = 1, kBwdDsDzp = 2, kBwdDsTies = 3, kBwdDsBounds = 4, kBwdDsArrive = 5 };
template <int MODE>
constexpr bool kTieMode = MODE == kBwdDsTies || MODE == kBwdDsArrive;

template <typename CT, int RM, int MODE, bool ZP0, typename Div>
__device__ __forceinline__ float bwd_elem(float xf, float gf, const Div& div, float s, float z, float qmin,
                                          float qmax, bool clamp_ste, int mode, float& ds_acc,
                                          float& dzp_acc, float& dq_acc) {
  const float t1 = rnd<CT>(div(xf));
  const float t2 = ZP0 ? t1 + 0.f : rnd<CT>(t1 + z);
  const float t3 = do_round<CT, RM>(t2, mode);
  const bool hi = t3 > qmax;
  float t4 = hi ? qmax : t3;
  const bool lo = t4 < qmin;
  t4 = lo ? qmin : t4;
  const bool pass = clamp_ste || !(hi || lo);
  const float gs = rnd<CT>(gf * s);
  const float dt = pass ? gs : 0.f;
  const float dxv = rnd<CT>(div(dt));
  if constexpr (MODE >= kBwdDs) {
    const float t5 = ZP0 ? t4 : rnd<CT>(t4 - z);
    const float term1 = rnd<CT>(gf * t5);
    const float term2 = rnd<CT>(-dt * rnd<CT>(div(t1)));
    ds_acc += term1;
    ds_acc += term2;
  }
  if constexpr (MODE == kBwdDsDzp) dzp_acc += dt - gs;
  if constexpr (MODE == kBwdDsBounds) {
    dzp_acc += lo ? gs : 0.f;  // d(qmin)
    dq_acc += hi ? gs : 0.f;   // d(qmax)
  }
  return dxv;
}

// bwd_elem on a pair of elements; the sums are kept as pairs too (added up once per unit)
//
// BVQ_BWD_LEAN (round 3; the kernel issued 23 VALU instructions per element, 88 % VALU-busy at the 2-read-1-write
// ceiling -- profiles/r02/pmc_final_build.md): the same values with fewer instructions --
//  * the clamp is v_med3_f32 and the pass mask ONE compare, "not (clamped <> rounded)" (true for equal and for NaN,
//    as the reference's two `where` leave a NaN in place and pass its gradient): 2 instructions per element
//    instead of 2 compares + 2 selects.  (A NaN element's clamped value differs -- med3 returns a bound -- but it
//    only feeds term1 of a dscale sum that term2 = -dt * ((x / s) / s) has made NaN already.)
//  * with a zero zero-point the backward needs no "+ 0.0": it only turns -0 into +0, which no comparison, no
//    gradient value and no sum can see;
//  * bf16: the two rounded terms of the scale gradient are ADDED by v_dot2c_f32_bf16 (acc += lo * 1 + hi * 1) straight
//    from the packed conversion: no unpacking (two shifts / masks per pair) and no packed add.  Term 1 and term 2 go
//    to the two halves of the pair accumulator: two independent chains.
#ifndef BVQ_BWD_LEAN
#define BVQ_BWD_LEAN 1
#endif
// acc + RN_bf16(v.x) + RN_bf16(v.y)
__device__ __forceinline__ float add_rounded_pair_bf16(float acc, f2 v) {
  typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 ones = {(bf16_t)1.0f, (bf16_t)1.0f};
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_convertvector(v, bf16x2), ones, acc, false);
}
// {acc.x + RN_bf16(v.x), acc.y + RN_bf16(v.y)}: the column-mapped kernels keep one sum per element of the pair.
// The selectors {1, 0} and {0, 1} must live in registers the compiler cannot see through: as a constant, {1.0bf16, 0}
// = 0x00003f80 is emitted as the inline constant "1.0", which the instruction reads as 0x3f800000 = {0, 1.0bf16}
// (ROCm 7.2 / gfx950: both sums then received the pair's second element).  make_dot_sel() once per kernel.
struct DotSel {
  uint32_t lo, hi;
};
__device__ __forceinline__ DotSel make_dot_sel() {
  DotSel d = {0x00003f80u, 0x3f800000u};
  asm volatile("" : "+s"(d.lo), "+s"(d.hi));
  return d;
}
__device__ __forceinline__ f2 add_rounded_lanes_bf16(f2 acc, f2 v, const DotSel& sel) {
  typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 p = __builtin_convertvector(v, bf16x2);
  return f2{__builtin_amdgcn_fdot2_f32_bf16(p, __builtin_bit_cast(bf16x2, sel.lo), acc.x, false),
            __builtin_amdgcn_fdot2_f32_bf16(p, __builtin_bit_cast(bf16x2, sel.hi), acc.y, false)};
}
// MIX: the caller adds the two halves of ds_acc up in the end (row-mapped units: one channel per wave), so the sums
// of the pair's elements may share an accumulator; otherwise ds_acc.x / .y stay the sums of element x / y.
template <typename CT, int RM, int MODE, bool ZP0, bool SAME16, bool MIX = false, typename Div, typename S>
__device__ __forceinline__ f2 bwd_elem2(f2 xf, f2 gf, const Div& div, S s, S z, float qmin, float qmax,
                                        bool clamp_ste, int mode, f2& ds_acc, f2& dzp_acc, f2& dq_acc,
                                        const DotSel& sel = DotSel{}) {
  const f2 t1 = rnd2<CT>(div(xf));
  constexpr bool kLean = BVQ_BWD_LEAN && MODE != kBwdDsBounds;
  const f2 t2 = ZP0 ? (kLean ? t1 : t1 + 0.f) : rnd2<CT>(t1 + z);
  const f2 t3 = do_round2<CT, RM>(t2, mode);
  const f2 qhi = splat2(qmax), qlo = splat2(qmin);
  f2 t4;
  b2 hi, lo, pass;
  const b2 all = {-1, -1};
  if constexpr (kLean) {
    t4 = f2{__builtin_amdgcn_fmed3f(t3.x, qmin, qmax), __builtin_amdgcn_fmed3f(t3.y, qmin, qmax)};
    const b2 same = {__builtin_islessgreater(t4.x, t3.x) ? 0 : -1, __builtin_islessgreater(t4.y, t3.y) ? 0 : -1};
    pass = clamp_ste ? all : same;
  } else {
    hi = t3 > qhi;
    t4 = hi ? qhi : t3;
    lo = t4 < qlo;
    t4 = lo ? qlo : t4;
    pass = clamp_ste ? all : ~(hi | lo);
  }
  const f2 gs = rnd2<CT>(gf * s);
  const f2 dt = pass ? gs : splat2(0.f);
  // rounded to CT, then stored as XT by the caller's pack2: when both are the same 16-bit type that second
  // conversion IS the rounding (rounding twice to one grid changes nothing), so it is not done here
  const f2 dxv = SAME16 ? div(dt) : rnd2<CT>(div(dt));
  if constexpr (MODE >= kBwdDs) {
    const f2 t5 = ZP0 ? t4 : rnd2<CT>(t4 - z);
    // (every product rounded to CT like the reference's ops.  Keeping the two terms in float32 would save three
    //  roundings per element, but the compiler then holds 25 more registers live -- 117 instead of 92 at depth 4,
    //  one wave per SIMD less -- and the kernel is no faster: profiles/r02_backward_variants.txt)
    if constexpr (kLean && elem<CT>::id == BVQ_BF16 && MIX) {
      const float a1 = add_rounded_pair_bf16(ds_acc.x, gf * t5);
      const float a2 = add_rounded_pair_bf16(ds_acc.y, -dt * rnd2<CT>(div(t1)));
      ds_acc = f2{a1, a2};
    } else if constexpr (kLean && elem<CT>::id == BVQ_BF16) {
      ds_acc = add_rounded_lanes_bf16(ds_acc, gf * t5, sel);
      ds_acc = add_rounded_lanes_bf16(ds_acc, -dt * rnd2<CT>(div(t1)), sel);
    } else {
      const f2 term1 = rnd2<CT>(gf * t5);
      const f2 term2 = rnd2<CT>(-dt * rnd2<CT>(div(t1)));
      ds_acc += term1;
      ds_acc += term2;
    }
  }
  if constexpr (MODE == kBwdDsDzp) dzp_acc += dt - gs;
  if constexpr (MODE == kBwdDsBounds) {
    dzp_acc += lo ? gs : splat2(0.f);  // d(qmin)
    dq_acc += hi ? gs : splat2(0.f);   // d(qmax)
  }
  return dxv;
}

// ---- the stats-scaled backward finished in the same launch (kBwdDsArrive) ------------------------------------------
// a store / load that is performed at agent scope (global_store / global_load ... sc1): written through to, read from
// the memory every XCD sees (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 stores and sc1 loads on both sides
// of a hand-off, the storing wave's vmcnt(0) wait before its arrival add)
template <typename T>
__device__ __forceinline__ void store_through(T* p, T v) {
  if constexpr (sizeof(T) == 2) {
    __hip_atomic_store(reinterpret_cast<uint16_t*>(p), __builtin_bit_cast(uint16_t, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  } else {
    __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
}
template <typename T>
__device__ __forceinline__ T load_through(const T* p) {
  if constexpr (sizeof(T) == 2) {
    return __builtin_bit_cast(T, __hip_atomic_load(reinterpret_cast<const uint16_t*>(p), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT));
  } else {
    return __builtin_bit_cast(T, __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT));
  }
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)b, off, kWave);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(b >> 32), off, kWave);
    v += __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);  // a butterfly: every lane ends with the same bits
  }
  return v;
}

// One wave finishes channel c of the stats-scaled backward from the units' partials (written through by their waves,
// read through here): sum of the dscale partials (double, fixed order: lane l takes partials l, l + 64, ... in order,
// then a butterfly), first position attaining the statistic; then either dscale -> statistic's gradient (the backward
// of scale = clamp_min_ste(stat) / int_threshold with torch's rounding points) and its deposit on that element of dx,
// or (batch-sharded tensors) this shard's message for the all-gather.
template <typename XT, bool PRE>
__device__ __forceinline__ void channel_finish(const QuantArgs& a, int32_t c, int lane) {
  const uint32_t* pos32 = reinterpret_cast<const uint32_t*>(a.pos_part);
  const int64_t n = (int64_t)a.arrive_per_channel;
  const int64_t ppr = a.t.ppr;
  double acc = 0.0;
  unsigned long long pmin = ~0ull;
  for (int64_t k = lane; k < n; k += kWave) {
    int64_t unit;
    if (a.t.nob == 1) {
      unit = (int64_t)c * ppr + k;
    } else {
      const int64_t o = k / ppr, p = k - o * ppr;
      unit = (o * a.t.channels + c) * ppr + p;
    }
    acc += (double)load_through<float>(a.ds_part + unit);
    const unsigned long long q = (unsigned long long)load_through<uint32_t>(pos32 + 2 * unit) |
                                 ((unsigned long long)load_through<uint32_t>(pos32 + 2 * unit + 1) << 32);
    pmin = q < pmin ? q : pmin;
  }
  acc = wave_sum_f64(acc);
  pmin = wave_min_u64(pmin);
  if (lane != 0) return;
  if (a.shard_msg) {
    a.shard_msg[c] = acc;
    a.shard_msg[a.t.channels + c] = pmin != ~0ull ? (double)a.shard_rank : kShardNoOwner;
    a.shard_pos[c] = pmin != ~0ull ? (long long)pmin : -1ll;
    return;
  }
  const float dsum = (float)acc;
  a.dscale_out[c] = dsum;
  if (pmin != ~0ull) {  // ~0: no element equals the statistic (e.g. NaN)
    float v = round_rt(dsum, a.gs_scale_dtype);
    v = round_rt(v / a.gs_int_threshold, a.gs_quot_dtype);
    const float g = rnd<XT>(v);
    const unsigned long long inner = (unsigned long long)a.t.row_len;
    const int64_t o = (int64_t)(pmin / inner);
    const int64_t i = (int64_t)(pmin - (unsigned long long)o * inner);
    const int64_t flat = (o * a.t.channels + c) * (int64_t)inner + i;
    const XT* xp = reinterpret_cast<const XT*>(a.x);
    XT* dp = reinterpret_cast<XT*>(a.y);
    const float term = deposit<XT, BVQ_MATCH_ABS>(g, xp[flat], PRE);
    store_through<XT>(dp + flat, from_f<XT>(to_f<XT>(load_through<XT>(dp + flat)) + term));
  }
}

// One wave has finished its unit: publish the unit's partials, count the unit in, and -- if that completes the
// channel -- finish the channel: sum of the dscale partials (double, fixed order: lane l takes partials l, l + 64, ...
// in order, then a butterfly), first position attaining the statistic, dscale -> statistic's gradient (the backward of
// scale = clamp_min_ste(stat) / int_threshold with torch's rounding points) and its deposit on that element of dx.
// ds / first: the unit's wave-reduced dscale sum and first arg-max position (~0: none).
template <typename XT, bool PRE>
__device__ __forceinline__ void bwd_arrive(const QuantArgs& a, const Unit& u, float ds, unsigned long long first,
                                           int lane) {
  uint32_t* pos32 = reinterpret_cast<uint32_t*>(a.pos_part);
  uint32_t last = 0;
  if (lane == 0) {
    store_through<float>(a.ds_part + u.id, ds);
    store_through<uint32_t>(pos32 + 2 * u.id, (uint32_t)first);
    store_through<uint32_t>(pos32 + 2 * u.id + 1, (uint32_t)(first >> 32));
  }
  // every store of this wave (dx chunks of all lanes, the partials) has been performed before the unit is counted in
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) {
    const uint32_t before = __hip_atomic_fetch_add(a.arrive + u.channel, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = before + 1u == a.arrive_per_channel ? 1u : 0u;
  }
  if (!__builtin_amdgcn_readfirstlane(last)) return;
  // ---- last arriver of this channel (rare: once per channel) ----
  if (lane == 0) __hip_atomic_store(a.arrive + u.channel, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // handed back as zero
  channel_finish<XT, PRE>(a, u.channel, lane);
}

// NT: cache policy of the loads of g and the stores of dx; NTX: of the loads of x (the same unless stated)
template <typename XT, typename CT, int VEC, int RM, int MODE, bool NT, bool ZP0, bool PRE, bool NTX = NT, typename Div>
__device__ __forceinline__ void bwd_unit(const QuantArgs& a, const Unit& u, const Div& div, float s,
                                         float z, float qmin, float qmax) {
  const int lane = threadIdx.x & 63;
  const bool clamp_ste = a.clamp_ste != 0;
  const int mode = a.round_mode;
  const XT* __restrict__ xp = reinterpret_cast<const XT*>(a.x) + u.base;
  const CT* __restrict__ gp = reinterpret_cast<const CT*>(a.g) + u.base;
  XT* __restrict__ dxp = reinterpret_cast<XT*>(a.y) + u.base;

  // abs-max tie search: |x| == statistic of this unit's channel
  uint32_t stat_bits = 0;
  const bool per_channel = a.t.channels > 1;
  if constexpr (kTieMode<MODE>)
    stat_bits = abs_bits<XT>(reinterpret_cast<const XT*>(a.tie_stat)[u.channel]);

  float ds_acc = 0.f, dzp_acc = 0.f, dq_acc = 0.f;
  uint32_t umax = 0;  // kBwdDsTies: largest |x| key this lane has seen in the unit's full chunks
  unsigned long long tie_first = ~0ull;
  f2 ds_acc2 = splat2(0.f), dzp_acc2 = splat2(0.f), dq_acc2 = splat2(0.f);  // pairwise path; folded into the scalars at the end
  // the unit through buffer descriptors (bvq_common.h): lanes past the unit's end load zeros without touching
  // memory -- x = g = 0 adds nothing to any sum and is no tie -- and their stores are dropped, so the walk below
  // needs no branch and no execution mask
  const int64_t extent = (int64_t)(u.nrows - 1) * u.row_stride + u.len;  // elements, first to last of the unit
  const buf_t bx = make_buf(xp, (uint32_t)(extent * (int64_t)sizeof(XT)));
  const buf_t bg = make_buf(gp, (uint32_t)(extent * (int64_t)sizeof(CT)));
  const buf_t bd = make_buf(dxp, (uint32_t)(extent * (int64_t)sizeof(XT)));
  constexpr uint32_t kSkip = 0x60000000u;  // element offset whose byte offset is >= 2^31 for 2- and 4-byte elements
  // the work on one chunk (VEC elements of x and g -> VEC elements of dx, sums and the chunk's largest |x| key)
  auto chunk = [&](const vec_t<XT, VEC>& xv, const vec_t<CT, VEC>& gv, uint32_t off) {
    vec_t<XT, VEC> dv;
    if constexpr (VEC % 2 == 0) {
#pragma unroll
      for (int k = 0; k < VEC; k += 2) {
        const f2 xraw = widen2<XT>(xv.v[k], xv.v[k + 1]);
        constexpr bool kSame16 = sizeof(CT) == 2 && sizeof(XT) == 2;  // then XT is CT (dispatch pairs)
        f2 d = bwd_elem2<CT, RM, MODE, ZP0, kSame16, true>(PRE ? relu2(xraw) : xraw, widen2<CT>(gv.v[k], gv.v[k + 1]),
                                            div, s, z, qmin, qmax, clamp_ste, mode, ds_acc2, dzp_acc2, dq_acc2);
        if constexpr (PRE) d = xraw > splat2(0.f) ? d : splat2(0.f);  // torch.relu backward: grad * (x > 0)
        pack2<XT>(d, dv.v[k], dv.v[k + 1]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float xraw = to_f<XT>(xv.v[k]);
        float d = bwd_elem<CT, RM, MODE, ZP0>(PRE ? relu_f(xraw) : xraw, to_f<CT>(gv.v[k]), div, s, z,
                                              qmin, qmax, clamp_ste, mode, ds_acc, dzp_acc, dq_acc);
        if constexpr (PRE) d = xraw > 0.f ? d : 0.f;  // torch.relu backward: grad * (x > 0)
        dv.v[k] = from_f<XT>(d);
      }
    }
    if constexpr (kTieMode<MODE>) {
      // cheap chunk-level filter: a tie in this chunk needs the chunk's max |x| to reach the statistic
      if constexpr (sizeof(XT) == 2 && VEC % 2 == 0 && !PRE) {
        // two 16-bit keys per word: clear both sign bits, packed unsigned max (2 ops per pair)
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        const vec_t<uint32_t, VEC / 2> w = __builtin_bit_cast(vec_t<uint32_t, VEC / 2>, xv);
        u16x2 m2 = {0, 0};
#pragma unroll
        for (int k = 0; k < VEC / 2; ++k)
          m2 = __builtin_elementwise_max(m2, __builtin_bit_cast(u16x2, w.v[k] & 0x7fff7fffu));
        const uint32_t m16 = m2.x > m2.y ? m2.x : m2.y;
        const uint32_t mx = elem<XT>::id == BVQ_BF16 ? (m16 << 16) : m16;  // the abs_bits<> key space
        umax = mx > umax ? mx : umax;
      } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const uint32_t b = pre_abs_bits<XT, PRE>(xv.v[k]);
          umax = b > umax ? b : umax;
        }
      }
    }
    // (kBwdDsArrive: written through, so that the finishing wave -- maybe on another XCD -- finds every dx element in
    //  memory: MI355X_MICROARCH.md, inter-workgroup visibility)
    buf_store<XT, VEC, NT, MODE == kBwdDsArrive>(bd, off * (uint32_t)sizeof(XT), dv);  // dropped where off is kSkip
  };
  ChunkCursor cur;
  cur.init(u, VEC, lane);
  const int64_t total = (int64_t)u.nrows * cur.cpr;
  if constexpr (elem<CT>::id == BVQ_F16 && !BVQ_F16_BWD_PIPE) {
    // float16: batches of two chunks per stream, loaded together, then worked on.  Its arithmetic (two converts per
    // rounding, the guarded reciprocal's wave-wide checks) is what bounds it, and the pipelined form below is
    // 7-20 % SLOWER here (profiles/r02_backward_variants.txt).
    constexpr int kU = 2;
    const uint32_t rs = (uint32_t)u.row_stride;
    for (int64_t done = 0; done < total; done += (int64_t)kWave * kU) {
      vec_t<XT, VEC> xv[kU];
      vec_t<CT, VEC> gv[kU];
      uint32_t off[kU];
#pragma unroll
      for (int j = 0; j < kU; ++j) {
        off[j] = cur.valid() ? cur.offset32(rs, VEC) : kSkip;
        xv[j] = buf_load<XT, VEC, NTX>(bx, off[j] * (uint32_t)sizeof(XT));
        gv[j] = buf_load<CT, VEC, NT>(bg, off[j] * (uint32_t)sizeof(CT));
        cur.next();
      }
#pragma unroll
      for (int j = 0; j < kU; ++j)
        if (done + (int64_t)j * kWave < total) chunk(xv[j], gv[j], off[j]);  // wave-uniform: a step no lane has is not computed
    }
  } else {
    // Software-pipelined walk: the loads of chunk i + kD are issued BEFORE chunk i is worked on, so every wave
    // keeps kD chunks of x and of g in flight while it computes (the counters of the round-1 kernel showed its
    // waves 45 % of their time in arithmetic or waiting to issue with nothing in flight:
    // profiles/r02/pmc_backward_r01_kernel.md).  One chunk per step, so a 56x56 row (392 chunks) costs 7 steps
    // of arithmetic instead of 4 x 2.
    // (a 32-byte chunk of g -- float32 arithmetic next to a 16-bit tensor -- at depth 4 would spill)
    constexpr int kD = sizeof(CT) * VEC > 16 ? 2 : kBwdDepth;
    // kD + 1 register sets, walked round-robin: step i works on set i % kS while chunk i + kD is loaded into set
    // (i - 1) % kS, the one step i - 1 has just finished with.  (With kD sets the refill of a slot overlaps the
    // work on its old contents and the compiler copies 8 registers aside per step: 2.6 of 21 issues per element.)
    constexpr int kS = kD + 1;
    const int32_t steps = (int32_t)((total + kWave - 1) / kWave);  // chunks per lane, the last possibly partial
    const uint32_t rs = (uint32_t)u.row_stride;
    vec_t<XT, VEC> xb[kS];
    vec_t<CT, VEC> gb[kS];
    uint32_t offb[kS];
#pragma unroll
    for (int j = 0; j < kD; ++j) {
      offb[j] = cur.valid() ? cur.offset32(rs, VEC) : kSkip;
      xb[j] = buf_load<XT, VEC, NTX>(bx, offb[j] * (uint32_t)sizeof(XT));
      gb[j] = buf_load<CT, VEC, NT>(bg, offb[j] * (uint32_t)sizeof(CT));
      cur.next();
      // Keep the issue order.  The loop's waits are counts of the loads issued AFTER the chunk a step needs; left
      // alone, the scheduler issues chunk 0 among the last and the first step of every trip waits for all but the
      // newest three loads.  (Costs 17 registers, one wave per SIMD, and is still 1-5 % faster on every box:
      // profiles/r02_backward_variants.txt.)
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int32_t base = 0; base < steps; base += kS) {
#pragma unroll
      for (int j = 0; j < kS; ++j) {
        if (base + j >= steps) break;  // wave-uniform
        const int f = (j + kD) % kS;   // the set the previous step worked on
        offb[f] = cur.valid() ? cur.offset32(rs, VEC) : kSkip;
        xb[f] = buf_load<XT, VEC, NTX>(bx, offb[f] * (uint32_t)sizeof(XT));
        gb[f] = buf_load<CT, VEC, NT>(bg, offb[f] * (uint32_t)sizeof(CT));
        cur.next();
        chunk(xb[j], gb[j], offb[j]);
      }
    }
  }
  if constexpr (kTieMode<MODE>) {
    // Rare: a handful of elements per channel attain the maximum.  The hot loop only tracked this lane's
    // largest key; a lane that saw the statistic walks its chunks once more (cold code, out of the hot
    // loop's register budget) and records the positions.
    unsigned long long first = ~0ull;  // this lane's first position attaining the statistic
    if (umax >= stat_bits) {
      ChunkCursor c2;
      c2.init(u, VEC, lane);
      while (c2.valid()) {
        const vec_t<XT, VEC> xr = load_vec<XT, VEC>(xp + c2.offset(u.row_stride, VEC));
        const int64_t pos = u.pos0 + c2.pos(a.t.row_len, VEC);
        for (int k = 0; k < VEC; ++k)
          if (pre_abs_bits<XT, PRE>(xr.v[k]) == stat_bits) {
            if (a.pos_part) {
              const unsigned long long p = (unsigned long long)(pos + k);
              first = p < first ? p : first;
            } else {
              record_tie(a.tie_info, per_channel, u.channel, (unsigned long long)(pos + k));
            }
          }
        c2.next();
      }
    }
    tie_first = first;
  }
  // ragged ends: the (< VEC) elements after the last full chunk of every row of the unit
  const int32_t tail = (int32_t)(u.len - (int64_t)cur.cpr * VEC);
  for (int32_t e = lane; e < u.nrows * tail; e += kWave) {
    const int32_t tr = e / tail, tk = e - tr * tail;
    const int64_t in_row = (int64_t)cur.cpr * VEC + tk;
    const int64_t i = (int64_t)tr * u.row_stride + in_row;
    const float xraw = to_f<XT>(xp[i]);
    float d = bwd_elem<CT, RM, MODE, ZP0>(PRE ? relu_f(xraw) : xraw, to_f<CT>(gp[i]), div, s, z, qmin, qmax,
                                          clamp_ste, mode, ds_acc, dzp_acc, dq_acc);
    if constexpr (PRE) d = xraw > 0.f ? d : 0.f;
    if constexpr (MODE == kBwdDsArrive)
      store_through<XT>(dxp + i, from_f<XT>(d));
    else
      dxp[i] = from_f<XT>(d);
    if constexpr (kTieMode<MODE>) {
      if (pre_abs_bits<XT, PRE>(xp[i]) == stat_bits) {
        const unsigned long long p = (unsigned long long)(u.pos0 + (int64_t)tr * a.t.row_len + in_row);
        if (a.pos_part) {
          tie_first = p < tie_first ? p : tie_first;
        } else {
          record_tie(a.tie_info, per_channel, u.channel, p);
        }
      }
    }
  }
  if constexpr (MODE == kBwdDsArrive) {
    ds_acc += ds_acc2.x + ds_acc2.y;
    ds_acc = wave_sum(ds_acc);
    tie_first = wave_min_u64(tie_first);
    bwd_arrive<XT, PRE>(a, u, ds_acc, tie_first, lane);
    return;
  }
  if constexpr (kTieMode<MODE>) {
    if (a.pos_part) {  // no atomics, nothing to initialise: the finishing kernel takes the minimum over units
      tie_first = wave_min_u64(tie_first);
      if (lane == 0) a.pos_part[u.id] = tie_first;
    }
  }
  if constexpr (MODE >= kBwdDs) {
    ds_acc += ds_acc2.x + ds_acc2.y;
    dzp_acc += dzp_acc2.x + dzp_acc2.y;
    ds_acc = wave_sum(ds_acc);
    if (lane == 0) a.ds_part[u.id] = ds_acc;
    if constexpr (MODE == kBwdDsDzp || MODE == kBwdDsBounds) {
      dzp_acc = wave_sum(dzp_acc);
      if (lane == 0) a.dzp_part[u.id] = dzp_acc;
    }
    if constexpr (MODE == kBwdDsBounds) {
      dq_acc += dq_acc2.x + dq_acc2.y;
      dq_acc = wave_sum(dq_acc);
      if (lane == 0) a.dq_part[u.id] = dq_acc;
    }
  }
}

// (96 scalar registers: one more would cost a resident workgroup per CU -- MI355X_MICROARCH.md, Residency)
template <typename XT, typename CT, int VEC, int RM, int MODE, bool NT, bool NTX = NT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(96), amdgpu_waves_per_eu(BVQ_BWD_WAVES, 8))) void fakequant_bwd_kernel(QuantArgs a) {
  const Unit u = locate_unit(a.t);
  if (!u.valid) return;
  float s, z;
  load_scale_zp<CT>(a, u.channel, s, z);
  const float qmin = rnd<CT>(a.bounds ? a.bounds[0] : a.qmin), qmax = rnd<CT>(a.bounds ? a.bounds[1] : a.qmax);
  const bool zp0 = sizeof(CT) == 2 && zp_is_pos_zero(z);
#define BVQ_BWD_UNIT(ZP0, PRE, DIV) bwd_unit<XT, CT, VEC, RM, MODE, NT, ZP0, PRE, NTX>(a, u, DIV, s, z, qmin, qmax)
#define BVQ_BWD_PRE(ZP0, DIV)      \
  do {                             \
    if (a.pre_relu)                \
      BVQ_BWD_UNIT(ZP0, true, DIV); \
    else                           \
      BVQ_BWD_UNIT(ZP0, false, DIV); \
  } while (0)
  if constexpr (elem<CT>::id == BVQ_BF16) {
    if (bf16_scale_ok(s)) {
      const DivBf16 div{1.0f / s};
      if (zp0)
        BVQ_BWD_PRE(true, div);
      else
        BVQ_BWD_PRE(false, div);
      return;
    }
  }
  if constexpr (elem<CT>::id == BVQ_F16) {
    if (f16_scale_ok(s)) {
#if BVQ_F16_BWD_DIV
      const DivF16R div{s, 1.0f / s};
#else
      const DivF16 div{s, 1.0f / s};
#endif
      if (zp0)
        BVQ_BWD_PRE(true, div);
      else
        BVQ_BWD_PRE(false, div);
      return;
    }
  }
  const DivExact div{s};
  if constexpr (sizeof(CT) == 2) {
    if (zp0) {
      BVQ_BWD_PRE(true, div);
      return;
    }
  }
  BVQ_BWD_PRE(false, div);
#undef BVQ_BWD_PRE
#undef BVQ_BWD_UNIT
}


#ifndef BVQ_COLS_BWD_UNROLL
#define BVQ_COLS_BWD_UNROLL 2  // rows in flight per lane
#endif
#ifndef BVQ_COLS_BWD_WAVES
#define BVQ_COLS_BWD_WAVES 4  // occupancy floor handed to the register allocator
#endif

template <typename T, int RM, bool NT, bool ZP0, bool FAST>
__device__ __forceinline__ void cols_bwd_rows(const ColsQuantArgs& a, const ColsLane<T>& ln, float qmin, float qmax) {
  constexpr int VEC = elem<T>::vec;
  constexpr int kU = BVQ_COLS_BWD_UNROLL;
  constexpr bool kSame16 = sizeof(T) == 2;
  const T* __restrict__ xp = reinterpret_cast<const T*>(a.x) + (int64_t)ln.chunk * VEC;
  const T* __restrict__ gp = reinterpret_cast<const T*>(a.g) + (int64_t)ln.chunk * VEC;
  T* __restrict__ dxp = reinterpret_cast<T*>(a.y) + (int64_t)ln.chunk * VEC;
  f2 r2[VEC / 2], ds2[VEC / 2], dz_unused = splat2(0.f);
  // abs-max tie search: per column, the largest |x| key and the first row that showed it.
  // 16-bit types: one 32-bit word per column, key << 16 | (0xffff - row counter), so that a single unsigned
  // max keeps both (ColsPlan bounds a lane's rows per unit by 65535).  float32: strictly-greater updates of
  // (key, row).  A column whose key equals its channel's statistic reports that row.
  typedef short i16x2 __attribute__((ext_vector_type(2)));
  uint32_t um[VEC], first[kSame16 ? 1 : VEC];
#pragma unroll
  for (int k = 0; k < VEC / 2; ++k) {
    r2[k] = f2{1.0f / ln.s2[k].x, 1.0f / ln.s2[k].y};
    ds2[k] = splat2(0.f);
  }
#pragma unroll
  for (int k = 0; k < VEC; ++k) um[k] = 0u;
#pragma unroll
  for (int k = 0; k < (kSame16 ? 1 : VEC); ++k)
    first[k] = ln.row0 < ln.row_end ? (uint32_t)ln.row0 : ~0u;  // an all-zero column attains its 0 in the first row
  const bool ties = a.tie_stat != nullptr;
  const bool clamp_ste = a.clamp_ste != 0;
  const int mode = a.round_mode;
  // the work on one row of this lane's columns: rr = the row, cnt = how many rows this lane has seen before it
  const DotSel dsel = make_dot_sel();
  auto row_work = [&](const vec_t<T, VEC>& xr, const vec_t<T, VEC>& gr, int64_t rr, uint32_t cnt) {
    vec_t<T, VEC> dv;
#pragma unroll
    for (int k = 0; k < VEC; k += 2) {
      const f2 xraw = widen2<T>(xr.v[k], xr.v[k + 1]);
      const f2 xin = a.pre_relu ? relu2(xraw) : xraw;
      const f2 gf = widen2<T>(gr.v[k], gr.v[k + 1]);
      f2 d;
      if constexpr (FAST && elem<T>::id == BVQ_F16)
        d = bwd_elem2<T, RM, kBwdDs, ZP0, kSame16>(xin, gf, BVQ_DIVF16V{ln.s2[k / 2], r2[k / 2]}, ln.s2[k / 2], ln.z2[k / 2],
                                                   qmin, qmax, clamp_ste, mode, ds2[k / 2], dz_unused, dz_unused, dsel);
      else if constexpr (FAST)
        d = bwd_elem2<T, RM, kBwdDs, ZP0, kSame16>(xin, gf, DivBf16V{r2[k / 2]}, ln.s2[k / 2], ln.z2[k / 2], qmin, qmax,
                                                   clamp_ste, mode, ds2[k / 2], dz_unused, dz_unused, dsel);
      else
        d = bwd_elem2<T, RM, kBwdDs, ZP0, kSame16>(xin, gf, DivExactV{ln.s2[k / 2]}, ln.s2[k / 2], ln.z2[k / 2], qmin,
                                                   qmax, clamp_ste, mode, ds2[k / 2], dz_unused, dz_unused, dsel);
      if (a.pre_relu) d = xraw > splat2(0.f) ? d : splat2(0.f);
      pack2<T>(d, dv.v[k], dv.v[k + 1]);
    }
    store_vec<T, VEC, NT>(dxp + rr * a.p.L, dv);
    if (ties) {
      if constexpr (kSame16) {
        const vec_t<uint32_t, VEC / 2> w = __builtin_bit_cast(vec_t<uint32_t, VEC / 2>, xr);
        const uint32_t inv = 0xffffu - cnt;
#pragma unroll
        for (int k = 0; k < VEC / 2; ++k) {
          // relu: negative patterns (sign bit set) count as 0; otherwise the sign bits are masked below
          const uint32_t w2 = a.pre_relu ? __builtin_bit_cast(uint32_t, __builtin_elementwise_max(
                                               __builtin_bit_cast(i16x2, w.v[k]), i16x2{0, 0}))
                                         : w.v[k];
          const uint32_t klo = ((w2 << 16) & 0x7fff0000u) | inv, khi = (w2 & 0x7fff0000u) | inv;
          um[2 * k] = klo > um[2 * k] ? klo : um[2 * k];
          um[2 * k + 1] = khi > um[2 * k + 1] ? khi : um[2 * k + 1];
        }
      } else {
        const uint32_t rr32 = (uint32_t)rr;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const uint32_t b = a.pre_relu ? pre_abs_bits<T, true>(xr.v[k]) : pre_abs_bits<T, false>(xr.v[k]);
          const bool gt = b > um[k];
          um[k] = gt ? b : um[k];
          first[k] = gt ? rr32 : first[k];
        }
      }
    }
  };
  // (a software-pipelined walk like the row-mapped backward's was measured here: with this kernel's per-column state it
  //  spills at depth 4 and is within +-2 % of these batches at depth 2-3 with a lower occupancy floor, 10 % slower for
  //  float32: profiles/r02_column_mapped.txt)
  uint32_t it = 0;  // row counter of this lane (wave-uniform)
  for (int64_t r = ln.row0; r < ln.row_end; r += (int64_t)kU * a.p.rpp) {
    vec_t<T, VEC> xv[kU], gv[kU];
    bool ok[kU];
#pragma unroll
    for (int j = 0; j < kU; ++j) {
      const int64_t rr = r + (int64_t)j * a.p.rpp;
      ok[j] = rr < ln.row_end;
      const int64_t lo = (ok[j] ? rr : ln.row0) * a.p.L;
      xv[j] = load_vec<T, VEC, NT>(xp + lo);
      gv[j] = load_vec<T, VEC, NT>(gp + lo);
    }
#pragma unroll
    for (int j = 0; j < kU; ++j)
      if (ok[j]) row_work(xv[j], gv[j], r + (int64_t)j * a.p.rpp, it + (uint32_t)j);
    it += kU;
  }
  // this lane's partial row of the [prows][L] arrays
  const int64_t prow = (ln.row0 - ln.sub) / a.p.rb * a.p.rpp + ln.sub;
  const int64_t base = prow * a.p.L + (int64_t)ln.chunk * VEC;
  if (a.ds_part) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) a.ds_part[base + k] = (k & 1) ? ds2[k / 2].y : ds2[k / 2].x;
  }
  if (!ties) return;
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    const int64_t col = (int64_t)ln.chunk * VEC + k;
    const uint32_t sk = abs_bits<T>(reinterpret_cast<const T*>(a.tie_stat)[col / a.inner]);
    bool hit;
    unsigned long long row;
    if constexpr (kSame16) {
      // (low half 0: no row seen; an all-zero column records its first row, 0 | 0xffff > 0)
      const uint32_t k16 = um[k] >> 16;
      hit = (elem<T>::id == BVQ_BF16 ? (k16 << 16) : k16) == sk && (um[k] & 0xffffu) != 0u;
      row = (unsigned long long)ln.row0 + (unsigned long long)(0xffffu - (um[k] & 0xffffu)) * (unsigned long long)a.p.rpp;
    } else {
      hit = um[k] == sk && first[k] != ~0u;
      row = first[k];
    }
    const unsigned long long pos = hit ? row * (unsigned long long)a.inner + (unsigned long long)(col % a.inner) : ~0ull;
    if (a.pos_part) a.pos_part[base + k] = pos;
    else if (pos != ~0ull) atomicMin(&a.tie_info[col / a.inner], pos);
  }
}

template <typename T, int RM, bool NT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(BVQ_COLS_BWD_WAVES, 8))) void fakequant_bwd_cols_kernel(
    ColsQuantArgs a) {
  ColsLane<T> ln;
  if (!ln.init(a) || !ln.active) return;
  const float qmin = rnd<T>(a.qmin), qmax = rnd<T>(a.qmax);
  if constexpr (sizeof(T) == 2) {
    if (ln.fast) {
      if (ln.zp0)
        cols_bwd_rows<T, RM, NT, true, true>(a, ln, qmin, qmax);
      else
        cols_bwd_rows<T, RM, NT, false, true>(a, ln, qmin, qmax);
      return;
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (ln.zp0) {
      cols_bwd_rows<T, RM, NT, true, false>(a, ln, qmin, qmax);
      return;
    }
  }
  cols_bwd_rows<T, RM, NT, false, false>(a, ln, qmin, qmax);
}

// Finish of the stats-scaled backward in ONE launch (per-channel layouts): per channel, sum the units'
// dscale partials (double, fixed order), take the first position attaining the statistic, turn dscale into
// the statistic's gradient (the backward of scale = clamp_min_ste(stat) / int_threshold, same rounding
// points as gstat_value) and deposit it on that element of dx.  Replaces tie_init + channel_sum + tie_apply.
template <typename T>
__global__ __launch_bounds__(kBlock) void bwd_stats_finish_kernel(const float* __restrict__ ds_part,
                                                                  const unsigned long long* __restrict__ pos_part,
                                                                  float* __restrict__ dscale, GstatSrc gs,
                                                                  const void* x, void* dx, int64_t nob,
                                                                  int32_t channels, int64_t ppr, int64_t inner) {
  __shared__ double sh[kBlock];
  __shared__ unsigned long long shp[kBlock];
  const int32_t c = blockIdx.x;
  const int64_t n = nob * ppr;
  double acc = 0.0;
  unsigned long long pmin = ~0ull;
  for (int64_t k = threadIdx.x; k < n; k += kBlock) {
    int64_t unit;
    if (nob == 1) {
      unit = (int64_t)c * ppr + k;
    } else {
      const int64_t o = k / ppr, p = k - o * ppr;
      unit = (o * channels + c) * ppr + p;
    }
    acc += (double)ds_part[unit];
    const unsigned long long q = pos_part[unit];
    pmin = q < pmin ? q : pmin;
  }
  sh[threadIdx.x] = acc;
  shp[threadIdx.x] = pmin;
  __syncthreads();
  for (int st = kBlock / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      sh[threadIdx.x] += sh[threadIdx.x + st];
      const unsigned long long o = shp[threadIdx.x + st];
      if (o < shp[threadIdx.x]) shp[threadIdx.x] = o;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float ds = (float)sh[0];
    dscale[c] = ds;
    const unsigned long long pos = shp[0];
    if (pos != ~0ull) {  // ~0: no element equals the statistic (e.g. NaN)
      float v = round_rt(ds, gs.scale_dtype);
      v = round_rt(v / gs.int_threshold, gs.quot_dtype);
      const float g = rnd<T>(v);
      const int64_t o = (int64_t)(pos / (unsigned long long)inner);
      const int64_t i = (int64_t)(pos - (unsigned long long)o * inner);
      const int64_t flat = (o * channels + c) * inner + i;
      const T* xp = reinterpret_cast<const T*>(x);
      T* dp = reinterpret_cast<T*>(dx);
      const float term = deposit<T, BVQ_MATCH_ABS>(g, xp[flat], gs.pre_relu != 0);
      dp[flat] = from_f<T>(to_f<T>(dp[flat]) + term);
    }
  }
}

// channel_finish as its own launch: one wave per channel (the routes whose streaming kernel does not finish its
// channels itself: column-mapped layouts, callers without an arrival buffer)
template <typename XT>
__global__ __launch_bounds__(kBlock) void channel_finish_kernel(QuantArgs a) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int32_t c = (int32_t)blockIdx.x * kWavesPerBlock + wave;
  if (c >= a.t.channels) return;
  if (a.pre_relu)
    channel_finish<XT, true>(a, c, threadIdx.x & 63);
  else
    channel_finish<XT, false>(a, c, threadIdx.x & 63);
}

// Batch-sharded tensors, after the all-gather of the shards' messages (float64 [world][2][channels]): per channel the
// dscale sums of all shards added in rank order (double, rounded to float32 ONCE: the same bits on every rank), the
// deposit's owner = the lowest rank that holds an arg-max, and -- on the owner -- dscale -> statistic's gradient and
// its deposit at this shard's first arg-max position.  Replaces unpack + cast + divide + cast + deposit launches.
template <typename T>
__global__ void shard_unpack_deposit_kernel(const double* __restrict__ all, int32_t world, int32_t channels, int32_t rank,
                                            const long long* __restrict__ first_pos, const void* x, void* dx,
                                            int64_t inner, GstatSrc gs, float* __restrict__ dscale_total) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= channels) return;
  double sum = 0.0, owner = kShardNoOwner;
  for (int r = 0; r < world; ++r) {
    sum += all[((int64_t)r * 2) * channels + c];
    const double o = all[((int64_t)r * 2 + 1) * channels + c];
    owner = o < owner ? o : owner;
  }
  const float ds = (float)sum;
  if (dscale_total) dscale_total[c] = ds;
  const long long pos = first_pos[c];
  if (owner != (double)rank || pos < 0) return;
  float v = round_rt(ds, gs.scale_dtype);
  v = round_rt(v / gs.int_threshold, gs.quot_dtype);
  const float g = rnd<T>(v);
  const int64_t o = (int64_t)(pos / inner);
  const int64_t i = (int64_t)(pos - o * inner);
  const int64_t flat = (o * channels + c) * inner + i;
  const T* xp = reinterpret_cast<const T*>(x);
  T* dp = reinterpret_cast<T*>(dx);
  const float term = deposit<T, BVQ_MATCH_ABS>(g, xp[flat], gs.pre_relu != 0);
  dp[flat] = from_f<T>(to_f<T>(dp[flat]) + term);
}

#endif  // backward part

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int validate(const bvq_quant_desc* d) {
  if (!d) {
    set_error("null descriptor");
    return BVQ_ERR_INVALID;
  }
  if (d->outer < 0 || d->channels < 1 || d->inner < 0) {
    set_error("bad shape [%lld,%lld,%lld]", (long long)d->outer, (long long)d->channels,
              (long long)d->inner);
    return BVQ_ERR_INVALID;
  }
  if (d->pre_op != BVQ_PRE_NONE && d->pre_op != BVQ_PRE_RELU) {
    set_error("bad pre_op %d", d->pre_op);
    return BVQ_ERR_INVALID;
  }
  if (d->codes_dtype < BVQ_CODES_I32 || d->codes_dtype > BVQ_CODES_U8) {
    set_error("bad codes_dtype %d", d->codes_dtype);
    return BVQ_ERR_INVALID;
  }
  if (d->round_mode < BVQ_ROUND || d->round_mode > BVQ_DPU_ROUND) {
    set_error("bad round_mode %d", d->round_mode);
    return BVQ_ERR_INVALID;
  }
  const bool ok = (d->x_dtype == d->ct_dtype && d->x_dtype >= BVQ_F32 && d->x_dtype <= BVQ_F16) ||
                  (d->ct_dtype == BVQ_F32 && (d->x_dtype == BVQ_BF16 || d->x_dtype == BVQ_F16));
  if (!ok) {
    set_error("unsupported dtype pair x=%d ct=%d", d->x_dtype, d->ct_dtype);
    return BVQ_ERR_UNSUPPORTED;
  }
  if (d->scale_dtype < BVQ_F32 || d->scale_dtype > BVQ_F16 || d->zp_dtype < BVQ_F32 ||
      d->zp_dtype > BVQ_F16) {
    set_error("bad scale/zp dtype");
    return BVQ_ERR_INVALID;
  }
  return BVQ_OK;
}

// [outer, channels, row_len] of the descriptor: per-tensor quantizers are one long row
static void rows_of(const bvq_quant_desc* d, int64_t& outer, int64_t& row_len, int32_t& channels) {
  const bool pc = (d->scale_per_channel || d->zp_per_channel) && d->channels > 1;
  if (pc) {
    outer = d->outer;
    row_len = d->inner;
    channels = (int32_t)d->channels;
  } else {
    outer = 1;
    row_len = d->outer * d->channels * d->inner;
    channels = 1;
  }
}

static void fill_args(QuantArgs& a, const bvq_quant_desc* d) {
  a.qmin = d->qmin;
  a.qmax = d->qmax;
  a.scale_dtype = d->scale_dtype;
  a.zp_dtype = d->zp_dtype;
  a.scale_pc = (d->scale_per_channel && d->channels > 1) ? 1 : 0;
  a.zp_pc = (d->zp_per_channel && d->channels > 1) ? 1 : 0;
  a.scalar_cast = d->scalar_mode == BVQ_SCALAR_CAST;
  a.clamp_ste = d->clamp_ste;
  a.out_int = d->out_kind == BVQ_OUT_INT;
  a.round_mode = d->round_mode;
  a.pre_relu = d->pre_op == BVQ_PRE_RELU;
  a.codes_dtype = d->codes_dtype;
}

// column-mapped route for this call? (channel axis last or nearly last; see ColsPlan)
static ColsPlan cols_quant_plan(const bvq_quant_desc* d, const void* p0, const void* p1, const void* p2,
                                bool no_partials = false) {
  ColsPlan none = {};
  if (!(d->scale_per_channel && d->channels > 1) || d->x_dtype != d->ct_dtype || d->out_kind != BVQ_OUT_DEQUANT)
    return none;
  if ((reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2)) & 15)
    return none;
  return cols_plan(d->x_dtype, d->outer, d->channels, d->inner, no_partials);
}

static void fill_cols_args(ColsQuantArgs& a, const ColsPlan& cp, const bvq_quant_desc* d) {
  a.p = cp;
  a.inner = d->inner;
  a.channels = (int32_t)d->channels;
  a.qmin = d->qmin;
  a.qmax = d->qmax;
  a.scale_dtype = d->scale_dtype;
  a.zp_dtype = d->zp_dtype;
  a.zp_pc = d->zp_per_channel ? 1 : 0;
  a.scalar_cast = d->scalar_mode == BVQ_SCALAR_CAST;
  a.clamp_ste = d->clamp_ste;
  a.round_mode = d->round_mode;
  a.pre_relu = d->pre_op == BVQ_PRE_RELU;
}

#define BVQ_COLS_LAUNCH(KERNEL, a, nt, st)                                                       \
  do {                                                                                           \
    const dim3 grid(grid_for_units((a).p.units)), block(kBlock);                                 \
    const bool rne = (a).round_mode == BVQ_ROUND;                                                \
    if (d->x_dtype == BVQ_F32) {                                                                 \
      if (rne && nt) KERNEL<float, BVQ_ROUND, true><<<grid, block, 0, st>>>(a);                  \
      else if (rne) KERNEL<float, BVQ_ROUND, false><<<grid, block, 0, st>>>(a);                  \
      else KERNEL<float, kAnyRM, false><<<grid, block, 0, st>>>(a);                              \
    } else if (d->x_dtype == BVQ_BF16) {                                                         \
      if (rne && nt) KERNEL<bf16_t, BVQ_ROUND, true><<<grid, block, 0, st>>>(a);                 \
      else if (rne) KERNEL<bf16_t, BVQ_ROUND, false><<<grid, block, 0, st>>>(a);                 \
      else KERNEL<bf16_t, kAnyRM, false><<<grid, block, 0, st>>>(a);                             \
    } else {                                                                                     \
      if (rne && nt) KERNEL<f16_t, BVQ_ROUND, true><<<grid, block, 0, st>>>(a);                  \
      else if (rne) KERNEL<f16_t, BVQ_ROUND, false><<<grid, block, 0, st>>>(a);                  \
      else KERNEL<f16_t, kAnyRM, false><<<grid, block, 0, st>>>(a);                              \
    }                                                                                            \
  } while (0)

// instantiated vector widths: 16 bytes of x per lane, or one element (ragged / misaligned rows)
static int snap_vec(int vec, int full) { return vec == full ? full : 1; }

#if BVQ_PART == 0 || BVQ_PART == 1
template <typename XT, typename CT>
static void launch_fwd(const QuantArgs& a, int vec, bool nt, hipStream_t st) {
  constexpr int V = elem<XT>::vec;
  const dim3 grid(grid_for_units(a.t.units)), block(kBlock);
  const bool rne = a.round_mode == BVQ_ROUND;
#ifdef BVQ_CACHE_EXPERIMENT
  // developer build: cache policy of the x loads (BVQ_X_FWD_NTL) and the y stores (BVQ_X_FWD_NTS) chosen per call
  if (vec == V && rne && getenv("BVQ_X_FWD_NTL")) {
    const int l = atoi(getenv("BVQ_X_FWD_NTL")), w = getenv("BVQ_X_FWD_NTS") ? atoi(getenv("BVQ_X_FWD_NTS")) : 1;
    if (w && l) fakequant_fwd_kernel<XT, CT, V, BVQ_ROUND, true, true><<<grid, block, 0, st>>>(a);
    else if (w) fakequant_fwd_kernel<XT, CT, V, BVQ_ROUND, true, false><<<grid, block, 0, st>>>(a);
    else if (l) fakequant_fwd_kernel<XT, CT, V, BVQ_ROUND, false, true><<<grid, block, 0, st>>>(a);
    else fakequant_fwd_kernel<XT, CT, V, BVQ_ROUND, false, false><<<grid, block, 0, st>>>(a);
    return;
  }
#endif
  if (vec == V) {
    if (rne && nt)
      fakequant_fwd_kernel<XT, CT, V, BVQ_ROUND, true><<<grid, block, 0, st>>>(a);
    else if (rne)
      fakequant_fwd_kernel<XT, CT, V, BVQ_ROUND, false><<<grid, block, 0, st>>>(a);
    else if (nt)
      fakequant_fwd_kernel<XT, CT, V, kAnyRM, true><<<grid, block, 0, st>>>(a);
    else
      fakequant_fwd_kernel<XT, CT, V, kAnyRM, false><<<grid, block, 0, st>>>(a);
  } else {
    if (rne)
      fakequant_fwd_kernel<XT, CT, 1, BVQ_ROUND, false><<<grid, block, 0, st>>>(a);
    else
      fakequant_fwd_kernel<XT, CT, 1, kAnyRM, false><<<grid, block, 0, st>>>(a);
  }
}

#endif

#if BVQ_BWD_CODE
template <typename XT, typename CT, int MODE>
static void launch_bwd_mode(const QuantArgs& a, int vec, bool nt, hipStream_t st) {
  constexpr int V = elem<XT>::vec;
  const dim3 grid(grid_for_units(a.t.units)), block(kBlock);
  const bool rne = a.round_mode == BVQ_ROUND;
#ifdef BVQ_CACHE_EXPERIMENT
  // developer build: cache policy of the x loads (BVQ_X_BWD_NTX) and of the g loads / dx stores (BVQ_X_BWD_NT)
  if (vec == V && rne && getenv("BVQ_X_BWD_NTX")) {
    const int xl = atoi(getenv("BVQ_X_BWD_NTX")), o = getenv("BVQ_X_BWD_NT") ? atoi(getenv("BVQ_X_BWD_NT")) : 1;
    if (o && xl) fakequant_bwd_kernel<XT, CT, V, BVQ_ROUND, MODE, true, true><<<grid, block, 0, st>>>(a);
    else if (o) fakequant_bwd_kernel<XT, CT, V, BVQ_ROUND, MODE, true, false><<<grid, block, 0, st>>>(a);
    else if (xl) fakequant_bwd_kernel<XT, CT, V, BVQ_ROUND, MODE, false, true><<<grid, block, 0, st>>>(a);
    else fakequant_bwd_kernel<XT, CT, V, BVQ_ROUND, MODE, false, false><<<grid, block, 0, st>>>(a);
    return;
  }
#endif
  if (vec == V) {
    if (rne && nt)
      fakequant_bwd_kernel<XT, CT, V, BVQ_ROUND, MODE, true><<<grid, block, 0, st>>>(a);
    else if (rne)
      fakequant_bwd_kernel<XT, CT, V, BVQ_ROUND, MODE, false><<<grid, block, 0, st>>>(a);
    else if (nt)
      fakequant_bwd_kernel<XT, CT, V, kAnyRM, MODE, true><<<grid, block, 0, st>>>(a);
    else
      fakequant_bwd_kernel<XT, CT, V, kAnyRM, MODE, false><<<grid, block, 0, st>>>(a);
  } else {
    if (rne)
      fakequant_bwd_kernel<XT, CT, 1, BVQ_ROUND, MODE, false><<<grid, block, 0, st>>>(a);
    else
      fakequant_bwd_kernel<XT, CT, 1, kAnyRM, MODE, false><<<grid, block, 0, st>>>(a);
  }
}

template <typename XT, typename CT>
void launch_bwd(const QuantArgs& a, int vec, int mode, bool nt, hipStream_t st) {
  switch (mode) {
    case kBwdDx:
      launch_bwd_mode<XT, CT, kBwdDx>(a, vec, nt, st);
      break;
    case kBwdDs:
      launch_bwd_mode<XT, CT, kBwdDs>(a, vec, nt, st);
      break;
    case kBwdDsDzp:
      launch_bwd_mode<XT, CT, kBwdDsDzp>(a, vec, nt, st);
      break;
    case kBwdDsBounds:
      launch_bwd_mode<XT, CT, kBwdDsBounds>(a, vec, nt, st);
      break;
    case kBwdDsArrive:
      launch_bwd_mode<XT, CT, kBwdDsArrive>(a, vec, nt, st);
      break;
    default:
      launch_bwd_mode<XT, CT, kBwdDsTies>(a, vec, nt, st);
      break;
  }
}


// which translation unit instantiates the row-mapped backward of which dtype pair (see the top of the file)
#define BVQ_LAUNCH_BWD(XT, CT) void launch_bwd<XT, CT>(const QuantArgs&, int, int, bool, hipStream_t)
#if BVQ_PART == 2
extern template BVQ_LAUNCH_BWD(float, float);
extern template BVQ_LAUNCH_BWD(bf16_t, bf16_t);
extern template BVQ_LAUNCH_BWD(bf16_t, float);
extern template BVQ_LAUNCH_BWD(f16_t, f16_t);
extern template BVQ_LAUNCH_BWD(f16_t, float);
#elif BVQ_PART == 21
template BVQ_LAUNCH_BWD(bf16_t, bf16_t);
#elif BVQ_PART == 22
template BVQ_LAUNCH_BWD(f16_t, f16_t);
#elif BVQ_PART == 23
template BVQ_LAUNCH_BWD(float, float);
template BVQ_LAUNCH_BWD(bf16_t, float);
template BVQ_LAUNCH_BWD(f16_t, float);
#endif
#undef BVQ_LAUNCH_BWD

#endif

#define BVQ_DISPATCH_PAIR(d, CALL)                                      \
  do {                                                                  \
    if ((d)->x_dtype == BVQ_F32) {                                      \
      CALL(float, float);                                               \
    } else if ((d)->x_dtype == BVQ_BF16 && (d)->ct_dtype == BVQ_BF16) { \
      CALL(bf16_t, bf16_t);                                             \
    } else if ((d)->x_dtype == BVQ_BF16) {                              \
      CALL(bf16_t, float);                                              \
    } else if ((d)->x_dtype == BVQ_F16 && (d)->ct_dtype == BVQ_F16) {   \
      CALL(f16_t, f16_t);                                               \
    } else {                                                            \
      CALL(f16_t, float);                                               \
    }                                                                   \
  } while (0)

}  // namespace bvq

using namespace bvq;

#if BVQ_PART == 0 || BVQ_PART == 1
static int fakequant_fwd_impl(const bvq_quant_desc* d, const void* x, const void* scale, const void* zp, void* y,
                             void* codes, const float* bounds, bvq_stream_t stream) {
  int rc = validate(d);
  if (rc) return rc;
  const int64_t n = d->outer * d->channels * d->inner;
  if (n == 0) return BVQ_OK;
  if (!x || !scale || !zp || (!y && !codes)) {
    set_error("bvq_fakequant_fwd: null pointer");
    return BVQ_ERR_INVALID;
  }
  if (y && !codes && !bounds) {
    const ColsPlan cp = cols_quant_plan(d, x, y, nullptr, true);
    if (cp.ok) {
      ColsQuantArgs ca = {};
      fill_cols_args(ca, cp, d);
      ca.x = x;
      ca.y = y;
      ca.scale = scale;
      ca.zp = zp;
      hipStream_t cst = (hipStream_t)stream;
      const bool cnt = n * (int64_t)(2 * dtype_size(d->x_dtype)) >= nt_threshold_bytes();
      BVQ_COLS_LAUNCH(fakequant_fwd_cols_kernel, ca, cnt, cst);
      return check_launch("bvq_fakequant_fwd/cols");
    }
  }
  int64_t outer, row_len;
  int32_t channels;
  rows_of(d, outer, row_len, channels);
  const void* ptrs[3] = {x, y, codes};
  const int els[3] = {dtype_size(d->x_dtype), dtype_size(d->ct_dtype), d->codes_dtype == BVQ_CODES_I32 ? 4 : 1};
  const int full = 16 / dtype_size(d->x_dtype);
  const int vec = snap_vec(pick_vec(full, outer * channels, row_len, ptrs, els, 3, true), full);
  QuantArgs a = {};
  a.t = make_tiling(outer, channels, row_len, vec, 0, true);
#ifdef BVQ_CACHE_EXPERIMENT
  if (getenv("BVQ_X_FWD_REV")) a.t.reverse = atoi(getenv("BVQ_X_FWD_REV"));
#endif
  a.x = x;
  a.scale = scale;
  a.zp = zp;
  a.y = y;
  a.codes = codes;
  a.bounds = bounds;
  fill_args(a, d);
  hipStream_t st = (hipStream_t)stream;
  const bool nt = n * (int64_t)(dtype_size(d->x_dtype) + dtype_size(d->ct_dtype)) >= nt_threshold_bytes();
#define BVQ_CALL(XT, CT) launch_fwd<XT, CT>(a, vec, nt, st)
  BVQ_DISPATCH_PAIR(d, BVQ_CALL);
#undef BVQ_CALL
  return check_launch("bvq_fakequant_fwd");
}

extern "C" int bvq_fakequant_fwd(const bvq_quant_desc* d, const void* x, const void* scale,
                                 const void* zp, void* y, void* codes, bvq_stream_t stream) {
  return fakequant_fwd_impl(d, x, scale, zp, y, codes, nullptr, stream);
}

extern "C" int bvq_fakequant_fwd_bounds(const bvq_quant_desc* d, const void* x, const void* scale, const void* zp,
                                        const float* bounds, void* y, bvq_stream_t stream) {
  if (!bounds) {
    set_error("bvq_fakequant_fwd_bounds: null bounds");
    return BVQ_ERR_INVALID;
  }
  return fakequant_fwd_impl(d, x, scale, zp, y, nullptr, bounds, stream);
}


// ---- statistic + quantizer in one launch ------------------------------------------------------------
struct FusedPlan {
  int32_t cpr, spr, slices, waves, nblocks;
};

static int num_cus() {
  static int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    return v;
  }();
  return n;
}

struct FusedShape {
  bool ok;
  int64_t outer, channels, inner;
  int vec;
};

// what both one-launch forms need: x and y of one dtype, dequantized output, whole 16-byte chunks
static FusedShape fused_shape(const bvq_quant_desc* d, const void* x, const void* y) {
  FusedShape f = {};
  static const int enabled = env_flag("BVQ_FUSED_FWD", 1);
  if (!enabled) return f;
  if (d->x_dtype != d->ct_dtype || d->out_kind != BVQ_OUT_DEQUANT) return f;
  if (d->zp_per_channel) return f;
  const bool pc = d->scale_per_channel && d->channels > 1;
  f.outer = pc ? d->outer : 1;
  f.channels = pc ? d->channels : 1;
  f.inner = pc ? d->inner : d->outer * d->channels * d->inner;
  f.vec = 16 / dtype_size(d->x_dtype);
  if (f.inner <= 0 || f.outer <= 0 || f.inner % f.vec != 0) return f;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return f;
  f.ok = true;
  return f;
}

// the register-resident form applies when a channel fits the registers of ONE workgroup
static bool fused_plan(const bvq_quant_desc* d, const void* x, const void* y, FusedPlan& p) {
  const FusedShape f = fused_shape(d, x, y);
  if (!f.ok) return false;
  const int64_t cpr = f.inner / f.vec;
  const int64_t spr = (cpr + kFusedSliceChunks - 1) / kFusedSliceChunks;
  const int64_t slices = f.outer * spr;
  if (cpr > (1 << 30) || slices > kFusedMaxWaves) return false;
  p.cpr = (int32_t)cpr;
  p.spr = (int32_t)spr;
  p.slices = (int32_t)slices;
  p.waves = (int)slices;
  // residency budget: 2 workgroups of 512 threads per CU (or the same number of waves in smaller ones)
  const int64_t budget = (int64_t)num_cus() * 2 * kFusedMaxWaves / p.waves;
  p.nblocks = (int32_t)(budget < f.channels ? budget : f.channels);
  return true;
}

extern "C" int64_t bvq_stats_fakequant_fwd_workspace_bytes(const bvq_quant_desc* d, const void* x, const void* y) {
  if (validate(d)) return -1;
  FusedPlan p;
  if (fused_plan(d, x, y, p)) return 16;  // no workspace needed; non-zero says "covered"
  return 0;  // not applicable: use bvq_absmax_scale + bvq_fakequant_fwd
}

extern "C" int bvq_stats_fakequant_fwd(const bvq_quant_desc* d, const void* x, double min_val, int use_min,
                                       double int_threshold, void* stat_out, void* scale_out, void* y,
                                       void* workspace, int64_t workspace_bytes, bvq_stream_t stream) {
  int rc = validate(d);
  if (rc) return rc;
  if (!x || !y || !stat_out || !scale_out || !workspace) {
    set_error("bvq_stats_fakequant_fwd: null pointer");
    return BVQ_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  const bool pc = d->scale_per_channel && d->channels > 1;
  const int64_t channels = pc ? d->channels : 1;
  const bool rne = d->round_mode == BVQ_ROUND;
  FusedPlan p;
  if (fused_plan(d, x, y, p)) {
    FusedArgs a = {};
    a.x = x;
    a.y = y;
    a.stat_out = stat_out;
    a.scale_out = scale_out;
    a.outer = pc ? d->outer : 1;
    a.inner = pc ? d->inner : d->outer * d->channels * d->inner;
    a.channels = (int32_t)channels;
    a.cpr = p.cpr;
    a.spr = p.spr;
    a.slices = p.slices;
    a.qmin = d->qmin;
    a.qmax = d->qmax;
    a.min_val = round_host((float)min_val, d->x_dtype);  // python scalar -> the statistic's dtype
    a.use_min = use_min;
    a.int_threshold = (float)int_threshold;
    a.scale_dtype = d->scale_dtype;
    a.scale_pc = pc ? 1 : 0;
    a.scalar_cast = d->scalar_mode == BVQ_SCALAR_CAST;
    a.round_mode = d->round_mode;
    a.pre_relu = d->pre_op == BVQ_PRE_RELU;
    const dim3 grid((unsigned)p.nblocks), block((unsigned)(p.waves * kWave));
#define BVQ_FUSED(T)                                                        \
  do {                                                                      \
    if (rne)                                                                \
      fused_absmax_fakequant_kernel<T, BVQ_ROUND><<<grid, block, 0, st>>>(a); \
    else                                                                    \
      fused_absmax_fakequant_kernel<T, kAnyRM><<<grid, block, 0, st>>>(a);   \
  } while (0)
    if (d->x_dtype == BVQ_F32)
      BVQ_FUSED(float);
    else if (d->x_dtype == BVQ_BF16)
      BVQ_FUSED(bf16_t);
    else
      BVQ_FUSED(f16_t);
#undef BVQ_FUSED
    return check_launch("bvq_stats_fakequant_fwd");
  }
  set_error("bvq_stats_fakequant_fwd: shape / layout not covered by the one-launch form");
  return BVQ_ERR_UNSUPPORTED;
}

// self-test of the float16 division (DivF16R): out[j * n_a + i] = the quotient the kernels compute for numerator
// a[i] and scale s[j], so that a test can compare EVERY pair with a / s on the device itself
__global__ __launch_bounds__(256) void selftest_div_f16r_kernel(const float* __restrict__ a, const float* __restrict__ sc,
                                                                float* __restrict__ out, int32_t n_a) {
  const float s = sc[blockIdx.y];
  const float r = 1.0f / s;
  for (int32_t i = blockIdx.x * 256 + threadIdx.x; i < n_a; i += gridDim.x * 256)
    out[(int64_t)blockIdx.y * n_a + i] = div_refined(a[i], s, r);
}

extern "C" int bvq_selftest_div_f16r(const float* a, int32_t n_a, const float* scales, int32_t n_s, float* out,
                                     bvq_stream_t stream) {
  if (!a || !scales || !out || n_a < 1 || n_s < 1 || n_s > 65535) {
    set_error("bvq_selftest_div_f16r: bad argument");
    return BVQ_ERR_INVALID;
  }
  int nb = (n_a + 255) / 256;
  if (nb > 1024) nb = 1024;
  selftest_div_f16r_kernel<<<dim3((unsigned)nb, (unsigned)n_s), dim3(256), 0, (hipStream_t)stream>>>(a, scales, out, n_a);
  return check_launch("bvq_selftest_div_f16r");
}

#endif  // forward part

#if BVQ_PART == 0 || BVQ_PART == 2
// the backward's decomposition: quantizer-style tiling with every unit addressable through 32-bit buffer offsets
// (4 = the widest element; the same bound for every dtype so that workspace sizing and launch agree)
static Tiling bwd_tiling(int64_t outer, int32_t channels, int64_t row_len, int vec) {
  Tiling t = make_tiling(outer, channels, row_len, vec, 0, true);
  cap_unit_extent(t, 4);
  return t;
}

static int64_t bwd_units(const bvq_quant_desc* d) {
  int64_t outer, row_len;
  int32_t channels;
  rows_of(d, outer, row_len, channels);
  // upper bound over the vector widths the launcher may pick
  const int full = 16 / dtype_size(d->x_dtype);
  const int64_t a = bwd_tiling(outer, channels, row_len, full).units;
  const int64_t b = bwd_tiling(outer, channels, row_len, 1).units;
  return a > b ? a : b;
}

extern "C" int64_t bvq_fakequant_bwd_workspace_bytes(const bvq_quant_desc* d) {
  if (validate(d)) return -1;
  int64_t outer, row_len;
  int32_t channels;
  rows_of(d, outer, row_len, channels);
  const int64_t units = bwd_units(d);
  const int64_t mid = channel_sums_mid_bytes(units / channels + 1, channels) + 16;
  int64_t bytes = 3 * units * (int64_t)sizeof(float) + mid + 256;  // (a third partial array: bvq_fakequant_bwd_bounds)
  const ColsPlan cp = cols_quant_plan(d, nullptr, nullptr, nullptr);
  if (cp.ok && (cp.prows + cols_fold_scratch_rows(cp.prows)) * cp.L * (int64_t)sizeof(float) + 256 > bytes)
    bytes = (cp.prows + cols_fold_scratch_rows(cp.prows)) * cp.L * (int64_t)sizeof(float) + 256;
  return bytes;
}

static int fakequant_bwd_impl(const bvq_quant_desc* d, const void* g, const void* x, const void* scale,
                             const void* zp, void* dx, float* dscale, float* dzp, const void* tie_stat,
                             int64_t* tie_info, void* workspace, int64_t workspace_bytes, bvq_stream_t stream,
                             const LearnedScaleEpilogue* epilogue, const float* bounds = nullptr,
                             float* dbounds = nullptr) {
  int rc = validate(d);
  if (rc) return rc;
  const int64_t n = d->outer * d->channels * d->inner;
  hipStream_t st = (hipStream_t)stream;
  int64_t outer, row_len;
  int32_t channels;
  rows_of(d, outer, row_len, channels);
  const bool need_sums = dscale != nullptr || dzp != nullptr;
  if ((tie_stat != nullptr) != (tie_info != nullptr)) {
    set_error("bvq_fakequant_bwd: tie_stat and tie_info go together");
    return BVQ_ERR_INVALID;
  }
  if (tie_stat && (!dscale || dzp)) {
    set_error("bvq_fakequant_bwd: the tie search rides on the dscale variant (dscale set, dzp null)");
    return BVQ_ERR_UNSUPPORTED;
  }
  if (tie_stat && channels != d->channels) {
    set_error("bvq_fakequant_bwd: tie search needs the statistic's layout (per-channel scale iff "
              "channels > 1)");
    return BVQ_ERR_UNSUPPORTED;
  }
  if (tie_info) launch_tie_init(reinterpret_cast<unsigned long long*>(tie_info), channels, st);
  if (n == 0) {
    if (dscale) (void)hipMemsetAsync(dscale, 0, sizeof(float) * channels, st);
    if (dzp) (void)hipMemsetAsync(dzp, 0, sizeof(float) * channels, st);
    return BVQ_OK;
  }
  if (!g || !x || !scale || !zp || !dx) {
    set_error("bvq_fakequant_bwd: null pointer");
    return BVQ_ERR_INVALID;
  }
  if (dbounds && (dzp || tie_stat || !dscale)) {
    set_error("bvq_fakequant_bwd: the bound gradients ride on the dscale variant (dscale set, dzp / tie_stat null)");
    return BVQ_ERR_UNSUPPORTED;
  }
  if (!dzp && !bounds) {
    const ColsPlan cp = cols_quant_plan(d, x, g, dx, !dscale && !tie_stat);
    if (cp.ok) {
      const int64_t need = dscale ? (cp.prows + cols_fold_scratch_rows(cp.prows)) * cp.L * (int64_t)sizeof(float) : 0;
      if (dscale && (!workspace || workspace_bytes < need)) {
        set_error("bvq_fakequant_bwd: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
        return BVQ_ERR_WORKSPACE;
      }
      ColsQuantArgs ca = {};
      fill_cols_args(ca, cp, d);
      ca.x = x;
      ca.g = g;
      ca.y = dx;
      ca.scale = scale;
      ca.zp = zp;
      ca.ds_part = dscale ? reinterpret_cast<float*>(workspace) : nullptr;
      ca.tie_stat = tie_stat;
      ca.tie_info = reinterpret_cast<unsigned long long*>(tie_info);
      const bool cnt = n * (int64_t)(3 * dtype_size(d->x_dtype)) >= nt_threshold_bytes();
      BVQ_COLS_LAUNCH(fakequant_bwd_cols_kernel, ca, cnt, st);
      rc = check_launch("bvq_fakequant_bwd/cols");
      if (rc) return rc;
      if (dscale) {
        float* folded = nullptr;
        launch_cols_fold_sum_min(ca.ds_part, nullptr, cp.prows, cp.L, ca.ds_part + cp.prows * cp.L, nullptr, &folded,
                                 nullptr, st);
        launch_channel_sums(folded, nullptr, dscale, nullptr, 1, channels, d->inner, nullptr, st, epilogue);
        rc = check_launch("bvq_fakequant_bwd/cols_sum");
      }
      return rc;
    }
  }
  const void* ptrs[3] = {x, g, dx};
  const int els[3] = {dtype_size(d->x_dtype), dtype_size(d->ct_dtype), dtype_size(d->x_dtype)};
  const int full = 16 / dtype_size(d->x_dtype);
  const int vec = snap_vec(pick_vec(full, outer * channels, row_len, ptrs, els, 3, true), full);
  QuantArgs a = {};
  a.t = bwd_tiling(outer, channels, row_len, vec);
#ifdef BVQ_CACHE_EXPERIMENT
  if (getenv("BVQ_X_BWD_REV")) a.t.reverse = atoi(getenv("BVQ_X_BWD_REV"));
#endif
  int64_t mid_off = 0;
  if (need_sums) {
    // float partials (8-byte aligned end), then the doubles of a split reduction
    mid_off = (((dbounds ? 3 : 2) * a.t.units * (int64_t)sizeof(float) + 7) / 8) * 8;
    const int64_t need = mid_off + channel_sums_mid_bytes(a.t.nob * a.t.ppr, channels);
    if (!workspace || workspace_bytes < need) {
      set_error("bvq_fakequant_bwd: workspace %lld < %lld bytes", (long long)workspace_bytes,
                (long long)need);
      return BVQ_ERR_WORKSPACE;
    }
    a.ds_part = reinterpret_cast<float*>(workspace);
    a.dzp_part = a.ds_part + a.t.units;
    a.dq_part = a.dzp_part + a.t.units;
  }
  a.x = x;
  a.g = g;
  a.scale = scale;
  a.zp = zp;
  a.y = dx;
  a.bounds = bounds;
  a.tie_stat = tie_stat;
  a.tie_info = reinterpret_cast<unsigned long long*>(tie_info);
  fill_args(a, d);
  const int mode = dbounds ? kBwdDsBounds : (tie_stat ? kBwdDsTies : (dzp ? kBwdDsDzp : (dscale ? kBwdDs : kBwdDx)));
  const bool nt =
      n * (int64_t)(2 * dtype_size(d->x_dtype) + dtype_size(d->ct_dtype)) >= nt_threshold_bytes();
#define BVQ_CALL(XT, CT) launch_bwd<XT, CT>(a, vec, mode, nt, st)
  BVQ_DISPATCH_PAIR(d, BVQ_CALL);
#undef BVQ_CALL
  rc = check_launch("bvq_fakequant_bwd");
  if (rc) return rc;
  if (need_sums) {
    // dbounds: [d(qmin) per channel | d(qmax) per channel] (the bounds themselves are scalars: the caller adds the
    // channels up)
    launch_channel_sums(dscale ? a.ds_part : nullptr, (dzp || dbounds) ? a.dzp_part : nullptr, dscale,
                        dbounds ? dbounds : dzp, a.t.nob, channels, a.t.ppr,
                        reinterpret_cast<char*>(workspace) + mid_off, st, epilogue);
    if (dbounds)
      launch_channel_sums(a.dq_part, nullptr, dbounds + channels, nullptr, a.t.nob, channels, a.t.ppr,
                          reinterpret_cast<char*>(workspace) + mid_off, st, nullptr);
    rc = check_launch("bvq_fakequant_bwd/channel_sum");
  }
  return rc;
}

extern "C" int bvq_fakequant_bwd(const bvq_quant_desc* d, const void* g, const void* x,
                                 const void* scale, const void* zp, void* dx, float* dscale,
                                 float* dzp, const void* tie_stat, int64_t* tie_info, void* workspace,
                                 int64_t workspace_bytes, bvq_stream_t stream) {
  return fakequant_bwd_impl(d, g, x, scale, zp, dx, dscale, dzp, tie_stat, tie_info, workspace, workspace_bytes,
                            stream, nullptr);
}

extern "C" int bvq_fakequant_bwd_bounds(const bvq_quant_desc* d, const void* g, const void* x, const void* scale,
                                        const void* zp, const float* bounds, void* dx, float* dscale, float* dbounds,
                                        void* workspace, int64_t workspace_bytes, bvq_stream_t stream) {
  if (!bounds || !dscale) {
    set_error("bvq_fakequant_bwd_bounds: null pointer");
    return BVQ_ERR_INVALID;
  }
  return fakequant_bwd_impl(d, g, x, scale, zp, dx, dscale, nullptr, nullptr, nullptr, workspace, workspace_bytes,
                            stream, nullptr, bounds, dbounds);
}

extern "C" int bvq_fakequant_bwd_learned(const bvq_quant_desc* d, const void* g, const void* x, const void* scale,
                                         const void* zp, void* dx, float* dscale, const void* value, int value_dtype,
                                         double min_val, int use_min, double int_threshold, const void* gscale,
                                         void* dvalue, void* workspace, int64_t workspace_bytes,
                                         bvq_stream_t stream) {
  if (!value || !dvalue || !dscale || value_dtype < BVQ_F32 || value_dtype > BVQ_F16) {
    set_error("bvq_fakequant_bwd_learned: bad argument");
    return BVQ_ERR_INVALID;
  }
  if (d && d->zp_per_channel) {
    set_error("bvq_fakequant_bwd_learned: per-channel zero-points are not covered");
    return BVQ_ERR_UNSUPPORTED;
  }
  LearnedScaleEpilogue ep = {};
  ep.value = value;
  ep.dvalue = dvalue;
  ep.gscale = gscale;
  ep.value_dtype = value_dtype;
  ep.scale_dtype = d ? d->scale_dtype : BVQ_F32;
  ep.use_min = use_min;
  ep.min_val = round_host((float)min_val, value_dtype);    // python scalar -> the parameter's dtype
  ep.int_threshold = (float)int_threshold;                  // the caller rounds it to the division's dtype
  return fakequant_bwd_impl(d, g, x, scale, zp, dx, dscale, nullptr, nullptr, nullptr, workspace, workspace_bytes,
                            stream, &ep);
}

static bool bwd_stats_supported(const bvq_quant_desc* d, int64_t& units, int64_t& per_channel) {
  if (!(d->scale_per_channel && d->channels > 1) || d->zp_per_channel) return false;
  const ColsPlan cp = cols_quant_plan(d, nullptr, nullptr, nullptr);
  if (cp.ok) {  // column-mapped partials: [prows][L] plus their fold [L]
    units = (cp.prows + cols_fold_scratch_rows(cp.prows)) * cp.L;
    per_channel = d->inner;
    return true;
  }
  units = bwd_units(d);
  per_channel = units / d->channels;
  return per_channel <= 4096;  // one workgroup of bwd_stats_finish_kernel per channel walks them
}

extern "C" int64_t bvq_fakequant_bwd_stats_workspace_bytes(const bvq_quant_desc* d) {
  if (validate(d)) return -1;
  int64_t units, per_channel;
  if (!bwd_stats_supported(d, units, per_channel)) return 0;  // use bvq_fakequant_bwd + bvq_stat_tie_apply_dscale
  return units * (int64_t)(sizeof(float) + sizeof(unsigned long long)) + 256;
}

struct ShardOut {  // batch-sharded tensors: the all-gather message instead of the deposit (null: unsharded)
  double* msg;
  long long* pos;
  int32_t rank;
};
static int bwd_stats_impl(const bvq_quant_desc* d, const void* g, const void* x, const void* scale, const void* zp,
                          const void* stat, void* dx, float* dscale, int scale_dtype, double int_threshold,
                          int quot_dtype, void* workspace, int64_t workspace_bytes, uint32_t* arrive,
                          int64_t arrive_words, bvq_stream_t stream, const ShardOut* shard = nullptr);

extern "C" int bvq_fakequant_bwd_stats(const bvq_quant_desc* d, const void* g, const void* x, const void* scale,
                                       const void* zp, const void* stat, void* dx, float* dscale,
                                       int scale_dtype, double int_threshold, int quot_dtype, void* workspace,
                                       int64_t workspace_bytes, bvq_stream_t stream) {
  return bwd_stats_impl(d, g, x, scale, zp, stat, dx, dscale, scale_dtype, int_threshold, quot_dtype, workspace,
                        workspace_bytes, nullptr, 0, stream);
}

extern "C" int bvq_fakequant_bwd_stats_onepass_supported(const bvq_quant_desc* d) {
  int64_t units, per_channel;
  if (validate(d) || !bwd_stats_supported(d, units, per_channel)) return 0;
  if (cols_quant_plan(d, nullptr, nullptr, nullptr).ok) return 0;  // column-mapped layouts: two launches
  return 1;
}

extern "C" int bvq_fakequant_bwd_stats_onepass(const bvq_quant_desc* d, const void* g, const void* x,
                                               const void* scale, const void* zp, const void* stat, void* dx,
                                               float* dscale, int scale_dtype, double int_threshold, int quot_dtype,
                                               void* workspace, int64_t workspace_bytes, uint32_t* arrive,
                                               int64_t arrive_words, bvq_stream_t stream) {
  if (!arrive) {
    set_error("bvq_fakequant_bwd_stats_onepass: null arrival buffer");
    return BVQ_ERR_INVALID;
  }
  if (!bvq_fakequant_bwd_stats_onepass_supported(d)) {
    set_error("bvq_fakequant_bwd_stats_onepass: layout not covered: use bvq_fakequant_bwd_stats");
    return BVQ_ERR_UNSUPPORTED;
  }
  if (arrive_words < d->channels) {
    set_error("bvq_fakequant_bwd_stats_onepass: arrival buffer of %lld words, %lld needed", (long long)arrive_words,
              (long long)d->channels);
    return BVQ_ERR_WORKSPACE;
  }
  return bwd_stats_impl(d, g, x, scale, zp, stat, dx, dscale, scale_dtype, int_threshold, quot_dtype, workspace,
                        workspace_bytes, arrive, arrive_words, stream);
}

extern "C" int bvq_fakequant_bwd_shard(const bvq_quant_desc* d, const void* g, const void* x, const void* scale,
                                       const void* zp, const void* stat, void* dx, double* message, int64_t* first_pos,
                                       int rank, void* workspace, int64_t workspace_bytes, uint32_t* arrive,
                                       int64_t arrive_words, bvq_stream_t stream) {
  if (!message || !first_pos || rank < 0) {
    set_error("bvq_fakequant_bwd_shard: bad argument");
    return BVQ_ERR_INVALID;
  }
  if (arrive && (!bvq_fakequant_bwd_stats_onepass_supported(d) || arrive_words < d->channels)) arrive = nullptr;
  ShardOut so = {message, reinterpret_cast<long long*>(first_pos), rank};
  float unused = 0.f;
  return bwd_stats_impl(d, g, x, scale, zp, stat, dx, &unused, BVQ_F32, 1.0, BVQ_F32, workspace, workspace_bytes, arrive,
                        arrive_words, stream, &so);
}

extern "C" int bvq_shard_unpack_deposit(int dtype, const void* x, void* dx, const double* gathered, int world,
                                        int64_t channels, int rank, const int64_t* first_pos, int64_t inner,
                                        int scale_dtype, double int_threshold, int quot_dtype, int pre_op,
                                        float* dscale_total, bvq_stream_t stream) {
  if (dtype < BVQ_F32 || dtype > BVQ_F16 || scale_dtype < BVQ_F32 || scale_dtype > BVQ_F16 || quot_dtype < BVQ_F32 ||
      quot_dtype > BVQ_F16 || channels < 1 || world < 1 || rank < 0 || rank >= world || inner < 1 || !x || !dx ||
      !gathered || !first_pos || !(int_threshold == int_threshold)) {
    set_error("bvq_shard_unpack_deposit: bad argument");
    return BVQ_ERR_INVALID;
  }
  GstatSrc gs = {};
  gs.from_dscale = 1;
  gs.scale_dtype = scale_dtype;
  gs.quot_dtype = quot_dtype;
  gs.int_threshold = (float)int_threshold;
  gs.pre_relu = pre_op == BVQ_PRE_RELU;
  const dim3 grid((unsigned)((channels + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const long long* fp = reinterpret_cast<const long long*>(first_pos);
  if (dtype == BVQ_F32)
    shard_unpack_deposit_kernel<float><<<grid, block, 0, st>>>(gathered, world, (int32_t)channels, rank, fp, x, dx, inner,
                                                              gs, dscale_total);
  else if (dtype == BVQ_BF16)
    shard_unpack_deposit_kernel<bf16_t><<<grid, block, 0, st>>>(gathered, world, (int32_t)channels, rank, fp, x, dx, inner,
                                                               gs, dscale_total);
  else
    shard_unpack_deposit_kernel<f16_t><<<grid, block, 0, st>>>(gathered, world, (int32_t)channels, rank, fp, x, dx, inner,
                                                              gs, dscale_total);
  return check_launch("bvq_shard_unpack_deposit");
}

static int bwd_stats_impl(const bvq_quant_desc* d, const void* g, const void* x, const void* scale, const void* zp,
                          const void* stat, void* dx, float* dscale, int scale_dtype, double int_threshold,
                          int quot_dtype, void* workspace, int64_t workspace_bytes, uint32_t* arrive,
                          int64_t arrive_words, bvq_stream_t stream, const ShardOut* shard) {
  int rc = validate(d);
  if (rc) return rc;
  int64_t units, per_channel;
  if (!bwd_stats_supported(d, units, per_channel)) {
    set_error("bvq_fakequant_bwd_stats: layout not covered (per-tensor scale or too many units per channel)");
    return BVQ_ERR_UNSUPPORTED;
  }
  if (scale_dtype < BVQ_F32 || scale_dtype > BVQ_F16 || quot_dtype < BVQ_F32 || quot_dtype > BVQ_F16) {
    set_error("bvq_fakequant_bwd_stats: bad dtype");
    return BVQ_ERR_INVALID;
  }
  const int64_t n = d->outer * d->channels * d->inner;
  hipStream_t st = (hipStream_t)stream;
  const int32_t channels = (int32_t)d->channels;
  if (n == 0) {
    if (dscale) (void)hipMemsetAsync(dscale, 0, sizeof(float) * channels, st);
    return BVQ_OK;
  }
  if (!g || !x || !scale || !zp || !stat || !dx || !dscale || !workspace) {
    set_error("bvq_fakequant_bwd_stats: null pointer");
    return BVQ_ERR_INVALID;
  }
  GstatSrc gs = {};
  gs.from_dscale = 1;
  gs.scale_dtype = scale_dtype;
  gs.quot_dtype = quot_dtype;
  gs.int_threshold = (float)int_threshold;
  gs.pre_relu = d->pre_op == BVQ_PRE_RELU;
  const bool nt =
      n * (int64_t)(2 * dtype_size(d->x_dtype) + dtype_size(d->ct_dtype)) >= nt_threshold_bytes();
  {
    const ColsPlan cp = cols_quant_plan(d, x, g, dx);
    const ColsPlan sized = cols_quant_plan(d, nullptr, nullptr, nullptr);
    if (sized.ok && !cp.ok) {
      set_error("bvq_fakequant_bwd_stats: the column-mapped route needs 16-byte aligned x, g and dx");
      return BVQ_ERR_UNSUPPORTED;
    }
    if (cp.ok) {
      const int64_t words = (cp.prows + cols_fold_scratch_rows(cp.prows)) * cp.L;
      const int64_t pos_off_c = ((words * (int64_t)sizeof(float) + 7) / 8) * 8;
      if (workspace_bytes < pos_off_c + words * (int64_t)sizeof(unsigned long long)) {
        set_error("bvq_fakequant_bwd_stats: workspace too small");
        return BVQ_ERR_WORKSPACE;
      }
      ColsQuantArgs ca = {};
      fill_cols_args(ca, cp, d);
      ca.x = x;
      ca.g = g;
      ca.y = dx;
      ca.scale = scale;
      ca.zp = zp;
      ca.ds_part = reinterpret_cast<float*>(workspace);
      ca.pos_part = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(workspace) + pos_off_c);
      ca.tie_stat = stat;
      BVQ_COLS_LAUNCH(fakequant_bwd_cols_kernel, ca, nt, st);
      rc = check_launch("bvq_fakequant_bwd_stats/cols");
      if (rc) return rc;
      float* ds_fold = nullptr;
      unsigned long long* pos_fold = nullptr;
      launch_cols_fold_sum_min(ca.ds_part, ca.pos_part, cp.prows, cp.L, ca.ds_part + cp.prows * cp.L,
                               ca.pos_part + cp.prows * cp.L, &ds_fold, &pos_fold, st);
      if (shard) {  // this shard's all-gather message from the folded partials: one wave per channel
        QuantArgs fa = {};
        fa.t.nob = 1;
        fa.t.channels = channels;
        fa.t.ppr = d->inner;
        fa.arrive_per_channel = (uint32_t)d->inner;
        fa.ds_part = ds_fold;
        fa.pos_part = pos_fold;
        fa.shard_msg = shard->msg;
        fa.shard_pos = shard->pos;
        fa.shard_rank = shard->rank;
        const dim3 cgrid((unsigned)((channels + kWavesPerBlock - 1) / kWavesPerBlock));
        channel_finish_kernel<float><<<cgrid, dim3(kBlock), 0, st>>>(fa);  // (the message path touches no tensor element)
        return check_launch("bvq_fakequant_bwd_shard/cols_finish");
      }
      const dim3 fgrid((unsigned)channels), fblock(kBlock);
      if (d->x_dtype == BVQ_F32)
        bwd_stats_finish_kernel<float><<<fgrid, fblock, 0, st>>>(ds_fold, pos_fold, dscale, gs, x, dx, 1, channels,
                                                                 d->inner, d->inner);
      else if (d->x_dtype == BVQ_BF16)
        bwd_stats_finish_kernel<bf16_t><<<fgrid, fblock, 0, st>>>(ds_fold, pos_fold, dscale, gs, x, dx, 1, channels,
                                                                  d->inner, d->inner);
      else
        bwd_stats_finish_kernel<f16_t><<<fgrid, fblock, 0, st>>>(ds_fold, pos_fold, dscale, gs, x, dx, 1, channels,
                                                                 d->inner, d->inner);
      return check_launch("bvq_fakequant_bwd_stats/cols_finish");
    }
  }
  const void* ptrs[3] = {x, g, dx};
  const int els[3] = {dtype_size(d->x_dtype), dtype_size(d->ct_dtype), dtype_size(d->x_dtype)};
  const int full = 16 / dtype_size(d->x_dtype);
  const int vec = snap_vec(pick_vec(full, d->outer * channels, d->inner, ptrs, els, 3, true), full);
  QuantArgs a = {};
  a.t = bwd_tiling(d->outer, channels, d->inner, vec);
#ifdef BVQ_CACHE_EXPERIMENT
  if (getenv("BVQ_X_BWD_REV")) a.t.reverse = atoi(getenv("BVQ_X_BWD_REV"));
#endif
  const int64_t pos_off = ((a.t.units * (int64_t)sizeof(float) + 7) / 8) * 8;
  if (workspace_bytes < pos_off + a.t.units * (int64_t)sizeof(unsigned long long)) {
    set_error("bvq_fakequant_bwd_stats: workspace too small");
    return BVQ_ERR_WORKSPACE;
  }
  a.ds_part = reinterpret_cast<float*>(workspace);
  a.pos_part = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(workspace) + pos_off);
  a.x = x;
  a.g = g;
  a.scale = scale;
  a.zp = zp;
  a.y = dx;
  a.tie_stat = stat;
  fill_args(a, d);
  a.arrive_per_channel = (uint32_t)(a.t.nob * a.t.ppr);
  if (shard) {
    a.shard_msg = shard->msg;
    a.shard_pos = shard->pos;
    a.shard_rank = shard->rank;
  }
  if (arrive) {  // one launch: the wave that completes a channel finishes it
    a.arrive = arrive;
    a.dscale_out = dscale;
    a.gs_scale_dtype = scale_dtype;
    a.gs_quot_dtype = quot_dtype;
    a.gs_int_threshold = (float)int_threshold;
#define BVQ_CALL(XT, CT) launch_bwd<XT, CT>(a, vec, kBwdDsArrive, nt, st)
    BVQ_DISPATCH_PAIR(d, BVQ_CALL);
#undef BVQ_CALL
    return check_launch("bvq_fakequant_bwd_stats_onepass");
  }
#define BVQ_CALL(XT, CT) launch_bwd<XT, CT>(a, vec, kBwdDsTies, nt, st)
  BVQ_DISPATCH_PAIR(d, BVQ_CALL);
#undef BVQ_CALL
  rc = check_launch("bvq_fakequant_bwd_stats");
  if (rc) return rc;
  if (shard) {
    const dim3 cgrid((unsigned)((channels + kWavesPerBlock - 1) / kWavesPerBlock));
    channel_finish_kernel<float><<<cgrid, dim3(kBlock), 0, st>>>(a);  // (the message path touches no tensor element)
    return check_launch("bvq_fakequant_bwd_shard/finish");
  }
  const dim3 grid((unsigned)channels), block(kBlock);
  if (d->x_dtype == BVQ_F32)
    bwd_stats_finish_kernel<float><<<grid, block, 0, st>>>(a.ds_part, a.pos_part, dscale, gs, x, dx, a.t.nob, channels,
                                                           a.t.ppr, d->inner);
  else if (d->x_dtype == BVQ_BF16)
    bwd_stats_finish_kernel<bf16_t><<<grid, block, 0, st>>>(a.ds_part, a.pos_part, dscale, gs, x, dx, a.t.nob, channels,
                                                            a.t.ppr, d->inner);
  else
    bwd_stats_finish_kernel<f16_t><<<grid, block, 0, st>>>(a.ds_part, a.pos_part, dscale, gs, x, dx, a.t.nob, channels,
                                                           a.t.ppr, d->inner);
  return check_launch("bvq_fakequant_bwd_stats/finish");
}

#endif  // backward part

