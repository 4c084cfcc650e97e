// bvq_fakequant.h -- what the quantizer translation units share: the kernel argument blocks, the division forms,
// the column-mapped lane state, and the host-side helpers of the entry points (validation, layout -> arguments).
// Included by bvq_fakequant_fwd.hip (forward kernels + entry points), bvq_fakequant_bwd.hip (backward entry points,
// column-mapped and finishing kernels) and, through bvq_fakequant_bwd.h, by the three translation units that
// instantiate the row-mapped backward kernel per dtype family (bvq_fakequant_bwd_{bf16,f16,f32}.hip).
#pragma once

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
template <typename T, int V = elem<T>::vec>
struct ColsLane {
  static constexpr int VEC = V;
  int64_t row0, row_end;
  int64_t blk0;  // first row of the unit's row block (wave-uniform)
  int32_t chunk, sub;
  bool active;
  int32_t ch[VEC];
  f2 s2[VEC / 2], z2[VEC / 2];
  bool fast, zp0;  // wave-uniform: every column's scale suits the bf16 reciprocal / every zero-point is +0

  // team: the unit is the WORKGROUP's -- its waves take the block's rows in turn (wave w: rows w, w + 4, ... of each
  // lane group), so that the workgroup walks one contiguous window of memory, and combine their partials on chip
  __device__ __forceinline__ bool init(const ColsQuantArgs& a, bool team = false) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int64_t unit = team ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * kWavesPerBlock + wave;
    if (unit >= a.p.units) return false;
    const int64_t rblk = unit / a.p.strips;
    const int32_t strip = (int32_t)(unit - rblk * a.p.strips);
    sub = lane / a.p.lpr;
    chunk = strip * kWave + (lane - sub * a.p.lpr);
    active = sub < a.p.rpp && chunk < a.p.cps;
    blk0 = rblk * a.p.rb;
    row0 = blk0 + (team ? (int64_t)wave * a.p.rpp : 0) + sub;
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
                                bool no_partials = false, bool team = false, int vec16 = 0) {
  ColsPlan none = {};
  if (!(d->scale_per_channel && d->channels > 1) || d->x_dtype != d->ct_dtype || d->out_kind != BVQ_OUT_DEQUANT)
    return none;
  if ((reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2)) & 15)
    return none;
  return cols_plan(d->x_dtype, d->outer, d->channels, d->inner, no_partials, team, vec16);
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

#define BVQ_COLS_LAUNCH(KERNEL, a, nt, st) BVQ_COLS_LAUNCH_G(KERNEL, a, nt, st, grid_for_units((a).p.units))
// GRID workgroups (a kernel whose unit is the workgroup passes the plan's unit count)
#define BVQ_COLS_LAUNCH_G(KERNEL, a, nt, st, GRID)                                               \
  do {                                                                                           \
    const dim3 grid((unsigned)(GRID)), block(kBlock);                                            \
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
