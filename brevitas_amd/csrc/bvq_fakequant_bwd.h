// bvq_fakequant_bwd.h -- device code of the quantizer backward: element math, the row-mapped kernel with its in-kernel
// finish (arrival), the column-mapped kernel, the finishing kernels, and the launch_bwd template whose instantiations
// are spread over bvq_fakequant_bwd_{bf16,f16,f32}.hip so that they build in parallel.
#pragma once
#include "bvq_fakequant.h"

namespace bvq {

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
  uint32_t lo, hi;      // bf16 {1, 0}, {0, 1}
  uint32_t lo16, hi16;  // float16 {1, 0}, {0, 1}
};
__device__ __forceinline__ DotSel make_dot_sel() {
  DotSel d = {0x00003f80u, 0x3f800000u, 0x00003c00u, 0x3c000000u};
  asm volatile("" : "+s"(d.lo), "+s"(d.hi), "+s"(d.lo16), "+s"(d.hi16));
  return d;
}
__device__ __forceinline__ f2 add_rounded_lanes_bf16(f2 acc, f2 v, const DotSel& sel) {
  typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 p = __builtin_convertvector(v, bf16x2);
  return f2{__builtin_amdgcn_fdot2_f32_bf16(p, __builtin_bit_cast(bf16x2, sel.lo), acc.x, false),
            __builtin_amdgcn_fdot2_f32_bf16(p, __builtin_bit_cast(bf16x2, sel.hi), acc.y, false)};
}
// float16: the same with v_dot2_f32_f16 (the selectors kept out of the compiler's sight like the bf16 ones)
#ifndef BVQ_BWD_LEAN_F16
#define BVQ_BWD_LEAN_F16 1
#endif
// acc + RN_f16(v.x) + RN_f16(v.y)   ({1, 1} = 0x3c003c00 is no inline constant: emitted as a literal)
__device__ __forceinline__ float add_rounded_pair_f16(float acc, f2 v) {
  typedef f16_t f16x2 __attribute__((ext_vector_type(2)));
  const f16x2 ones = {(f16_t)1.0f, (f16_t)1.0f};
  return __builtin_amdgcn_fdot2(__builtin_convertvector(v, f16x2), ones, acc, false);
}
__device__ __forceinline__ f2 add_rounded_lanes_f16(f2 acc, f2 v, const DotSel& sel) {
  typedef f16_t f16x2 __attribute__((ext_vector_type(2)));
  const f16x2 p = __builtin_convertvector(v, f16x2);
  return f2{__builtin_amdgcn_fdot2(p, __builtin_bit_cast(f16x2, sel.lo16), acc.x, false),
            __builtin_amdgcn_fdot2(p, __builtin_bit_cast(f16x2, sel.hi16), acc.y, false)};
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
    } else if constexpr (kLean && BVQ_BWD_LEAN_F16 && elem<CT>::id == BVQ_F16 && MIX) {
      const float a1 = add_rounded_pair_f16(ds_acc.x, gf * t5);
      const float a2 = add_rounded_pair_f16(ds_acc.y, -dt * rnd2<CT>(div(t1)));
      ds_acc = f2{a1, a2};
    } else if constexpr (kLean && BVQ_BWD_LEAN_F16 && elem<CT>::id == BVQ_F16) {
      ds_acc = add_rounded_lanes_f16(ds_acc, gf * t5, sel);
      ds_acc = add_rounded_lanes_f16(ds_acc, -dt * rnd2<CT>(div(t1)), sel);
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

// One wave's unit of the column-mapped backward: a block of rows of its strip of 64 column chunks.  The block is
// addressed through buffer descriptors (base = the block's first row, extent = its bytes; cols_plan keeps that below
// 2^31): a lane's rows are offset, offset + step, ... and rows past the block's end read zeros without a memory access
// and drop their stores -- the walk has no execution mask and no 64-bit address arithmetic (the first form of this
// kernel spent 40 of its 243 vector instructions per row on both).  PRE (the fused ReLU) is a template parameter for
// the same reason: as a run-time flag it cost five selects per pair.
// Arg-max search of the statistic: the hot loop keeps each column's largest |x| key only (packed unsigned 16-bit max:
// 2 instructions per pair); a lane one of whose columns attains its channel's statistic -- about one lane per channel
// in the whole launch -- walks its rows once more, cold, and takes the first row that shows it.
template <typename T>
constexpr int kColsBwdVec = sizeof(T) == 2 ? kColsTeamVec16 : elem<T>::vec;

template <typename T, int RM, bool NT, bool ZP0, bool FAST, bool PRE>
__device__ __forceinline__ void cols_bwd_rows(const ColsQuantArgs& a, const ColsLane<T, kColsBwdVec<T>>& ln, float qmin,
                                              float qmax, float* sh_ds) {
  constexpr int VEC = kColsBwdVec<T>;
  constexpr bool kSame16 = sizeof(T) == 2;
  typedef short i16x2 __attribute__((ext_vector_type(2)));
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  const int64_t nrows = ln.row_end - ln.blk0;  // wave-uniform, > 0
  const uint32_t bytes = (uint32_t)(nrows * a.p.L * (int64_t)sizeof(T));
  const buf_t bx = make_buf(reinterpret_cast<const T*>(a.x) + ln.blk0 * a.p.L, bytes);
  const buf_t bg = make_buf(reinterpret_cast<const T*>(a.g) + ln.blk0 * a.p.L, bytes);
  const buf_t bd = make_buf(reinterpret_cast<T*>(a.y) + ln.blk0 * a.p.L, bytes);
  // the workgroup's waves take the block's row groups in turn: wave w works on groups w, w + 4, ...
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t stride_rows = (int64_t)kWavesPerBlock * a.p.rpp;
  const uint32_t step = (uint32_t)(stride_rows * a.p.L * (int64_t)sizeof(T));  // between a lane's consecutive rows
  const uint32_t off0 = (uint32_t)(((ln.row0 - ln.blk0) * a.p.L + (int64_t)ln.chunk * VEC) * (int64_t)sizeof(T));
  const int64_t mine = nrows - (int64_t)wave * a.p.rpp;  // rows from this wave's first group on
  const int32_t steps = mine > 0 ? (int32_t)((mine + stride_rows - 1) / stride_rows) : 0;  // the last possibly past the end
  f2 r2[VEC / 2], ds2[VEC / 2], dz_unused = splat2(0.f);
  uint32_t um[kSame16 ? VEC / 2 : VEC];  // largest |x| key per column (16-bit types: two keys per word)
#pragma unroll
  for (int k = 0; k < VEC / 2; ++k) {
    r2[k] = f2{1.0f / ln.s2[k].x, 1.0f / ln.s2[k].y};
    ds2[k] = splat2(0.f);
  }
#pragma unroll
  for (int k = 0; k < (kSame16 ? VEC / 2 : VEC); ++k) um[k] = 0u;
  const bool ties = a.tie_stat != nullptr;
  const bool clamp_ste = a.clamp_ste != 0;
  const int mode = a.round_mode;
  const DotSel dsel = make_dot_sel();
  // the work on one row of this lane's columns
  auto row_work = [&](const vec_t<T, VEC>& xr, const vec_t<T, VEC>& gr, uint32_t off) {
    vec_t<T, VEC> dv;
#pragma unroll
    for (int k = 0; k < VEC; k += 2) {
      const f2 xraw = widen2<T>(xr.v[k], xr.v[k + 1]);
      const f2 xin = PRE ? relu2(xraw) : xraw;
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
      if constexpr (PRE) d = xraw > splat2(0.f) ? d : splat2(0.f);  // torch.relu backward: grad * (x > 0)
      pack2<T>(d, dv.v[k], dv.v[k + 1]);
    }
    buf_store<T, VEC, NT>(bd, off, dv);  // dropped past the block's end
    if (ties) {
      if constexpr (kSame16) {
        const vec_t<uint32_t, VEC / 2> w = __builtin_bit_cast(vec_t<uint32_t, VEC / 2>, xr);
#pragma unroll
        for (int k = 0; k < VEC / 2; ++k) {
          // relu: negative patterns (sign bit set) count as 0; then both sign bits are cleared (NaN keys stay above inf)
          const uint32_t w2 = PRE ? __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, w.v[k]),
                                                                                      i16x2{0, 0}))
                                  : w.v[k];
          um[k] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, um[k]),
                                                                         __builtin_bit_cast(u16x2, w2 & 0x7fff7fffu)));
        }
      } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const uint32_t b = pre_abs_bits<T, PRE>(xr.v[k]);
          um[k] = b > um[k] ? b : um[k];
        }
      }
    }
  };
  uint32_t off = off0;
  {
    constexpr int kU = BVQ_COLS_BWD_UNROLL;
    for (int32_t i = 0; i < steps; i += kU) {
      vec_t<T, VEC> xv[kU], gv[kU];
#pragma unroll
      for (int j = 0; j < kU; ++j) {
        xv[j] = buf_load<T, VEC, NT>(bx, off + (uint32_t)j * step);
        gv[j] = buf_load<T, VEC, NT>(bg, off + (uint32_t)j * step);
      }
#pragma unroll
      for (int j = 0; j < kU; ++j)
        if (i + j < steps) row_work(xv[j], gv[j], off + (uint32_t)j * step);  // wave-uniform
      off += (uint32_t)kU * step;
    }
  }
  // the workgroup's partial row of the [prows][L] array: the four waves' sums of a column, added in wave order
  const int64_t prow = ln.blk0 / a.p.rb * a.p.rpp + ln.sub;
  const int64_t base = prow * a.p.L + (int64_t)ln.chunk * VEC;
  if (a.ds_part) {  // (wave-uniform; every wave of the workgroup gets here -- lane 0 of a wave is always active)
    const int lane = threadIdx.x & 63;
    if (wave > 0) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) sh_ds[(wave - 1) * 8 * kWave + k * kWave + lane] = (k & 1) ? ds2[k / 2].y : ds2[k / 2].x;
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        float acc = (k & 1) ? ds2[k / 2].y : ds2[k / 2].x;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock - 1; ++w) acc += sh_ds[w * 8 * kWave + k * kWave + lane];
        a.ds_part[base + k] = acc;
      }
    }
  }
  if (!ties) return;
  // columns whose largest key is their channel's statistic (a lane without rows has seen nothing)
  uint32_t sk[VEC], frow[VEC];
  bool hit[VEC], any = false;
  const bool has_rows = ln.row0 < ln.row_end;
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    const int64_t col = (int64_t)ln.chunk * VEC + k;
    sk[k] = abs_bits<T>(reinterpret_cast<const T*>(a.tie_stat)[col / a.inner]);
    uint32_t key;
    if constexpr (kSame16) {
      const uint32_t k16 = (k & 1) ? (um[k / 2] >> 16) : (um[k / 2] & 0xffffu);
      key = elem<T>::id == BVQ_BF16 ? (k16 << 16) : k16;
    } else {
      key = um[k];
    }
    hit[k] = has_rows && key == sk[k];
    any = any || hit[k];
    frow[k] = ~0u;
  }
  if (any) {
    // cold: the first of this lane's rows showing the statistic, four rows in flight (rows past the end read zeros --
    // they can only match a statistic of 0, which the lane's first row has matched before)
    uint32_t o = off0;
    for (int32_t i = 0; i < steps; i += 4) {
      vec_t<T, VEC> xr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xr[j] = buf_load<T, VEC, false>(bx, o + (uint32_t)j * step);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int k = 0; k < VEC; ++k)
          if (hit[k] && frow[k] == ~0u && pre_abs_bits<T, PRE>(xr[j].v[k]) == sk[k]) frow[k] = (uint32_t)(i + j);
      }
      o += 4u * step;
    }
  }
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    const int64_t col = (int64_t)ln.chunk * VEC + k;
    const unsigned long long row = (unsigned long long)ln.row0 + (unsigned long long)frow[k] * (unsigned long long)stride_rows;
    const unsigned long long pos =
        hit[k] && frow[k] != ~0u ? row * (unsigned long long)a.inner + (unsigned long long)(col % a.inner) : ~0ull;
    // pos_part: one entry per column here (initialised to ~0 by the host side), tie_info: one per channel
    if (pos != ~0ull) atomicMin(a.pos_part ? &a.pos_part[col] : &a.tie_info[col / a.inner], pos);
  }
}

template <typename T, int RM, bool NT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(BVQ_COLS_BWD_WAVES, 8))) void fakequant_bwd_cols_kernel(
    ColsQuantArgs a) {
  __shared__ float sh_ds[(kWavesPerBlock - 1) * 8 * kWave];  // waves 1..3 hand their column sums to wave 0
  ColsLane<T, kColsBwdVec<T>> ln;
  if (!ln.init(a, true) || !ln.active) return;
  const float qmin = rnd<T>(a.qmin), qmax = rnd<T>(a.qmax);
#define BVQ_COLS_BWD(ZP0, FAST)                                   \
  do {                                                            \
    if (a.pre_relu)                                               \
      cols_bwd_rows<T, RM, NT, ZP0, FAST, true>(a, ln, qmin, qmax, sh_ds);  \
    else                                                          \
      cols_bwd_rows<T, RM, NT, ZP0, FAST, false>(a, ln, qmin, qmax, sh_ds); \
  } while (0)
  if constexpr (sizeof(T) == 2) {
    if (ln.fast) {
      if (ln.zp0)
        BVQ_COLS_BWD(true, true);
      else
        BVQ_COLS_BWD(false, true);
      return;
    }
    if (ln.zp0) {
      BVQ_COLS_BWD(true, false);
      return;
    }
  }
  BVQ_COLS_BWD(false, false);
#undef BVQ_COLS_BWD
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

template <typename XT, typename CT, int MODE>
static void launch_bwd_mode(const QuantArgs& a, int vec, bool nt, hipStream_t st) {
  constexpr int V = elem<XT>::vec;
  const dim3 grid(grid_for_units(a.t.units)), block(kBlock);
  const bool rne = a.round_mode == BVQ_ROUND;
  // Instantiations: round-half-even at full vector width with either cache policy; every other rounding mode shares
  // one kernel (default policy), and so do ragged / misaligned rows (one element per lane).  The one-launch form
  // (kBwdDsArrive) exists for the first pair only: bwd_stats_impl sends everything else to the two-launch route.
  if (vec == V) {
    if (rne && nt) {
      fakequant_bwd_kernel<XT, CT, V, BVQ_ROUND, MODE, true><<<grid, block, 0, st>>>(a);
      return;
    }
    if (rne) {
      fakequant_bwd_kernel<XT, CT, V, BVQ_ROUND, MODE, false><<<grid, block, 0, st>>>(a);
      return;
    }
  }
  if constexpr (MODE != kBwdDsArrive) {
    if (vec == V)
      fakequant_bwd_kernel<XT, CT, V, kAnyRM, MODE, false><<<grid, block, 0, st>>>(a);
    else if (rne)
      fakequant_bwd_kernel<XT, CT, 1, BVQ_ROUND, MODE, false><<<grid, block, 0, st>>>(a);
    else
      fakequant_bwd_kernel<XT, CT, 1, kAnyRM, MODE, false><<<grid, block, 0, st>>>(a);
  }
}
// the launches the one-launch backward is instantiated for
static inline bool bwd_arrive_covers(int vec, int full, int round_mode) { return vec == full && round_mode == BVQ_ROUND; }

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

#define BVQ_LAUNCH_BWD(XT, CT) void launch_bwd<XT, CT>(const QuantArgs&, int, int, bool, hipStream_t)

}  // namespace bvq
