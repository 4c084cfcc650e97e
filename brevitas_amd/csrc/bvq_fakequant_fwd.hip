// bvq_fakequant_fwd.hip -- fused affine quantize/dequantize: the forward kernels and their entry points.
//
// Replaces the ~9 full-tensor ATen passes of IntQuant.forward (B/core/quant/int_base.py:63-97)
// with one read of x and one write of y, and the ~8 passes autograd runs for its backward with one
// read of g, one read of x and one write of dx (the per-channel scale / zero-point gradient sums,
// and the search for the elements that attain the abs-max statistic, ride on the same reads).
// HBM-bound: algorithmic bytes per element are
//   forward  sizeof(x) + sizeof(y)          backward  sizeof(g) + sizeof(x) + sizeof(dx).

#include "bvq_fakequant.h"

namespace bvq {

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


template <typename T>
constexpr int kColsFwdVec = sizeof(T) == 2 ? kColsFwdVec16 : elem<T>::vec;

// One wave's unit of the column-mapped forward: a block of rows of its strip of 64 column chunks, addressed through buffer
// descriptors like the backward's (bvq_fakequant_bwd.h, cols_bwd_rows): rows past the block's end read zeros without a
// memory access and drop their stores, so the walk has no execution mask and no 64-bit address arithmetic; the fused ReLU
// is a template parameter; a lane holds 4 columns of a 16-bit type (8-byte loads: half the per-column scales /
// reciprocals / zero-points to fetch and divide at the start of every unit, half the registers).
template <typename T, int RM, bool NT, bool ZP0, bool FAST, bool PRE>
__device__ __forceinline__ void cols_fwd_rows(const ColsQuantArgs& a, const ColsLane<T, kColsFwdVec<T>>& ln, float qmin,
                                              float qmax) {
  constexpr int VEC = kColsFwdVec<T>;
#ifndef BVQ_COLS_FWD_UNROLL
#define BVQ_COLS_FWD_UNROLL 4  // rows in flight per lane
#endif
  constexpr int kU = BVQ_COLS_FWD_UNROLL;
  const int64_t nrows = ln.row_end - ln.blk0;  // wave-uniform, > 0
  const uint32_t bytes = (uint32_t)(nrows * a.p.L * (int64_t)sizeof(T));
  const buf_t bx = make_buf(reinterpret_cast<const T*>(a.x) + ln.blk0 * a.p.L, bytes);
  const buf_t by = make_buf(reinterpret_cast<T*>(a.y) + ln.blk0 * a.p.L, bytes);
  const uint32_t step = (uint32_t)((int64_t)a.p.rpp * a.p.L * (int64_t)sizeof(T));  // between a lane's consecutive rows
  uint32_t off = (uint32_t)(((int64_t)ln.sub * a.p.L + (int64_t)ln.chunk * VEC) * (int64_t)sizeof(T));
  const int32_t steps = (int32_t)((nrows + a.p.rpp - 1) / a.p.rpp);  // rows per lane, the last possibly past the end
  f2 r2[VEC / 2];
#pragma unroll
  for (int k = 0; k < VEC / 2; ++k) r2[k] = f2{1.0f / ln.s2[k].x, 1.0f / ln.s2[k].y};
  const int mode = a.round_mode;
  for (int32_t i = 0; i < steps; i += kU) {
    vec_t<T, VEC> xv[kU];
#pragma unroll
    for (int j = 0; j < kU; ++j) xv[j] = buf_load<T, VEC, NT>(bx, off + (uint32_t)j * step);
#pragma unroll
    for (int j = 0; j < kU; ++j) {
      if (i + j < steps) {  // wave-uniform
        vec_t<T, VEC> yv;
#pragma unroll
        for (int k = 0; k < VEC; k += 2) {
          f2 xf = widen2<T>(xv[j].v[k], xv[j].v[k + 1]);
          if constexpr (PRE) xf = relu2(xf);
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
        buf_store<T, VEC, NT>(by, off + (uint32_t)j * step, yv);  // dropped past the block's end
      }
    }
    off += (uint32_t)kU * step;
  }
}

template <typename T, int RM, bool NT>
__global__ __launch_bounds__(kBlock) void fakequant_fwd_cols_kernel(ColsQuantArgs a) {
  ColsLane<T, kColsFwdVec<T>> ln;
  if (!ln.init(a) || !ln.active) return;
  const float qmin = rnd<T>(a.qmin), qmax = rnd<T>(a.qmax);
#define BVQ_COLS_FWD(ZP0, FAST)                                \
  do {                                                         \
    if (a.pre_relu)                                            \
      cols_fwd_rows<T, RM, NT, ZP0, FAST, true>(a, ln, qmin, qmax);  \
    else                                                       \
      cols_fwd_rows<T, RM, NT, ZP0, FAST, false>(a, ln, qmin, qmax); \
  } while (0)
  if constexpr (sizeof(T) == 2) {
    if (ln.fast) {
      if (ln.zp0)
        BVQ_COLS_FWD(true, true);
      else
        BVQ_COLS_FWD(false, true);
      return;
    }
    if (ln.zp0) {
      BVQ_COLS_FWD(true, false);
      return;
    }
  }
  BVQ_COLS_FWD(false, false);
#undef BVQ_COLS_FWD
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

template <typename XT, typename CT>
static void launch_fwd(const QuantArgs& a, int vec, bool nt, hipStream_t st) {
  constexpr int V = elem<XT>::vec;
  const dim3 grid(grid_for_units(a.t.units)), block(kBlock);
  const bool rne = a.round_mode == BVQ_ROUND;
  if (vec == V) {
    if (rne && nt)
      fakequant_fwd_kernel<XT, CT, V, BVQ_ROUND, true><<<grid, block, 0, st>>>(a);
    else if (rne)
      fakequant_fwd_kernel<XT, CT, V, BVQ_ROUND, false><<<grid, block, 0, st>>>(a);
    else  // (the other rounding modes share one kernel, default cache policy)
      fakequant_fwd_kernel<XT, CT, V, kAnyRM, false><<<grid, block, 0, st>>>(a);
  } else {
    if (rne)
      fakequant_fwd_kernel<XT, CT, 1, BVQ_ROUND, false><<<grid, block, 0, st>>>(a);
    else
      fakequant_fwd_kernel<XT, CT, 1, kAnyRM, false><<<grid, block, 0, st>>>(a);
  }
}


}  // namespace bvq

using namespace bvq;

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
    const ColsPlan cp = cols_quant_plan(d, x, y, nullptr, true, false, kColsFwdVec16);
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

