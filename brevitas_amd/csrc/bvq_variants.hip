// bvq_variants.hip -- the other quantizers of the same elementwise family as fused kernels:
//   BinaryQuant / ClampedBinaryQuant   B/core/quant/binary.py:19-101     y = binary_sign_ste([clamp] x) * scale
//   TernaryQuant                       B/core/quant/ternary.py:18-66     y = [|x| > t * scale] * sign(x) * scale
//   DecoupledIntQuant                  B/core/quant/int_base.py:100-182  round with (pre_scale, pre_zp),
//                                                                        de-quantize with (scale, zp)
//   TruncIntQuant                      B/core/quant/int.py:199-229       recover the integer, drop LSBs, de-quantize
// The reference issues 4..10 torch ops per call (each a full pass over the tensor, plus what autograd saves and
// re-reads); here the forward is ONE read of x and one write of y, the backward one read of g, one of x and one
// write of dx, with the gradients of the scales riding on the same reads as per-unit partial sums (fixed-order
// double combine, bvq_sums.h).  Every intermediate is rounded to the compute dtype CT exactly where the
// reference's op chain rounds it (bvq_quant_math.h), so y and dx are the reference's bits.
#include "bvq_quant_math.h"
#include "bvq_sums.h"
#include "bvq_ties.h"

namespace bvq {

struct VarArgs {
  Tiling t;
  const void* x;
  const void* g;          // bwd
  void* y;                // fwd: y (CT) ; bwd: dx (XT)
  const void* scale;      // 1 or `channels` elements, scale_dtype
  const void* pre_scale;  // DECOUPLED: the scale the rounding grid comes from (same layout / dtype)
  const void* zp;         // DECOUPLED / TRUNC: one element, zp_dtype (null: +0)
  const void* pre_zp;     // DECOUPLED: one element (null: +0)
  float* part_a;          // bwd: per-unit partial of d(scale)      (null: not wanted)
  float* part_b;          // bwd: per-unit partial of d(pre_scale)  (DECOUPLED; null: not wanted)
  float qmin, qmax, threshold, trunc_scale;
  int32_t kind, scale_dtype, zp_dtype, scale_pc, scalar_cast, clamp_ste, round_mode;
};

struct VarScalars {
  float s, ps, z, pz;  // scale, pre-scale, zero-point, pre-zero-point as the arithmetic sees them
  float thr;           // TERNARY: threshold * scale as the comparison sees it
  float cb;            // CLAMPED_BINARY: the clamp bound as its comparisons see it
  float qmin, qmax, ts;
  int mode;
  bool ste;
};

template <typename XT, typename CT>
__device__ __forceinline__ VarScalars load_var_scalars(const VarArgs& a, int32_t channel) {
  VarScalars k;
  const int64_t ci = a.scale_pc ? channel : 0;
  k.s = load_scalar_as_f(a.scale, a.scale_dtype, ci);
  k.ps = a.pre_scale ? load_scalar_as_f(a.pre_scale, a.scale_dtype, ci) : k.s;
  k.z = a.zp ? load_scalar_as_f(a.zp, a.zp_dtype, 0) : 0.f;
  k.pz = a.pre_zp ? load_scalar_as_f(a.pre_zp, a.zp_dtype, 0) : 0.f;
  // threshold * scale: a python float times the scale tensor, rounded to the scale's dtype (ternary.py:64) ...
  float thr = a.threshold * k.s;
  thr = a.scale_dtype == BVQ_F32 ? thr : (a.scale_dtype == BVQ_BF16 ? rnd<bf16_t>(thr) : rnd<f16_t>(thr));
  // ... and a comparison converts both sides to their common dtype, which a 0-dim operand never widens: |x| > thr
  // and x > scale (tensor_clamp's torch.where conditions) compare in x's dtype unless the scale is dimensioned
  k.thr = a.scale_pc ? thr : rnd<XT>(thr);
  k.cb = a.scale_pc ? k.s : rnd<XT>(k.s);
  if (a.scalar_cast && !a.scale_pc) {  // device-torch semantics for a 0-dim operand wider than CT (bvq.h)
    k.s = rnd<CT>(k.s);
    k.ps = rnd<CT>(k.ps);
  }
  if (a.scalar_cast) {
    k.z = rnd<CT>(k.z);
    k.pz = rnd<CT>(k.pz);
  }
  k.qmin = rnd<CT>(a.qmin);
  k.qmax = rnd<CT>(a.qmax);
  k.ts = a.trunc_scale;
  k.mode = a.round_mode;
  k.ste = a.clamp_ste != 0;
  return k;
}

__device__ __forceinline__ float bsign(float v) { return (float)(v >= 0.f) - (float)(v < 0.f); }  // +1 at 0

// ---- forward, one element -----------------------------------------------------------------------
template <typename CT>
__device__ __forceinline__ float var_fwd(int kind, float x, const VarScalars& k) {
  switch (kind) {
    case BVQ_VAR_BINARY:
      return rnd<CT>(bsign(x) * k.s);                               // binary.py:62
    case BVQ_VAR_CLAMPED_BINARY: {
      const float xc = clamp_where(x, -k.cb, k.cb);                 // binary.py:113 (tensor_clamp: NaN passes)
      return rnd<CT>(bsign(xc) * k.s);                              // :114
    }
    case BVQ_VAR_TERNARY: {
      const float m = __builtin_fabsf(x) > k.thr ? 1.f : 0.f;       // ternary.py:64: mask.float()
      return (m * sgn_f(x)) * k.s;                                  // :65-66; CT is float32 here (mask.float() promotes)
    }
    case BVQ_VAR_DECOUPLED: {
      float t = rnd<CT>(x / k.ps);                                  // int_base.py:153
      t = rnd<CT>(t + k.pz);                                        // :154
      t = round_any<CT>(t, k.mode);                                 // :157
      const float q = clamp_where(t, k.qmin, k.qmax);               // :158
      return rnd<CT>(rnd<CT>(q - k.z) * k.s);                       // :178-179
    }
    default: {  // BVQ_VAR_TRUNC
      float t = rnd<CT>(x / k.s);                                   // int.py:218
      t = rnd<CT>(t + k.z);                                         // :219
      t = __builtin_rintf(t);                                       // :220 round_ste
      t = rnd<CT>(t / k.ts);                                        // :224
      t = round_any<CT>(t, k.mode);                                 // :225
      return rnd<CT>(rnd<CT>(t - k.z) * k.s);                       // :226-227
    }
  }
}

// ---- backward, one element: returns dx, adds this element's terms of d(scale) / d(pre_scale) ------------
template <typename CT>
__device__ __forceinline__ float var_bwd(int kind, float x, float g, const VarScalars& k, float& da, float& db) {
  switch (kind) {
    case BVQ_VAR_BINARY:
      da += rnd<CT>(g * bsign(x));
      return rnd<CT>(g * k.s);                                      // straight through the sign
    case BVQ_VAR_CLAMPED_BINARY: {
      const bool hi = x > k.cb;
      const float x1 = hi ? k.cb : x;
      const bool lo = x1 < -k.cb;
      const float xc = lo ? -k.cb : x1;
      const float dsg = rnd<CT>(g * k.s);
      da += rnd<CT>(g * bsign(xc));
      if (k.ste) return dsg;
      // TensorClamp: the two torch.where route the gradient to x where it passed, to the bounds (+scale, -scale)
      // where it was replaced
      if (hi) da += dsg;
      if (lo) da += rnd<CT>(-dsg);
      return (hi || lo) ? 0.f : dsg;
    }
    case BVQ_VAR_TERNARY: {
      const float m = __builtin_fabsf(x) > k.thr ? 1.f : 0.f;
      da += g * (m * sgn_f(x));
      return (g * k.s) * m;                                         // rounded to x's dtype by the caller
    }
    case BVQ_VAR_DECOUPLED: {
      const float t1 = rnd<CT>(x / k.ps);
      const float t2 = rnd<CT>(t1 + k.pz);
      const float t3 = round_any<CT>(t2, k.mode);
      const bool hi = t3 > k.qmax;
      float q = hi ? k.qmax : t3;
      const bool lo = q < k.qmin;
      q = lo ? k.qmin : q;
      const float dq = rnd<CT>(g * k.s);
      const float dt = (k.ste || !(hi || lo)) ? dq : 0.f;
      da += rnd<CT>(g * rnd<CT>(q - k.z));
      db += rnd<CT>(-dt * rnd<CT>(t1 / k.ps));
      return rnd<CT>(dt / k.ps);
    }
    default: {  // BVQ_VAR_TRUNC: straight through both roundings
      const float t1 = rnd<CT>(x / k.s);
      float t = rnd<CT>(t1 + k.z);
      t = __builtin_rintf(t);
      t = rnd<CT>(t / k.ts);
      t = round_any<CT>(t, k.mode);
      const float d4 = rnd<CT>(g * k.s);
      const float d1 = rnd<CT>(d4 / k.ts);
      da += rnd<CT>(g * rnd<CT>(t - k.z));
      da += rnd<CT>(-d1 * rnd<CT>(t1 / k.s));
      return rnd<CT>(d1 / k.s);
    }
  }
}

constexpr int kVarUnroll = 4;

template <typename XT, typename CT, int VEC, bool BWD>
__global__ __launch_bounds__(kBlock) void variant_kernel(VarArgs a) {
  const Unit u = locate_unit(a.t);
  if (!u.valid) return;
  const int lane = threadIdx.x & 63;
  const VarScalars k = load_var_scalars<XT, CT>(a, u.channel);
  const int kind = a.kind;
  const XT* __restrict__ xp = reinterpret_cast<const XT*>(a.x) + u.base;
  const CT* __restrict__ gp = BWD ? reinterpret_cast<const CT*>(a.g) + u.base : nullptr;
  typedef typename std::conditional<BWD, XT, CT>::type OT;  // forward writes y (CT), backward dx (XT)
  OT* __restrict__ op = reinterpret_cast<OT*>(a.y) + u.base;
  float da = 0.f, db = 0.f;
  ChunkWalker cur;
  cur.init(u, VEC, lane);
  const int64_t total = (int64_t)u.nrows * cur.cpr;
  for (int64_t done = 0; done < total; done += (int64_t)kWave * kVarUnroll) {
    vec_t<XT, VEC> xv[kVarUnroll];
    vec_t<CT, VEC> gv[kVarUnroll];
    int64_t off[kVarUnroll];
    bool ok[kVarUnroll];
#pragma unroll
    for (int j = 0; j < kVarUnroll; ++j) {
      ok[j] = cur.valid();
      off[j] = cur.offset(u.row_stride, VEC);
      if (ok[j]) {
        xv[j] = load_vec<XT, VEC>(xp + off[j]);
        if constexpr (BWD) gv[j] = load_vec<CT, VEC>(gp + off[j]);
      }
      cur.next();
    }
#pragma unroll
    for (int j = 0; j < kVarUnroll; ++j) {
      if (ok[j]) {
        vec_t<OT, VEC> ov;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          if constexpr (BWD)
            ov.v[e] = from_f<OT>(var_bwd<CT>(kind, to_f<XT>(xv[j].v[e]), to_f<CT>(gv[j].v[e]), k, da, db));
          else
            ov.v[e] = from_f<OT>(var_fwd<CT>(kind, to_f<XT>(xv[j].v[e]), k));
        }
        store_vec<OT, VEC>(op + off[j], ov);
      }
    }
  }
  // ragged ends: the (< VEC) elements after the last full chunk of every row of the unit
  const int32_t tail = (int32_t)(u.len - (int64_t)cur.cpr * VEC);
  for (int32_t e = lane; e < u.nrows * tail; e += kWave) {
    const int32_t tr = e / tail, tk = e - tr * tail;
    const int64_t i = (int64_t)tr * u.row_stride + (int64_t)cur.cpr * VEC + tk;
    if constexpr (BWD)
      op[i] = from_f<OT>(var_bwd<CT>(kind, to_f<XT>(xp[i]), to_f<CT>(gp[i]), k, da, db));
    else
      op[i] = from_f<OT>(var_fwd<CT>(kind, to_f<XT>(xp[i]), k));
  }
  if constexpr (BWD) {
    if (a.part_a) {
      da = wave_sum(da);
      if (lane == 0) a.part_a[u.id] = da;
    }
    if (a.part_b) {
      db = wave_sum(db);
      if (lane == 0) a.part_b[u.id] = db;
    }
  }
}

// ---- host side ----------------------------------------------------------------------------------------
static int var_validate(const char* fn, const bvq_variant_desc* d) {
  if (!d) {
    set_error("%s: null descriptor", fn);
    return BVQ_ERR_INVALID;
  }
  if (d->outer < 0 || d->channels < 1 || d->inner < 0 || d->kind < BVQ_VAR_BINARY || d->kind > BVQ_VAR_TRUNC ||
      d->round_mode < BVQ_ROUND || d->round_mode > BVQ_DPU_ROUND) {
    set_error("%s: bad descriptor", fn);
    return BVQ_ERR_INVALID;
  }
  const bool ok = (d->x_dtype == d->ct_dtype && d->x_dtype >= BVQ_F32 && d->x_dtype <= BVQ_F16) ||
                  (d->ct_dtype == BVQ_F32 && (d->x_dtype == BVQ_BF16 || d->x_dtype == BVQ_F16));
  if (!ok || d->scale_dtype < BVQ_F32 || d->scale_dtype > BVQ_F16 || d->zp_dtype < BVQ_F32 || d->zp_dtype > BVQ_F16) {
    set_error("%s: unsupported dtypes (x %d, compute %d, scale %d, zero-point %d)", fn, d->x_dtype, d->ct_dtype,
              d->scale_dtype, d->zp_dtype);
    return BVQ_ERR_UNSUPPORTED;
  }
  if (d->kind == BVQ_VAR_TERNARY && d->ct_dtype != BVQ_F32) {
    set_error("%s: TernaryQuant computes in float32 (mask.float() promotes)", fn);
    return BVQ_ERR_UNSUPPORTED;
  }
  return BVQ_OK;
}

static void var_fill(VarArgs& a, const bvq_variant_desc* d, const void* const* ptrs, const int* els, int nptr, int& vec) {
  const bool pc = d->scale_per_channel && d->channels > 1;
  const int64_t outer = pc ? d->outer : 1;
  const int32_t channels = pc ? (int32_t)d->channels : 1;
  const int64_t row_len = pc ? d->inner : d->outer * d->channels * d->inner;
  const int full = 16 / dtype_size(d->x_dtype);
  vec = pick_vec(full, outer * channels, row_len, ptrs, els, nptr, true);
  vec = vec == full ? full : 1;
  a.t = make_tiling(outer, channels, row_len, vec, 0, true);
  a.qmin = d->qmin;
  a.qmax = d->qmax;
  a.threshold = d->threshold;
  a.trunc_scale = d->trunc_scale;
  a.kind = d->kind;
  a.scale_dtype = d->scale_dtype;
  a.zp_dtype = d->zp_dtype;
  a.scale_pc = pc ? 1 : 0;
  a.scalar_cast = d->scalar_mode == BVQ_SCALAR_CAST;
  a.clamp_ste = d->clamp_ste;
  a.round_mode = d->round_mode;
}

template <bool BWD>
static void var_launch(const VarArgs& a, const bvq_variant_desc* d, int vec, hipStream_t st) {
  const dim3 grid(grid_for_units(a.t.units)), block(kBlock);
#define BVQ_VAR(XT, CT)                                                  \
  do {                                                                   \
    if (vec == elem<XT>::vec)                                            \
      variant_kernel<XT, CT, elem<XT>::vec, BWD><<<grid, block, 0, st>>>(a); \
    else                                                                 \
      variant_kernel<XT, CT, 1, BWD><<<grid, block, 0, st>>>(a);         \
  } while (0)
  if (d->x_dtype == BVQ_F32)
    BVQ_VAR(float, float);
  else if (d->x_dtype == BVQ_BF16 && d->ct_dtype == BVQ_BF16)
    BVQ_VAR(bf16_t, bf16_t);
  else if (d->x_dtype == BVQ_BF16)
    BVQ_VAR(bf16_t, float);
  else if (d->ct_dtype == BVQ_F16)
    BVQ_VAR(f16_t, f16_t);
  else
    BVQ_VAR(f16_t, float);
#undef BVQ_VAR
}

}  // namespace bvq

using namespace bvq;

extern "C" int bvq_variant_fwd(const bvq_variant_desc* d, const void* x, const void* scale, const void* pre_scale,
                               const void* zp, const void* pre_zp, void* y, bvq_stream_t stream) {
  int rc = var_validate("bvq_variant_fwd", d);
  if (rc) return rc;
  if (d->outer * d->channels * d->inner == 0) return BVQ_OK;
  if (!x || !scale || !y || (d->kind == BVQ_VAR_DECOUPLED && !pre_scale)) {
    set_error("bvq_variant_fwd: null pointer");
    return BVQ_ERR_INVALID;
  }
  VarArgs a = {};
  const void* ptrs[2] = {x, y};
  const int els[2] = {dtype_size(d->x_dtype), dtype_size(d->ct_dtype)};
  int vec;
  var_fill(a, d, ptrs, els, 2, vec);
  a.x = x;
  a.y = y;
  a.scale = scale;
  a.pre_scale = d->kind == BVQ_VAR_DECOUPLED ? pre_scale : nullptr;
  a.zp = zp;
  a.pre_zp = pre_zp;
  var_launch<false>(a, d, vec, (hipStream_t)stream);
  return check_launch("bvq_variant_fwd");
}

static int64_t var_units(const bvq_variant_desc* d) {
  const bool pc = d->scale_per_channel && d->channels > 1;
  const int64_t outer = pc ? d->outer : 1;
  const int32_t channels = pc ? (int32_t)d->channels : 1;
  const int64_t row_len = pc ? d->inner : d->outer * d->channels * d->inner;
  const int full = 16 / dtype_size(d->x_dtype);
  const int64_t u0 = make_tiling(outer, channels, row_len, full, 0, true).units;
  const int64_t u1 = make_tiling(outer, channels, row_len, 1, 0, true).units;
  return u0 > u1 ? u0 : u1;
}

extern "C" int64_t bvq_variant_bwd_workspace_bytes(const bvq_variant_desc* d) {
  if (var_validate("bvq_variant_bwd_workspace_bytes", d)) return -1;
  const int64_t units = var_units(d);
  const int64_t channels = (d->scale_per_channel && d->channels > 1) ? d->channels : 1;
  return ((2 * units * (int64_t)sizeof(float) + 7) / 8) * 8 + channel_sums_mid_bytes(units / channels + 1, channels) + 64;
}

extern "C" int bvq_variant_bwd(const bvq_variant_desc* d, const void* g, const void* x, const void* scale,
                               const void* pre_scale, const void* zp, const void* pre_zp, void* dx, float* dscale,
                               float* dpre_scale, void* workspace, int64_t workspace_bytes, bvq_stream_t stream) {
  int rc = var_validate("bvq_variant_bwd", d);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int64_t channels = (d->scale_per_channel && d->channels > 1) ? d->channels : 1;
  if (d->outer * d->channels * d->inner == 0) {
    if (dscale) (void)hipMemsetAsync(dscale, 0, sizeof(float) * channels, st);
    if (dpre_scale) (void)hipMemsetAsync(dpre_scale, 0, sizeof(float) * channels, st);
    return BVQ_OK;
  }
  if (!g || !x || !scale || !dx || (d->kind == BVQ_VAR_DECOUPLED && !pre_scale)) {
    set_error("bvq_variant_bwd: null pointer");
    return BVQ_ERR_INVALID;
  }
  if (dpre_scale && d->kind != BVQ_VAR_DECOUPLED) {
    set_error("bvq_variant_bwd: only the decoupled quantizer has a pre-scale");
    return BVQ_ERR_INVALID;
  }
  VarArgs a = {};
  const void* ptrs[3] = {x, g, dx};
  const int els[3] = {dtype_size(d->x_dtype), dtype_size(d->ct_dtype), dtype_size(d->x_dtype)};
  int vec;
  var_fill(a, d, ptrs, els, 3, vec);
  int64_t mid_off = 0;
  if (dscale || dpre_scale) {
    mid_off = ((2 * a.t.units * (int64_t)sizeof(float) + 7) / 8) * 8;
    const int64_t need = mid_off + channel_sums_mid_bytes(a.t.nob * a.t.ppr, channels);
    if (!workspace || workspace_bytes < need) {
      set_error("bvq_variant_bwd: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
      return BVQ_ERR_WORKSPACE;
    }
    a.part_a = dscale ? reinterpret_cast<float*>(workspace) : nullptr;
    a.part_b = dpre_scale ? reinterpret_cast<float*>(workspace) + a.t.units : nullptr;
  }
  a.x = x;
  a.g = g;
  a.y = dx;
  a.scale = scale;
  a.pre_scale = d->kind == BVQ_VAR_DECOUPLED ? pre_scale : nullptr;
  a.zp = zp;
  a.pre_zp = pre_zp;
  var_launch<true>(a, d, vec, st);
  rc = check_launch("bvq_variant_bwd");
  if (rc) return rc;
  if (dscale || dpre_scale) {
    launch_channel_sums(a.part_a, a.part_b, dscale, dpre_scale, a.t.nob, (int32_t)channels, a.t.ppr,
                        reinterpret_cast<char*>(workspace) + mid_off, st);
    rc = check_launch("bvq_variant_bwd/channel_sum");
  }
  return rc;
}
