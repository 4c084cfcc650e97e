// bvq_fakequant.hip -- fused affine quantize/dequantize forward and its autograd backward.
//
// Replaces the ~9 full-tensor ATen passes of IntQuant.forward (B/core/quant/int_base.py:63-97)
// with one read of x and one write of y, and the ~8 passes autograd runs for its backward with one
// read of g, one read of x and one write of dx (the per-channel scale / zero-point gradient sums
// ride on the same reads).  HBM-bound: algorithmic bytes per element are
//   forward  sizeof(x) + sizeof(y)          backward  sizeof(g) + sizeof(x) + sizeof(dx).
#include "bvq_quant_math.h"

namespace bvq {

struct QuantArgs {
  Tiling t;
  const void* x;
  const void* scale;
  const void* zp;
  void* y;          // fwd: output; bwd: dx
  int32_t* codes;   // fwd only, nullable
  const void* g;    // bwd only
  float* ds_part;   // bwd only, per-unit partial of dscale (nullable)
  float* dzp_part;  // bwd only, per-unit partial of dzp (nullable)
  float qmin, qmax;
  int32_t scale_dtype, zp_dtype;
  int32_t scale_pc, zp_pc;
  int32_t scalar_cast;
  int32_t clamp_ste;
  int32_t out_int;
};

constexpr int kUnroll = 4;  // 16-byte loads in flight per lane before arithmetic starts

struct UnitInfo {
  int64_t start;  // first element
  int64_t len;    // elements in this unit
  int32_t channel;
  bool valid;
};

__device__ __forceinline__ UnitInfo locate_unit(const Tiling& t) {
  UnitInfo u;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  u.valid = unit < t.units;
  if (!u.valid) {
    u.start = 0;
    u.len = 0;
    u.channel = 0;
    return u;
  }
  const int64_t row = unit / t.ppr;
  const int64_t piece = unit - row * t.ppr;
  const int64_t off = piece * t.piece_len;
  u.start = row * t.row_len + off;
  const int64_t rest = t.row_len - off;
  u.len = rest < t.piece_len ? rest : t.piece_len;
  u.channel = (int32_t)(row % t.channels);
  return u;
}

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

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <typename XT, typename CT, int VEC, int RM>
__global__ __launch_bounds__(kBlock) void fakequant_fwd_kernel(QuantArgs a) {
  const UnitInfo u = locate_unit(a.t);
  if (!u.valid) return;
  const int lane = threadIdx.x & 63;
  float s, z;
  load_scale_zp<CT>(a, u.channel, s, z);
  const float qmin = a.qmin, qmax = a.qmax;

  const XT* __restrict__ xp = reinterpret_cast<const XT*>(a.x) + u.start;
  CT* __restrict__ yp = reinterpret_cast<CT*>(a.y) + u.start;
  int32_t* __restrict__ cp = a.codes ? a.codes + u.start : nullptr;
  const bool out_int = a.out_int != 0;

  const int64_t nvec = u.len / VEC;
  for (int64_t base = 0; base < nvec; base += (int64_t)kWave * kUnroll) {
    vec_t<XT, VEC> xv[kUnroll];
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      const int64_t i = base + (int64_t)j * kWave + lane;
      if (i < nvec) xv[j] = load_vec<XT, VEC>(xp + i * VEC);
    }
#pragma unroll
    for (int j = 0; j < kUnroll; ++j) {
      const int64_t i = base + (int64_t)j * kWave + lane;
      if (i < nvec) {
        vec_t<CT, VEC> yv;
        vec_t<int32_t, VEC> cv;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const float xf = to_f<XT>(xv[j].v[k]);
          const float q = quant_to_int<CT, RM>(xf, s, z, qmin, qmax);
          cv.v[k] = (int32_t)q;
          yv.v[k] = from_f<CT>(out_int ? q : dequant<CT>(q, s, z));
        }
        store_vec<CT, VEC>(yp + i * VEC, yv);
        if (cp) store_vec<int32_t, VEC>(cp + i * VEC, cv);
      }
    }
  }
  // ragged end (only the last piece of a single-row tensor can have one)
  const int64_t tail0 = nvec * VEC;
  const int64_t i = tail0 + lane;
  if (i < u.len) {
    const float xf = to_f<XT>(xp[i]);
    const float q = quant_to_int<CT, RM>(xf, s, z, qmin, qmax);
    yp[i] = from_f<CT>(out_int ? q : dequant<CT>(q, s, z));
    if (cp) cp[i] = (int32_t)q;
  }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// Autograd of the forward chain:  round_ste passes the gradient; TensorClamp masks
// clipped positions (TensorClampSte passes them); x/scale and y*scale give
//   dx     = (pass ? g*scale : 0) / scale
//   dscale = sum g*(q - zp)  -  sum dt * ((x/scale)/scale)        (torch: -grad * ((a/b)/b))
//   dzp    = sum dt  -  sum g*scale
template <typename CT, int RM, bool NEED_SUMS>
__device__ __forceinline__ float bwd_elem(float xf, float gf, float s, float z, float qmin, float qmax,
                                          bool clamp_ste, float& ds_acc, float& dzp_acc) {
  const float t1 = rnd<CT>(xf / s);
  const float t2 = rnd<CT>(t1 + z);
  const float t3 = round_op<CT, RM>(t2);
  const bool hi = t3 > qmax;
  float t4 = hi ? qmax : t3;
  const bool lo = t4 < qmin;
  t4 = lo ? qmin : t4;
  const bool pass = clamp_ste || !(hi || lo);
  const float gs = rnd<CT>(gf * s);
  const float dt = pass ? gs : 0.f;
  const float dxv = rnd<CT>(dt / s);
  if constexpr (NEED_SUMS) {
    const float t5 = rnd<CT>(t4 - z);
    const float term1 = rnd<CT>(gf * t5);
    const float term2 = rnd<CT>(-dt * rnd<CT>(t1 / s));
    ds_acc += term1;
    ds_acc += term2;
    dzp_acc += dt - gs;
  }
  return dxv;
}

template <typename XT, typename CT, int VEC, int RM, bool NEED_SUMS>
__global__ __launch_bounds__(kBlock) void fakequant_bwd_kernel(QuantArgs a) {
  const UnitInfo u = locate_unit(a.t);
  if (!u.valid) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float s, z;
  load_scale_zp<CT>(a, u.channel, s, z);
  const float qmin = a.qmin, qmax = a.qmax;
  const bool clamp_ste = a.clamp_ste != 0;

  const XT* __restrict__ xp = reinterpret_cast<const XT*>(a.x) + u.start;
  const CT* __restrict__ gp = reinterpret_cast<const CT*>(a.g) + u.start;
  XT* __restrict__ dxp = reinterpret_cast<XT*>(a.y) + u.start;

  float ds_acc = 0.f, dzp_acc = 0.f;
  constexpr int kU = kUnroll / 2 > 0 ? kUnroll / 2 : 1;  // two input streams
  const int64_t nvec = u.len / VEC;
  for (int64_t base = 0; base < nvec; base += (int64_t)kWave * kU) {
    vec_t<XT, VEC> xv[kU];
    vec_t<CT, VEC> gv[kU];
#pragma unroll
    for (int j = 0; j < kU; ++j) {
      const int64_t i = base + (int64_t)j * kWave + lane;
      if (i < nvec) {
        xv[j] = load_vec<XT, VEC>(xp + i * VEC);
        gv[j] = load_vec<CT, VEC>(gp + i * VEC);
      }
    }
#pragma unroll
    for (int j = 0; j < kU; ++j) {
      const int64_t i = base + (int64_t)j * kWave + lane;
      if (i < nvec) {
        vec_t<XT, VEC> dv;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const float d = bwd_elem<CT, RM, NEED_SUMS>(to_f<XT>(xv[j].v[k]), to_f<CT>(gv[j].v[k]), s, z,
                                                      qmin, qmax, clamp_ste, ds_acc, dzp_acc);
          dv.v[k] = from_f<XT>(d);
        }
        store_vec<XT, VEC>(dxp + i * VEC, dv);
      }
    }
  }
  const int64_t i = nvec * VEC + lane;
  if (i < u.len) {
    const float d = bwd_elem<CT, RM, NEED_SUMS>(to_f<XT>(xp[i]), to_f<CT>(gp[i]), s, z, qmin, qmax,
                                                clamp_ste, ds_acc, dzp_acc);
    dxp[i] = from_f<XT>(d);
  }
  if constexpr (NEED_SUMS) {
    const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    ds_acc = wave_sum(ds_acc);
    dzp_acc = wave_sum(dzp_acc);
    if (lane == 0) {
      if (a.ds_part) a.ds_part[unit] = ds_acc;
      if (a.dzp_part) a.dzp_part[unit] = dzp_acc;
    }
  }
}

// Combine per-unit partial sums of one channel in a fixed order (double accumulation):
// channel c owns units (o*channels + c)*ppr + p for o in [0, outer), p in [0, ppr).
__global__ __launch_bounds__(kBlock) void channel_sum_kernel(const float* __restrict__ part0,
                                                             const float* __restrict__ part1,
                                                             float* __restrict__ out0,
                                                             float* __restrict__ out1, int64_t outer,
                                                             int32_t channels, int64_t ppr) {
  __shared__ double sh[2][kBlock];
  const int32_t c = blockIdx.x;
  const int64_t n = outer * ppr;
  double acc0 = 0.0, acc1 = 0.0;
  for (int64_t k = threadIdx.x; k < n; k += kBlock) {
    const int64_t o = k / ppr, p = k - o * ppr;
    const int64_t unit = (o * channels + c) * ppr + p;
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
    if (out0) out0[c] = (float)sh[0][0];
    if (out1) out1[c] = (float)sh[1][0];
  }
}

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
  if ((d->scale_per_channel || d->zp_per_channel) && d->channels == 1) {
    // harmless, but keep descriptors canonical
  }
  return BVQ_OK;
}

// rows/row_len of the descriptor: per-tensor quantizers are one long row
static void rows_of(const bvq_quant_desc* d, int64_t& rows, int64_t& row_len, int32_t& channels) {
  const bool pc = (d->scale_per_channel || d->zp_per_channel) && d->channels > 1;
  if (pc) {
    rows = d->outer * d->channels;
    row_len = d->inner;
    channels = (int32_t)d->channels;
  } else {
    rows = 1;
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
}

template <typename XT, typename CT, int VEC>
static void launch_fwd_rm(const QuantArgs& a, int rm, hipStream_t st) {
  const dim3 grid(grid_for_units(a.t.units)), block(kBlock);
  switch (rm) {
    case BVQ_ROUND:
      fakequant_fwd_kernel<XT, CT, VEC, BVQ_ROUND><<<grid, block, 0, st>>>(a);
      break;
    case BVQ_FLOOR:
      fakequant_fwd_kernel<XT, CT, VEC, BVQ_FLOOR><<<grid, block, 0, st>>>(a);
      break;
    case BVQ_CEIL:
      fakequant_fwd_kernel<XT, CT, VEC, BVQ_CEIL><<<grid, block, 0, st>>>(a);
      break;
    case BVQ_ROUND_TO_ZERO:
      fakequant_fwd_kernel<XT, CT, VEC, BVQ_ROUND_TO_ZERO><<<grid, block, 0, st>>>(a);
      break;
    default:
      fakequant_fwd_kernel<XT, CT, VEC, BVQ_DPU_ROUND><<<grid, block, 0, st>>>(a);
      break;
  }
}

template <typename XT, typename CT>
static void launch_fwd(const QuantArgs& a, int vec, int rm, hipStream_t st) {
  constexpr int V = elem<XT>::vec;
  if (vec == V)
    launch_fwd_rm<XT, CT, V>(a, rm, st);
  else if (vec == 2 && V > 2)
    launch_fwd_rm<XT, CT, 2>(a, rm, st);
  else
    launch_fwd_rm<XT, CT, 1>(a, rm, st);
}

template <typename XT, typename CT, int VEC, bool NS>
static void launch_bwd_rm(const QuantArgs& a, int rm, hipStream_t st) {
  const dim3 grid(grid_for_units(a.t.units)), block(kBlock);
  switch (rm) {
    case BVQ_ROUND:
      fakequant_bwd_kernel<XT, CT, VEC, BVQ_ROUND, NS><<<grid, block, 0, st>>>(a);
      break;
    case BVQ_FLOOR:
      fakequant_bwd_kernel<XT, CT, VEC, BVQ_FLOOR, NS><<<grid, block, 0, st>>>(a);
      break;
    case BVQ_CEIL:
      fakequant_bwd_kernel<XT, CT, VEC, BVQ_CEIL, NS><<<grid, block, 0, st>>>(a);
      break;
    case BVQ_ROUND_TO_ZERO:
      fakequant_bwd_kernel<XT, CT, VEC, BVQ_ROUND_TO_ZERO, NS><<<grid, block, 0, st>>>(a);
      break;
    default:
      fakequant_bwd_kernel<XT, CT, VEC, BVQ_DPU_ROUND, NS><<<grid, block, 0, st>>>(a);
      break;
  }
}

template <typename XT, typename CT>
static void launch_bwd(const QuantArgs& a, int vec, int rm, bool need_sums, hipStream_t st) {
  constexpr int V = elem<XT>::vec;
  if (need_sums) {
    if (vec == V)
      launch_bwd_rm<XT, CT, V, true>(a, rm, st);
    else if (vec == 2 && V > 2)
      launch_bwd_rm<XT, CT, 2, true>(a, rm, st);
    else
      launch_bwd_rm<XT, CT, 1, true>(a, rm, st);
  } else {
    if (vec == V)
      launch_bwd_rm<XT, CT, V, false>(a, rm, st);
    else if (vec == 2 && V > 2)
      launch_bwd_rm<XT, CT, 2, false>(a, rm, st);
    else
      launch_bwd_rm<XT, CT, 1, false>(a, rm, st);
  }
}

// vec widths actually instantiated: full (16 B of x), 2 and 1
static int snap_vec(int vec, int full) { return vec == full ? full : (vec >= 2 && full > 2 ? 2 : 1); }

}  // namespace bvq

using namespace bvq;

extern "C" int bvq_fakequant_fwd(const bvq_quant_desc* d, const void* x, const void* scale,
                                 const void* zp, void* y, int32_t* codes, bvq_stream_t stream) {
  int rc = validate(d);
  if (rc) return rc;
  const int64_t n = d->outer * d->channels * d->inner;
  if (n == 0) return BVQ_OK;
  if (!x || !scale || !zp || !y) {
    set_error("bvq_fakequant_fwd: null pointer");
    return BVQ_ERR_INVALID;
  }
  int64_t rows, row_len;
  int32_t channels;
  rows_of(d, rows, row_len, channels);
  const void* ptrs[3] = {x, y, codes};
  const int els[3] = {dtype_size(d->x_dtype), dtype_size(d->ct_dtype), 4};
  const int full = 16 / dtype_size(d->x_dtype);
  const int vec = snap_vec(pick_vec(full, rows, row_len, ptrs, els, 3), full);
  QuantArgs a = {};
  a.t = make_tiling(rows, row_len, channels, vec);
  a.x = x;
  a.scale = scale;
  a.zp = zp;
  a.y = y;
  a.codes = codes;
  fill_args(a, d);
  hipStream_t st = (hipStream_t)stream;
  if (d->x_dtype == BVQ_F32)
    launch_fwd<float, float>(a, vec, d->round_mode, st);
  else if (d->x_dtype == BVQ_BF16 && d->ct_dtype == BVQ_BF16)
    launch_fwd<bf16_t, bf16_t>(a, vec, d->round_mode, st);
  else if (d->x_dtype == BVQ_BF16)
    launch_fwd<bf16_t, float>(a, vec, d->round_mode, st);
  else if (d->x_dtype == BVQ_F16 && d->ct_dtype == BVQ_F16)
    launch_fwd<f16_t, f16_t>(a, vec, d->round_mode, st);
  else
    launch_fwd<f16_t, float>(a, vec, d->round_mode, st);
  return check_launch("bvq_fakequant_fwd");
}

static int64_t bwd_units(const bvq_quant_desc* d) {
  int64_t rows, row_len;
  int32_t channels;
  rows_of(d, rows, row_len, channels);
  // upper bound over every vector width the launcher may pick
  int64_t worst = 0;
  const int full = 16 / dtype_size(d->x_dtype);
  for (int v = 1; v <= full; v <<= 1) {
    Tiling t = make_tiling(rows, row_len, channels, v);
    if (t.units > worst) worst = t.units;
  }
  return worst;
}

extern "C" int64_t bvq_fakequant_bwd_workspace_bytes(const bvq_quant_desc* d) {
  if (validate(d)) return -1;
  return 2 * bwd_units(d) * (int64_t)sizeof(float) + 256;
}

extern "C" int bvq_fakequant_bwd(const bvq_quant_desc* d, const void* g, const void* x,
                                 const void* scale, const void* zp, void* dx, float* dscale,
                                 float* dzp, void* workspace, int64_t workspace_bytes,
                                 bvq_stream_t stream) {
  int rc = validate(d);
  if (rc) return rc;
  const int64_t n = d->outer * d->channels * d->inner;
  hipStream_t st = (hipStream_t)stream;
  int64_t rows, row_len;
  int32_t channels;
  rows_of(d, rows, row_len, channels);
  const bool need_sums = dscale != nullptr || dzp != nullptr;
  if (n == 0) {
    if (dscale) (void)hipMemsetAsync(dscale, 0, sizeof(float) * channels, st);
    if (dzp) (void)hipMemsetAsync(dzp, 0, sizeof(float) * channels, st);
    return BVQ_OK;
  }
  if (!g || !x || !scale || !zp || !dx) {
    set_error("bvq_fakequant_bwd: null pointer");
    return BVQ_ERR_INVALID;
  }
  const void* ptrs[3] = {x, g, dx};
  const int els[3] = {dtype_size(d->x_dtype), dtype_size(d->ct_dtype), dtype_size(d->x_dtype)};
  const int full = 16 / dtype_size(d->x_dtype);
  const int vec = snap_vec(pick_vec(full, rows, row_len, ptrs, els, 3), full);
  QuantArgs a = {};
  a.t = make_tiling(rows, row_len, channels, vec);
  if (need_sums) {
    const int64_t need = 2 * a.t.units * (int64_t)sizeof(float);
    if (!workspace || workspace_bytes < need) {
      set_error("bvq_fakequant_bwd: workspace %lld < %lld bytes", (long long)workspace_bytes,
                (long long)need);
      return BVQ_ERR_WORKSPACE;
    }
    a.ds_part = reinterpret_cast<float*>(workspace);
    a.dzp_part = a.ds_part + a.t.units;
  }
  a.x = x;
  a.g = g;
  a.scale = scale;
  a.zp = zp;
  a.y = dx;
  fill_args(a, d);
  if (d->x_dtype == BVQ_F32)
    launch_bwd<float, float>(a, vec, d->round_mode, need_sums, st);
  else if (d->x_dtype == BVQ_BF16 && d->ct_dtype == BVQ_BF16)
    launch_bwd<bf16_t, bf16_t>(a, vec, d->round_mode, need_sums, st);
  else if (d->x_dtype == BVQ_BF16)
    launch_bwd<bf16_t, float>(a, vec, d->round_mode, need_sums, st);
  else if (d->x_dtype == BVQ_F16 && d->ct_dtype == BVQ_F16)
    launch_bwd<f16_t, f16_t>(a, vec, d->round_mode, need_sums, st);
  else
    launch_bwd<f16_t, float>(a, vec, d->round_mode, need_sums, st);
  rc = check_launch("bvq_fakequant_bwd");
  if (rc) return rc;
  if (need_sums) {
    // per-tensor quantizers have one "channel" spanning every unit
    const int64_t outer_rows = rows / channels;
    channel_sum_kernel<<<dim3((unsigned)channels), dim3(kBlock), 0, st>>>(
        dscale ? a.ds_part : nullptr, dzp ? a.dzp_part : nullptr, dscale, dzp, outer_rows, channels,
        a.t.ppr);
    rc = check_launch("bvq_fakequant_bwd/channel_sum");
  }
  return rc;
}
