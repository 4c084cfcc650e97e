// bvq_fakequant_bwd.hip -- entry points of the quantizer backward (include/bvq.h) and the launches of its column-mapped
// and finishing kernels; the row-mapped kernel is instantiated per dtype family in bvq_fakequant_bwd_{bf16,f16,f32}.hip.
#include "bvq_fakequant_bwd.h"

namespace bvq {
extern template BVQ_LAUNCH_BWD(float, float);
extern template BVQ_LAUNCH_BWD(bf16_t, bf16_t);
extern template BVQ_LAUNCH_BWD(bf16_t, float);
extern template BVQ_LAUNCH_BWD(f16_t, f16_t);
extern template BVQ_LAUNCH_BWD(f16_t, float);
}  // namespace bvq

using namespace bvq;

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
  const ColsPlan cp = cols_quant_plan(d, nullptr, nullptr, nullptr, false, true);
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
    const ColsPlan cp = cols_quant_plan(d, x, g, dx, !dscale && !tie_stat, true);
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
      BVQ_COLS_LAUNCH_G(fakequant_bwd_cols_kernel, ca, cnt, st, cp.units);
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
  const ColsPlan cp = cols_quant_plan(d, nullptr, nullptr, nullptr, false, true);
  if (cp.ok) {  // column-mapped partials: [prows][L] plus their fold [L]; first positions [L]
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
  const ColsPlan cp = cols_quant_plan(d, nullptr, nullptr, nullptr, false, true);
  if (cp.ok)  // float partial rows and their fold, then one first position per column
    return units * (int64_t)sizeof(float) + cp.L * (int64_t)sizeof(unsigned long long) + 256;
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
    const ColsPlan cp = cols_quant_plan(d, x, g, dx, false, true);
    const ColsPlan sized = cols_quant_plan(d, nullptr, nullptr, nullptr, false, true);
    if (sized.ok && !cp.ok) {
      set_error("bvq_fakequant_bwd_stats: the column-mapped route needs 16-byte aligned x, g and dx");
      return BVQ_ERR_UNSUPPORTED;
    }
    if (cp.ok) {
      const int64_t words = (cp.prows + cols_fold_scratch_rows(cp.prows)) * cp.L;
      const int64_t pos_off_c = ((words * (int64_t)sizeof(float) + 7) / 8) * 8;
      if (workspace_bytes < pos_off_c + cp.L * (int64_t)sizeof(unsigned long long)) {
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
      // first positions attaining the statistic: one entry per column, taken by atomic min from the few lanes that see it
      ca.pos_part = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(workspace) + pos_off_c);
      ca.tie_stat = stat;
      launch_tie_init(ca.pos_part, cp.L, st);
      BVQ_COLS_LAUNCH_G(fakequant_bwd_cols_kernel, ca, nt, st, cp.units);
      rc = check_launch("bvq_fakequant_bwd_stats/cols");
      if (rc) return rc;
      float* ds_fold = nullptr;
      unsigned long long* pos_fold = ca.pos_part;
      launch_cols_fold_sum_min(ca.ds_part, nullptr, cp.prows, cp.L, ca.ds_part + cp.prows * cp.L, nullptr, &ds_fold,
                               nullptr, st);
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
  if (arrive && !bwd_arrive_covers(vec, full, d->round_mode)) arrive = nullptr;  // two launches for the rare forms
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

