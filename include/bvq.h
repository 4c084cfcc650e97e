/*
 * bvq.h -- C ABI of libbvq.so, the MI355X (gfx950) fake-quantization engine.
 *
 * This is the drop-in boundary for Brevitas' per-tensor / per-channel integer
 * quant-dequant hot path.  Nothing like it exists in the reference (it has no
 * native kernels besides 12 ATen-forwarding STE ops); every entry point below
 * names the reference interface whose tensor math it replaces.  Paths are
 * relative to the reference checkout, `B/` = `src/brevitas/`.
 *
 * Conventions
 *   - plain C, no torch types: raw device pointers, int64 sizes, a hipStream_t
 *     passed as void*.  The library owns no memory across calls: every output
 *     and every workspace is allocated by the caller (B/ ownership model: each
 *     op returns a fresh tensor from the host framework's allocator).
 *   - all pointers are DEVICE pointers unless the name says `host`.
 *   - every function is re-entrant, never synchronises the device, never
 *     allocates, and only enqueues work on `stream` (autograd runs backward on
 *     its own thread: B/ has no threads of its own, SURVEY 8b "Threading").
 *   - return 0 on success, a negative bvq_status otherwise; the message for the
 *     calling thread's last failure is bvq_last_error().  Never throws.
 *   - a tensor is described as [outer, channels, inner] (row-major,
 *     contiguous).  channels == 1 means per-tensor.  A weight [Cout,Cin,kh,kw]
 *     quantized per output channel is (1, Cout, Cin*kh*kw); an NCHW activation
 *     quantized per channel is (N, C, H*W) -- no permute copy is ever made
 *     (the reference does permute().contiguous(), B/core/function_wrapper/shape.py:19-27).
 */
#ifndef BVQ_H_
#define BVQ_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BVQ_ABI_VERSION 2

typedef void* bvq_stream_t; /* hipStream_t */

typedef enum bvq_status {
  BVQ_OK = 0,
  BVQ_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, bad enum) */
  BVQ_ERR_UNSUPPORTED = -2, /* dtype / layout combination not built */
  BVQ_ERR_WORKSPACE = -3,   /* workspace too small */
  BVQ_ERR_LAUNCH = -4       /* HIP reported a launch error */
} bvq_status;

/* element types of tensors ("dtype" is the storage/arithmetic type, not a precision claim) */
typedef enum bvq_dtype { BVQ_F32 = 0, BVQ_BF16 = 1, BVQ_F16 = 2 } bvq_dtype;

/* float_to_int_impl variants: B/core/function_wrapper/ops_ste.py:14-76 */
typedef enum bvq_round_mode {
  BVQ_ROUND = 0,         /* torch.round, half to even  (round_ste,  B/function/ops_ste.py:46-67)  */
  BVQ_FLOOR = 1,         /* torch.floor                (floor_ste,  :94-115)                      */
  BVQ_CEIL = 2,          /* torch.ceil                 (ceil_ste,   :70-91)                       */
  BVQ_ROUND_TO_ZERO = 3, /* sign(x)*floor(|x|)         (B/function/ops.py:37-53)                  */
  BVQ_DPU_ROUND = 4      /* DPU rounding               (B/function/ops.py:56-72)                  */
} bvq_round_mode;

/* elementwise ops of the STE namespace, forward math only
 * (B/ops/autograd_ste_ops.py:37-382, B/csrc/autograd_ste_ops.cpp:14-194) */
typedef enum bvq_unary_op {
  BVQ_OP_ROUND = 0,
  BVQ_OP_FLOOR = 1,
  BVQ_OP_CEIL = 2,
  BVQ_OP_ROUND_TO_ZERO = 3,
  BVQ_OP_DPU_ROUND = 4,
  BVQ_OP_BINARY_SIGN = 5,  /* (x>=0) - (x<0), +1 at 0   (B/function/ops.py:16-34) */
  BVQ_OP_TERNARY_SIGN = 6, /* torch.sign                                          */
  BVQ_OP_ABS = 7           /* torch.abs (forward of abs_binary_sign_grad)         */
} bvq_unary_op;

typedef enum bvq_stat_kind {
  BVQ_STAT_ABSMAX = 0, /* out[c]      = max |x|              (AbsMax,    B/core/stats/stats_op.py:129-141) */
  BVQ_STAT_MINMAX = 1  /* out[c]=max, out[channels+c]=min    (AbsMinMax, B/core/stats/stats_op.py:144-158) */
} bvq_stat_kind;

/* How a 0-dim (one element) scale / zero-point whose dtype is WIDER than the
 * compute dtype enters the arithmetic.  torch's own kernels differ here:
 *   OPMATH: the scalar keeps its float32 value (ATen CPU reduced-float scalar path);
 *   CAST  : the scalar is first rounded to the compute dtype (ATen device kernels).
 * Same-dtype operands and per-channel operands are unaffected. */
typedef enum bvq_scalar_mode { BVQ_SCALAR_OPMATH = 0, BVQ_SCALAR_CAST = 1 } bvq_scalar_mode;

/* Elementwise activation applied to x BEFORE the statistic / the quantizer, fused into the same
 * kernels: FusedActivationQuantProxy.forward = tensor_quant(activation_impl(x))
 * (B/proxy/runtime_quant.py:73-84).  RELU: torch.relu forward (NaN and -0.0 pass), backward
 * grad * (x > 0) (threshold_backward). */
typedef enum bvq_pre_op { BVQ_PRE_NONE = 0, BVQ_PRE_RELU = 1 } bvq_pre_op;

/* element type of the integer codes bvq_fakequant_fwd can emit: what QuantTensor.int() returns
 * (B/quant_tensor/__init__.py:174-187: int8 / uint8 up to 8 bits, int32 above) and what the QCDQ export
 * handlers quantize to (B/export/common/handler/qcdq.py:71-84) */
typedef enum bvq_codes_dtype { BVQ_CODES_I32 = 0, BVQ_CODES_I8 = 1, BVQ_CODES_U8 = 2 } bvq_codes_dtype;

/* output selection for bvq_fakequant_fwd */
#define BVQ_OUT_DEQUANT 0 /* y = (clamp(round(x/s+zp)) - zp) * s   IntQuant.forward, B/core/quant/int_base.py:86-97 */
#define BVQ_OUT_INT 1     /* y =  clamp(round(x/s+zp))             IntQuant.to_int,  B/core/quant/int_base.py:63-76 */

/*
 * Descriptor of one affine integer quantizer application.
 * Replaces the argument set of IntQuant.forward(scale, zero_point, bit_width, x)
 * (B/core/quant/int_base.py:86-97): bit_width only enters through
 * qmin = min_int(signed, narrow_range, bit_width), qmax = max_int(...)
 * (B/function/ops.py:132-191), computed on the host.
 */
typedef struct bvq_quant_desc {
  int64_t outer;    /* x viewed as [outer, channels, inner], contiguous */
  int64_t channels; /* 1 = per-tensor */
  int64_t inner;
  int32_t x_dtype;     /* dtype of x and of dx */
  int32_t ct_dtype;    /* torch.result_type(x, scale): dtype of every intermediate, of y and of g */
  int32_t scale_dtype; /* dtype of the scale buffer */
  int32_t zp_dtype;    /* dtype of the zero-point buffer */
  int32_t scale_per_channel; /* 0: one element, 1: `channels` elements */
  int32_t zp_per_channel;    /* 0: one element, 1: `channels` elements */
  float qmin;          /* integer clamp bounds, as floats (the reference keeps them as 0-dim float tensors) */
  float qmax;
  int32_t round_mode;  /* bvq_round_mode */
  int32_t scalar_mode; /* bvq_scalar_mode */
  int32_t clamp_ste;   /* backward only. 1: TensorClampSte (grad passes clipped elements, weights);
                          0: TensorClamp (grad masked where clipped, activations). B/core/quant/int_base.py:53-54 */
  int32_t out_kind;    /* forward only. BVQ_OUT_DEQUANT or BVQ_OUT_INT */
  int32_t pre_op;      /* bvq_pre_op applied to x first (forward and backward) */
  int32_t codes_dtype; /* forward only: element type of the optional integer-code output (bvq_codes_dtype) */
} bvq_quant_desc;

/* ---- library ------------------------------------------------------------------------------ */

int bvq_abi_version(void);
/* message of the calling thread's last failing call ("" if none) */
const char* bvq_last_error(void);

/* ---- elementwise STE namespace (seam 1) ----------------------------------------------------
 * Forward math of torch.ops.autograd_ste_ops.<name>_impl / brevitas.ops.autograd_ste_ops.<name>_impl
 * (registration B/csrc/autograd_ste_ops.cpp:258-271, aliases B/ops/autograd_ste_ops.py:385-431).
 * The straight-through backward of these ops is the identity and needs no kernel. */

/* y[i] = op(x[i]).  x may alias y (tensor_clamp_ste_-style in-place use). */
int bvq_unary(int op, int dtype, const void* x, void* y, int64_t n, bvq_stream_t stream);

/* torch.clamp(x, lo, hi) / torch.clamp_min(x, lo): scalar_clamp_ste_impl, scalar_clamp_min_ste_impl
 * (B/ops/autograd_ste_ops.py:37-97).  Bounds are host doubles rounded to `dtype` like torch does.
 * use_lo / use_hi select which bounds apply.  NaN propagates. */
int bvq_scalar_clamp(int dtype, const void* x, void* y, int64_t n, double lo, int use_lo, double hi,
                     int use_hi, bvq_stream_t stream);

/* tensor_clamp(x, min_val, max_val) = where(x>max,max,x) then where(out<min,min,out)
 * (B/function/ops.py:75-100; tensor_clamp_ste_impl forward, B/ops/autograd_ste_ops.py:100-128).
 * lo / hi are device buffers of dtype `dtype`: one element each when bounds_full == 0
 * (the hot path: 0-dim min_int/max_int), n elements each when bounds_full == 1. */
int bvq_tensor_clamp(int dtype, const void* x, const void* lo, const void* hi, int bounds_full, void* y,
                     int64_t n, bvq_stream_t stream);

/* backward of the non-STE tensor_clamp w.r.t. x (autograd of the two torch.where):
 * dx = (!(x>hi) && !(x<lo)) ? g : 0 */
int bvq_tensor_clamp_bwd(int dtype, const void* g, const void* x, const void* lo, const void* hi,
                         int bounds_full, void* dx, int64_t n, bvq_stream_t stream);

/* backward of abs_binary_sign_grad (B/ops/autograd_ste_ops.py:356-382): dx = binary_sign(x) * g */
int bvq_abs_binary_sign_grad_bwd(int dtype, const void* g, const void* x, void* dx, int64_t n,
                                 bvq_stream_t stream);

/* ---- statistics (seam 2: AbsMax / AbsMinMax) ------------------------------------------------ */

/* bytes of scratch bvq_stats needs for this problem (0 is a valid answer) */
int64_t bvq_stats_workspace_bytes(int kind, int dtype, int64_t outer, int64_t channels, int64_t inner);

/* Per-channel (or whole-tensor, channels == 1) statistic over the `outer` and `inner` axes of
 * x[outer, channels, inner].  One streaming read of x, no permuted copy.
 *   ABSMAX: out[c] = max |x|, NaN if any NaN (torch.max semantics)       -> `channels` elements
 *   MINMAX: out[c] = max x ; out[channels + c] = min x, NaN-propagating  -> 2*`channels` elements
 * out has dtype out_dtype: BVQ_F32 or the dtype of x (the values are exact in either). */
int bvq_stats(int kind, int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner,
              int out_dtype, void* out, void* workspace, int64_t workspace_bytes, bvq_stream_t stream);

/* the same statistic of pre_op(x) (bvq_pre_op), without materialising the activation */
int bvq_stats_pre(int kind, int pre_op, int dtype, const void* x, int64_t outer, int64_t channels,
                  int64_t inner, int out_dtype, void* out, void* workspace, int64_t workspace_bytes,
                  bvq_stream_t stream);

/* bvq_stats(ABSMAX) with the scale derivation of a stats-scaled quantizer in the same launches:
 *   stat_out[c]  = max |x|                                        (dtype of x)
 *   thr          = use_min ? clamp_min(stat, min_val) : stat      (scalar_clamp_min_ste, B/core/restrict_val.py:22-42;
 *                                                                  min_val is rounded to the dtype of x like torch does)
 *   scale_out[c] = thr / int_threshold, rounded to scale_dtype    (RescalingIntQuant.forward, B/core/quant/int.py:160)
 * int_threshold is the value the division sees (the caller applies torch's promotion of the 0-dim
 * int_threshold tensor); scale_dtype is the dtype torch gives the quotient.  pre_op: the statistic is
 * taken of pre_op(x) (bvq_pre_op). */
int bvq_absmax_scale(int pre_op, int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner,
                     void* stat_out, double min_val, int use_min, double int_threshold, int scale_dtype,
                     void* scale_out, void* workspace, int64_t workspace_bytes, bvq_stream_t stream);
/* bvq_absmax_scale whose finishing launch ALSO folds the statistic into _RuntimeStats' running average
 * (B/core/stats/stats_wrapper.py:61-66): first_batch: running *= stat; else running *= (1 - momentum);
 * running += momentum * stat -- the rounding points of bvq_running_stats_update, one launch fewer per step.
 * running: [channels] in run_dtype, updated in place. */
int bvq_absmax_scale_running(int pre_op, int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner,
                             void* stat_out, double min_val, int use_min, double int_threshold, int scale_dtype,
                             void* scale_out, int run_dtype, void* running, double momentum, int first_batch,
                             void* workspace, int64_t workspace_bytes, bvq_stream_t stream);

/* The same statistic (+ optional scale epilogue, + optional running average) in ONE launch, for per-channel
 * layouts whose rows the row-mapped kernels walk (bvq_absmax_onepass_supported: 1 / 0).  Persistent waves fold their
 * maxima into a per-channel key word with an atomic max and count the units they covered in a per-channel counter;
 * the wave whose count completes a channel finishes it (B/core/stats/stats_op.py:129-141 -> stat_out;
 * B/core/restrict_val.py:22-42 and B/core/quant/int.py:160 -> scale_out; B/core/stats/stats_wrapper.py:61-66 ->
 * running).  No wave waits for another.  Results are the bits of bvq_absmax_scale[_running] (a max is exact).
 * stat_dtype: BVQ_F32 (the batch-sharded route all-reduces it) or the dtype of x.  scale_out / running: nullable.
 * Not covered (bvq_absmax_onepass_supported: 0): a whole-tensor statistic and per-channel layouts with more than 32
 * units per channel -- arrivals at one word serialise (~0.25 us each) and soon cost more than the finishing launch
 * they would save.
 * arrive: `arrive_words` >= 2 * channels uint32 words in device memory that are ALL ZERO when the launch starts;
 *   the finishing waves hand them back as zeros, so a caller keeps ONE such buffer per stream, cleared once when
 *   it is allocated, and never clears it again (launches on one stream are ordered; do not share it between streams).
 * BVQ_ERR_UNSUPPORTED: layout not covered (per-tensor, column-mapped): call bvq_absmax_scale[_running]. */
int bvq_absmax_onepass_supported(int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner);
int bvq_absmax_scale_onepass(int pre_op, int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner,
                             int stat_dtype, void* stat_out, double min_val, int use_min, double int_threshold,
                             int scale_dtype, void* scale_out, int run_dtype, void* running, double momentum,
                             int first_batch, uint32_t* arrive, int64_t arrive_words, bvq_stream_t stream);

/* AbsMax of a LIST of tensors that share the channel axis, + the scale epilogue, in ONE launch: the statistic of
 * _ParameterListStats with several tracked parameters (B/core/stats/stats_wrapper.py:83-114 -- a weight quantizer shared
 * by several layers reduces torch.cat of their weights' views; B/core/scaling/standalone.py StatsFromParameterScaling).
 * Tensor i is contiguous [outers[i], channels, inners[i]] of `dtype` (per output channel: outers = 1, inners = the
 * row of channel c; whole tensor: channels = 1, inners = numel).  No concatenation is materialised: the launch's waves
 * are dealt to the tensors, and the last wave to arrive at a channel's words finishes that channel (contract of the
 * arrival buffer: bvq_absmax_scale_onepass).  stat_out [channels] in `dtype`; scale_out nullable.  The bits are those
 * of the reference's reduction over the concatenation (a max is exact).
 * A whole-tensor statistic (channels = 1) leaves one partial per unit in `workspace` (>= 16 KiB) and takes a second,
 * finishing launch instead of the arrivals: every unit of every tensor would arrive at ONE pair of words, and
 * agent-scope atomics on one address serialise.  `arrive` may be null then; `workspace` may be null for channels > 1.
 * bvq_absmax_list_supported: 1 / 0 -- 1..8 tensors, at most 32 units per channel over the whole list.
 * BVQ_ERR_UNSUPPORTED otherwise: reduce the tensors one by one. */
int bvq_absmax_list_supported(int dtype, int n, const void* const* xs, const int64_t* outers, int64_t channels,
                              const int64_t* inners);
int bvq_absmax_scale_list(int dtype, int n, const void* const* xs, const int64_t* outers, int64_t channels,
                          const int64_t* inners, void* stat_out, double min_val, int use_min, double int_threshold,
                          int scale_dtype, void* scale_out, uint32_t* arrive, int64_t arrive_words,
                          void* workspace, int64_t workspace_bytes, bvq_stream_t stream);

/* Running average kept by _RuntimeStats (B/core/stats/stats_wrapper.py:61-66), one launch:
 *   first_batch: running *= stat ; otherwise running *= (1 - momentum); running += momentum * stat
 * with torch's rounding points (in-place results in run_dtype, momentum * stat in stat_dtype). */
int bvq_running_stats_update(int run_dtype, void* running, int stat_dtype, const void* stat, int64_t n,
                             double momentum, int first_batch, bvq_stream_t stream);

/* ---- batch-sharded tensors (one process per GPU, the activation split along the batch) ----------------
 * The two collectives of brevitas_amd/distributed.py carry small messages; these entry points build and consume
 * them in one launch each instead of a dozen scale-shaped torch ops.
 * bvq_scale_from_stat: the all-reduced (MAX) float32 statistic -> statistic in `stat_dtype` and
 *   scale = clamp_min(stat, min_val) / int_threshold in `scale_dtype` (rounding points of bvq_absmax_scale).
 * bvq_shard_pack: this shard's float64 [2][channels] message for the backward all-gather: its dscale sums and,
 *   per channel, `rank` if it holds an element attaining the statistic (tie_info[c] >= 0) else 2^30
 *   (per_channel = 0: the number of ties it holds, tie_info[0]).
 * bvq_shard_unpack: from the gathered [world][2][channels] messages: dscale_total (double sum in rank order:
 *   the same bits on every rank) and, per_channel, tie_info[c] = -1 for every channel whose lowest claiming rank
 *   is another shard; per_channel = 0: total_ties[0] = ties over all shards. */
int bvq_scale_from_stat(const float* stat32, int64_t channels, int stat_dtype, void* stat_out, double min_val,
                        int use_min, double int_threshold, int scale_dtype, void* scale_out, bvq_stream_t stream);
/* bvq_scale_from_stat with _RuntimeStats' running average folded in (the rounding points of bvq_absmax_scale_running);
 * running: nullable. */
int bvq_scale_from_stat_running(const float* stat32, int64_t channels, int stat_dtype, void* stat_out, double min_val,
                                int use_min, double int_threshold, int scale_dtype, void* scale_out, int run_dtype,
                                void* running, double momentum, int first_batch, bvq_stream_t stream);
/* The stats-scaled backward of ONE BATCH SHARD (per-channel layouts of bvq_fakequant_bwd_stats): dx WITHOUT the deposit,
 * and this shard's message for the backward all-gather, float64 [2][channels]: row 0 = the channel's dscale sum kept
 * in double (the shards' sums are added in double and rounded to float32 once: bvq_shard_unpack_deposit), row 1 = its
 * claim on the channel's deposit (`rank`, or 2^30 when no element of the shard attains the statistic); first_pos[c] =
 * the shard's first arg-max position (outer * inner + i; -1: none).  arrive (nullable): the arrival buffer of
 * bvq_fakequant_bwd_stats_onepass -- the streaming kernel's last-arriving wave per channel then writes the message
 * (one launch); without it, or on column-mapped layouts, a one-wave-per-channel launch follows the streaming kernel.
 * workspace: bvq_fakequant_bwd_stats_workspace_bytes.
 * bvq_shard_unpack_deposit: from the gathered [world][2][channels] messages, per channel: dscale_total (nullable out)
 * = the double sum over the shards in rank order, rounded once (the same bits on every rank); on the shard that owns
 * the deposit (lowest claiming rank) dscale -> statistic's gradient (B/core/quant/int.py:160 backward, rounding points
 * of bvq_fakequant_bwd_stats) deposited on dx at first_pos[c].  One launch instead of unpack + cast + divide + deposit. */
int bvq_fakequant_bwd_shard(const bvq_quant_desc* d, const void* g, const void* x, const void* scale, const void* zp,
                            const void* stat, void* dx, double* message, int64_t* first_pos, int rank, void* workspace,
                            int64_t workspace_bytes, uint32_t* arrive, int64_t arrive_words, bvq_stream_t stream);
int bvq_shard_unpack_deposit(int dtype, const void* x, void* dx, const double* gathered, int world, int64_t channels,
                             int rank, const int64_t* first_pos, int64_t inner, int scale_dtype, double int_threshold,
                             int quot_dtype, int pre_op, float* dscale_total, bvq_stream_t stream);
int bvq_shard_pack(const float* dscale, const int64_t* tie_info, int64_t channels, int rank, int per_channel,
                   double* message, bvq_stream_t stream);
int bvq_shard_unpack(const double* gathered, int world, int64_t channels, int rank, int per_channel,
                     float* dscale_total, int64_t* tie_info, int64_t* total_ties, bvq_stream_t stream);

/* ---- histogram (KLMinimizerThreshold, B/core/stats/stats_op.py:280-350) -----------------------------------
 * counts[b] = number of elements of x in bin b of torch.histc(x, bins, min=-absmax, max=absmax) -- equal-width
 * bins, values outside the range ignored, bin = (int)((v + absmax) * bins / (2 absmax)) in float32, the value
 * +absmax in the last bin.  absmax: ONE element on the device in x's dtype (the AbsMax statistic: no host
 * sync); counts: int32 [bins], overwritten; 1 <= bins <= 8192.  One streaming read of x. */
int bvq_histc(int dtype, const void* x, int64_t n, const void* absmax, int bins, int32_t* counts,
              bvq_stream_t stream);

/* ---- moment statistics ---------------------------------------------------------------------------
 * sums is float32 [3 * channels]: sums[c] = SUM d, sums[channels + c] = SUM d^2 over the `outer` and `inner`
 * axes with d = |x| - p[c], and sums[2 * channels + c] = p[c], the channel's pivot (|x| of its first element,
 * 0 if that is not finite); per-unit float32 partials, double accumulation across units, fixed order.  The
 * host finishes on `channels` values:  mean |x| = p + SUM d / n,  var |x| = (SUM d^2 - (SUM d)^2 / n) / (n - 1)
 * -- shifted sums, so the variance does not cancel when the mean is far larger than the spread.  That is
 * what AbsAve (mean |x|) and MeanSigmaStd / MeanLearnedSigmaStd (mean |x| + sigma * sqrt(var |x| + eps))
 * need, in ONE read of x instead of abs (read + write) + mean (read) + var (read) --
 * B/core/stats/stats_op.py:186-262.  Sums are order-dependent: equal to torch's within float32 rounding. */
int64_t bvq_abs_moments_workspace_bytes(int dtype, int64_t outer, int64_t channels, int64_t inner);
int bvq_abs_moments(int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner, float* sums,
                    void* workspace, int64_t workspace_bytes, bvq_stream_t stream);
/* dx = sgn(x) * (a[c] + b[c] * |x|), rounded once to x's dtype: the backward of any function of those two
 * moments (a = dL/dmean / n - 2 mean dL/dvar / (n-1), b = 2 dL/dvar / (n-1)); sgn(0) = 0 */
int bvq_abs_affine_bwd(int dtype, const void* x, const float* a, const float* b, void* dx, int64_t outer,
                       int64_t channels, int64_t inner, bvq_stream_t stream);

/* ---- percentile statistics ---------------------------------------------------------------------
 * k-th smallest value (k is 1-indexed, the same for every channel) of |x| (abs_key = 1) or of x
 * (abs_key = 0) over the `outer` and `inner` axes of x[outer, channels, inner]: torch.kthvalue on the
 * flat tensor / along dim 1 of the [C, -1] view, as used by AbsPercentile, NegativePercentileOrZero and
 * PercentileInterval (B/core/stats/stats_op.py:41-126).  Exact selection (MSD radix select); NaNs order last.
 * Streaming reads of x: channels == 1 (>= 4M elements, 16-byte aligned): 1 for |x| of a 16-bit type, else 2;
 * otherwise 2 for 16-bit types, 3 for float32.  out: dtype of x, `channels` elements. */
int64_t bvq_kth_workspace_bytes(int dtype, int64_t outer, int64_t channels, int64_t inner);
int bvq_kth_value(int abs_key, int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner,
                  int64_t k, void* out, void* workspace, int64_t workspace_bytes, bvq_stream_t stream);
/* Two ranks of the same tensor in one call -- PercentileInterval's low and high percentile
 * (B/core/stats/stats_op.py:97-126): out[0][channels] = the k_first-th value, out[1][channels] = the k_second-th.
 * For channels == 1 (>= 4M elements, 16-byte aligned) both come from ONE histogram read (+ one read for the low
 * key bits of both); otherwise the same as two bvq_kth_value calls.  Workspace: bvq_kth_workspace_bytes. */
int bvq_kth_pair(int abs_key, int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner,
                 int64_t k_first, int64_t k_second, void* out, void* workspace, int64_t workspace_bytes,
                 bvq_stream_t stream);

/* The same selection in steps, for a tensor whose batch is sharded over several devices (one process
 * per GPU): every shard histograms its own elements, the caller sums the histogram of the pass over the
 * shards (one RCCL all-reduce of channels * 2048 uint32 counters, at byte offset bvq_kth_hist_offset()
 * of the workspace) and every shard then picks the same digit -- the result is the k-th value of the
 * CONCATENATED tensor, on every shard, with no other exchange:
 *
 *   bvq_kth_begin(...);
 *   for (pass = 0; pass < bvq_kth_passes(dtype); ++pass) {
 *     bvq_kth_hist(..., pass, ...);   all-reduce(SUM) workspace[offset(pass) .. + channels*2048*4);
 *     bvq_kth_pick(..., pass, ...);
 *   }
 *   bvq_kth_finish(...);
 *
 * The rank is either explicit (BVQ_KTH_EXPLICIT, k >= 1) or derived ON THE DEVICE from the number of
 * elements the first (summed) histogram counted, by the rule the percentile statistics use -- so the
 * global element count never has to travel to the host:
 *   BVQ_KTH_HIGH: k = floor(.01 * q * n + 0.5)   AbsPercentile, PercentileInterval's upper end
 *   BVQ_KTH_LOW : k = ceil (.01 * q * n)         NegativePercentileOrZero, PercentileInterval's lower end
 * (B/core/stats/stats_op.py:56,84,114-116), clamped to [1, n] (torch.kthvalue raises for k = 0; callers
 * that need that error check q * n on the host).  Pass the same rule and q to bvq_kth_begin and to every
 * bvq_kth_pick.  The workspace (bvq_kth_workspace_bytes) depends on dtype and channels only.  Counters
 * are 32-bit: fewer than 2^32 elements per channel over all shards -- bvq_kth_value / bvq_kth_pair /
 * bvq_kth_hist return BVQ_ERR_UNSUPPORTED for a call that alone exceeds it; the caller of the sharded
 * protocol checks (elements per channel on a shard) x (shards) before starting. */
typedef enum bvq_kth_rule { BVQ_KTH_EXPLICIT = 0, BVQ_KTH_HIGH = 1, BVQ_KTH_LOW = 2 } bvq_kth_rule;
int bvq_kth_passes(int dtype);
int64_t bvq_kth_hist_offset(int dtype, int64_t channels, int pass);
int bvq_kth_begin(int dtype, int64_t channels, int rule, int64_t k, double q, void* workspace,
                  int64_t workspace_bytes, bvq_stream_t stream);
int bvq_kth_hist(int abs_key, int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner,
                 int pass, void* workspace, int64_t workspace_bytes, bvq_stream_t stream);
int bvq_kth_pick(int dtype, int64_t channels, int pass, int rule, double q, void* workspace,
                 int64_t workspace_bytes, bvq_stream_t stream);
int bvq_kth_finish(int abs_key, int dtype, int64_t channels, void* out, void* workspace,
                   int64_t workspace_bytes, bvq_stream_t stream);

/* The sharded selection of a WHOLE-TENSOR statistic (one channel; the default Int8ActPerTensorFloat's
 * AbsPercentile, B/quant/base.py:68-75) with the 15-bit first digit of bvq_kth_value's per-tensor route: the first
 * pass histograms the top 15 key bits (32768 counters held in LDS) -- for |x| of a 16-bit type that is the whole key
 * and the only pass; otherwise a second pass counts the remaining 1 / 16 / 17 bits of the elements in the chosen
 * bin.  One read and one all-reduce for bf16 / f16 |x| (two of each with the 11-bit passes above), two for
 * float32 (three).  Same calling pattern, same rank rules, same 32-bit counter limit:
 *
 *   passes = bvq_kthw_plan(abs_key, dtype, -1, NULL, NULL);
 *   bvq_kthw_begin(...);
 *   for (pass = 0; pass < passes; ++pass) {
 *     bvq_kthw_hist(..., pass, ...);   bvq_kthw_plan(abs_key, dtype, pass, &offset, &words);
 *     all-reduce(SUM) the `words` uint32 counters at workspace + offset;
 *     bvq_kthw_pick(..., pass, ...);
 *   }
 *   bvq_kthw_finish(...);   -> out[1]
 *
 * x: n contiguous elements of this shard (any alignment, n may be 0).  Workspace: bvq_kth_workspace_bytes(dtype,
 * 1, 1, n).  rule / k / q are read by the pick of pass 0 only. */
int bvq_kthw_plan(int abs_key, int dtype, int pass, int64_t* hist_offset_bytes, int64_t* hist_words);
int bvq_kthw_begin(int abs_key, int dtype, void* workspace, int64_t workspace_bytes, bvq_stream_t stream);
int bvq_kthw_hist(int abs_key, int dtype, const void* x, int64_t n, int pass, void* workspace,
                  int64_t workspace_bytes, bvq_stream_t stream);
int bvq_kthw_pick(int abs_key, int dtype, int pass, int rule, int64_t k, double q, void* workspace,
                  int64_t workspace_bytes, bvq_stream_t stream);
int bvq_kthw_finish(int abs_key, int dtype, void* out, void* workspace, int64_t workspace_bytes,
                    bvq_stream_t stream);

/* which elements attain the statistic */
typedef enum bvq_match_kind {
  BVQ_MATCH_ABS = 0,   /* |x| == stat, deposit scaled by sgn(x): torch.max(torch.abs(x)) (AbsMax)   */
  BVQ_MATCH_VALUE = 1, /*  x  == stat: torch.max(x) / torch.min(x) (the two halves of AbsMinMax)    */
  /* OR-ed flag: even with channels == 1 only the FIRST attaining element receives the gradient (an
   * index-returning reduction such as torch.kthvalue / torch.max(dim) over a flattened tensor) */
  BVQ_MATCH_FIRST = 16
} bvq_match_kind;

/* Backward of a max / min statistic w.r.t. x, as autograd derives it from torch.max / torch.min
 * (and torch.abs) in B/core/stats/stats_op.py:137-158:
 *   channels == 1 (full reduction): every element attaining stat receives gstat / #ties;
 *   channels  > 1 (reduction along a dim): the FIRST such element of each channel, in
 *                                    (outer, inner) order, receives gstat[c];
 *   MATCH_ABS additionally multiplies by sgn(x) (0 at 0), so zeros carry the sign of x.
 * mode_add == 0: dx is fully written (zeros elsewhere); mode_add == 1: the terms are added in place
 * to an existing dx (used by the fused quantizer backward: one streaming read of x, no write pass).
 * stat, gstat and dx have dtype `dtype`.  workspace: bvq_stats_workspace_bytes(ABSMAX,...) bytes. */
int bvq_stat_bwd(int match, int dtype, const void* x, const void* stat, const void* gstat, void* dx,
                 int64_t outer, int64_t channels, int64_t inner, int mode_add, void* workspace,
                 int64_t workspace_bytes, bvq_stream_t stream);

/* The two halves of bvq_stat_bwd, for callers that put work between them (batch-sharded tensors:
 * the ranks agree on which shard owns each channel's first maximum before anything is deposited).
 * tie_info: device buffer of bvq_tie_info_bytes(channels) bytes, int64 words:
 *   channels > 1 : word c = smallest (outer*inner + i) position attaining stat[c], or -1 if none;
 *                  setting a word to -1 suppresses that channel's deposit;
 *   channels == 1: word 0 = number of ties found, words 2.. = flat indices of (up to 1024 of) them.
 * bvq_stat_tie_scan: one streaming read of x; if dx_zero_fill is non-null it is also written with
 *   the zeros non-attaining elements receive (signed by sgn(x) for MATCH_ABS).
 * bvq_stat_tie_apply: deposits gstat at the recorded positions.  total_ties (nullable, device, one
 *   int64) replaces the local tie count when the ties of a whole-tensor maximum are spread over
 *   several shards.  pre_op (MATCH_ABS only): the statistic was taken of pre_op(x), so the deposit's
 *   sign is sgn(pre_op(x)). */
int64_t bvq_tie_info_bytes(int64_t channels);
int bvq_stat_tie_scan(int match, int dtype, const void* x, const void* stat, int64_t outer,
                      int64_t channels, int64_t inner, void* dx_zero_fill, int64_t* tie_info,
                      bvq_stream_t stream);
int bvq_stat_tie_apply(int match, int pre_op, int dtype, const void* x, const void* stat,
                       const void* gstat, const int64_t* tie_info, const int64_t* total_ties, void* dx,
                       int64_t outer, int64_t channels, int64_t inner, int mode_add, bvq_stream_t stream);

/* bvq_stat_tie_apply(MATCH_ABS, mode_add = 1) for the fused stats-scaled quantizer, taking the
 * float32 scale-gradient sums of bvq_fakequant_bwd directly: per channel the deposited gradient is
 *   ((dscale.to(scale_dtype)) / int_threshold -> quot_dtype).to(dtype of x)
 * i.e. the backward of  scale = clamp_min_ste(stat) / int_threshold  (B/core/quant/int.py:160,
 * B/core/restrict_val.py:22-42) with torch's rounding points, without the three tiny launches.
 * pre_op: the statistic was taken of pre_op(x); the deposit's sign is sgn(pre_op(x)). */
int bvq_stat_tie_apply_dscale(int pre_op, int dtype, const void* x, const void* stat, const float* dscale,
                              int scale_dtype, double int_threshold, int quot_dtype,
                              const int64_t* tie_info, const int64_t* total_ties, void* dx, int64_t outer,
                              int64_t channels, int64_t inner, bvq_stream_t stream);

/* ---- fused affine quantize / dequantize (seam 2: IntQuant) ---------------------------------- */

/* Forward: one read of x, one write of y.
 *   t = x / scale ; t = t + zp ; t = round_mode(t) ; t = clamp(t, qmin, qmax) ;
 *   y = (t - zp) * scale          (BVQ_OUT_DEQUANT)   or   y = t   (BVQ_OUT_INT)
 * with every operation rounded to ct_dtype exactly where the reference's op chain rounds
 * (B/core/quant/int_base.py:63-97).  codes (nullable, desc->codes_dtype, same element count) receives
 * the clamped integer codes; y may be null when only the codes are wanted (1 read of x, 1-4 bytes
 * written per element: the layout downstream integer kernels and the QCDQ exporters consume). */
int bvq_fakequant_fwd(const bvq_quant_desc* desc, const void* x, const void* scale, const void* zp,
                      void* y, void* codes, bvq_stream_t stream);

/* Statistic AND quantizer in ONE launch: AbsMax over (outer, inner) -> clamp_min(min_val) -> / int_threshold
 * -> quantize-dequantize with that scale and a zero zero-point -- bvq_absmax_scale followed by
 * bvq_fakequant_fwd, i.e. RescalingIntQuant.forward on the stats-scaled graphs (B/core/quant/int.py:155-163,
 * B/core/scaling/runtime.py:50-72), in ONE launch, reading x ONCE: a channel that fits the registers of one
 * workgroup (weights, small activations) is held there between the reduction and the quantization (2 tensor
 * passes instead of 3, one launch instead of three).  Uses desc's shape, dtypes (x_dtype == ct_dtype), qmin/qmax,
 * round_mode, scalar_mode, pre_op, scale_dtype / scale_per_channel; zero-point is +0.  stat_out: [channels] in
 * x's dtype, scale_out: [channels] in scale_dtype.  bvq_stats_fakequant_fwd_workspace_bytes returns 0 when the
 * shape is not covered (a channel larger than one workgroup's registers, ragged rows, misaligned pointers):
 * the caller then takes the two-call route. */
int64_t bvq_stats_fakequant_fwd_workspace_bytes(const bvq_quant_desc* desc, const void* x, const void* y);
int bvq_stats_fakequant_fwd(const bvq_quant_desc* desc, const void* x, double min_val, int use_min,
                            double int_threshold, void* stat_out, void* scale_out, void* y, void* workspace,
                            int64_t workspace_bytes, bvq_stream_t stream);

/* bytes of scratch bvq_fakequant_bwd needs */
int64_t bvq_fakequant_bwd_workspace_bytes(const bvq_quant_desc* desc);

/* Backward of the chain above, as autograd derives it:
 *   dt = pass ? g*scale : 0      pass = clamp_ste || !(t_rounded > qmax || t_rounded < qmin)
 *   dx = dt / scale                                              (dtype x_dtype)
 *   dscale[c] = sum g*(t_clamped - zp)  -  sum dt * ((x/scale)/scale)     (float32, nullable)
 *   dzp[c]    = sum dt - sum g*scale                                      (float32, nullable)
 * dscale / dzp have `channels` elements when scale OR zero-point is per-channel, else one.
 * One read of g, one read of x, one write of dx; the per-channel sums ride on the same reads and
 * are combined in a fixed order (bit-reproducible run to run).
 * tie_stat / tie_info (both null or both set; requires dscale set and dzp null): while streaming x
 * the kernel also records which elements attain the abs-max statistic tie_stat (dtype of x,
 * `channels` elements) into tie_info (see bvq_stat_tie_scan), so that the statistic's gradient can
 * be deposited with bvq_stat_tie_apply without another pass over x. */
int bvq_fakequant_bwd(const bvq_quant_desc* desc, const void* g, const void* x, const void* scale,
                      const void* zp, void* dx, float* dscale, float* dzp, const void* tie_stat,
                      int64_t* tie_info, void* workspace, int64_t workspace_bytes, bvq_stream_t stream);

/* Learned bit widths (BitWidthParameter, B/core/bit_width/parameter.py:23-100): the integer range is then a pair of
 * 0-dim tensors in the autograd graph (min_int / max_int of the bit-width tensor, B/function/ops.py:132-191), not
 * host numbers.  `bounds` = [qmin, qmax] as float32 ON THE DEVICE replaces desc->qmin / qmax (no host sync):
 * bvq_fakequant_fwd_bounds is bvq_fakequant_fwd reading them; bvq_fakequant_bwd_bounds is bvq_fakequant_bwd(dscale)
 * that also returns, when dbounds is non-null (a plain TensorClamp: tensor_clamp's two torch.where send the gradient
 * of a replaced value to the bound that replaced it), dbounds[0 .. n) = per-channel sums of the gradient reaching
 * qmin and dbounds[n .. 2n) of that reaching qmax (n = channels for per-channel scales, else 1; the caller adds the
 * channels up).  With a straight-through clamp pass dbounds = null.  Workspace: bvq_fakequant_bwd_workspace_bytes. */
int bvq_fakequant_fwd_bounds(const bvq_quant_desc* desc, const void* x, const void* scale, const void* zp,
                             const float* bounds, void* y, bvq_stream_t stream);
int bvq_fakequant_bwd_bounds(const bvq_quant_desc* desc, const void* g, const void* x, const void* scale,
                             const void* zp, const float* bounds, void* dx, float* dscale, float* dbounds,
                             void* workspace, int64_t workspace_bytes, bvq_stream_t stream);

/* Learned scales (ParameterScaling, and ParameterFromRuntimeStatsScaling once its collection phase is over:
 * the steady state of the default Int8ActPerTensorFloat; B/core/scaling/standalone.py:75-152, 155-298), float
 * restriction:   scale = abs_binary_sign_grad(clamp_min_ste(value, min_val)) / int_threshold.
 * bvq_learned_scale: that forward in ONE launch over the n (1 or `channels`) elements of `value`, the quotient
 *   rounded to scale_dtype -- instead of clamp, |.|, division (3 launches).  min_val: python scalar (rounded to
 *   value_dtype here); int_threshold: already rounded to the dtype the division runs in.
 * bvq_fakequant_bwd_learned: bvq_fakequant_bwd(dscale) whose LAST reduction launch also carries the scale
 *   gradient through that chain's backward with torch's rounding points --
 *     dvalue = binary_sign(clamp_min(value)) * ((dscale.to(scale_dtype) [+ gscale]) / int_threshold)
 *   (gscale, nullable, scale_dtype: gradient reaching `scale` from its other users) -- instead of a cast, a
 *   division, a sign-multiply and their launches.  dscale (float32, 1 or `channels`) is still written.
 *   Workspace: bvq_fakequant_bwd_workspace_bytes. */
int bvq_learned_scale(int value_dtype, const void* value, int64_t n, double min_val, int use_min,
                      double int_threshold, int scale_dtype, void* scale_out, bvq_stream_t stream);
int bvq_fakequant_bwd_learned(const bvq_quant_desc* desc, const void* g, const void* x, const void* scale,
                              const void* zp, void* dx, float* dscale, const void* value, int value_dtype,
                              double min_val, int use_min, double int_threshold, const void* gscale,
                              void* dvalue, void* workspace, int64_t workspace_bytes, bvq_stream_t stream);

/* Backward of the stats-scaled per-channel graphs (scale = clamp_min(AbsMax(x)) / int_threshold) in
 * TWO launches: the backward kernel (dx, per-unit dscale sums, per-unit first position attaining `stat`) and one
 * finishing kernel per call that sums dscale (-> dscale[channels], float32), turns it into the statistic's
 * gradient -- dscale.to(scale_dtype) / int_threshold (in quot_dtype) -> x's dtype -- and adds it, times sgn(x),
 * to the first arg-max element of every channel in dx: bvq_fakequant_bwd(tie_stat) + bvq_stat_tie_apply_dscale
 * without their three helper launches.  bvq_fakequant_bwd_stats_workspace_bytes returns 0 for layouts it does
 * not cover (one scale for the whole tensor: its gradient is shared evenly by all ties; channels with more
 * than 4096 work units): the caller then takes the two-call route. */
int64_t bvq_fakequant_bwd_stats_workspace_bytes(const bvq_quant_desc* desc);
int bvq_fakequant_bwd_stats(const bvq_quant_desc* desc, const void* g, const void* x, const void* scale,
                            const void* zp, const void* stat, void* dx, float* dscale, int scale_dtype,
                            double int_threshold, int quot_dtype, void* workspace, int64_t workspace_bytes,
                            bvq_stream_t stream);

/* bvq_fakequant_bwd_stats in ONE launch (row-mapped per-channel layouts; bvq_fakequant_bwd_stats_onepass_supported:
 * 1 / 0): the backward kernel writes dx and its per-unit partials through to memory (agent scope), every wave counts
 * its unit in on its channel's arrival counter, and the wave whose count completes the channel sums the channel's
 * dscale partials (double, fixed order), takes the first arg-max position, converts dscale into the statistic's
 * gradient and deposits it on that element of dx -- the finishing launch of bvq_fakequant_bwd_stats, done by whoever
 * arrives last; no wave waits.  Same results, bit for bit (the sums are taken in a fixed order of their own).
 * arrive: `arrive_words` >= channels uint32 words, ALL ZERO when the launch starts, handed back as zeros (the
 * contract of bvq_absmax_scale_onepass; the two may share one buffer on one stream).  workspace: as
 * bvq_fakequant_bwd_stats_workspace_bytes. */
int bvq_fakequant_bwd_stats_onepass_supported(const bvq_quant_desc* d);
int bvq_fakequant_bwd_stats_onepass(const bvq_quant_desc* d, const void* g, const void* x, const void* scale,
                                    const void* zp, const void* stat, void* dx, float* dscale, int scale_dtype,
                                    double int_threshold, int quot_dtype, void* workspace, int64_t workspace_bytes,
                                    uint32_t* arrive, int64_t arrive_words, bvq_stream_t stream);

/* ---- the other quantizers of the family -------------------------------------------
 * BinaryQuant / ClampedBinaryQuant (B/core/quant/binary.py:19-101), TernaryQuant (B/core/quant/ternary.py:18-66),
 * DecoupledIntQuant (B/core/quant/int_base.py:100-182) and TruncIntQuant (B/core/quant/int.py:199-229) as ONE
 * forward kernel (read x, write y) and ONE backward kernel (read g, read x, write dx; the scale gradients as
 * float32 sums over the tensor or per channel) instead of the reference's 4..10 torch ops each, with the
 * reference's rounding points in the compute dtype ct_dtype = torch.result_type of its op chain.
 *   x viewed as [outer, channels, inner]; scale / pre_scale: 1 element (whole tensor) or `channels` elements in
 *   scale_dtype; zero-points: ONE element in zp_dtype, or null for +0.  kind-specific fields:
 *     BINARY          y = binary_sign(x) * scale
 *     CLAMPED_BINARY  x clamped to [-scale, scale] first; clamp_ste = 0: the clamp masks dx and feeds d(scale)
 *     TERNARY         y = [|x| > threshold * scale] * sign(x) * scale   (ct_dtype must be float32: the reference's
 *                     mask.float() promotes)
 *     DECOUPLED       round_mode / qmin / qmax / clamp_ste as bvq_quant_desc; rounding grid from (pre_scale, pre_zp),
 *                     de-quantization with (scale, zp); dpre_scale receives the pre-scale's gradient
 *     TRUNC           y = (round_mode(round((x / scale + zp)) / trunc_scale) - zp) * scale
 * dscale / dpre_scale: float32 [1 or channels], nullable.  Workspace: bvq_variant_bwd_workspace_bytes. */
typedef enum bvq_variant_kind {
  BVQ_VAR_BINARY = 0, BVQ_VAR_CLAMPED_BINARY = 1, BVQ_VAR_TERNARY = 2, BVQ_VAR_DECOUPLED = 3, BVQ_VAR_TRUNC = 4
} bvq_variant_kind;
typedef struct bvq_variant_desc {
  int64_t outer, channels, inner;
  int32_t kind;               /* bvq_variant_kind */
  int32_t x_dtype, ct_dtype, scale_dtype, zp_dtype;
  int32_t scale_per_channel;
  int32_t round_mode;         /* bvq_round_mode */
  int32_t clamp_ste;
  int32_t scalar_mode;        /* bvq_scalar_mode */
  float qmin, qmax;           /* DECOUPLED */
  float threshold;            /* TERNARY */
  float trunc_scale;          /* TRUNC: 2^(input_bit_width - output_bit_width) */
} bvq_variant_desc;
int bvq_variant_fwd(const bvq_variant_desc* desc, const void* x, const void* scale, const void* pre_scale,
                    const void* zp, const void* pre_zp, void* y, bvq_stream_t stream);
int64_t bvq_variant_bwd_workspace_bytes(const bvq_variant_desc* desc);
int bvq_variant_bwd(const bvq_variant_desc* desc, const void* g, const void* x, const void* scale,
                    const void* pre_scale, const void* zp, const void* pre_zp, void* dx, float* dscale,
                    float* dpre_scale, void* workspace, int64_t workspace_bytes, bvq_stream_t stream);

/* Diagnostic entry (no reference counterpart): the float32 quotient the float16 quantizer kernels compute for a
 * numerator a[i] and a scale scales[j] -- the product with the correctly rounded reciprocal, corrected by one exact
 * remainder step (brevitas_amd/csrc/bvq_fakequant.h, DivF16R) -- out[j * n_a + i], float32 device buffers.  The
 * tests compare every float16 numerator x every float16 scale in [2^-14, 2^14] with a / s (tests/test_gpu_fastdiv.py). */
int bvq_selftest_div_f16r(const float* a, int32_t n_a, const float* scales, int32_t n_s, float* out,
                          bvq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BVQ_H_ */
