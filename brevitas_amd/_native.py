"""ctypes binding of libbvq.so -- the C-ABI HIP library (include/bvq.h).

This is the ONLY compute backend of the package: there is no CPU or pure-torch fallback.  If the
library is missing, or a tensor does not live on a ROCm device, the call fails loudly.

PyTorch is used here as plumbing only: device memory (torch.empty), the current HIP stream and the
device guard.  Signatures carry raw pointers and sizes.
"""
import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# BREVITAS_AMD_LIB: developer override to load an experimental build of the same ABI (tools/microbench.py)
LIB_PATH = os.environ.get('BREVITAS_AMD_LIB') or os.path.join(_PKG, 'libbvq.so')

F32, BF16, F16 = 0, 1, 2
ROUND, FLOOR, CEIL, ROUND_TO_ZERO, DPU_ROUND = range(5)
(OP_ROUND, OP_FLOOR, OP_CEIL, OP_ROUND_TO_ZERO, OP_DPU_ROUND, OP_BINARY_SIGN, OP_TERNARY_SIGN,
 OP_ABS) = range(8)
STAT_ABSMAX, STAT_MINMAX = 0, 1
SCALAR_OPMATH, SCALAR_CAST = 0, 1
OUT_DEQUANT, OUT_INT = 0, 1
MATCH_ABS, MATCH_VALUE = 0, 1
MATCH_FIRST = 16  # OR-ed: only the first attaining element, even for a whole-tensor reduction
PRE_NONE, PRE_RELU = 0, 1
CODES_I32, CODES_I8, CODES_U8 = 0, 1, 2
_CODES_TORCH = {CODES_I32: torch.int32, CODES_I8: torch.int8, CODES_U8: torch.uint8}
ABI_VERSION = 2

_DTYPES = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}

EXPORTS = (
    'bvq_abi_version', 'bvq_last_error', 'bvq_unary', 'bvq_stats_pre', 'bvq_scalar_clamp', 'bvq_tensor_clamp',
    'bvq_tensor_clamp_bwd', 'bvq_abs_binary_sign_grad_bwd', 'bvq_stats_workspace_bytes', 'bvq_stats',
    'bvq_absmax_scale', 'bvq_running_stats_update', 'bvq_scale_from_stat', 'bvq_shard_pack', 'bvq_shard_unpack', 'bvq_abs_moments_workspace_bytes', 'bvq_abs_moments',
    'bvq_abs_affine_bwd', 'bvq_kth_workspace_bytes', 'bvq_kth_value', 'bvq_kth_pair', 'bvq_kth_passes',
    'bvq_kth_hist_offset', 'bvq_kth_begin', 'bvq_kth_hist', 'bvq_kth_pick', 'bvq_kth_finish', 'bvq_stat_bwd', 'bvq_tie_info_bytes', 'bvq_stat_tie_scan', 'bvq_stat_tie_apply', 'bvq_stat_tie_apply_dscale', 'bvq_fakequant_fwd', 'bvq_stats_fakequant_fwd_workspace_bytes', 'bvq_stats_fakequant_fwd',
    'bvq_fakequant_bwd_workspace_bytes', 'bvq_fakequant_bwd_stats_workspace_bytes', 'bvq_fakequant_bwd_stats', 'bvq_fakequant_bwd',
    'bvq_learned_scale', 'bvq_fakequant_bwd_learned', 'bvq_variant_fwd', 'bvq_variant_bwd_workspace_bytes', 'bvq_variant_bwd',
    'bvq_fakequant_fwd_bounds', 'bvq_fakequant_bwd_bounds', 'bvq_histc', 'bvq_absmax_scale_running', 'bvq_selftest_div_f16r',
    'bvq_kthw_plan', 'bvq_kthw_begin', 'bvq_kthw_hist', 'bvq_kthw_pick', 'bvq_kthw_finish',
    'bvq_absmax_onepass_supported', 'bvq_absmax_scale_onepass', 'bvq_fakequant_bwd_stats_onepass_supported',
    'bvq_fakequant_bwd_stats_onepass', 'bvq_scale_from_stat_running', 'bvq_fakequant_bwd_shard',
    'bvq_shard_unpack_deposit', 'bvq_absmax_list_supported', 'bvq_absmax_scale_list')


class QuantDesc(ctypes.Structure):
    """bvq_quant_desc of include/bvq.h"""
    _fields_ = [
        ('outer', ctypes.c_int64), ('channels', ctypes.c_int64), ('inner', ctypes.c_int64),
        ('x_dtype', ctypes.c_int32), ('ct_dtype', ctypes.c_int32), ('scale_dtype', ctypes.c_int32),
        ('zp_dtype', ctypes.c_int32), ('scale_per_channel', ctypes.c_int32),
        ('zp_per_channel', ctypes.c_int32), ('qmin', ctypes.c_float), ('qmax', ctypes.c_float),
        ('round_mode', ctypes.c_int32), ('scalar_mode', ctypes.c_int32), ('clamp_ste', ctypes.c_int32),
        ('out_kind', ctypes.c_int32), ('pre_op', ctypes.c_int32), ('codes_dtype', ctypes.c_int32)]


class VariantDesc(ctypes.Structure):
    """bvq_variant_desc of include/bvq.h"""
    _fields_ = [
        ('outer', ctypes.c_int64), ('channels', ctypes.c_int64), ('inner', ctypes.c_int64), ('kind', ctypes.c_int32),
        ('x_dtype', ctypes.c_int32), ('ct_dtype', ctypes.c_int32), ('scale_dtype', ctypes.c_int32),
        ('zp_dtype', ctypes.c_int32), ('scale_per_channel', ctypes.c_int32), ('round_mode', ctypes.c_int32),
        ('clamp_ste', ctypes.c_int32), ('scalar_mode', ctypes.c_int32), ('qmin', ctypes.c_float),
        ('qmax', ctypes.c_float), ('threshold', ctypes.c_float), ('trunc_scale', ctypes.c_float)]


VAR_BINARY, VAR_CLAMPED_BINARY, VAR_TERNARY, VAR_DECOUPLED, VAR_TRUNC = range(5)


class BvqError(RuntimeError):
    pass


def _load(path=None, strict=True):
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise ImportError(
            'brevitas_amd: %s is missing. Build it with `python -m brevitas_amd.csrc.build` '
            '(needs hipcc, gfx950). There is no fallback backend.' % path)
    lib = ctypes.CDLL(path)
    vp, i64, i32, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double
    lib.bvq_abi_version.restype = i32
    lib.bvq_last_error.restype = ctypes.c_char_p
    sig = {
        'bvq_unary': (i32, [i32, i32, vp, vp, i64, vp]),
        'bvq_scalar_clamp': (i32, [i32, vp, vp, i64, dbl, i32, dbl, i32, vp]),
        'bvq_tensor_clamp': (i32, [i32, vp, vp, vp, i32, vp, i64, vp]),
        'bvq_tensor_clamp_bwd': (i32, [i32, vp, vp, vp, vp, i32, vp, i64, vp]),
        'bvq_abs_binary_sign_grad_bwd': (i32, [i32, vp, vp, vp, i64, vp]),
        'bvq_stats_workspace_bytes': (i64, [i32, i32, i64, i64, i64]),
        'bvq_stats': (i32, [i32, i32, vp, i64, i64, i64, i32, vp, vp, i64, vp]),
        'bvq_stats_pre': (i32, [i32, i32, i32, vp, i64, i64, i64, i32, vp, vp, i64, vp]),
        'bvq_stat_bwd': (i32, [i32, i32, vp, vp, vp, vp, i64, i64, i64, i32, vp, i64, vp]),
        'bvq_fakequant_fwd': (i32, [ctypes.POINTER(QuantDesc), vp, vp, vp, vp, vp, vp]),
        'bvq_stats_fakequant_fwd_workspace_bytes': (i64, [ctypes.POINTER(QuantDesc), vp, vp]),
        'bvq_stats_fakequant_fwd': (i32, [ctypes.POINTER(QuantDesc), vp, dbl, i32, dbl, vp, vp, vp, vp, i64, vp]),
        'bvq_fakequant_bwd_workspace_bytes': (i64, [ctypes.POINTER(QuantDesc)]),
        'bvq_fakequant_bwd_stats_workspace_bytes': (i64, [ctypes.POINTER(QuantDesc)]),
        'bvq_fakequant_bwd_stats': (i32, [ctypes.POINTER(QuantDesc), vp, vp, vp, vp, vp, vp, vp, i32, dbl, i32, vp, i64, vp]),
        'bvq_fakequant_bwd_stats_onepass_supported': (i32, [ctypes.POINTER(QuantDesc)]),
        'bvq_fakequant_bwd_stats_onepass': (i32, [ctypes.POINTER(QuantDesc), vp, vp, vp, vp, vp, vp, vp, i32, dbl, i32, vp, i64, vp, i64, vp]),
        'bvq_absmax_scale': (i32, [i32, i32, vp, i64, i64, i64, vp, dbl, i32, dbl, i32, vp, vp, i64, vp]),
        'bvq_absmax_scale_running': (i32, [i32, i32, vp, i64, i64, i64, vp, dbl, i32, dbl, i32, vp, i32, vp, dbl, i32, vp, i64, vp]),
        'bvq_absmax_onepass_supported': (i32, [i32, vp, i64, i64, i64]),
        'bvq_absmax_scale_onepass': (i32, [i32, i32, vp, i64, i64, i64, i32, vp, dbl, i32, dbl, i32, vp, i32, vp, dbl, i32, vp, i64, vp]),
        'bvq_absmax_list_supported': (i32, [i32, i32, vp, vp, i64, vp]),
        'bvq_absmax_scale_list': (i32, [i32, i32, vp, vp, i64, vp, vp, dbl, i32, dbl, i32, vp, vp, i64, vp, i64, vp]),
        'bvq_running_stats_update': (i32, [i32, vp, i32, vp, i64, dbl, i32, vp]),
        'bvq_scale_from_stat': (i32, [vp, i64, i32, vp, dbl, i32, dbl, i32, vp, vp]),
        'bvq_scale_from_stat_running': (i32, [vp, i64, i32, vp, dbl, i32, dbl, i32, vp, i32, vp, dbl, i32, vp]),
        'bvq_fakequant_bwd_shard': (i32, [ctypes.POINTER(QuantDesc), vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i64, vp, i64, vp]),
        'bvq_shard_unpack_deposit': (i32, [i32, vp, vp, vp, i32, i64, i32, vp, i64, i32, dbl, i32, i32, vp, vp]),
        'bvq_shard_pack': (i32, [vp, vp, i64, i32, i32, vp, vp]),
        'bvq_shard_unpack': (i32, [vp, i32, i64, i32, i32, vp, vp, vp, vp]),
        'bvq_abs_moments_workspace_bytes': (i64, [i32, i64, i64, i64]),
        'bvq_abs_moments': (i32, [i32, vp, i64, i64, i64, vp, vp, i64, vp]),
        'bvq_abs_affine_bwd': (i32, [i32, vp, vp, vp, vp, i64, i64, i64, vp]),
        'bvq_kth_workspace_bytes': (i64, [i32, i64, i64, i64]),
        'bvq_kth_value': (i32, [i32, i32, vp, i64, i64, i64, i64, vp, vp, i64, vp]),
        'bvq_kth_pair': (i32, [i32, i32, vp, i64, i64, i64, i64, i64, vp, vp, i64, vp]),
        'bvq_kth_passes': (i32, [i32]),
        'bvq_kth_hist_offset': (i64, [i32, i64, i32]),
        'bvq_kth_begin': (i32, [i32, i64, i32, i64, dbl, vp, i64, vp]),
        'bvq_kth_hist': (i32, [i32, i32, vp, i64, i64, i64, i32, vp, i64, vp]),
        'bvq_kth_pick': (i32, [i32, i64, i32, i32, dbl, vp, i64, vp]),
        'bvq_kth_finish': (i32, [i32, i32, i64, vp, vp, i64, vp]),
        'bvq_kthw_plan': (i32, [i32, i32, i32, vp, vp]),
        'bvq_kthw_begin': (i32, [i32, i32, vp, i64, vp]),
        'bvq_kthw_hist': (i32, [i32, i32, vp, i64, i32, vp, i64, vp]),
        'bvq_kthw_pick': (i32, [i32, i32, i32, i32, i64, dbl, vp, i64, vp]),
        'bvq_kthw_finish': (i32, [i32, i32, vp, vp, i64, vp]),
        'bvq_tie_info_bytes': (i64, [i64]),
        'bvq_stat_tie_scan': (i32, [i32, i32, vp, vp, i64, i64, i64, vp, vp, vp]),
        'bvq_stat_tie_apply': (i32, [i32, i32, i32, vp, vp, vp, vp, vp, vp, i64, i64, i64, i32, vp]),
        'bvq_stat_tie_apply_dscale': (i32, [i32, i32, vp, vp, vp, i32, dbl, i32, vp, vp, vp, i64, i64, i64, vp]),
        'bvq_fakequant_bwd': (i32, [ctypes.POINTER(QuantDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
        'bvq_learned_scale': (i32, [i32, vp, i64, dbl, i32, dbl, i32, vp, vp]),
        'bvq_histc': (i32, [i32, vp, i64, vp, i32, vp, vp]),
        'bvq_selftest_div_f16r': (i32, [vp, i32, vp, i32, vp, vp]),
        'bvq_fakequant_fwd_bounds': (i32, [ctypes.POINTER(QuantDesc), vp, vp, vp, vp, vp, vp]),
        'bvq_fakequant_bwd_bounds': (i32, [ctypes.POINTER(QuantDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
        'bvq_variant_fwd': (i32, [ctypes.POINTER(VariantDesc), vp, vp, vp, vp, vp, vp, vp]),
        'bvq_variant_bwd_workspace_bytes': (i64, [ctypes.POINTER(VariantDesc)]),
        'bvq_variant_bwd': (i32, [ctypes.POINTER(VariantDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
        'bvq_fakequant_bwd_learned': (i32, [ctypes.POINTER(QuantDesc), vp, vp, vp, vp, vp, vp, vp, i32, dbl, i32, dbl, vp, vp, vp, i64, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name, None)
        if fn is None:
            if strict:
                raise ImportError('brevitas_amd: %s does not export %s' % (path, name))
            continue  # developer A/B runs against an older build (tools/variant_bench.py)
        fn.restype = res
        fn.argtypes = args
    ver = lib.bvq_abi_version()
    if ver != ABI_VERSION:
        raise ImportError('brevitas_amd: libbvq.so has ABI %d, this package needs %d' % (ver, ABI_VERSION))
    return lib


lib = _load()


def last_error():
    return lib.bvq_last_error().decode()


def check(rc, what):
    if rc != 0:
        raise BvqError('%s failed (%d): %s' % (what, rc, last_error()))


def dtype_code(dtype):
    try:
        return _DTYPES[dtype]
    except KeyError:
        raise BvqError('brevitas_amd: unsupported dtype %s (float32, bfloat16, float16)' % dtype)


def require_device(*tensors):
    """every tensor must live on the same ROCm device; no CPU path exists"""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise BvqError(
                'brevitas_amd: got a %s tensor; the fake-quantization engine only runs on a ROCm '
                'device (there is no CPU fallback)' % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise BvqError('brevitas_amd: tensors on different devices (%s, %s)' % (dev, t.device))
    return dev


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_ptr(device):
    """the current HIP stream of `device` as the void* the C ABI takes"""
    if _raw_stream is not None:  # no Stream object construction on the hot path
        return _raw_stream(device.index if device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t):
    """raw device address (ctypes converts the int for the c_void_p parameters)"""
    return t.data_ptr() if t is not None else None


class _DeviceGuard:
    """`with torch.cuda.device(dev)` only when dev is not already current (the common case costs one call)"""
    __slots__ = ('ctx',)

    def __init__(self, dev):
        idx = dev.index
        self.ctx = None if idx is None or idx == torch.cuda.current_device() else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


# Optional measurement hook (bench.py): an object with before(name) / after(name), called around the
# C-ABI calls listed below on the launching thread, e.g. to record HIP events on the current stream.
_timer = None


def set_kernel_timer(timer):
    global _timer
    _timer = timer


# ---- arrival buffers of the one-launch kernels --------------------------------------------------------------------
# include/bvq.h, bvq_absmax_scale_onepass: per-channel key / counter words that are zero when a launch starts and that
# the launch hands back as zeros.  One buffer per (device, stream), zero-filled once when it is allocated; launches
# on one stream are ordered, so they can share it.  Not allocated while a stream is capturing (the capture would
# own the memory): those calls take the two-launch route.
ARRIVE_WORDS = 1 << 16
_arrive = {}
ONEPASS = os.environ.get('BREVITAS_AMD_ONEPASS', '1') != '0'
ONEPASS_BWD = os.environ.get('BREVITAS_AMD_ONEPASS_BWD', '1') != '0'   # developer A/B switch of the backward alone


def arrival_buffer(dev, stream, words):
    """-> int32 tensor of >= words zeros for this device and stream, or None (capturing / switched off)"""
    if not ONEPASS or words > ARRIVE_WORDS:
        return None
    key = (dev.index, stream)
    buf = _arrive.get(key)
    if buf is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        buf = _arrive[key] = torch.zeros(ARRIVE_WORDS, dtype=torch.int32, device=dev)
    return buf


# ---- thin wrappers: allocate outputs with torch, pass raw pointers --------------------------------

def unary(op, x):
    dev = require_device(x)
    x = x.contiguous()
    y = torch.empty_like(x)
    with _DeviceGuard(dev):
        check(lib.bvq_unary(op, dtype_code(x.dtype), ptr(x), ptr(y), x.numel(), stream_ptr(dev)), 'bvq_unary')
    return y


def scalar_clamp(x, lo, hi):
    dev = require_device(x)
    x = x.contiguous()
    y = torch.empty_like(x)
    with _DeviceGuard(dev):
        check(lib.bvq_scalar_clamp(dtype_code(x.dtype), ptr(x), ptr(y), x.numel(),
                                   0.0 if lo is None else float(lo), int(lo is not None),
                                   0.0 if hi is None else float(hi), int(hi is not None), stream_ptr(dev)),
              'bvq_scalar_clamp')
    return y


def _bounds(x, lo, hi):
    """bring clamp bounds to x's dtype/device; returns (lo, hi, bounds_full)"""
    lo = lo.to(device=x.device, dtype=x.dtype)
    hi = hi.to(device=x.device, dtype=x.dtype)
    if lo.numel() == 1 and hi.numel() == 1:
        return lo.reshape(1), hi.reshape(1), 0
    return lo.expand_as(x).contiguous(), hi.expand_as(x).contiguous(), 1


def tensor_clamp(x, lo, hi, out=None):
    dev = require_device(x)
    xc = x.contiguous()
    lo, hi, full = _bounds(xc, lo, hi)
    y = out if out is not None else torch.empty_like(xc)
    with _DeviceGuard(dev):
        check(lib.bvq_tensor_clamp(dtype_code(xc.dtype), ptr(xc), ptr(lo), ptr(hi), full, ptr(y), xc.numel(),
                                   stream_ptr(dev)), 'bvq_tensor_clamp')
    return y


def tensor_clamp_bwd(g, x, lo, hi):
    dev = require_device(g, x)
    xc = x.contiguous()
    g = g.to(xc.dtype).contiguous()
    lo, hi, full = _bounds(xc, lo, hi)
    dx = torch.empty_like(xc)
    with _DeviceGuard(dev):
        check(lib.bvq_tensor_clamp_bwd(dtype_code(xc.dtype), ptr(g), ptr(xc), ptr(lo), ptr(hi), full, ptr(dx),
                                       xc.numel(), stream_ptr(dev)), 'bvq_tensor_clamp_bwd')
    return dx


def abs_binary_sign_grad_bwd(g, x):
    dev = require_device(g, x)
    xc = x.contiguous()
    g = g.to(xc.dtype).contiguous()
    dx = torch.empty_like(xc)
    with _DeviceGuard(dev):
        check(lib.bvq_abs_binary_sign_grad_bwd(dtype_code(xc.dtype), ptr(g), ptr(xc), ptr(dx), xc.numel(),
                                               stream_ptr(dev)), 'bvq_abs_binary_sign_grad_bwd')
    return dx


def stats(kind, x, outer, channels, inner, out_f32=False, pre_op=PRE_NONE):
    """x contiguous, viewed as [outer, channels, inner] -> [channels] (ABSMAX) or [2, channels]"""
    dev = require_device(x)
    assert x.is_contiguous() and x.numel() == outer * channels * inner
    dt = dtype_code(x.dtype)
    nout = channels * (2 if kind == STAT_MINMAX else 1)
    out = torch.empty(nout, dtype=torch.float32 if out_f32 else x.dtype, device=dev)
    if kind == STAT_ABSMAX and lib.bvq_absmax_onepass_supported(dt, ptr(x), outer, channels, inner):
        with _DeviceGuard(dev):
            st = stream_ptr(dev)
            arrive = arrival_buffer(dev, st, max(2 * channels, 18))
            if arrive is not None:
                if _timer is not None:
                    _timer.before('bvq_stats')
                check(lib.bvq_absmax_scale_onepass(pre_op, dt, ptr(x), outer, channels, inner, dtype_code(out.dtype),
                                                   ptr(out), 0.0, 0, 1.0, 0, None, 0, None, 0.0, 0, ptr(arrive),
                                                   arrive.numel(), st), 'bvq_absmax_scale_onepass')
                if _timer is not None:
                    _timer.after('bvq_stats')
                return out
    wsb = lib.bvq_stats_workspace_bytes(kind, dt, outer, channels, inner)
    if wsb < 0:
        raise BvqError('bvq_stats_workspace_bytes: bad arguments')
    ws = torch.empty(max(int(wsb), 8), dtype=torch.uint8, device=dev)
    with _DeviceGuard(dev):
        if _timer is not None:
            _timer.before('bvq_stats')
        check(lib.bvq_stats_pre(kind, pre_op, dt, ptr(x), outer, channels, inner, dtype_code(out.dtype), ptr(out),
                                ptr(ws), ws.numel(), stream_ptr(dev)), 'bvq_stats')
        if _timer is not None:
            _timer.after('bvq_stats')
    return out


def stat_bwd(match, x, stat, gstat, outer, channels, inner, dx=None):
    """dense (dx None) or in-place additive (dx given) backward of a max/min statistic"""
    dev = require_device(x, stat, gstat, dx)
    assert x.is_contiguous()
    dt = dtype_code(x.dtype)
    stat = stat.to(x.dtype).contiguous()
    gstat = gstat.to(x.dtype).contiguous()
    mode_add = int(dx is not None)
    if dx is None:
        dx = torch.empty_like(x)
    assert dx.is_contiguous() and dx.dtype == x.dtype
    wsb = lib.bvq_stats_workspace_bytes(STAT_ABSMAX, dt, outer, channels, inner)
    ws = torch.empty(max(int(wsb), 8), dtype=torch.uint8, device=dev)
    with _DeviceGuard(dev):
        check(lib.bvq_stat_bwd(match, dt, ptr(x), ptr(stat), ptr(gstat), ptr(dx), outer, channels, inner,
                               mode_add, ptr(ws), ws.numel(), stream_ptr(dev)), 'bvq_stat_bwd')
    return dx


def fakequant_fwd(desc, x, scale, zp, want_codes=False, want_y=True):
    """-> y, (y, codes) or codes alone; codes have the element type of desc.codes_dtype"""
    dev = require_device(x, scale, zp)
    ct = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}[desc.ct_dtype]
    y = torch.empty(x.shape, dtype=ct, device=dev) if want_y else None
    codes = torch.empty(x.shape, dtype=_CODES_TORCH[desc.codes_dtype], device=dev) if want_codes else None
    with _DeviceGuard(dev):
        if _timer is not None:
            _timer.before('bvq_fakequant_fwd')
        check(lib.bvq_fakequant_fwd(ctypes.byref(desc), ptr(x), ptr(scale), ptr(zp), ptr(y), ptr(codes),
                                    stream_ptr(dev)), 'bvq_fakequant_fwd')
        if _timer is not None:
            _timer.after('bvq_fakequant_fwd')
    if not want_y:
        return codes
    return (y, codes) if want_codes else y


def stats_fakequant_fwd(desc, x, min_val, int_threshold, scale_dtype):
    """abs-max statistic, scale and quantize-dequantize in ONE launch (x read once) -> (stat, scale, y), or
    None when the shape is not covered by that kernel (the caller takes the two-call route)"""
    dev = require_device(x)
    assert x.is_contiguous()
    y = torch.empty_like(x)
    wsb = int(lib.bvq_stats_fakequant_fwd_workspace_bytes(ctypes.byref(desc), ptr(x), ptr(y)))
    if wsb <= 0:
        return None
    channels = int(desc.channels) if (desc.scale_per_channel and desc.channels > 1) else 1
    stat = torch.empty(channels, dtype=x.dtype, device=dev)
    scale = torch.empty(channels, dtype=scale_dtype, device=dev)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    with _DeviceGuard(dev):
        if _timer is not None:
            _timer.before('bvq_stats_fakequant_fwd')
        check(lib.bvq_stats_fakequant_fwd(ctypes.byref(desc), ptr(x), float(min_val or 0.0), int(bool(min_val)),
                                          float(int_threshold), ptr(stat), ptr(scale), ptr(y), ptr(ws), wsb,
                                          stream_ptr(dev)), 'bvq_stats_fakequant_fwd')
        if _timer is not None:
            _timer.after('bvq_stats_fakequant_fwd')
    return stat, scale, y


def absmax_scale_list(xs, outers, channels, inners, min_val, int_threshold, scale_dtype):
    """abs-max statistic of a LIST of contiguous tensors [outers[i], channels, inners[i]] (the concatenation the
    reference builds is never materialised) and the scale derived from it, ONE launch:
    -> (stat [channels], scale [channels]), or None when the list is not covered / no arrival buffer"""
    dev = require_device(*xs)
    n = len(xs)
    dt = dtype_code(xs[0].dtype)
    for x, o, i in zip(xs, outers, inners):
        assert x.is_contiguous() and x.dtype == xs[0].dtype and x.numel() == o * channels * i
    ptrs = (ctypes.c_void_p * n)(*[x.data_ptr() for x in xs])
    oa = (ctypes.c_int64 * n)(*outers)
    ia = (ctypes.c_int64 * n)(*inners)
    if not lib.bvq_absmax_list_supported(dt, n, ptrs, oa, channels, ia):
        return None
    with _DeviceGuard(dev):
        st = stream_ptr(dev)
        arrive = ws = None
        if channels > 1:
            arrive = arrival_buffer(dev, st, max(2 * channels, 18))
            if arrive is None:
                return None
        else:  # a whole-tensor statistic: one partial per unit (<= 4096) and a finishing launch
            ws = torch.empty(1 << 14, dtype=torch.uint8, device=dev)
        stat = torch.empty(channels, dtype=xs[0].dtype, device=dev)
        scale = torch.empty(channels, dtype=scale_dtype, device=dev)
        check(lib.bvq_absmax_scale_list(dt, n, ptrs, oa, channels, ia, ptr(stat), float(min_val or 0.0),
                                        int(bool(min_val)), float(int_threshold), dtype_code(scale_dtype), ptr(scale),
                                        ptr(arrive), arrive.numel() if arrive is not None else 0, ptr(ws),
                                        ws.numel() if ws is not None else 0, st), 'bvq_absmax_scale_list')
    return stat, scale


def absmax_scale(x, outer, channels, inner, min_val, int_threshold, scale_dtype, pre_op=PRE_NONE, running=None,
                 momentum=0.0, first_batch=False):
    """abs-max statistic and the scale derived from it, one call: -> (stat [channels], scale [channels]);
    running (contiguous [channels] buffer): also folded with the statistic in the same finishing launch"""
    dev = require_device(x)
    assert x.is_contiguous() and x.numel() == outer * channels * inner
    dt = dtype_code(x.dtype)
    stat = torch.empty(channels, dtype=x.dtype, device=dev)
    scale = torch.empty(channels, dtype=scale_dtype, device=dev)
    if lib.bvq_absmax_onepass_supported(dt, ptr(x), outer, channels, inner):
        # one launch: the statistic kernel's last-arriving wave per channel finishes it
        with _DeviceGuard(dev):
            st = stream_ptr(dev)
            arrive = arrival_buffer(dev, st, max(2 * channels, 18))
            if arrive is not None:
                if _timer is not None:
                    _timer.before('bvq_stats')
                check(lib.bvq_absmax_scale_onepass(
                    pre_op, dt, ptr(x), outer, channels, inner, dt, ptr(stat), float(min_val or 0.0),
                    int(bool(min_val)), float(int_threshold), dtype_code(scale_dtype), ptr(scale),
                    dtype_code(running.dtype) if running is not None else 0, ptr(running), float(momentum),
                    int(first_batch), ptr(arrive), arrive.numel(), st), 'bvq_absmax_scale_onepass')
                if _timer is not None:
                    _timer.after('bvq_stats')
                return stat, scale
    wsb = lib.bvq_stats_workspace_bytes(STAT_ABSMAX, dt, outer, channels, inner)
    ws = torch.empty(max(int(wsb), 8), dtype=torch.uint8, device=dev)
    with _DeviceGuard(dev):
        if _timer is not None:
            _timer.before('bvq_stats')
        if running is not None:
            check(lib.bvq_absmax_scale_running(pre_op, dt, ptr(x), outer, channels, inner, ptr(stat),
                                               float(min_val or 0.0), int(bool(min_val)), float(int_threshold),
                                               dtype_code(scale_dtype), ptr(scale), dtype_code(running.dtype),
                                               ptr(running), float(momentum), int(first_batch), ptr(ws), ws.numel(),
                                               stream_ptr(dev)), 'bvq_absmax_scale_running')
        else:
            check(lib.bvq_absmax_scale(pre_op, dt, ptr(x), outer, channels, inner, ptr(stat), float(min_val or 0.0),
                                       int(bool(min_val)), float(int_threshold), dtype_code(scale_dtype), ptr(scale),
                                       ptr(ws), ws.numel(), stream_ptr(dev)), 'bvq_absmax_scale')
        if _timer is not None:
            _timer.after('bvq_stats')
    return stat, scale


def kth_value(x, k, outer, channels, inner, abs_key):
    """exact k-th smallest (1-indexed) of |x| or x per channel of x[outer, channels, inner] -> [channels]"""
    dev = require_device(x)
    assert x.is_contiguous() and x.numel() == outer * channels * inner
    dt = dtype_code(x.dtype)
    out = torch.empty(channels, dtype=x.dtype, device=dev)
    wsb = int(lib.bvq_kth_workspace_bytes(dt, outer, channels, inner))
    if wsb < 0:
        raise BvqError('bvq_kth_workspace_bytes: bad arguments')
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    with _DeviceGuard(dev):
        if _timer is not None:
            _timer.before('bvq_kth_value')
        check(lib.bvq_kth_value(int(abs_key), dt, ptr(x), outer, channels, inner, int(k), ptr(out), ptr(ws), wsb,
                                stream_ptr(dev)), 'bvq_kth_value')
        if _timer is not None:
            _timer.after('bvq_kth_value')
    return out


def kth_pair(x, k_first, k_second, outer, channels, inner, abs_key):
    """two ranks of the same tensor, one histogram read where the per-tensor route applies -> [2, channels]"""
    dev = require_device(x)
    assert x.is_contiguous() and x.numel() == outer * channels * inner
    dt = dtype_code(x.dtype)
    out = torch.empty(2, channels, dtype=x.dtype, device=dev)
    wsb = int(lib.bvq_kth_workspace_bytes(dt, outer, channels, inner))
    if wsb < 0:
        raise BvqError('bvq_kth_workspace_bytes: bad arguments')
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    with _DeviceGuard(dev):
        check(lib.bvq_kth_pair(int(abs_key), dt, ptr(x), outer, channels, inner, int(k_first), int(k_second), ptr(out),
                               ptr(ws), wsb, stream_ptr(dev)), 'bvq_kth_pair')
    return out


KTH_EXPLICIT, KTH_HIGH, KTH_LOW = 0, 1, 2
_KBINS = 2048


def scale_from_stat(stat32, stat_dtype, min_val, int_threshold, scale_dtype, running=None, momentum=0.0,
                    first_batch=False):
    """all-reduced float32 statistic [channels] -> (stat in stat_dtype, scale in scale_dtype), one launch; running
    (contiguous [channels] buffer): _RuntimeStats' running average updated in the same launch"""
    dev = require_device(stat32, running)
    assert stat32.dtype == torch.float32 and stat32.is_contiguous()
    n = stat32.numel()
    stat = torch.empty(n, dtype=stat_dtype, device=dev)
    scale = torch.empty(n, dtype=scale_dtype, device=dev)
    with _DeviceGuard(dev):
        check(lib.bvq_scale_from_stat_running(
            ptr(stat32), n, dtype_code(stat_dtype), ptr(stat), float(min_val or 0.0), int(bool(min_val)),
            float(int_threshold), dtype_code(scale_dtype), ptr(scale),
            dtype_code(running.dtype) if running is not None else 0, ptr(running), float(momentum), int(first_batch),
            stream_ptr(dev)), 'bvq_scale_from_stat_running')
    return stat, scale


def fakequant_bwd_shard(desc, g, x, scale, zp, stat, rank):
    """backward of the stats-scaled per-channel graph on ONE BATCH SHARD: -> (dx without the deposit, this shard's
    float64 [2 * channels] all-gather message, first arg-max position per channel), or None if the layout is not
    covered (include/bvq.h, bvq_fakequant_bwd_shard)"""
    dev = require_device(g, x, scale, zp, stat)
    wsb = int(lib.bvq_fakequant_bwd_stats_workspace_bytes(ctypes.byref(desc)))
    if wsb <= 0 or (x.data_ptr() | g.data_ptr()) & 15:
        return None
    ch = int(desc.channels)
    dx = torch.empty_like(x)
    msg = torch.empty(2 * ch, dtype=torch.float64, device=dev)
    pos = torch.empty(ch, dtype=torch.int64, device=dev)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    stat = stat.to(x.dtype).contiguous()
    with _DeviceGuard(dev):
        st = stream_ptr(dev)
        arrive = arrival_buffer(dev, st, ch) if ONEPASS_BWD else None
        if _timer is not None:
            _timer.before('bvq_fakequant_bwd')
        check(lib.bvq_fakequant_bwd_shard(ctypes.byref(desc), ptr(g), ptr(x), ptr(scale), ptr(zp), ptr(stat), ptr(dx),
                                          ptr(msg), ptr(pos), int(rank), ptr(ws), wsb, ptr(arrive),
                                          arrive.numel() if arrive is not None else 0, st), 'bvq_fakequant_bwd_shard')
        if _timer is not None:
            _timer.after('bvq_fakequant_bwd')
    return dx, msg, pos


def shard_unpack_deposit(x, dx, gathered, world, channels, rank, first_pos, inner, scale_dtype, int_threshold,
                         quot_dtype, pre_op=PRE_NONE, want_dscale=False):
    """after the all-gather: the shards' dscale sums added in double, the owner's deposit on dx in place (include/bvq.h);
    -> float32 dscale_total [channels] if asked"""
    dev = require_device(x, dx, gathered, first_pos)
    assert gathered.dtype == torch.float64 and gathered.is_contiguous() and gathered.numel() == world * 2 * channels
    ds = torch.empty(channels, dtype=torch.float32, device=dev) if want_dscale else None
    with _DeviceGuard(dev):
        check(lib.bvq_shard_unpack_deposit(dtype_code(x.dtype), ptr(x), ptr(dx), ptr(gathered), int(world), channels,
                                           int(rank), ptr(first_pos), inner, dtype_code(scale_dtype),
                                           float(int_threshold), dtype_code(quot_dtype), pre_op, ptr(ds),
                                           stream_ptr(dev)), 'bvq_shard_unpack_deposit')
    return ds


def shard_pack(ds, tie_info, channels, rank, per_channel):
    """this shard's float64 [2 * channels] message for the backward all-gather (include/bvq.h)"""
    dev = require_device(ds, tie_info)
    assert ds.dtype == torch.float32 and ds.is_contiguous() and tie_info.dtype == torch.int64
    msg = torch.empty(2 * channels, dtype=torch.float64, device=dev)
    with _DeviceGuard(dev):
        check(lib.bvq_shard_pack(ptr(ds), ptr(tie_info), channels, int(rank), int(per_channel), ptr(msg),
                                 stream_ptr(dev)), 'bvq_shard_pack')
    return msg


def shard_unpack(gathered, world, channels, rank, per_channel, tie_info):
    """-> (dscale_total float32 [channels], total_ties int64 [1] or None); tie_info is updated in place"""
    dev = require_device(gathered, tie_info)
    assert gathered.dtype == torch.float64 and gathered.is_contiguous() and gathered.numel() == world * 2 * channels
    ds_total = torch.empty(channels, dtype=torch.float32, device=dev)
    total = None if per_channel else torch.empty(1, dtype=torch.int64, device=dev)
    with _DeviceGuard(dev):
        check(lib.bvq_shard_unpack(ptr(gathered), int(world), channels, int(rank), int(per_channel), ptr(ds_total),
                                   ptr(tie_info), ptr(total), stream_ptr(dev)), 'bvq_shard_unpack')
    return ds_total, total


def abs_moments(x, outer, channels, inner):
    """-> float32 [3 * channels] of x[outer, channels, inner]: per channel sum d, sum d^2 with d = |x| - p, and the
    pivot p (include/bvq.h): mean |x| = p + sum d / n, var |x| = (sum d^2 - (sum d)^2 / n) / (n - 1)"""
    dev = require_device(x)
    assert x.is_contiguous() and x.numel() == outer * channels * inner
    dt = dtype_code(x.dtype)
    sums = torch.empty(3 * channels, dtype=torch.float32, device=dev)
    wsb = int(lib.bvq_abs_moments_workspace_bytes(dt, outer, channels, inner))
    if wsb < 0:
        raise BvqError('bvq_abs_moments_workspace_bytes: bad arguments')
    ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=dev)
    with _DeviceGuard(dev):
        check(lib.bvq_abs_moments(dt, ptr(x), outer, channels, inner, ptr(sums), ptr(ws), wsb, stream_ptr(dev)),
              'bvq_abs_moments')
    return sums


def abs_affine_bwd(x, a, b, outer, channels, inner):
    """dx = sgn(x) * (a[c] + b[c] * |x|); a, b float32 [channels]"""
    dev = require_device(x, a, b)
    assert x.is_contiguous() and a.dtype == torch.float32 and b.dtype == torch.float32
    assert a.numel() == channels and b.numel() == channels
    dx = torch.empty_like(x)
    with _DeviceGuard(dev):
        check(lib.bvq_abs_affine_bwd(dtype_code(x.dtype), ptr(x), ptr(a.contiguous()), ptr(b.contiguous()), ptr(dx),
                                     outer, channels, inner, stream_ptr(dev)), 'bvq_abs_affine_bwd')
    return dx


class KthSelectSteps:
    """bvq_kth_value in steps (include/bvq.h) for a batch-sharded tensor: between hist(p) and pick(p)
    the caller sums the returned counters over the shards (brevitas_amd.distributed.sharded_kth_value).
    rule / q: the rank is derived on the device from the global element count (KTH_HIGH / KTH_LOW), or
    KTH_EXPLICIT with k."""

    def __init__(self, x, outer, channels, inner, abs_key, rule, q, k=0):
        self.dev = require_device(x)
        assert x.is_contiguous() and x.numel() == outer * channels * inner
        self.x, self.layout, self.abs_key = x, (outer, channels, inner), int(abs_key)
        self.rule, self.q, self.k = int(rule), float(q), int(k)
        self.per_channel = outer * inner if channels > 1 else x.numel()  # elements per channel on THIS shard
        self.dt = dtype_code(x.dtype)
        self.passes = int(lib.bvq_kth_passes(self.dt))
        self.wsb = int(lib.bvq_kth_workspace_bytes(self.dt, outer, channels, inner))
        if self.wsb < 0:
            raise BvqError('bvq_kth_workspace_bytes: bad arguments')
        self.ws = torch.empty(self.wsb, dtype=torch.uint8, device=self.dev)

    def begin(self):
        with _DeviceGuard(self.dev):
            check(lib.bvq_kth_begin(self.dt, self.layout[1], self.rule, self.k, self.q, ptr(self.ws), self.wsb,
                                    stream_ptr(self.dev)), 'bvq_kth_begin')

    def hist(self, p):
        """histogram this shard's elements for pass p -> the [channels * 2048] counters (int32 view of the
        unsigned counters: a two's-complement sum over the shards is their unsigned sum)"""
        outer, channels, inner = self.layout
        with _DeviceGuard(self.dev):
            check(lib.bvq_kth_hist(self.abs_key, self.dt, ptr(self.x), outer, channels, inner, p, ptr(self.ws),
                                   self.wsb, stream_ptr(self.dev)), 'bvq_kth_hist')
        off = int(lib.bvq_kth_hist_offset(self.dt, channels, p))
        return self.ws[off:off + 4 * channels * _KBINS].view(torch.int32)

    def pick(self, p):
        with _DeviceGuard(self.dev):
            check(lib.bvq_kth_pick(self.dt, self.layout[1], p, self.rule, self.q, ptr(self.ws), self.wsb,
                                   stream_ptr(self.dev)), 'bvq_kth_pick')

    def finish(self):
        out = torch.empty(self.layout[1], dtype=self.x.dtype, device=self.dev)
        with _DeviceGuard(self.dev):
            check(lib.bvq_kth_finish(self.abs_key, self.dt, self.layout[1], ptr(out), ptr(self.ws), self.wsb,
                                     stream_ptr(self.dev)), 'bvq_kth_finish')
        return out


class KthWideSteps:
    """the sharded selection of a whole-tensor statistic with the 15-bit first digit (include/bvq.h, bvq_kthw_*):
    same interface as KthSelectSteps -- begin / hist(p) / pick(p) / finish, `passes`, `per_channel` -- for
    brevitas_amd.distributed.sharded_kth_value; one pass for |x| of a 16-bit type, two otherwise"""

    def __init__(self, x, abs_key, rule, q, k=0):
        self.dev = require_device(x)
        assert x.is_contiguous()
        self.x, self.abs_key = x.reshape(-1), int(abs_key)
        self.rule, self.q, self.k = int(rule), float(q), int(k)
        self.per_channel = x.numel()
        self.dt = dtype_code(x.dtype)
        self.passes = int(lib.bvq_kthw_plan(self.abs_key, self.dt, -1, None, None))
        self.wsb = int(lib.bvq_kth_workspace_bytes(self.dt, 1, 1, max(x.numel(), 1)))
        if self.passes < 1 or self.wsb < 0:
            raise BvqError('bvq_kthw_plan / bvq_kth_workspace_bytes: bad arguments')
        self.ws = torch.empty(self.wsb, dtype=torch.uint8, device=self.dev)

    def begin(self):
        with _DeviceGuard(self.dev):
            check(lib.bvq_kthw_begin(self.abs_key, self.dt, ptr(self.ws), self.wsb, stream_ptr(self.dev)), 'bvq_kthw_begin')

    def hist(self, p):
        """-> the counters of pass p to be summed over the shards (int32 view of the unsigned counters)"""
        with _DeviceGuard(self.dev):
            check(lib.bvq_kthw_hist(self.abs_key, self.dt, ptr(self.x) if self.x.numel() else None, self.x.numel(), p,
                                    ptr(self.ws), self.wsb, stream_ptr(self.dev)), 'bvq_kthw_hist')
        off, words = ctypes.c_int64(0), ctypes.c_int64(0)
        lib.bvq_kthw_plan(self.abs_key, self.dt, p, ctypes.byref(off), ctypes.byref(words))
        return self.ws[off.value:off.value + 4 * words.value].view(torch.int32)

    def pick(self, p):
        with _DeviceGuard(self.dev):
            check(lib.bvq_kthw_pick(self.abs_key, self.dt, p, self.rule, self.k, self.q, ptr(self.ws), self.wsb,
                                    stream_ptr(self.dev)), 'bvq_kthw_pick')

    def finish(self):
        out = torch.empty(1, dtype=self.x.dtype, device=self.dev)
        with _DeviceGuard(self.dev):
            check(lib.bvq_kthw_finish(self.abs_key, self.dt, ptr(out), ptr(self.ws), self.wsb, stream_ptr(self.dev)),
                  'bvq_kthw_finish')
        return out


def running_stats_update(running, stat, momentum, first_batch):
    """in-place batch-norm style update of a running statistic (one launch)"""
    dev = require_device(running, stat)
    assert running.is_contiguous() and running.numel() == stat.numel()
    stat = stat.contiguous()
    with _DeviceGuard(dev):
        check(lib.bvq_running_stats_update(dtype_code(running.dtype), ptr(running), dtype_code(stat.dtype),
                                           ptr(stat), running.numel(), float(momentum), int(first_batch),
                                           stream_ptr(dev)), 'bvq_running_stats_update')
    return running


def tie_info_buffer(channels, device):
    nbytes = int(lib.bvq_tie_info_bytes(int(channels)))
    return torch.empty(nbytes // 8, dtype=torch.int64, device=device)


def stat_tie_scan(match, x, stat, outer, channels, inner, dx_zero_fill=None):
    """record which elements of x attain `stat`; returns the tie_info buffer (int64, device)"""
    dev = require_device(x, stat, dx_zero_fill)
    assert x.is_contiguous()
    stat = stat.to(x.dtype).contiguous()
    info = tie_info_buffer(channels, dev)
    with _DeviceGuard(dev):
        check(lib.bvq_stat_tie_scan(match, dtype_code(x.dtype), ptr(x), ptr(stat), outer, channels, inner,
                                    ptr(dx_zero_fill), ptr(info), stream_ptr(dev)), 'bvq_stat_tie_scan')
    return info


def stat_tie_apply(match, x, stat, gstat, info, dx, outer, channels, inner, mode_add, total_ties=None,
                   pre_op=PRE_NONE):
    dev = require_device(x, stat, gstat, info, dx, total_ties)
    assert x.is_contiguous() and dx.is_contiguous() and dx.dtype == x.dtype
    stat = stat.to(x.dtype).contiguous()
    gstat = gstat.to(x.dtype).contiguous()
    with _DeviceGuard(dev):
        check(lib.bvq_stat_tie_apply(match, pre_op, dtype_code(x.dtype), ptr(x), ptr(stat), ptr(gstat), ptr(info),
                                     ptr(total_ties), ptr(dx), outer, channels, inner, int(mode_add),
                                     stream_ptr(dev)), 'bvq_stat_tie_apply')
    return dx


def stat_tie_apply_dscale(x, stat, dscale, scale_dtype, int_threshold, quot_dtype, info, dx, outer, channels,
                          inner, total_ties=None, pre_op=PRE_NONE):
    """deposit the statistic's gradient derived from float32 dscale sums (fused quantizer backward)"""
    dev = require_device(x, stat, dscale, info, dx, total_ties)
    assert x.is_contiguous() and dx.is_contiguous() and dx.dtype == x.dtype and dscale.dtype == torch.float32
    stat = stat.to(x.dtype).contiguous()
    with _DeviceGuard(dev):
        check(lib.bvq_stat_tie_apply_dscale(pre_op, dtype_code(x.dtype), ptr(x), ptr(stat), ptr(dscale),
                                            dtype_code(scale_dtype), float(int_threshold), dtype_code(quot_dtype),
                                            ptr(info), ptr(total_ties), ptr(dx), outer, channels, inner,
                                            stream_ptr(dev)), 'bvq_stat_tie_apply_dscale')
    return dx


def fakequant_bwd_stats(desc, g, x, scale, zp, stat, scale_dtype, int_threshold, quot_dtype, want_dscale=False):
    """backward of the stats-scaled per-channel graph in two launches: dx with the statistic's gradient already
    deposited on the arg-max elements (and the float32 dscale sums if asked); None if the layout is not covered"""
    dev = require_device(g, x, scale, zp, stat)
    wsb = int(lib.bvq_fakequant_bwd_stats_workspace_bytes(ctypes.byref(desc)))
    if wsb <= 0 or (x.data_ptr() | g.data_ptr()) & 15:  # (views into the middle of a buffer: the general route)
        return None
    dx = torch.empty_like(x)
    ds = torch.empty(int(desc.channels), dtype=torch.float32, device=dev)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    stat = stat.to(x.dtype).contiguous()
    with _DeviceGuard(dev):
        st = stream_ptr(dev)
        arrive = None
        if ONEPASS_BWD and lib.bvq_fakequant_bwd_stats_onepass_supported(ctypes.byref(desc)):
            arrive = arrival_buffer(dev, st, int(desc.channels))
        if _timer is not None:
            _timer.before('bvq_fakequant_bwd')
        if arrive is not None:  # one launch: the wave that completes a channel finishes it
            check(lib.bvq_fakequant_bwd_stats_onepass(ctypes.byref(desc), ptr(g), ptr(x), ptr(scale), ptr(zp), ptr(stat),
                                                      ptr(dx), ptr(ds), dtype_code(scale_dtype), float(int_threshold),
                                                      dtype_code(quot_dtype), ptr(ws), wsb, ptr(arrive), arrive.numel(),
                                                      st), 'bvq_fakequant_bwd_stats_onepass')
        else:
            check(lib.bvq_fakequant_bwd_stats(ctypes.byref(desc), ptr(g), ptr(x), ptr(scale), ptr(zp), ptr(stat), ptr(dx),
                                              ptr(ds), dtype_code(scale_dtype), float(int_threshold),
                                              dtype_code(quot_dtype), ptr(ws), wsb, st), 'bvq_fakequant_bwd_stats')
        if _timer is not None:
            _timer.after('bvq_fakequant_bwd')
    return (dx, ds) if want_dscale else dx


def variant_fwd(desc, x, scale, pre_scale=None, zp=None, pre_zp=None):
    """forward of the sign / decoupled / truncating quantizers (include/bvq.h, bvq_variant_fwd) -> y in desc.ct_dtype"""
    dev = require_device(x, scale, pre_scale, zp, pre_zp)
    ct = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}[desc.ct_dtype]
    y = torch.empty(x.shape, dtype=ct, device=dev)
    with _DeviceGuard(dev):
        check(lib.bvq_variant_fwd(ctypes.byref(desc), ptr(x), ptr(scale), ptr(pre_scale), ptr(zp), ptr(pre_zp), ptr(y),
                                  stream_ptr(dev)), 'bvq_variant_fwd')
    return y


def variant_bwd(desc, g, x, scale, pre_scale=None, zp=None, pre_zp=None, need_dscale=False, need_dpre=False):
    """-> (dx, dscale float32 or None, dpre_scale float32 or None)"""
    dev = require_device(g, x, scale, pre_scale, zp, pre_zp)
    dx = torch.empty_like(x)
    nsum = int(desc.channels) if (desc.scale_per_channel and desc.channels > 1) else 1
    ds = torch.empty(nsum, dtype=torch.float32, device=dev) if need_dscale else None
    dp = torch.empty(nsum, dtype=torch.float32, device=dev) if need_dpre else None
    ws, wsb = None, 0
    if need_dscale or need_dpre:
        wsb = int(lib.bvq_variant_bwd_workspace_bytes(ctypes.byref(desc)))
        if wsb < 0:
            raise BvqError('bvq_variant_bwd_workspace_bytes: ' + last_error())
        ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=dev)
    with _DeviceGuard(dev):
        check(lib.bvq_variant_bwd(ctypes.byref(desc), ptr(g), ptr(x), ptr(scale), ptr(pre_scale), ptr(zp), ptr(pre_zp),
                                  ptr(dx), ptr(ds), ptr(dp), ptr(ws), wsb, stream_ptr(dev)), 'bvq_variant_bwd')
    return dx, ds, dp


def fakequant_fwd_bounds(desc, x, scale, zp, bounds):
    """bvq_fakequant_fwd with the integer range [qmin, qmax] read from the device (float32 [2]) -> y"""
    dev = require_device(x, scale, zp, bounds)
    assert bounds.dtype == torch.float32 and bounds.numel() == 2 and bounds.is_contiguous()
    ct = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}[desc.ct_dtype]
    y = torch.empty(x.shape, dtype=ct, device=dev)
    with _DeviceGuard(dev):
        check(lib.bvq_fakequant_fwd_bounds(ctypes.byref(desc), ptr(x), ptr(scale), ptr(zp), ptr(bounds), ptr(y),
                                           stream_ptr(dev)), 'bvq_fakequant_fwd_bounds')
    return y


def fakequant_bwd_bounds(desc, g, x, scale, zp, bounds, need_dbounds):
    """-> (dx, dscale float32 [n], dbounds float32 [2, n] or None): n = channels or 1"""
    dev = require_device(g, x, scale, zp, bounds)
    dx = torch.empty_like(x)
    nsum = int(desc.channels) if (desc.scale_per_channel and desc.channels > 1) else 1
    ds = torch.empty(nsum, dtype=torch.float32, device=dev)
    db = torch.empty(2, nsum, dtype=torch.float32, device=dev) if need_dbounds else None
    wsb = int(lib.bvq_fakequant_bwd_workspace_bytes(ctypes.byref(desc)))
    if wsb < 0:
        raise BvqError('bvq_fakequant_bwd_workspace_bytes: ' + last_error())
    ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=dev)
    with _DeviceGuard(dev):
        check(lib.bvq_fakequant_bwd_bounds(ctypes.byref(desc), ptr(g), ptr(x), ptr(scale), ptr(zp), ptr(bounds), ptr(dx),
                                           ptr(ds), ptr(db), ptr(ws), wsb, stream_ptr(dev)), 'bvq_fakequant_bwd_bounds')
    return dx, ds, db


def histc(x, absmax, bins):
    """torch.histc(x, bins, min=-absmax, max=absmax) with the bounds read from the device -> int32 [bins]"""
    dev = require_device(x, absmax)
    x = x.contiguous()
    counts = torch.empty(bins, dtype=torch.int32, device=dev)
    with _DeviceGuard(dev):
        check(lib.bvq_histc(dtype_code(x.dtype), ptr(x), x.numel(), ptr(absmax.to(x.dtype).reshape(1)), int(bins),
                            ptr(counts), stream_ptr(dev)), 'bvq_histc')
    return counts


def selftest_div_f16r(a, scales):
    """diagnostic: the quotient the float16 kernels compute for every (scale, numerator) pair -> float32 [n_s, n_a]"""
    dev = require_device(a, scales)
    assert a.dtype == torch.float32 and scales.dtype == torch.float32 and a.is_contiguous() and scales.is_contiguous()
    out = torch.empty(scales.numel(), a.numel(), dtype=torch.float32, device=dev)
    with _DeviceGuard(dev):
        check(lib.bvq_selftest_div_f16r(ptr(a), a.numel(), ptr(scales), scales.numel(), ptr(out), stream_ptr(dev)),
              'bvq_selftest_div_f16r')
    return out


def learned_scale(value, min_val, int_threshold, scale_dtype):
    """scale = |clamp_min(value, min_val)| / int_threshold, one launch -> [value.numel()] in scale_dtype"""
    dev = require_device(value)
    v = value.detach().reshape(-1).contiguous()
    scale = torch.empty(v.numel(), dtype=scale_dtype, device=dev)
    with _DeviceGuard(dev):
        check(lib.bvq_learned_scale(dtype_code(v.dtype), ptr(v), v.numel(), float(min_val or 0.0), int(bool(min_val)),
                                    float(int_threshold), dtype_code(scale_dtype), ptr(scale), stream_ptr(dev)),
              'bvq_learned_scale')
    return scale


def fakequant_bwd_learned(desc, g, x, scale, zp, value, min_val, int_threshold, gscale=None):
    """quantizer backward + the learned scale's backward in its last launch -> (dx, dscale float32, dvalue flat)"""
    dev = require_device(g, x, scale, zp, value, gscale)
    dx = torch.empty_like(x)
    pc = desc.scale_per_channel and desc.channels > 1
    nsum = int(desc.channels) if pc else 1
    v = value.detach().reshape(-1).contiguous()
    assert v.numel() == nsum and (gscale is None or (gscale.numel() == nsum and gscale.is_contiguous()))
    ds = torch.empty(nsum, dtype=torch.float32, device=dev)
    dv = torch.empty(nsum, dtype=v.dtype, device=dev)
    wsb = int(lib.bvq_fakequant_bwd_workspace_bytes(ctypes.byref(desc)))
    if wsb < 0:
        raise BvqError('bvq_fakequant_bwd_workspace_bytes: ' + last_error())
    ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=dev)
    with _DeviceGuard(dev):
        if _timer is not None:
            _timer.before('bvq_fakequant_bwd')
        check(lib.bvq_fakequant_bwd_learned(ctypes.byref(desc), ptr(g), ptr(x), ptr(scale), ptr(zp), ptr(dx), ptr(ds),
                                            ptr(v), dtype_code(v.dtype), float(min_val or 0.0), int(bool(min_val)),
                                            float(int_threshold), ptr(gscale), ptr(dv), ptr(ws), wsb, stream_ptr(dev)),
              'bvq_fakequant_bwd_learned')
        if _timer is not None:
            _timer.after('bvq_fakequant_bwd')
    return dx, ds, dv


def fakequant_bwd(desc, g, x, scale, zp, need_dscale, need_dzp, tie_stat=None):
    """-> (dx, dscale, dzp[, tie_info]); tie_stat: abs-max statistic whose ties are recorded on the fly"""
    dev = require_device(g, x, scale, zp, tie_stat)
    dx = torch.empty_like(x)
    pc = (desc.scale_per_channel or desc.zp_per_channel) and desc.channels > 1
    nsum = int(desc.channels) if pc else 1
    ds = torch.empty(nsum, dtype=torch.float32, device=dev) if need_dscale else None
    dz = torch.empty(nsum, dtype=torch.float32, device=dev) if need_dzp else None
    info = None
    if tie_stat is not None:
        tie_stat = tie_stat.to(x.dtype).contiguous()
        info = tie_info_buffer(desc.channels, dev)
    ws = None
    wsb = 0
    if need_dscale or need_dzp:
        wsb = int(lib.bvq_fakequant_bwd_workspace_bytes(ctypes.byref(desc)))
        if wsb < 0:
            raise BvqError('bvq_fakequant_bwd_workspace_bytes: ' + last_error())
        ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=dev)
    with _DeviceGuard(dev):
        if _timer is not None:
            _timer.before('bvq_fakequant_bwd')
        check(lib.bvq_fakequant_bwd(ctypes.byref(desc), ptr(g), ptr(x), ptr(scale), ptr(zp), ptr(dx), ptr(ds),
                                    ptr(dz), ptr(tie_stat), ptr(info), ptr(ws), wsb, stream_ptr(dev)),
              'bvq_fakequant_bwd')
        if _timer is not None:
            _timer.after('bvq_fakequant_bwd')
    if tie_stat is not None:
        return dx, ds, dz, info
    return dx, ds, dz
