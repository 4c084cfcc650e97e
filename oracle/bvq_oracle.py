"""ctypes/numpy front-end of oracle/bvq_oracle.c (test infrastructure only).

Tensors are numpy arrays: float32 as np.float32, bfloat16 / float16 as np.uint16 bit patterns
(helpers convert from / to torch tensors).  The descriptor mirrors `bvq_quant_desc` of
include/bvq.h so tests drive the HIP library and the oracle with the same arguments.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libbvq_oracle.so')

F32, BF16, F16 = 0, 1, 2
ROUND, FLOOR, CEIL, ROUND_TO_ZERO, DPU_ROUND = range(5)
(OP_ROUND, OP_FLOOR, OP_CEIL, OP_ROUND_TO_ZERO, OP_DPU_ROUND, OP_BINARY_SIGN, OP_TERNARY_SIGN,
 OP_ABS) = range(8)
STAT_ABSMAX, STAT_MINMAX = 0, 1
SCALAR_OPMATH, SCALAR_CAST = 0, 1
OUT_DEQUANT, OUT_INT = 0, 1
PRE_NONE, PRE_RELU = 0, 1


class QuantDesc(ctypes.Structure):
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


def build(force=False):
    src = os.path.join(_HERE, 'bvq_oracle.c')
    hdr = os.path.join(_HERE, '..', 'include', 'bvq.h')
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.run(['make', '-C', _HERE, '-B', 'libbvq_oracle.so'], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def np_dtype(dt):
    return np.float32 if dt == F32 else np.uint16


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _c(a, dt):
    a = np.ascontiguousarray(a)
    assert a.dtype == np_dtype(dt), (a.dtype, dt)
    return a


# ---- torch <-> oracle arrays ----------------------------------------------------------------------

def from_torch(t):
    """torch tensor -> (numpy array, dtype code); bf16/f16 as uint16 bit patterns"""
    import torch
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.float32:
        return t.numpy().copy(), F32
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16).copy(), BF16
    if t.dtype == torch.float16:
        return t.view(torch.int16).numpy().view(np.uint16).copy(), F16
    raise TypeError(t.dtype)


def to_torch(a, dt):
    import torch
    if dt == F32:
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32).copy())
    t = torch.from_numpy(np.ascontiguousarray(a).view(np.int16).copy())
    return t.view(torch.bfloat16 if dt == BF16 else torch.float16)


def to_float32(a, dt):
    """widen an oracle array to float32 values"""
    if dt == F32:
        return np.asarray(a, dtype=np.float32)
    if dt == BF16:
        return (np.asarray(a, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)
    return np.asarray(a, dtype=np.uint16).view(np.float16).astype(np.float32)


# ---- elementwise ----------------------------------------------------------------------------------

def unary(op, x, dt):
    x = _c(x, dt)
    y = np.empty_like(x)
    lib().orc_unary(op, dt, _ptr(x), _ptr(y), ctypes.c_int64(x.size))
    return y


def scalar_clamp(x, dt, lo=None, hi=None):
    x = _c(x, dt)
    y = np.empty_like(x)
    lib().orc_scalar_clamp(dt, _ptr(x), _ptr(y), ctypes.c_int64(x.size),
                           ctypes.c_double(0.0 if lo is None else lo), int(lo is not None),
                           ctypes.c_double(0.0 if hi is None else hi), int(hi is not None))
    return y


def tensor_clamp(x, lo, hi, dt):
    x, lo, hi = _c(x, dt), _c(lo, dt), _c(hi, dt)
    full = int(lo.size == x.size and x.size != 1)
    y = np.empty_like(x)
    lib().orc_tensor_clamp(dt, _ptr(x), _ptr(lo), _ptr(hi), full, _ptr(y), ctypes.c_int64(x.size))
    return y


def tensor_clamp_bwd(g, x, lo, hi, dt):
    g, x, lo, hi = _c(g, dt), _c(x, dt), _c(lo, dt), _c(hi, dt)
    full = int(lo.size == x.size and x.size != 1)
    dx = np.empty_like(x)
    lib().orc_tensor_clamp_bwd(dt, _ptr(g), _ptr(x), _ptr(lo), _ptr(hi), full, _ptr(dx),
                               ctypes.c_int64(x.size))
    return dx


def abs_binary_sign_grad_bwd(g, x, dt):
    g, x = _c(g, dt), _c(x, dt)
    dx = np.empty_like(x)
    lib().orc_abs_binary_sign_grad_bwd(dt, _ptr(g), _ptr(x), _ptr(dx), ctypes.c_int64(x.size))
    return dx


# ---- statistics -----------------------------------------------------------------------------------

def stats(kind, x, dt, outer, channels, inner, pre_op=PRE_NONE):
    x = _c(x, dt)
    assert x.size == outer * channels * inner
    out = np.empty(channels * (2 if kind == STAT_MINMAX else 1), dtype=np.float32)
    lib().orc_stats(kind, pre_op, dt, _ptr(x), ctypes.c_int64(outer), ctypes.c_int64(channels),
                    ctypes.c_int64(inner), _ptr(out))
    return out


def abs_moments(x, dt, outer, channels, inner):
    """-> float64 [2 * channels]: sum |x|, then sum x^2"""
    x = _c(x, dt)
    assert x.size == outer * channels * inner
    out = np.empty(2 * channels, dtype=np.float64)
    lib().orc_abs_moments(dt, _ptr(x), ctypes.c_int64(outer), ctypes.c_int64(channels), ctypes.c_int64(inner),
                          _ptr(out))
    return out


def kth_value(x, dt, outer, channels, inner, k, abs_key):
    x = _c(x, dt)
    out = np.empty(channels, dtype=np.float32)
    rc = lib().orc_kth_value(int(abs_key), dt, _ptr(x), ctypes.c_int64(outer), ctypes.c_int64(channels),
                             ctypes.c_int64(inner), ctypes.c_int64(k), _ptr(out))
    if rc != 0:
        raise ValueError('k out of range')
    return out


def absmax_bwd(x, stat, gstat, dt, outer, channels, inner):
    x, stat, gstat = _c(x, dt), _c(stat, dt), _c(gstat, dt)
    dx = np.empty_like(x)
    lib().orc_absmax_bwd(dt, _ptr(x), _ptr(stat), _ptr(gstat), _ptr(dx), ctypes.c_int64(outer),
                         ctypes.c_int64(channels), ctypes.c_int64(inner))
    return dx


# ---- affine quantizer -----------------------------------------------------------------------------

def make_desc(outer, channels, inner, x_dtype, ct_dtype, scale_dtype, zp_dtype=F32,
              scale_per_channel=False, zp_per_channel=False, qmin=-128.0, qmax=127.0,
              round_mode=ROUND, scalar_mode=SCALAR_OPMATH, clamp_ste=False, out_kind=OUT_DEQUANT,
              pre_op=PRE_NONE):
    return QuantDesc(outer, channels, inner, x_dtype, ct_dtype, scale_dtype, zp_dtype,
                     int(scale_per_channel), int(zp_per_channel), qmin, qmax, round_mode, scalar_mode,
                     int(clamp_ste), out_kind, pre_op)


def fakequant_fwd(desc, x, scale, zp, want_codes=True):
    x = _c(x, desc.x_dtype)
    scale = _c(scale, desc.scale_dtype)
    zp = _c(zp, desc.zp_dtype)
    y = np.empty(x.shape, dtype=np_dtype(desc.ct_dtype))
    codes = np.empty(x.shape, dtype=np.int32) if want_codes else None
    lib().orc_fakequant_fwd(ctypes.byref(desc), _ptr(x), _ptr(scale), _ptr(zp), _ptr(y), _ptr(codes))
    return y, codes


def fakequant_bwd(desc, g, x, scale, zp):
    g = _c(g, desc.ct_dtype)
    x = _c(x, desc.x_dtype)
    scale = _c(scale, desc.scale_dtype)
    zp = _c(zp, desc.zp_dtype)
    pc = (desc.scale_per_channel or desc.zp_per_channel) and desc.channels > 1
    nsum = desc.channels if pc else 1
    dx = np.empty(x.shape, dtype=np_dtype(desc.x_dtype))
    ds = np.empty(nsum, dtype=np.float32)
    dz = np.empty(nsum, dtype=np.float32)
    lib().orc_fakequant_bwd(ctypes.byref(desc), _ptr(g), _ptr(x), _ptr(scale), _ptr(zp), _ptr(dx),
                            _ptr(ds), _ptr(dz))
    return dx, ds, dz


def variant_fwd(desc, x, scale, pre_scale=None, zp=None, pre_zp=None):
    """BinaryQuant / ClampedBinaryQuant / TernaryQuant / DecoupledIntQuant / TruncIntQuant forward -> y (ct dtype)"""
    x = _c(x, desc.x_dtype)
    y = np.empty(x.shape, dtype=np_dtype(desc.ct_dtype))
    lib().orc_variant_fwd(ctypes.byref(desc), _ptr(x), _ptr(_c(scale, desc.scale_dtype)),
                          _ptr(_c(pre_scale, desc.scale_dtype)) if pre_scale is not None else None,
                          _ptr(_c(zp, desc.zp_dtype)) if zp is not None else None,
                          _ptr(_c(pre_zp, desc.zp_dtype)) if pre_zp is not None else None, _ptr(y))
    return y


def variant_bwd(desc, g, x, scale, pre_scale=None, zp=None, pre_zp=None):
    """-> (dx, dscale float32, dpre_scale float32)"""
    x = _c(x, desc.x_dtype)
    g = _c(g, desc.ct_dtype)
    nsum = desc.channels if (desc.scale_per_channel and desc.channels > 1) else 1
    dx = np.empty(x.shape, dtype=np_dtype(desc.x_dtype))
    ds = np.empty(nsum, dtype=np.float32)
    dp = np.empty(nsum, dtype=np.float32)
    lib().orc_variant_bwd(ctypes.byref(desc), _ptr(g), _ptr(x), _ptr(_c(scale, desc.scale_dtype)),
                          _ptr(_c(pre_scale, desc.scale_dtype)) if pre_scale is not None else None,
                          _ptr(_c(zp, desc.zp_dtype)) if zp is not None else None,
                          _ptr(_c(pre_zp, desc.zp_dtype)) if pre_zp is not None else None, _ptr(dx), _ptr(ds), _ptr(dp))
    return dx, ds, dp


def step_stats_scaled(desc, x, g, min_val, int_threshold):
    """timed CPU baseline: abs-max stats -> scale -> fwd -> bwd (threaded)"""
    x = _c(x, desc.x_dtype)
    g = _c(g, desc.ct_dtype)
    y = np.empty(x.shape, dtype=np_dtype(desc.ct_dtype))
    dx = np.empty(x.shape, dtype=np_dtype(desc.x_dtype))
    scale = np.empty(desc.channels, dtype=np_dtype(desc.scale_dtype))
    stat = np.empty(desc.channels, dtype=np.float32)
    ds = np.empty(desc.channels, dtype=np.float32)
    lib().orc_step_stats_scaled(ctypes.byref(desc), _ptr(x), _ptr(g), _ptr(y), _ptr(dx), _ptr(scale),
                                _ptr(stat), _ptr(ds), ctypes.c_double(min_val),
                                ctypes.c_double(int_threshold))
    return y, dx, scale, stat, ds


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))
