"""IntQuant: scaled, shifted, uniform integer quantizer (drop-in for B/core/quant/int_base.py:15-97).

Same constructor, same `forward(scale, zero_point, bit_width, x)`, `to_int`, `min_int`, `max_int`.
When the injected float_to_int_impl / tensor_clamp_impl are the library's own wrappers and the bit
width is host-known, the whole chain  x/scale + zp -> round -> clamp -> (. - zp) * scale  runs as
ONE HIP kernel (and its autograd as one more); anything else runs the same chain op by op through
the HIP-backed straight-through ops, exactly as the reference composes it.
"""
from typing import NamedTuple, Optional

import torch
from torch import Tensor
from torch.nn import Module

import brevitas_amd.config as config
from brevitas_amd import _native as nat
from brevitas_amd.core.function_wrapper import RoundSte, TensorClamp
from brevitas_amd.core.quant import _fused
from brevitas_amd.core.quant.delay import DelayWrapper, _NoDelay
from brevitas_amd.function.ops import int_range_host, max_int, min_int


class QCDQOperands(NamedTuple):
    """what QuantizeLinear -> Clip -> DequantizeLinear consume (IntQuant.to_qcdq_operands)"""
    int_codes: Tensor      # int8 / uint8 (up to 8 bits) or int32, shape of x
    scale: Tensor          # 0-dim or [channels]
    zero_point: Tensor     # dtype of int_codes, shape of scale
    axis: Optional[int]    # channel axis of a per-channel scale, None for per-tensor

    def dequantize(self) -> Tensor:
        """DequantizeLinear: (codes - zero_point) * scale along `axis` (B/export/common/handler/qcdq.py:26-27)"""
        shape = [1] * self.int_codes.dim()
        if self.axis is not None:
            shape[self.axis] = -1
        return (self.int_codes.to(self.scale.dtype) - self.zero_point.to(self.scale.dtype).reshape(shape)) \
            * self.scale.reshape(shape)


class IntQuant(torch.nn.Module):
    """
    Args:
        narrow_range (bool): restrict the integer range to a narrow (symmetric) one.
        signed (bool): signed or unsigned integer range.
        float_to_int_impl (Module): float -> integer conversion. Default: RoundSte()
        tensor_clamp_impl (Module): clamp to [min_int, max_int]. Default: TensorClamp()
        quant_delay_steps (int): training steps during which the input is passed through. Default: 0

    Examples (B/core/quant/int_base.py:32-38):
        >>> int_quant = IntQuant(narrow_range=True, signed=True)
        >>> scale, zero_point, bit_width = torch.tensor(0.01), torch.tensor(0.), torch.tensor(4.)
        >>> int_quant(scale, zero_point, bit_width, torch.Tensor([0.042, -0.053, 0.31, -0.44]))
        tensor([ 0.0400, -0.0500,  0.0700, -0.0700])
    """

    def __init__(self, narrow_range: bool, signed: bool, float_to_int_impl: Module = RoundSte(),
                 tensor_clamp_impl: Module = TensorClamp(), quant_delay_steps: int = 0):
        super().__init__()
        self.float_to_int_impl = float_to_int_impl
        self.tensor_clamp_impl = tensor_clamp_impl
        self.signed = signed
        self.narrow_range = narrow_range
        self.delay_wrapper = DelayWrapper(quant_delay_steps)

    # ---- fused path ---------------------------------------------------------------------------------

    def _fused_args(self, scale: Tensor, zero_point: Tensor, bit_width: Tensor, x: Tensor):
        """(plan, qmin, qmax, round_mode, clamp_ste) if the fused kernel covers this call, else None"""
        if not config.FUSED_PATHS:
            return None
        round_mode = getattr(self.float_to_int_impl, 'bvq_round_mode', None)
        clamp_ste = getattr(self.tensor_clamp_impl, 'bvq_clamp_ste', None)
        bw = getattr(bit_width, 'bvq_host_value', None)
        if round_mode is None or clamp_ste is None or bw is None:
            return None
        p = _fused.plan(x, scale, zero_point)
        if p is None:
            return None
        qmin, qmax = int_range_host(self.signed, self.narrow_range, bw)
        return p, qmin, qmax, round_mode, clamp_ste

    # ---- reference surface ----------------------------------------------------------------------------

    def to_int(self, scale: Tensor, zero_point: Tensor, bit_width: Tensor, x: Tensor) -> Tensor:
        fa = self._fused_args(scale, zero_point, bit_width, x)
        differentiable = torch.is_grad_enabled() and any(
            t.requires_grad for t in (x, scale, zero_point, bit_width))
        if fa is not None and not differentiable:
            p, qmin, qmax, round_mode, clamp_ste = fa
            return _fused.FakeQuantFn.apply(x, scale, zero_point, p, qmin, qmax, round_mode, clamp_ste,
                                            nat.OUT_INT)
        # op-by-op chain (B/core/quant/int_base.py:69-75)
        y = x / scale
        y = y + zero_point
        min_int_val = self.min_int(bit_width)
        max_int_val = self.max_int(bit_width)
        y = self.float_to_int_impl(y)
        y = self.tensor_clamp_impl(y, min_val=min_int_val, max_val=max_int_val)
        return y

    def to_int_codes(self, scale: Tensor, zero_point: Tensor, bit_width: Tensor, x: Tensor) -> Tensor:
        """to_int as an INTEGER tensor, in the dtype QuantTensor.int() uses (B/quant_tensor/__init__.py:
        174-187): int8 / uint8 up to 8 bits, int32 above.  One read of x and one 1-byte (4-byte) write per
        element -- the operand layout of integer GEMMs and of the QCDQ exporters
        (B/export/common/handler/qcdq.py:47-90).  Not differentiable."""
        bw = getattr(bit_width, 'bvq_host_value', None)
        if bw is None:
            bw = int(bit_width.item())
        if bw <= 8:
            tdt, cdt = (torch.int8, nat.CODES_I8) if self.signed else (torch.uint8, nat.CODES_U8)
        else:
            tdt, cdt = torch.int32, nat.CODES_I32
        fa = self._fused_args(scale, zero_point, bit_width, x)
        if fa is None:
            with torch.no_grad():
                return self.to_int(scale, zero_point, bit_width, x).to(tdt)
        p, qmin, qmax, round_mode, clamp_ste = fa
        # the plan describes x in MEMORY order (a dense channels_last tensor is [N*H*W, C] rows): hand the
        # kernel the same element order and map the codes back to x's logical layout
        xc, back = _fused._memory_order(x.detach(), p.channels, p.nhwc)
        sc = scale.detach().reshape(-1).contiguous()
        zc = zero_point.detach().reshape(-1).contiguous()
        desc = _fused.make_desc(p, xc, sc, zc, qmin, qmax, round_mode, clamp_ste, nat.OUT_INT)
        desc.codes_dtype = cdt
        codes = nat.fakequant_fwd(desc, xc, sc, zc, want_codes=True, want_y=False)
        return codes if back is None else codes.permute(back)

    def to_qcdq_operands(self, scale: Tensor, zero_point: Tensor, bit_width: Tensor, x: Tensor) -> 'QCDQOperands':
        """The operand package of a QuantizeLinear / Clip / DequantizeLinear chain, as the reference's QCDQ export
        handlers assemble it (B/export/common/handler/qcdq.py:92-150, base.py:34-41,125-139): integer codes in the
        wire dtype (written directly by the quantizer kernel, to_int_codes), the scale flattened (0-dim if it has one
        element), the zero-point expanded like the scale and cast to the codes' dtype, and the quantization axis (the
        first dim of the scale that is not 1, None for a per-tensor scale).  Not differentiable."""
        codes = self.to_int_codes(scale, zero_point, bit_width, x)
        axis = next((i for i, s in enumerate(scale.shape) if s != 1), None)
        flat = scale.detach().flatten()
        sc = flat.view(()) if flat.numel() == 1 else flat
        zf = zero_point.detach().flatten()
        zf = zf.view(()) if zf.numel() == 1 else zf
        zf = zf.expand_as(sc)
        if not self.signed and bool((zf < 0).any()):
            raise RuntimeError("Zero points have to be positive under unsigned quantization")
        return QCDQOperands(codes, sc, zf.to(codes.dtype), axis)

    def min_int(self, bit_width):
        return min_int(self.signed, self.narrow_range, bit_width)

    def max_int(self, bit_width):
        return max_int(self.signed, self.narrow_range, bit_width)

    def forward(self, scale: Tensor, zero_point: Tensor, bit_width: Tensor, x: Tensor) -> Tensor:
        return self.bvq_forward_pre(scale, zero_point, bit_width, x, nat.PRE_NONE)

    def bvq_forward_pre(self, scale: Tensor, zero_point: Tensor, bit_width: Tensor, x: Tensor, pre_op: int
                        ) -> Tensor:
        """forward(scale, zero_point, bit_width, pre_op(x)) with the activation `pre_op`
        (include/bvq.h, bvq_pre_op) folded into the quantizer kernel when that kernel applies"""
        fa = self._fused_args(scale, zero_point, bit_width, x)
        if fa is not None and not bit_width.requires_grad:
            p, qmin, qmax, round_mode, clamp_ste = fa
            y = _fused.FakeQuantFn.apply(x, scale, zero_point, p, qmin, qmax, round_mode, clamp_ste,
                                         nat.OUT_DEQUANT, pre_op)
            x_act = x  # only handed back while quantization is delayed; see below
            if pre_op != nat.PRE_NONE and not isinstance(self.delay_wrapper.delay_impl, _NoDelay):
                x_act = torch.relu(x)
        else:
            bounds = self._fused_bounds_args(scale, zero_point, bit_width, x)
            if bounds is not None:
                # a bit width that is a tensor in the autograd graph (learned): the same kernels with the integer
                # range read from device memory, the range's own gradient returned next to dx and dscale
                p, round_mode, clamp_ste = bounds
                y = _fused.FakeQuantBoundsFn.apply(x, scale, zero_point, self.min_int(bit_width),
                                                   self.max_int(bit_width), p, round_mode, clamp_ste, pre_op)
                x_act = x
                if pre_op != nat.PRE_NONE and not isinstance(self.delay_wrapper.delay_impl, _NoDelay):
                    x_act = torch.relu(x)
                return self.delay_wrapper(x_act, y)
            x_act = torch.relu(x) if pre_op == nat.PRE_RELU else x
            y_int = self.to_int(scale, zero_point, bit_width, x_act)
            y = y_int - zero_point
            y = y * scale
        return self.delay_wrapper(x_act, y)

    def _fused_bounds_args(self, scale: Tensor, zero_point: Tensor, bit_width: Tensor, x: Tensor):
        """(plan, round_mode, clamp_ste) if the fused kernels can take the integer range from device tensors"""
        if not config.FUSED_PATHS or not bit_width.is_cuda or bit_width.numel() != 1 or zero_point.requires_grad:
            return None
        round_mode = getattr(self.float_to_int_impl, 'bvq_round_mode', None)
        clamp_ste = getattr(self.tensor_clamp_impl, 'bvq_clamp_ste', None)
        if round_mode is None or clamp_ste is None:
            return None
        p = _fused.plan(x, scale, zero_point)
        if p is None or p.zp_pc:
            return None
        return p, round_mode, clamp_ste


class DecoupledIntQuant(torch.nn.Module):
    """Integer quantizer whose rounding uses (pre_scale, pre_zero_point) and whose de-quantization uses
    (scale, zero_point) (drop-in for B/core/quant/int_base.py:100-182).

    Examples (B/core/quant/int_base.py:118-126):
        >>> q = DecoupledIntQuant(narrow_range=True, signed=True)
        >>> q(torch.tensor(0.02), torch.tensor(0.), torch.tensor(0.01), torch.tensor(0.), torch.tensor(4.), inp)
        tensor([ 0.0200, -0.0300,  0.0700, -0.0700])
    """

    def __init__(self, narrow_range: bool, signed: bool, float_to_int_impl: Module = RoundSte(),
                 tensor_clamp_impl: Module = TensorClamp(), quant_delay_steps: int = 0):
        super().__init__()
        self.float_to_int_impl = float_to_int_impl
        self.tensor_clamp_impl = tensor_clamp_impl
        self.signed = signed
        self.narrow_range = narrow_range
        self.delay_wrapper = DelayWrapper(quant_delay_steps)

    def to_int(self, pre_scale: Tensor, pre_zero_point: Tensor, bit_width: Tensor, x: Tensor) -> Tensor:
        y = x / pre_scale
        y = y + pre_zero_point
        min_int_val = self.min_int(bit_width)
        max_int_val = self.max_int(bit_width)
        y = self.float_to_int_impl(y)
        y = self.tensor_clamp_impl(y, min_val=min_int_val, max_val=max_int_val)
        return y

    def min_int(self, bit_width):
        return min_int(self.signed, self.narrow_range, bit_width)

    def max_int(self, bit_width):
        return max_int(self.signed, self.narrow_range, bit_width)

    def _fused_forward(self, pre_scale, pre_zero_point, scale, zero_point, bit_width, x):
        """the whole chain as one kernel (and its autograd as one more), or None if not covered"""
        round_mode = getattr(self.float_to_int_impl, 'bvq_round_mode', None)
        clamp_ste = getattr(self.tensor_clamp_impl, 'bvq_clamp_ste', None)
        bw = getattr(bit_width, 'bvq_host_value', None)
        if round_mode is None or clamp_ste is None or bw is None or bit_width.requires_grad:
            return None
        if not _fused.scalar_zero_point_ok(zero_point, pre_zero_point, x=x):
            return None
        p = _fused.variant_plan(x, scale, pre_scale)
        if p is None:
            return None
        ct = torch.result_type(x, pre_scale)
        if not (ct == x.dtype or ct == torch.float32) or zero_point.dtype != pre_zero_point.dtype:
            return None
        qmin, qmax = int_range_host(self.signed, self.narrow_range, bw)
        return _fused.VariantFn.apply(x, scale, pre_scale, zero_point, pre_zero_point, p,
                                      dict(kind=nat.VAR_DECOUPLED, ct=ct, round_mode=round_mode, clamp_ste=clamp_ste,
                                           qmin=qmin, qmax=qmax))

    def forward(self, pre_scale: Tensor, pre_zero_point: Tensor, scale: Tensor, zero_point: Tensor, bit_width: Tensor,
                x: Tensor) -> Tensor:
        y = self._fused_forward(pre_scale, pre_zero_point, scale, zero_point, bit_width, x)
        if y is not None:
            return self.delay_wrapper(x, y)
        y_int = self.to_int(pre_scale, pre_zero_point, bit_width, x)
        y = y_int - zero_point
        y = y * scale
        return self.delay_wrapper(x, y)
