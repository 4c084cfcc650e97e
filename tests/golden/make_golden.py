"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE on CPU.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Runs only in the build container: it imports brevitas.core.* / brevitas.function.* from
/root/reference/src (read-only).  `brevitas.inject` needs the third-party `dependencies` package,
which is not installed; the hot path does not use it, so a bare namespace stub stands in for that
parent package -- no reference file is modified or copied.  The named quantizers
(Int8WeightPerChannelFloat, ...) cannot be imported for the same reason; their resolved module
graphs are assembled by hand exactly as SURVEY 8a lists them.

What is committed are the resulting .npz files (inputs + expected outputs, bf16/f16 stored as
uint16 bit patterns) and this script.  All inputs come from torch.manual_seed(123456), the seed of
the reference's own tests (tests/conftest.py:7).
"""
import json
import os
import sys
import types

REF = '/root/reference/src'
HERE = os.path.dirname(os.path.abspath(__file__))

stub = types.ModuleType('brevitas.inject')
stub.__path__ = [os.path.join(REF, 'brevitas', 'inject')]
sys.modules['brevitas.inject'] = stub
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import brevitas  # noqa: E402
from brevitas.core.bit_width import BitWidthConst  # noqa: E402
from brevitas.core.function_wrapper import (CeilSte, DPURoundSte, FloorSte, OverOutputChannelView,  # noqa: E402
                                            OverTensorView, RoundSte, RoundToZeroSte, TensorClamp,
                                            TensorClampSte)
from brevitas.core.quant import IntQuant, RescalingIntQuant  # noqa: E402
from brevitas.core.restrict_val import FloatRestrictValue  # noqa: E402
from brevitas.core.scaling import (ConstScaling, IntScaling, ParameterFromRuntimeStatsScaling,  # noqa: E402
                                   ParameterScaling, RuntimeStatsScaling, StatsFromParameterScaling)
from brevitas.core.stats import AbsMax, AbsMinMax  # noqa: E402
from brevitas.core.zero_point import ZeroZeroPoint  # noqa: E402
from brevitas.function import ops as ref_ops  # noqa: E402
from brevitas.function import ops_ste as ref_ste  # noqa: E402

assert not brevitas.NATIVE_STE_BACKEND_LOADED

DT = {'f32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16}
ROUND_IMPL = {'round': RoundSte, 'floor': FloorSte, 'ceil': CeilSte, 'rtz': RoundToZeroSte,
              'dpu': DPURoundSte}


def enc(t):
    """torch tensor -> numpy (bf16/f16 as uint16 bit patterns, everything else as is)"""
    t = t.detach().contiguous()
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16).numpy().view(np.uint16).copy()
    return t.numpy().copy()


def dtname(t):
    return {torch.float32: 'f32', torch.bfloat16: 'bf16', torch.float16: 'f16'}[t.dtype]


class Store:
    def __init__(self, name):
        self.name = name
        self.arrays = {}
        self.meta = []

    def case(self, meta, **tensors):
        idx = len(self.meta)
        m = dict(meta)
        m['dtypes'] = {}
        for k, v in tensors.items():
            if v is None:
                continue
            self.arrays['c%d_%s' % (idx, k)] = enc(v)
            m['dtypes'][k] = dtname(v) if v.dtype.is_floating_point else str(v.dtype)
        self.meta.append(m)

    def save(self):
        path = os.path.join(HERE, self.name + '.npz')
        np.savez_compressed(path, __meta__=np.frombuffer(json.dumps(self.meta).encode(), dtype=np.uint8),
                            **self.arrays)
        print('%s: %d cases, %.1f KB' % (path, len(self.meta), os.path.getsize(path) / 1024))


def adversarial(dtype, n_rand=96):
    """ties, signed zeros, denormals, range boundaries, inf/nan, plus seeded randoms"""
    ties = torch.arange(-6, 7, dtype=torch.float32) + 0.5
    special = torch.tensor([0.0, -0.0, 1e-40, -1e-40, 1.17549435e-38, 1e-10, -1e-10, 126.5, 127.5,
                            -127.5, -128.5, 127.49, 128.0, -128.0, 254.5, 255.5, 1e6, -1e6, 3.0e38,
                            float('inf'), float('-inf'), float('nan'), 0.499999, -0.499999, 1.5, 2.5,
                            -1.5, -2.5, 0.25, -0.75])
    rand = torch.randn(n_rand) * 3
    return torch.cat([ties, special, rand]).to(dtype)


# ------------------------------------------------------------------------------------------------
# A. straight-through ops (seam 1)
# ------------------------------------------------------------------------------------------------
def gen_ste():
    st = Store('ste_ops')
    unary = ['round_ste', 'floor_ste', 'ceil_ste', 'round_to_zero_ste', 'dpu_round_ste',
             'binary_sign_ste', 'ternary_sign_ste', 'abs_binary_sign_grad']
    for dn, dtype in DT.items():
        x = adversarial(dtype)
        g = torch.randn(x.shape).to(dtype)
        for name in unary:
            xi = x.clone().requires_grad_(True)
            y = getattr(ref_ste, name)(xi)
            y.backward(g)
            st.case({'op': name, 'dtype': dn}, x=x, g=g, y=y, dx=xi.grad)
        for lo, hi in ((-1.5, 2.25), (1e-10, 6.0e4)):
            xi = x.clone().requires_grad_(True)
            y = ref_ste.scalar_clamp_ste(xi, lo, hi)
            y.backward(g)
            st.case({'op': 'scalar_clamp_ste', 'dtype': dn, 'lo': lo, 'hi': hi}, x=x, g=g, y=y, dx=xi.grad)
        for lo in (1e-10, 0.3):
            xi = x.clone().requires_grad_(True)
            y = ref_ste.scalar_clamp_min_ste(xi, lo)
            y.backward(g)
            st.case({'op': 'scalar_clamp_min_ste', 'dtype': dn, 'lo': lo}, x=x, g=g, y=y, dx=xi.grad)
        # tensor clamps: 0-dim bounds (the hot path: min_int/max_int) and same-shape bounds
        lo0, hi0 = torch.tensor(-2.0).to(dtype), torch.tensor(1.75).to(dtype)
        lof = (torch.rand(x.shape) - 1.5).to(dtype)
        hif = (torch.rand(x.shape) + 0.5).to(dtype)
        for tag, lo, hi in (('scalar', lo0, hi0), ('full', lof, hif)):
            for name in ('tensor_clamp_ste', 'tensor_clamp'):
                fn = getattr(ref_ste, name) if name.endswith('ste') else ref_ops.tensor_clamp
                xi = x.clone().requires_grad_(True)
                y = fn(xi, lo, hi)
                y.backward(g)
                st.case({'op': name, 'dtype': dn, 'bounds': tag}, x=x, g=g, lo=lo, hi=hi, y=y, dx=xi.grad)
            # in-place variant: values only (B/function/ops.py:103-111: torch.min / torch.max)
            xi = x.clone()
            y = ref_ste.tensor_clamp_ste_(xi, lo, hi)
            st.case({'op': 'tensor_clamp_ste_', 'dtype': dn, 'bounds': tag}, x=x, lo=lo, hi=hi, y=y)
    # integer range formulas (B/function/ops.py:132-191)
    for signed in (True, False):
        for narrow in (True, False):
            for bw in range(2, 9):
                b = torch.tensor(float(bw))
                st.case({'op': 'int_range', 'signed': signed, 'narrow': narrow, 'bit_width': bw},
                        max_int=ref_ops.max_int(signed, narrow, b), min_int=ref_ops.min_int(signed, narrow, b))
    st.save()


# ------------------------------------------------------------------------------------------------
# B. IntQuant forward / to_int / backward
# ------------------------------------------------------------------------------------------------
def gen_int_quant():
    st = Store('int_quant')
    shapes = {
        'tensor': ((4, 1024), None),           # config 1 smoke shape, per-tensor
        'ragged': ((3, 5, 7), None),           # 105 elements: not a multiple of any vector width
        'ch0': ((6, 4, 3, 3), 0),              # weight-like, per output channel (dim 0), inner 36
        'ch1': ((5, 6, 4, 4), 1),              # activation-like NCHW, per channel (dim 1), inner 16
        'ch1_odd': ((3, 5, 7), 1),             # inner 7: scalar path
        'chlast': ((9, 8), 1),                 # linear activation, per feature (inner 1)
    }
    cfgs = []
    # (dtype of x, dtype of scale ('same'|'f32'|'bf16'), scale layout, signed, narrow, bw, zp kind, round, clamp)
    for xd in ('f32', 'bf16', 'f16'):
        for lay in ('tensor', 'ch0', 'ch1'):
            cfgs.append((xd, 'same', lay, True, False, 8, 'zero', 'round', 'where'))
            cfgs.append((xd, 'same', lay, True, True, 8, 'zero', 'round', 'ste'))
    for xd in ('bf16', 'f16'):
        cfgs.append((xd, 'f32', 'tensor', True, False, 8, 'zero', 'round', 'where'))   # 0-dim f32 scale: stays xd
        cfgs.append((xd, 'f32', 'ch1', True, False, 8, 'zero', 'round', 'where'))      # f32 [1,C,1,1]: promotes to f32
        cfgs.append((xd, 'f32', 'ch0', True, True, 4, 'zero', 'round', 'ste'))
    for lay in ('ragged', 'ch1_odd', 'chlast'):
        for xd in ('f32', 'bf16'):
            cfgs.append((xd, 'same', lay, True, False, 8, 'zero', 'round', 'where'))
    for bw in (2, 3, 4, 5, 6, 7):
        cfgs.append(('f32', 'same', 'ch0', True, True, bw, 'zero', 'round', 'ste'))
        cfgs.append(('bf16', 'same', 'tensor', False, False, bw, 'zero', 'round', 'where'))
    cfgs.append(('f32', 'same', 'tensor', False, False, 8, 'zero', 'round', 'where'))
    cfgs.append(('f32', 'same', 'tensor', False, True, 8, 'half', 'round', 'where'))
    cfgs.append(('f32', 'same', 'ch1', False, False, 8, 'perch', 'round', 'where'))
    cfgs.append(('bf16', 'same', 'ch1', False, False, 8, 'perch', 'round', 'where'))
    cfgs.append(('bf16', 'f32', 'tensor', False, False, 8, 'half', 'round', 'where'))
    for rm in ('floor', 'ceil', 'rtz', 'dpu'):
        cfgs.append(('f32', 'same', 'tensor', True, False, 8, 'zero', rm, 'where'))
        cfgs.append(('bf16', 'same', 'ch1', True, False, 6, 'half', rm, 'ste'))

    for (xd, sd, lay, signed, narrow, bw, zpk, rm, clamp) in cfgs:
        shape, chdim = shapes[lay]
        dtype = DT[xd]
        sdtype = dtype if sd == 'same' else DT[sd]
        x = (torch.randn(shape) * 1.3)
        # sprinkle adversarial values: exact ties, zeros, boundary hits
        flat = x.view(-1)
        adv = adversarial(torch.float32, n_rand=0)
        adv = adv[torch.isfinite(adv)]
        k = min(adv.numel(), flat.numel() // 3)
        flat[torch.randperm(flat.numel())[:k]] = adv[:k] * 0.02
        x = x.to(dtype)
        if chdim is None:
            scale = torch.tensor(0.02).to(sdtype)
        else:
            sshape = [1] * len(shape)
            sshape[chdim] = shape[chdim]
            scale = (torch.rand(sshape) * 0.03 + 0.005).to(sdtype)
        qmax = float(ref_ops.max_int(signed, narrow, torch.tensor(float(bw))))
        if zpk == 'zero':
            zp = torch.tensor(0.0)
        elif zpk == 'half':
            zp = torch.tensor(float(int(qmax * 0.3)))
        else:
            zshape = [1] * len(shape)
            zshape[chdim] = shape[chdim]
            zp = torch.randint(0, int(qmax) // 2, zshape).float()
        if zpk == 'perch':
            zp = zp.to(sdtype)
        bit_width = torch.tensor(float(bw))
        iq = IntQuant(narrow_range=narrow, signed=signed, float_to_int_impl=ROUND_IMPL[rm](),
                      tensor_clamp_impl=TensorClampSte() if clamp == 'ste' else TensorClamp())
        xi = x.clone().requires_grad_(True)
        si = scale.clone().requires_grad_(True)
        zi = zp.clone().requires_grad_(True)
        y = iq(si, zi, bit_width, xi)
        g = torch.randn(y.shape).to(y.dtype)
        y.backward(g)
        with torch.no_grad():
            yint = iq.to_int(scale, zp, bit_width, x)
        st.case({'x_dtype': xd, 'layout': lay, 'shape': list(shape), 'chdim': chdim, 'signed': signed,
                 'narrow': narrow, 'bit_width': bw, 'zp_kind': zpk, 'round': rm, 'clamp': clamp,
                 'scalar_mode': 'opmath'},
                x=x, scale=scale, zp=zp, g=g, y=y, y_int=yint, dx=xi.grad, dscale=si.grad, dzp=zi.grad)
    st.save()


# ------------------------------------------------------------------------------------------------
# C. statistics
# ------------------------------------------------------------------------------------------------
def gen_stats():
    st = Store('stats')
    for dn, dtype in DT.items():
        # whole tensor
        for tag in ('rand', 'ties', 'zeros', 'nan'):
            x = torch.randn(7, 33)
            if tag == 'ties':
                x = (x * 2).round() / 2
                m = x.abs().max()
                x[2, 5], x[4, 1], x[6, 30] = m, -m, m
            if tag == 'zeros':
                x = torch.zeros(3, 4)
            if tag == 'nan':
                x[3, 3] = float('nan')
            x = x.to(dtype)
            for name, mod in (('absmax', AbsMax()), ('absminmax', AbsMinMax())):
                xi = x.clone().requires_grad_(True)
                out = mod(OverTensorView()(xi))
                gout = torch.tensor(0.7).to(out.dtype)
                out.backward(gout)
                st.case({'stat': name, 'dtype': dn, 'tag': tag, 'shape': list(x.shape), 'chdim': None},
                        x=x, out=out, gout=gout, dx=xi.grad)
        # per channel: weights (dim 0, no permute) and NCHW activations (dim 1, permute (1,0,2,3))
        for lay, shape, chdim, perm in (('ch0', (6, 4, 3, 3), 0, None), ('ch1', (5, 6, 4, 4), 1, (1, 0, 2, 3)),
                                        ('ch1_odd', (3, 5, 7), 1, (1, 0, 2))):
            for tag in ('rand', 'ties'):
                x = torch.randn(shape)
                if tag == 'ties':
                    x = (x * 2).round() / 2
                if tag == 'ties' and lay == 'ch0':
                    x[1] = 0.0
                x = x.to(dtype)
                for name, mod in (('absmax', AbsMax(1)), ('absminmax', AbsMinMax(1))):
                    xi = x.clone().requires_grad_(True)
                    out = mod(OverOutputChannelView(perm)(xi))
                    gout = torch.randn(out.shape).to(out.dtype)
                    out.backward(gout)
                    st.case({'stat': name, 'dtype': dn, 'tag': tag, 'shape': list(shape), 'chdim': chdim},
                            x=x, out=out, gout=gout, dx=xi.grad)
    st.save()


# ------------------------------------------------------------------------------------------------
# D. resolved quantizer graphs
# ------------------------------------------------------------------------------------------------
def weight_quant(weight, bit_width, narrow=True):
    """Int8WeightPerChannelFloat / its Int4 variant, resolved"""
    cout = weight.shape[0]
    scaling_shape = (cout,) + (1,) * (weight.dim() - 1)
    return RescalingIntQuant(
        IntQuant(narrow_range=narrow, signed=True, float_to_int_impl=RoundSte(),
                 tensor_clamp_impl=TensorClampSte()),
        StatsFromParameterScaling(AbsMax(1), OverOutputChannelView(None), 1, [weight],
                                  FloatRestrictValue(), scaling_shape, affine_rescaling=False,
                                  scaling_min_val=1e-10),
        IntScaling(signed=True, narrow_range=narrow), ZeroZeroPoint(), BitWidthConst(bit_width))


def act_quant_param_from_stats(collect_steps, per_channel_c=None):
    """Int8ActPerTensorFloat with scaling_stats_op=MAX, resolved"""
    if per_channel_c is None:
        view, stats, shape = OverTensorView(), AbsMax(), ()
    else:
        view, stats, shape = OverOutputChannelView((1, 0, 2, 3)), AbsMax(1), (1, per_channel_c, 1, 1)
    return RescalingIntQuant(
        IntQuant(narrow_range=False, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp()),
        ParameterFromRuntimeStatsScaling(collect_steps, stats, view, shape, FloatRestrictValue(), 0.1, 1e-10),
        IntScaling(signed=True, narrow_range=False), ZeroZeroPoint(), BitWidthConst(8))


def act_quant_runtime_stats(per_channel_c=None):
    if per_channel_c is None:
        view, stats, shape = OverTensorView(), AbsMax(), ()
    else:
        view, stats, shape = OverOutputChannelView((1, 0, 2, 3)), AbsMax(1), (1, per_channel_c, 1, 1)
    return RescalingIntQuant(
        IntQuant(narrow_range=False, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp()),
        RuntimeStatsScaling(stats, view, FloatRestrictValue(), shape, affine_rescaling=False,
                            scaling_stats_momentum=0.1, scaling_min_val=1e-10),
        IntScaling(signed=True, narrow_range=False), ZeroZeroPoint(), BitWidthConst(8))


def gen_graphs():
    st = Store('quant_graphs')
    # -- weights: config 2 (conv) and config 5 (linear, int4) at toy sizes ------------------------
    for dn in ('f32', 'bf16'):
        for tag, shape, bw in (('conv_int8', (16, 8, 3, 3), 8), ('linear_int4', (24, 40), 4),
                               ('conv_int8_ragged', (5, 3, 3, 3), 8)):
            w = torch.randn(shape) * 0.02
            if tag == 'conv_int8':
                w[3] = 0.0                      # all-zero channel exercises scaling_min_val
                w[5].view(-1)[7] = w[5].abs().max()
                w[5].view(-1)[2] = -w[5].abs().max()   # +-max tie: gradient goes to the first
            w = torch.nn.Parameter(w.to(DT[dn]))
            q = weight_quant(w, bw)
            y, scale, zp, bwt = q(w)
            g = torch.randn(y.shape).to(y.dtype)
            y.backward(g)
            st.case({'graph': 'weight_per_channel', 'tag': tag, 'dtype': dn, 'bit_width': bw, 'shape': list(shape)},
                    x=w.data, g=g, y=y, scale=scale, zp=zp, bit_width=bwt, dx=w.grad)
    # -- activations, runtime stats (training EMA then eval) --------------------------------------
    for dn in ('f32', 'bf16'):
        for tag, pc in (('per_tensor', None), ('per_channel', 6)):
            q = act_quant_runtime_stats(pc)
            q.train()
            for step in range(3):
                x = (torch.randn(4, 6, 5, 5) * (1.0 + step)).to(DT[dn])
                xi = x.clone().requires_grad_(True)
                y, scale, zp, bwt = q(xi)
                g = torch.randn(y.shape).to(y.dtype)
                y.backward(g)
                st.case({'graph': 'act_runtime_stats', 'tag': tag, 'dtype': dn, 'step': step, 'training': True,
                         'channels': pc},
                        x=x, g=g, y=y, scale=scale, dx=xi.grad,
                        running_stats=q.scaling_impl.runtime_stats.running_stats.clone())
            q.eval()
            xi = x.clone().requires_grad_(True)
            y, scale, zp, bwt = q(xi)
            g = torch.randn(y.shape).to(y.dtype)
            y.backward(g)
            st.case({'graph': 'act_runtime_stats', 'tag': tag, 'dtype': dn, 'step': 3, 'training': False,
                     'channels': pc},
                    x=x, g=g, y=y, scale=scale, dx=xi.grad,
                    running_stats=q.scaling_impl.runtime_stats.running_stats.clone())
    # -- activations, stats collection then learned parameter (Int8ActPerTensorFloat, MAX stats) ---
    for dn in ('f32', 'bf16'):
        for tag, pc in (('per_tensor', None), ('per_channel', 6)):
            q = act_quant_param_from_stats(2, pc)
            q.train()
            for step in range(4):
                x = (torch.randn(4, 6, 5, 5) * (1.0 + 0.5 * step)).to(DT[dn])
                xi = x.clone().requires_grad_(True)
                q.zero_grad()
                y, scale, zp, bwt = q(xi)
                g = torch.randn(y.shape).to(y.dtype)
                y.backward(g)
                si = q.scaling_impl
                st.case({'graph': 'act_param_from_stats', 'tag': tag, 'dtype': dn, 'step': step, 'channels': pc,
                         'counter': int(si.counter)},
                        x=x, g=g, y=y, scale=scale, dx=xi.grad, buffer=si.buffer.clone(),
                        value=si.value.detach().clone(),
                        dvalue=si.value.grad.clone() if si.value.grad is not None else None)
            sd = {k: v for k, v in q.state_dict().items()}
            st.case({'graph': 'act_param_from_stats_state_dict', 'tag': tag, 'dtype': dn, 'keys': sorted(sd.keys())},
                    **{k.replace('.', '__'): v for k, v in sd.items()})
    # -- learned per-tensor scale, steady state (ParameterScaling) ---------------------------------
    for dn in ('f32', 'bf16'):
        for cast_module in (False, True):
            q = RescalingIntQuant(
                IntQuant(narrow_range=False, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp()),
                ParameterScaling(3.0, scaling_shape=None, restrict_scaling_impl=FloatRestrictValue(),
                                 scaling_min_val=1e-10),
                IntScaling(signed=True, narrow_range=False), ZeroZeroPoint(), BitWidthConst(8))
            if cast_module:
                if dn == 'f32':
                    continue
                q = q.to(DT[dn])
            x = (torch.randn(4, 6, 5, 5) * 1.2).to(DT[dn])
            xi = x.clone().requires_grad_(True)
            y, scale, zp, bwt = q(xi)
            g = torch.randn(y.shape).to(y.dtype)
            y.backward(g)
            st.case({'graph': 'act_parameter_scale', 'dtype': dn, 'module_cast': cast_module},
                    x=x, g=g, y=y, scale=scale, dx=xi.grad, dvalue=q.scaling_impl.value.grad,
                    value=q.scaling_impl.value.detach())
    # -- const scale (reference doctest graph, B/core/quant/int.py:113-134) -------------------------
    q = RescalingIntQuant(IntQuant(narrow_range=True, signed=True), ConstScaling(0.1),
                          IntScaling(signed=True, narrow_range=True), ZeroZeroPoint(), BitWidthConst(4))
    x = torch.Tensor([0.042, -0.053, 0.31, -0.44])
    y, scale, zp, bwt = q(x)
    st.case({'graph': 'const_scale_doctest'}, x=x, y=y, scale=scale, zp=zp, bit_width=bwt)
    st.save()


# ------------------------------------------------------------------------------------------------
# E. activation function + quantizer (FusedActivationQuantProxy, B/proxy/runtime_quant.py:73-84) and
#    externally scaled quantizers (bias quantization, B/core/quant/int.py:17-91)
# ------------------------------------------------------------------------------------------------
def gen_act_fused():
    from brevitas.core.function_wrapper import Identity
    from brevitas.core.quant import PrescaledRestrictIntQuant, PrescaledRestrictIntQuantWithInputBitWidth
    st = Store('act_fused')

    def uint_quant(scaling, signed):
        return RescalingIntQuant(
            IntQuant(narrow_range=False, signed=signed, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp()),
            scaling, IntScaling(signed=signed, narrow_range=False), ZeroZeroPoint(), BitWidthConst(8))

    for dn in ('f32', 'bf16'):
        for signed in (False, True):
            for tag, pc in (('per_tensor', None), ('per_channel', 6)):
                if pc is None:
                    view, stats, shape = OverTensorView(), AbsMax(), ()
                else:
                    view, stats, shape = OverOutputChannelView((1, 0, 2, 3)), AbsMax(1), (1, pc, 1, 1)
                q = uint_quant(RuntimeStatsScaling(stats, view, FloatRestrictValue(), shape, False, 0.1, 1e-10), signed)
                act = torch.nn.ReLU()
                q.train()
                for step in range(2):
                    x = (torch.randn(4, 6, 5, 5) * (1.0 + step)).to(DT[dn])
                    if step == 1 and pc is not None:
                        x[:, 2] = -x[:, 2].abs()  # a channel that is entirely negative: statistic 0 -> min_val
                    x.view(-1)[::17] = 0.0
                    xi = x.clone().requires_grad_(True)
                    y, scale, zp, bwt = q(act(xi))
                    g = torch.randn(y.shape).to(y.dtype)
                    y.backward(g)
                    st.case({'graph': 'relu_runtime_stats', 'tag': tag, 'dtype': dn, 'signed': signed, 'step': step,
                             'channels': pc},
                            x=x, g=g, y=y, scale=scale, dx=xi.grad,
                            running_stats=q.scaling_impl.runtime_stats.running_stats.clone())
            # learned scale (steady state of Uint8ActPerTensorFloat after a ReLU)
            q = uint_quant(ParameterScaling(2.5, None, FloatRestrictValue(), 1e-10), signed)
            x = (torch.randn(4, 6, 5, 5) * 1.5).to(DT[dn])
            xi = x.clone().requires_grad_(True)
            y, scale, zp, bwt = q(torch.relu(xi))
            g = torch.randn(y.shape).to(y.dtype)
            y.backward(g)
            st.case({'graph': 'relu_parameter_scale', 'dtype': dn, 'signed': signed},
                    x=x, g=g, y=y, scale=scale, dx=xi.grad, dvalue=q.scaling_impl.value.grad)
    # externally scaled (bias) quantizers: doctest B/core/quant/int.py:32-48 and a per-channel case
    q = PrescaledRestrictIntQuantWithInputBitWidth(IntQuant(narrow_range=True, signed=True), Identity())
    x = torch.Tensor([0.042, -0.053, 0.31, -0.44])
    y, scale, zp, bwt = q(x, torch.tensor(0.01), torch.tensor(4.))
    st.case({'graph': 'prescaled_input_bit_width_doctest'}, x=x, y=y, scale=scale, zp=zp, bit_width=bwt)
    for dn in ('f32', 'bf16'):
        q = PrescaledRestrictIntQuant(IntQuant(narrow_range=False, signed=True, float_to_int_impl=RoundSte(),
                                               tensor_clamp_impl=TensorClamp()), BitWidthConst(8))
        b = (torch.randn(16) * 0.5).to(DT[dn])
        scale = (torch.rand(16) * 0.01 + 0.001).to(DT[dn])
        bi = b.clone().requires_grad_(True)
        si = scale.clone().requires_grad_(True)
        y, so, zp, bwt = q(bi, si)
        g = torch.randn(16).to(y.dtype)
        y.backward(g)
        st.case({'graph': 'prescaled_bias', 'dtype': dn}, x=b, scale=scale, g=g, y=y, dx=bi.grad, dscale=si.grad)
    st.save()


# ------------------------------------------------------------------------------------------------
# F. percentile statistics (B/core/stats/stats_op.py:41-126) and the default Int8ActPerTensorFloat graph
# ------------------------------------------------------------------------------------------------
def gen_percentile():
    from brevitas.core.stats import AbsPercentile, NegativePercentileOrZero, PercentileInterval
    st = Store('percentile')
    ten = torch.Tensor([1, 2, 3, 4, 5, 6, 7, 8, 9, 10])  # tests/brevitas/core/test_stats.py:12-30
    for v in range(1, 11):
        st.case({'stat': 'abs_percentile', 'q': v * 10, 'dim': None, 'tag': 'one_to_ten'}, x=ten,
                out=AbsPercentile(v * 10, None)(ten))
    st.case({'stat': 'abs_percentile', 'q': 90, 'dim': 1, 'tag': 'one_to_ten_x2'}, x=ten.repeat(2, 1),
            out=AbsPercentile(90, stats_reduce_dim=1)(ten.repeat(2, 1)))
    vals = torch.tensor([-1., -2., 5])  # test_stats.py:47-67
    st.case({'stat': 'neg_percentile', 'q': 0.01, 'dim': None, 'tag': 'neg'}, x=vals,
            out=NegativePercentileOrZero(0.01)(vals))
    st.case({'stat': 'neg_percentile', 'q': 0.01, 'dim': None, 'tag': 'pos'}, x=torch.tensor([1., 2., 5]),
            out=NegativePercentileOrZero(0.01)(torch.tensor([1., 2., 5])))
    st.case({'stat': 'interval', 'low_q': 0.01, 'high_q': 99.9, 'dim': None, 'tag': 'one_to_ten'}, x=ten,
            out=PercentileInterval(low_percentile_q=0.01, high_percentile_q=99.9)(ten))
    for dn, dtype in DT.items():
        for tag, shape in (('flat', (37, 91)), ('rows', (6, 500))):
            x = torch.randn(shape) * 2
            x.view(-1)[::13] = 0.0
            if tag == 'flat':
                x[3, 3] = float('inf')
            x = x.to(dtype)
            dims = (None,) if tag == 'flat' else (None, 1)
            for dim in dims:
                for q in (99.999, 99.0, 50.0, 3.0, 100.0):
                    xi = x.clone().requires_grad_(True)
                    out = AbsPercentile(q, dim)(xi)
                    gout = torch.randn(out.shape).to(out.dtype)
                    out.backward(gout)
                    st.case({'stat': 'abs_percentile', 'q': q, 'dim': dim, 'tag': tag, 'dtype': dn},
                            x=x, out=out, gout=gout, dx=xi.grad)
                for q in (0.01, 10.0, 60.0):
                    st.case({'stat': 'neg_percentile', 'q': q, 'dim': dim, 'tag': tag, 'dtype': dn}, x=x,
                            out=NegativePercentileOrZero(q, dim)(x))
                st.case({'stat': 'interval', 'low_q': 0.01, 'high_q': 99.9, 'dim': dim, 'tag': tag, 'dtype': dn},
                        x=x, out=PercentileInterval(0.01, 99.9, dim)(x))
    # the default activation quantizer: Int8ActPerTensorFloat = ParamFromRuntimePercentileScaling
    # (AbsPercentile(99.999, None), collect 300 steps; here 2) -- B/quant/base.py:68-75, scaled_int.py:170-180
    for dn in ('f32', 'bf16'):
        q = RescalingIntQuant(
            IntQuant(narrow_range=False, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp()),
            ParameterFromRuntimeStatsScaling(2, AbsPercentile(99.999, None), OverTensorView(), (),
                                             FloatRestrictValue(), 0.1, 1e-10),
            IntScaling(signed=True, narrow_range=False), ZeroZeroPoint(), BitWidthConst(8))
        q.train()
        for step in range(3):
            x = (torch.randn(8, 6, 9, 9) * (1.0 + 0.5 * step)).to(DT[dn])
            y, scale, zp, bwt = q(x)
            st.case({'stat': 'int8_act_per_tensor_float', 'dtype': dn, 'step': step}, x=x, y=y, scale=scale,
                    buffer=q.scaling_impl.buffer.clone(), value=q.scaling_impl.value.detach().clone())
    st.save()


# ------------------------------------------------------------------------------------------------
# G. asymmetric ("shifted") quantizers: integer zero-point from statistics (B/quant/shifted_scaled_int.py)
# ------------------------------------------------------------------------------------------------
def gen_shifted():
    from brevitas.core.stats import NegativeMinOrZero, NegativePercentileOrZero, PercentileInterval
    from brevitas.core.zero_point import ParameterFromRuntimeZeroPoint, StatsFromParameterZeroPoint
    st = Store('shifted')
    for dn in ('f32', 'bf16'):
        for tag, per_channel in (('per_tensor', False), ('per_channel', True)):
            w = torch.randn(10, 6, 3, 3) * 0.1 + 0.03
            w[4] = w[4].abs()  # a channel without negative values: zero-point offset 0
            w = torch.nn.Parameter(w.to(DT[dn]))
            if per_channel:
                shape, mk, s_stat, z_stat, cat = (10, 1, 1, 1), (lambda: OverOutputChannelView(None)), AbsMinMax(1), \
                    NegativeMinOrZero(1), 1
            else:
                shape, mk, s_stat, z_stat, cat = (), (lambda: OverTensorView()), AbsMinMax(), NegativeMinOrZero(), 0
            iq = IntQuant(narrow_range=False, signed=False, float_to_int_impl=RoundSte(),
                          tensor_clamp_impl=TensorClampSte())
            q = RescalingIntQuant(
                iq, StatsFromParameterScaling(s_stat, mk(), cat, [w], FloatRestrictValue(), shape, False, 1e-10),
                IntScaling(signed=False, narrow_range=False),
                StatsFromParameterZeroPoint(iq, True, mk(), cat, z_stat, shape, [w]), BitWidthConst(8))
            y, scale, zp, bwt = q(w)
            g = torch.randn(y.shape).to(y.dtype)
            y.backward(g)
            st.case({'graph': 'shifted_weight', 'tag': tag, 'dtype': dn}, x=w.data, g=g, y=y, scale=scale, zp=zp,
                    dx=w.grad)
        iq = IntQuant(narrow_range=False, signed=False, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp())
        q = RescalingIntQuant(
            iq, ParameterFromRuntimeStatsScaling(2, PercentileInterval(0.001, 99.999, None), OverTensorView(), (),
                                                 FloatRestrictValue(), 0.1, 1e-10),
            IntScaling(signed=False, narrow_range=False),
            ParameterFromRuntimeZeroPoint(2, iq, True, NegativePercentileOrZero(0.001, None), (), OverTensorView(), 0.1),
            BitWidthConst(8))
        q.train()
        for step in range(4):
            x = (torch.randn(4, 6, 9, 9) * (1.0 + 0.3 * step) + 0.5).to(DT[dn])
            xi = x.clone().requires_grad_(True)
            q.zero_grad()
            y, scale, zp, bwt = q(xi)
            g = torch.randn(y.shape).to(y.dtype)
            y.backward(g)
            st.case({'graph': 'shifted_act', 'dtype': dn, 'step': step}, x=x, g=g, y=y, scale=scale, zp=zp, dx=xi.grad,
                    zp_buffer=q.zero_point_impl.buffer.clone(), zp_value=q.zero_point_impl.value.detach().clone(),
                    scale_value=q.scaling_impl.value.detach().clone())
        sd = q.state_dict()
        st.case({'graph': 'shifted_act_state_dict', 'dtype': dn, 'keys': sorted(sd.keys())})
    st.save()


# ------------------------------------------------------------------------------------------------
# H. remaining quantizer variants of the same elementwise family (B/core/quant/{binary,ternary,int_base,int}.py)
# ------------------------------------------------------------------------------------------------
def gen_variants():
    from brevitas.core.quant import (BinaryQuant, ClampedBinaryQuant, DecoupledIntQuant, TernaryQuant,
                                     TruncIntQuant)
    st = Store('variants')
    for dn in ('f32', 'bf16', 'f16'):
        x = (torch.randn(5, 7, 3) * 0.8).to(DT[dn])
        x.view(-1)[::11] = 0.0
        for name, mk in (('binary', lambda: BinaryQuant(ParameterScaling(0.7))),
                         ('clamped_binary', lambda: ClampedBinaryQuant(ParameterScaling(0.7))),
                         ('ternary', lambda: TernaryQuant(ParameterScaling(0.9), 0.5))):
            q = mk()
            xi = x.clone().requires_grad_(True)
            y, scale, zp, bwt = q(xi)
            g = torch.randn(y.shape).to(y.dtype)
            y.backward(g)
            st.case({'quant': name, 'dtype': dn}, x=x, g=g, y=y, scale=scale, zp=zp, bit_width=bwt, dx=xi.grad,
                    dvalue=q.scaling_impl.value.grad)
        # decoupled: rounding grid from pre_scale, de-quantization with scale
        dq = DecoupledIntQuant(narrow_range=True, signed=True)
        xi = x.clone().requires_grad_(True)
        pre_scale, scale = torch.tensor(0.02), torch.tensor(0.013)
        y = dq(pre_scale, torch.tensor(0.), scale, torch.tensor(0.), torch.tensor(4.), xi)
        g = torch.randn(y.shape).to(y.dtype)
        y.backward(g)
        st.case({'quant': 'decoupled', 'dtype': dn}, x=x, g=g, y=y, dx=xi.grad, pre_scale=pre_scale, scale=scale)
        # truncation of an 8-bit quantized value to 5 bits
        for rm, impl in (('floor', FloorSte), ('round', RoundSte)):
            tq = TruncIntQuant(impl(), BitWidthConst(5))
            s8 = torch.tensor(0.05)
            xq = (torch.randint(-128, 128, (6, 9)).float() * s8).to(DT[dn])
            xi = xq.clone().requires_grad_(True)
            y, so, zo, bo = tq(xi, s8, torch.tensor(0.), torch.tensor(8.))
            g = torch.randn(y.shape).to(y.dtype)
            y.backward(g)
            st.case({'quant': 'trunc', 'round': rm, 'dtype': dn}, x=xq, g=g, y=y, dx=xi.grad, scale=so, bit_width=bo)
        # per-channel scales (dimensioned, in the tensor's dtype) and the straight-through clamp
        xw = (torch.randn(6, 40) * 0.8).to(DT[dn])
        for name, mk in (('binary', lambda: BinaryQuant(ParameterScaling(torch.rand(6, 1) + 0.3, (6, 1)))),
                         ('clamped_binary', lambda: ClampedBinaryQuant(ParameterScaling(torch.rand(6, 1) + 0.3, (6, 1)),
                                                                      tensor_clamp_impl=TensorClampSte())),
                         ('ternary', lambda: TernaryQuant(ParameterScaling(torch.rand(6, 1) + 0.3, (6, 1)), 0.6))):
            q = mk().to(DT[dn])
            xi = xw.clone().requires_grad_(True)
            y, scale, zp, bwt = q(xi)
            g = torch.randn(y.shape).to(y.dtype)
            y.backward(g)
            st.case({'quant': name, 'dtype': dn, 'tag': 'per_channel'}, x=xw, g=g, y=y, scale=scale, zp=zp,
                    bit_width=bwt, dx=xi.grad, dvalue=q.scaling_impl.value.grad,
                    value=q.scaling_impl.value.detach().clone())
        # decoupled with learnable scales: gradients of pre_scale and scale, per channel, non-zero zero-points
        dq = DecoupledIntQuant(narrow_range=False, signed=True, tensor_clamp_impl=TensorClamp())
        xi = xw.clone().requires_grad_(True)
        pre_scale = (torch.rand(6, 1) * 0.05 + 0.02).to(DT[dn]).requires_grad_(True)
        scale = (torch.rand(6, 1) * 0.05 + 0.02).to(DT[dn]).requires_grad_(True)
        y = dq(pre_scale, torch.tensor(1.), scale, torch.tensor(2.), torch.tensor(4.), xi)
        g = torch.randn(y.shape).to(y.dtype)
        y.backward(g)
        st.case({'quant': 'decoupled', 'dtype': dn, 'tag': 'per_channel'}, x=xw, g=g, y=y, dx=xi.grad,
                pre_scale=pre_scale.detach(), scale=scale.detach(), dpre_scale=pre_scale.grad, dscale=scale.grad)
    # doctests: B/core/quant/int_base.py:118-126, ternary.py:32-44
    inp = torch.Tensor([0.042, -0.053, 0.31, -0.44])
    y = DecoupledIntQuant(narrow_range=True, signed=True)(torch.tensor(0.02), torch.tensor(0.), torch.tensor(0.01),
                                                          torch.tensor(0.), torch.tensor(4.), inp)
    st.case({'quant': 'decoupled_doctest'}, x=inp, y=y)
    y, scale, zp, bwt = TernaryQuant(ConstScaling(1.0), 0.5)(torch.Tensor([0.04, -0.6, 3.3]))
    st.case({'quant': 'ternary_doctest'}, x=torch.Tensor([0.04, -0.6, 3.3]), y=y, scale=scale, zp=zp, bit_width=bwt)
    st.save()


# ------------------------------------------------------------------------------------------------
# I. restricted scales: power-of-two (fixed point, B/quant/fixed_point.py:23-73) and log-domain learned scales
# ------------------------------------------------------------------------------------------------
def gen_fixed_point():
    from brevitas.core.restrict_val import LogFloatRestrictValue, PowerOfTwoRestrictValue
    from brevitas.core.scaling import PowerOfTwoIntScaling
    from brevitas.core.stats import AbsPercentile
    st = Store('fixed_point')
    for dn in ('f32', 'bf16'):
        # Int8WeightPerTensorFixedPoint, and the same with one radix point per output channel.  The injectors
        # inherit MaxStatsScaling.scaling_min_val = 1e-10 (B/quant/base.py:52-57; PerTensorPoTScaling8bit,
        # :185-191, does not override it): an all-zero channel gets the scale 1e-10 / 128 and exact zeros
        for tag, per_channel in (('per_tensor', False), ('per_channel', True), ('per_channel_zero_row', True),
                                 ('per_tensor_all_zero', False)):
            w = torch.nn.Parameter((torch.randn(12, 5, 3, 3) * 0.2).to(DT[dn]))
            if tag == 'per_channel_zero_row':
                w.data[3] = 0.0
                w.data[7] = 0.0
            if tag == 'per_tensor_all_zero':
                w.data.zero_()
            if per_channel:
                scaling = StatsFromParameterScaling(AbsMax(1), OverOutputChannelView(None), 1, [w],
                                                    PowerOfTwoRestrictValue(CeilSte()), (12, 1, 1, 1), False, 1e-10)
            else:
                scaling = StatsFromParameterScaling(AbsMax(), OverTensorView(), 0, [w],
                                                    PowerOfTwoRestrictValue(CeilSte()), (), False, 1e-10)
            q = RescalingIntQuant(
                IntQuant(narrow_range=True, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClampSte()),
                scaling, PowerOfTwoIntScaling(signed=True), ZeroZeroPoint(), BitWidthConst(8))
            y, scale, zp, bwt = q(w)
            g = torch.randn(y.shape).to(y.dtype)
            y.backward(g)
            st.case({'graph': 'pot_weight', 'tag': tag, 'dtype': dn}, x=w.data, g=g, y=y, scale=scale, zp=zp, dx=w.grad)
        # Int8BiasPerTensorFixedPointInternalScaling (B/quant/fixed_point.py:78-90): a bias with a power-of-two
        # scale of its own; a zero-initialised bias must come back as exact zeros
        for tag in ('random', 'zero'):
            b = torch.nn.Parameter((torch.randn(24) * 0.3).to(DT[dn]))
            if tag == 'zero':
                b.data.zero_()
            q = RescalingIntQuant(
                IntQuant(narrow_range=False, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp()),
                StatsFromParameterScaling(AbsMax(), OverTensorView(), 0, [b], PowerOfTwoRestrictValue(CeilSte()), (),
                                          False, 1e-10),
                PowerOfTwoIntScaling(signed=True), ZeroZeroPoint(), BitWidthConst(8))
            y, scale, zp, bwt = q(b)
            g = torch.randn(y.shape).to(y.dtype)
            y.backward(g)
            st.case({'graph': 'pot_bias', 'tag': tag, 'dtype': dn}, x=b.data, g=g, y=y, scale=scale, zp=zp, dx=b.grad)
        # Int8ActPerTensorFixedPoint / Uint8ActPerTensorFixedPoint: percentile collected for 2 steps, then learned
        for signed in (True, False):
            q = RescalingIntQuant(
                IntQuant(narrow_range=False, signed=signed, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp()),
                ParameterFromRuntimeStatsScaling(2, AbsPercentile(99.999, None), OverTensorView(), (),
                                                 PowerOfTwoRestrictValue(CeilSte()), 0.1, 1e-10),
                PowerOfTwoIntScaling(signed=signed), ZeroZeroPoint(), BitWidthConst(8))
            q.train()
            for step in range(4):
                x = (torch.randn(4, 6, 9, 9) * (1.0 + 0.4 * step)).to(DT[dn])
                if not signed:
                    x = torch.relu(x)
                xi = x.clone().requires_grad_(True)
                q.zero_grad()
                y, scale, zp, bwt = q(xi)
                g = torch.randn(y.shape).to(y.dtype)
                y.backward(g)
                vg = q.scaling_impl.value.grad
                st.case({'graph': 'pot_act', 'signed': signed, 'dtype': dn, 'step': step}, x=x, g=g, y=y, scale=scale,
                        dx=xi.grad, value=q.scaling_impl.value.detach().clone(),
                        dvalue=None if vg is None else vg.clone())
            q.eval()
            x = (torch.randn(3, 6, 5, 5) * 2).to(DT[dn])
            y, scale, zp, bwt = q(x)
            st.case({'graph': 'pot_act_eval', 'signed': signed, 'dtype': dn}, x=x, y=y, scale=scale)
        # Uint8ActPerTensorFixedPointMaxInit: learned log2-domain value initialised from max_val
        for name, restrict, int_scaling in (
                ('pot_param', lambda: PowerOfTwoRestrictValue(CeilSte()), lambda: PowerOfTwoIntScaling(signed=False)),
                ('log_param', lambda: LogFloatRestrictValue(), lambda: IntScaling(signed=False, narrow_range=False))):
            q = RescalingIntQuant(
                IntQuant(narrow_range=False, signed=False, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp()),
                ParameterScaling(0.75, None, restrict(), None), int_scaling(), ZeroZeroPoint(), BitWidthConst(8))
            x = torch.relu(torch.randn(4, 5, 6) * 0.4).to(DT[dn])
            xi = x.clone().requires_grad_(True)
            y, scale, zp, bwt = q(xi)
            g = torch.randn(y.shape).to(y.dtype)
            y.backward(g)
            st.case({'graph': name, 'dtype': dn}, x=x, g=g, y=y, scale=scale, dx=xi.grad,
                    value=q.scaling_impl.value.detach().clone(), dvalue=q.scaling_impl.value.grad)
    st.save()


# ------------------------------------------------------------------------------------------------
# J. learned bit width (B/core/bit_width/parameter.py): integer bounds and integer threshold become tensors
# ------------------------------------------------------------------------------------------------
def gen_learned_bw():
    from brevitas.core.bit_width import BitWidthParameter, MsbClampBitWidth, RemoveBitwidthParameter
    st = Store('learned_bw')
    for dn in ('f32', 'bf16', 'f16'):
        # weight: per-channel abs-max scale, straight-through clamp -> d(bit width) through the scale only
        w = torch.nn.Parameter((torch.randn(6, 4, 3, 3) * 0.3).to(DT[dn]))
        q = RescalingIntQuant(
            IntQuant(narrow_range=True, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClampSte()),
            StatsFromParameterScaling(AbsMax(1), OverOutputChannelView(None), 1, [w], FloatRestrictValue(),
                                      (6, 1, 1, 1), False, 1e-10),
            IntScaling(signed=True, narrow_range=True), ZeroZeroPoint(), BitWidthParameter(4))
        y, scale, zp, bwt = q(w)
        g = torch.randn(y.shape).to(y.dtype)
        y.backward(g)
        st.case({'graph': 'weight', 'dtype': dn}, x=w.data, g=g, y=y, scale=scale, bit_width=bwt, dx=w.grad,
                doffset=q.msb_clamp_bit_width_impl.bit_width_offset.grad)
        # activation: learned scale, plain clamp -> d(bit width) through the scale AND the clamp bounds
        for bits in (3, 6):
            q = RescalingIntQuant(
                IntQuant(narrow_range=False, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp()),
                ParameterScaling(1.5, None, FloatRestrictValue(), 1e-10),
                IntScaling(signed=True, narrow_range=False), ZeroZeroPoint(), BitWidthParameter(bits))
            x = (torch.randn(3, 5, 7) * 1.2).to(DT[dn])
            xi = x.clone().requires_grad_(True)
            y, scale, zp, bwt = q(xi)
            g = torch.randn(y.shape).to(y.dtype)
            y.backward(g)
            st.case({'graph': 'act', 'bits': bits, 'dtype': dn}, x=x, g=g, y=y, scale=scale, bit_width=bwt,
                    dx=xi.grad, doffset=q.msb_clamp_bit_width_impl.bit_width_offset.grad,
                    dvalue=q.scaling_impl.value.grad)
    # the bit-width modules themselves
    bw = BitWidthParameter(5, min_bit_width=3)
    out = bw()
    out.backward()
    st.case({'graph': 'bit_width_parameter'}, out=out, offset=bw.bit_width_offset.data.clone(),
            doffset=bw.bit_width_offset.grad)
    for remove in (0, 3):
        rm = RemoveBitwidthParameter(remove)
        msb = MsbClampBitWidth(rm, 2, 16)
        inp = torch.tensor(9.0, requires_grad=True)
        out = msb(inp)
        out.backward()
        st.case({'graph': 'msb_clamp', 'remove': remove}, out=out, coeff=rm.bit_width_coeff.data.clone(),
                dcoeff=rm.bit_width_coeff.grad, dinp=inp.grad)
    st.save()


# ------------------------------------------------------------------------------------------------
# K. moment statistics (B/core/stats/stats_op.py:186-231): AbsAve, MeanSigmaStd
# ------------------------------------------------------------------------------------------------
def gen_moments():
    from brevitas.core.stats import AbsAve, MeanSigmaStd
    st = Store('moments')
    for dn in ('f32', 'bf16'):
        for tag, shape, dim in (('tensor', (2000,), None), ('rows', (7, 300), 1), ('cols', (300, 7), 0)):
            for name, mk in (('abs_ave', lambda d: AbsAve(d)), ('mean_sigma_std', lambda d: MeanSigmaStd(3.0, d))):
                x = (torch.randn(shape) * 0.7 + 0.1).to(DT[dn])
                x.view(-1)[::17] = 0.0
                xi = x.clone().requires_grad_(True)
                out = mk(dim)(xi)
                g = torch.randn(out.shape).to(out.dtype)
                out.backward(g)
                st.case({'stat': name, 'tag': tag, 'dim': dim, 'dtype': dn, 'shape': list(shape)}, x=x, out=out, g=g,
                        dx=xi.grad)
    st.save()


# ------------------------------------------------------------------------------------------------
# L. KLMinimizerThreshold (B/core/stats/stats_op.py:280-350): threshold search over the histogram of x
# ------------------------------------------------------------------------------------------------
def gen_kl():
    from brevitas.core.stats.stats_op import KLMinimizerThreshold
    st = Store('kl_threshold')
    for tag, signed, bits, make in (
            ('gauss_s8', True, 8, lambda: torch.randn(40000) * 1.5),
            ('gauss_s4', True, 4, lambda: torch.randn(40000) * 0.7),
            ('heavy_tail_s8', True, 8, lambda: torch.randn(40000) * torch.exp(torch.randn(40000))),
            ('relu_u8', False, 8, lambda: torch.relu(torch.randn(40000)) * 2.0),
            ('sparse_s6', True, 6, lambda: torch.randn(40000) * (torch.rand(40000) < 0.1).float())):
        x = make()
        m = KLMinimizerThreshold(signed, BitWidthConst(bits))
        out = m(x)
        a = float(x.abs().max())
        hist = torch.histc(x, bins=1001, min=-a, max=a).int()
        st.case({'tag': tag, 'signed': signed, 'bits': bits}, x=x, out=out, hist=hist)
    st.save()


# ------------------------------------------------------------------------------------------------
# L. affine rescaling of the statistic (_StatsScaling with affine_rescaling=True, B/core/scaling/runtime.py:19-72):
#    threshold = clamp_min(|stat * affine_weight + affine_bias|); weights and activations, gradients of the two
#    affine parameters included
# ------------------------------------------------------------------------------------------------
def gen_affine():
    st = Store('affine')

    def set_affine(q):
        ar = q.scaling_impl.stats_scaling_impl.affine_rescaling
        with torch.no_grad():
            ar.affine_weight.copy_(1.0 + 0.25 * torch.randn(ar.affine_weight.shape))
            ar.affine_bias.copy_(0.05 * torch.randn(ar.affine_bias.shape))
        return ar

    for dn in ('f32', 'bf16'):
        shape = (12, 6, 3, 3)
        w = torch.randn(shape) * 0.05
        w[4] = 0.0
        w = torch.nn.Parameter(w.to(DT[dn]))
        q = RescalingIntQuant(
            IntQuant(narrow_range=True, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClampSte()),
            StatsFromParameterScaling(AbsMax(1), OverOutputChannelView(None), 1, [w], FloatRestrictValue(),
                                      (shape[0], 1, 1, 1), affine_rescaling=True, scaling_min_val=1e-10),
            IntScaling(signed=True, narrow_range=True), ZeroZeroPoint(), BitWidthConst(8))
        ar = set_affine(q)
        y, scale, zp, bwt = q(w)
        g = torch.randn(y.shape).to(y.dtype)
        y.backward(g)
        st.case({'graph': 'weight_affine', 'dtype': dn, 'shape': list(shape)}, x=w.data, g=g, y=y, scale=scale, dx=w.grad,
                affine_weight=ar.affine_weight.detach(), affine_bias=ar.affine_bias.detach(),
                daw=ar.affine_weight.grad, dab=ar.affine_bias.grad)
    for dn in ('f32', 'bf16'):
        for tag, pc in (('per_tensor', None), ('per_channel', 6)):
            if pc is None:
                view, stats, shape = OverTensorView(), AbsMax(), ()
            else:
                view, stats, shape = OverOutputChannelView((1, 0, 2, 3)), AbsMax(1), (1, pc, 1, 1)
            q = RescalingIntQuant(
                IntQuant(narrow_range=False, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp()),
                RuntimeStatsScaling(stats, view, FloatRestrictValue(), shape, affine_rescaling=True,
                                    scaling_stats_momentum=0.1, scaling_min_val=1e-10),
                IntScaling(signed=True, narrow_range=False), ZeroZeroPoint(), BitWidthConst(8))
            ar = set_affine(q)
            q.train()
            for step in range(2):
                x = (torch.randn(4, 6, 5, 5) * (1.0 + step)).to(DT[dn])
                xi = x.clone().requires_grad_(True)
                q.zero_grad()
                y, scale, zp, bwt = q(xi)
                g = torch.randn(y.shape).to(y.dtype)
                y.backward(g)
                st.case({'graph': 'act_affine', 'tag': tag, 'dtype': dn, 'step': step, 'channels': pc},
                        x=x, g=g, y=y, scale=scale, dx=xi.grad,
                        affine_weight=ar.affine_weight.detach(), affine_bias=ar.affine_bias.detach(),
                        daw=ar.affine_weight.grad, dab=ar.affine_bias.grad,
                        running_stats=q.scaling_impl.runtime_stats.running_stats.clone())
    st.save()


# ------------------------------------------------------------------------------------------------
# M. percentile and moment statistics reduced over dim 0 of [rows, channels] tensors (channel axis LAST: [tokens, hidden],
#    flattened NHWC) -- the layout the column-mapped kernels serve
# ------------------------------------------------------------------------------------------------
def gen_channel_last():
    from brevitas.core.stats import AbsAve, AbsPercentile, MeanSigmaStd, NegativePercentileOrZero, PercentileInterval
    st = Store('channel_last')
    for dn, dtype in DT.items():
        for tag, shape in (('tokens_hidden', (96, 16)), ('odd_rows', (131, 24)), ('wide', (65, 200))):
            x = torch.randn(shape) * 2
            x.view(-1)[::17] = 0.0
            x[:, 3] = 0.0                       # an all-zero channel
            x[5, 1] = float('inf')
            x = x.to(dtype)
            for q in (99.999, 50.0, 3.0, 100.0):
                xi = x.clone().requires_grad_(True)
                out = AbsPercentile(q, 0)(xi)
                gout = torch.randn(out.shape).to(out.dtype)
                out.backward(gout)
                st.case({'stat': 'abs_percentile', 'q': q, 'dim': 0, 'tag': tag, 'dtype': dn}, x=x, out=out, gout=gout,
                        dx=xi.grad if q in (99.999, 50.0) else None)
            for q in (0.01, 60.0):
                st.case({'stat': 'neg_percentile', 'q': q, 'dim': 0, 'tag': tag, 'dtype': dn}, x=x,
                        out=NegativePercentileOrZero(q, 0)(x))
            st.case({'stat': 'interval', 'low_q': 0.01, 'high_q': 99.9, 'dim': 0, 'tag': tag, 'dtype': dn}, x=x,
                    out=PercentileInterval(0.01, 99.9, 0)(x))
            xf = torch.where(torch.isfinite(x), x, torch.zeros_like(x))
            xi = xf.clone().requires_grad_(True)
            out = AbsAve(0)(xi)
            gout = torch.randn(out.shape).to(out.dtype)
            out.backward(gout)
            st.case({'stat': 'abs_ave', 'dim': 0, 'tag': tag, 'dtype': dn}, x=xf, out=out, gout=gout, dx=xi.grad)
            xi = xf.clone().requires_grad_(True)
            out = MeanSigmaStd(3.0, 0)(xi)
            gout = torch.randn(out.shape).to(out.dtype)
            out.backward(gout)
            st.case({'stat': 'mean_sigma_std', 'sigma': 3.0, 'dim': 0, 'tag': tag, 'dtype': dn}, x=xf, out=out, gout=gout,
                    dx=xi.grad)
    st.save()


# ------------------------------------------------------------------------------------------------
# N. a weight quantizer SHARED by several layers: _ParameterListStats over more than one tracked parameter
#    (B/core/stats/stats_wrapper.py:83-114) -- the statistic is AbsMax of the concatenation of the weights' views, the
#    tensor quantized is each of them in turn; gradients reach every parameter through the concatenation
# ------------------------------------------------------------------------------------------------
def gen_shared():
    st = Store('shared')
    for dn in ('f32', 'bf16', 'f16'):
        for tag in ('per_channel', 'per_tensor'):
            for shapes in (((8, 6, 3, 3), (8, 4, 1, 1)), ((8, 16), (8, 40), (8, 8))):
                ws = []
                for k, shape in enumerate(shapes):
                    w = torch.randn(shape) * 0.05 * (1 + k)
                    ws.append(w)
                # ties: channel 1's maximum appears in the first AND the second tensor (the second is concatenated in front and
                # owns it), channel 2's
                # twice inside the last tensor, channel 3 is all zero everywhere; per tensor: the global maximum twice
                ws[0].view(8, -1)[1, 2] = 0.75
                ws[1].view(8, -1)[1, 1] = -0.75
                ws[-1].view(8, -1)[2, 0] = 0.875
                ws[-1].view(8, -1)[2, 3] = 0.875
                for w in ws:
                    w.view(8, -1)[3] = 0.0
                if tag == 'per_tensor':
                    ws[0].view(-1)[5] = 1.5
                    ws[-1].view(-1)[7] = -1.5
                ws = [torch.nn.Parameter(w.to(DT[dn])) for w in ws]
                if tag == 'per_channel':
                    scaling = StatsFromParameterScaling(AbsMax(1), OverOutputChannelView(None), 1, ws, FloatRestrictValue(),
                                                        (8,) + (1,) * (ws[0].dim() - 1), affine_rescaling=False,
                                                        scaling_min_val=1e-10)
                else:
                    scaling = StatsFromParameterScaling(AbsMax(), OverTensorView(), 0, ws, FloatRestrictValue(), (),
                                                        affine_rescaling=False, scaling_min_val=1e-10)
                q = RescalingIntQuant(
                    IntQuant(narrow_range=True, signed=True, float_to_int_impl=RoundSte(),
                             tensor_clamp_impl=TensorClampSte()),
                    scaling, IntScaling(signed=True, narrow_range=True), ZeroZeroPoint(), BitWidthConst(8))
                for index in range(len(ws)):
                    for w in ws:
                        w.grad = None
                    y, scale, zp, bwt = q(ws[index])
                    g = torch.randn(y.shape).to(y.dtype)
                    y.backward(g)
                    arrays = dict(g=g, y=y, scale=scale)
                    for k, w in enumerate(ws):
                        arrays['w%d' % k] = w.data
                        arrays['dw%d' % k] = w.grad if w.grad is not None else torch.zeros_like(w)
                    st.case({'graph': 'shared_weight', 'tag': tag, 'dtype': dn, 'index': index, 'n': len(ws),
                             'shapes': [list(sh) for sh in shapes]}, **arrays)
    st.save()


if __name__ == '__main__':
    torch.set_num_threads(1)
    only = sys.argv[1:]
    if not only or not set(only) <= {'act_fused', 'percentile', 'shifted', 'variants', 'fixed_point', 'learned_bw', 'moments', 'kl', 'affine', 'channel_last', 'shared'}:
        # the first four files were generated in ONE run, in this order, from a single seed
        torch.manual_seed(123456)
        gen_ste()
        gen_int_quant()
        gen_stats()
        gen_graphs()
    if not only or 'act_fused' in only:
        torch.manual_seed(123457)
        gen_act_fused()
    if not only or 'percentile' in only:
        torch.manual_seed(123458)
        gen_percentile()
    if not only or 'shifted' in only:
        torch.manual_seed(123459)
        gen_shifted()
    if not only or 'variants' in only:
        torch.manual_seed(123460)
        gen_variants()
    if not only or 'fixed_point' in only:
        torch.manual_seed(123461)
        gen_fixed_point()
    if not only or 'learned_bw' in only:
        torch.manual_seed(123462)
        gen_learned_bw()
    if not only or 'moments' in only:
        torch.manual_seed(123463)
        gen_moments()
    if not only or 'kl' in only:
        torch.manual_seed(123464)
        gen_kl()
    if not only or 'affine' in only:
        torch.manual_seed(123465)
        gen_affine()
    if not only or 'channel_last' in only:
        torch.manual_seed(123466)
        gen_channel_last()
    if not only or 'shared' in only:
        torch.manual_seed(123467)
        gen_shared()
