"""PTQ calibration (brevitas_amd.graph.calibrate, drop-in for B/graph/calibrate.py:46-66,99-166).

The reference's calibration_mode cannot be imported here (it sits on brevitas.nn / the injector stack,
whose third-party dependency is absent: parity unpinned by reference output), so the test pins the
collect-only forward to the reference PROCEDURE instead: run every activation quantizer in full in
training mode, discard its output and pass the float activation on (what the reference's forward hook
does), with the full quantizers being the ones already pinned bit-exactly by the golden vectors.  State
(buffers, counters, learned values) and the quantized model's outputs afterwards must be identical."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


class ActLayer(torch.nn.Module):
    """QuantReLU: activation fused with its quantizer"""

    def __init__(self, proxy):
        super().__init__()
        self.fused_activation_quant_proxy = proxy

    def forward(self, x):
        return self.fused_activation_quant_proxy(x)[0]


def build(stats_op):
    import brevitas_amd.quant as Q
    from brevitas_amd.nn import QuantConv2d, QuantLinear
    from brevitas_amd.proxy import FusedActivationQuantProxy
    torch.manual_seed(123456)
    conv = QuantConv2d(3, 8, 3, padding=1, weight_quant=Q.Int8WeightPerChannelFloat,
                       input_quant=Q.Int8ActPerTensorFloat(collect_stats_steps=300, scaling_stats_op=stats_op),
                       device=DEV)
    act = ActLayer(FusedActivationQuantProxy(
        torch.nn.ReLU(), Q.Uint8ActPerTensorFloat(collect_stats_steps=300, scaling_stats_op=stats_op))).to(DEV)
    shifted = Q.ShiftedUint8ActPerTensorFloat(collect_stats_steps=300)
    fc = QuantLinear(8 * 6 * 6, 5, weight_quant=Q.Int8WeightPerTensorFloat, input_quant=shifted, device=DEV)
    return torch.nn.ModuleDict(dict(conv=conv, act=act, fc=fc))


def run(m, x):
    h = m['act'](m['conv'](x))
    return m['fc'](h.flatten(1))


def reference_procedure_step(m, x):
    """B/graph/calibrate.py:115-127: quantizers run in full (training mode), outputs are discarded"""
    conv, act, fc = m['conv'], m['act'], m['fc']
    conv.input_quant(x)
    h = torch.nn.functional.conv2d(x, conv.weight, conv.bias, padding=1)
    act.fused_activation_quant_proxy(h)
    h = torch.relu(h).flatten(1)
    fc.input_quant(h)
    return torch.nn.functional.linear(h, fc.weight, fc.bias)


@pytest.mark.parametrize('stats_op', ['percentile', 'max'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_calibration_mode_equals_reference_procedure(stats_op, dtype):
    from brevitas_amd.graph.calibrate import calibration_mode, finalize_collect_stats
    a, b = build(stats_op).to(dtype), build(stats_op).to(dtype)
    a.eval(), b.eval()
    torch.manual_seed(7)
    batches = [(torch.randn(4, 3, 6, 6, device=DEV) * (1 + i)).to(dtype) for i in range(3)]
    with torch.no_grad():
        with calibration_mode(a):
            assert a['conv'].input_quant.training and a['conv'].input_quant.bvq_collect_only
            outs_a = [run(a, x) for x in batches]
        assert not a['conv'].input_quant.training and not a['conv'].input_quant.bvq_collect_only
        b.train()
        outs_b = [reference_procedure_step(b, x) for x in batches]
        b.apply(finalize_collect_stats)
        b.eval()
        # float forwards during calibration
        for ya, yb in zip(outs_a, outs_b):
            assert torch.equal(ya, yb)
        # identical quantizer state
        sa, sb = a.state_dict(), b.state_dict()
        assert sa.keys() == sb.keys()
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k
        for (na, ma), (nb, mb) in zip(a.named_modules(), b.named_modules()):
            if hasattr(ma, 'counter'):
                assert ma.counter == mb.counter == ma.collect_stats_steps, na
                assert torch.equal(ma.buffer, mb.buffer), na
        # and the calibrated quantized model agrees
        x = (torch.randn(4, 3, 6, 6, device=DEV) * 2).to(dtype)
        assert torch.equal(run(a, x), run(b, x))
        # quantization is back on: the output differs from the float model's
        assert not torch.equal(run(a, x), reference_procedure_step(build(stats_op).to(dtype).train(), x))


def test_calibration_runtime_stats_collect_only_matches_full_forward():
    """RuntimeStatsScaling (running average of the abs-max): the collect-only pass updates running_stats
    exactly as the full fused forward does, with and without the fused ReLU"""
    from bench import build_quantizer
    from brevitas_amd import _native as nat
    torch.manual_seed(123456)
    for per_channel in (True, False):
        for pre_op in (nat.PRE_NONE, nat.PRE_RELU):
            qa = build_quantizer(16, per_channel, torch.device(DEV))
            qb = build_quantizer(16, per_channel, torch.device(DEV))
            qa.bvq_collect_only = True
            for i in range(3):
                x = (torch.randn(4, 16, 5, 5, device=DEV) * (1 + i)).to(torch.bfloat16)
                ya, sa, za, _ = qa.bvq_forward_pre(x, pre_op)
                yb, sb, zb, _ = qb.bvq_forward_pre(x, pre_op)
                assert torch.equal(ya, torch.relu(x) if pre_op else x)
                assert torch.equal(sa, sb)
                assert torch.equal(qa.scaling_impl.runtime_stats.running_stats,
                                   qb.scaling_impl.runtime_stats.running_stats)
