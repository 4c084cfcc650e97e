"""BASELINE.json configs 4 and 5 as parity cases, driven through the thin layers
(brevitas_amd.nn) and the named-quantizer graphs (brevitas_amd.quant):

  4. QuantConv2d (Int8 per-channel weight + Int8 per-tensor act), ResNet-50 layer3 bottleneck
     (1x1 1024->256, 3x3 256->256, 1x1 256->1024) on [n,1024,14,14] bf16;
  5. Int4WeightPerChannelFloat + Int8ActPerTensorFloat on Linear[8192,8192], x [b,8192] bf16.

Every quantized operand is compared bit for bit with the oracle; the layer output must then equal the
float conv / linear of those operands (same torch op on the same bits)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def oracle_weight(oracle, w, bit_width):
    """Int{8,4}WeightPerChannelFloat on the CPU oracle -> (dequantized weight, scale) as torch tensors"""
    cout = w.shape[0]
    k = w.numel() // cout
    qmax = float(2 ** (bit_width - 1) - 1)
    wn, code = oracle.from_torch(w.reshape(-1))
    d = oracle.make_desc(1, cout, k, code, code, code, oracle.F32, scale_per_channel=True, qmin=-qmax, qmax=qmax,
                         clamp_ste=True)
    stat = oracle.stats(oracle.STAT_ABSMAX, wn, code, 1, cout, k)
    thr = np.maximum(stat, np.float32(1e-10))
    scale = oracle.to_torch(oracle.from_torch((torch.from_numpy(thr).to(w.dtype).float() / qmax).to(w.dtype))[0], code)
    y, _ = oracle.fakequant_fwd(d, wn, oracle.from_torch(scale)[0], np.zeros(1, np.float32))
    return oracle.to_torch(y, code).reshape(w.shape), scale


def oracle_act_per_tensor(oracle, x, scalar_mode):
    """Int8ActPerTensorFloat (MAX statistics, collection phase): scale = max(absmax, 1e-10) / 128"""
    xn, code = oracle.from_torch(x.reshape(-1))
    stat = oracle.stats(oracle.STAT_ABSMAX, xn, code, 1, 1, xn.size)
    # threshold: bf16 0-dim + 0. * float32 value -> float32 ; scale = threshold / 128 in float32
    scale = (np.maximum(stat, np.float32(1e-10)) / np.float32(128.0)).astype(np.float32)
    d = oracle.make_desc(1, 1, xn.size, code, code, oracle.F32, oracle.F32, qmin=-128.0, qmax=127.0,
                         scalar_mode=scalar_mode)
    y, _ = oracle.fakequant_fwd(d, xn, scale, np.zeros(1, np.float32))
    return oracle.to_torch(y, code).reshape(x.shape), torch.from_numpy(scale)


def test_config4_resnet50_layer3_block(oracle):
    import brevitas_amd.quant as Q
    from brevitas_amd.nn import QuantConv2d
    torch.manual_seed(123456)
    n = 8  # the real config runs 128 per GPU; the arithmetic does not depend on the batch size
    x = torch.randn(n, 1024, 14, 14).to(torch.bfloat16)
    specs = [(1024, 256, 1, 0), (256, 256, 3, 1), (256, 1024, 1, 0)]
    layers = [QuantConv2d(ci, co, k, padding=p, bias=False, weight_quant=Q.Int8WeightPerChannelFloat,
                          input_quant=Q.Int8ActPerTensorFloat(collect_stats_steps=300, scaling_stats_op='max'),
                          dtype=torch.bfloat16,
                          device=DEV) for ci, co, k, p in specs]
    h = x.to(DEV)
    for layer in layers:
        layer.train()
        # operands as the reference's graphs would produce them (checked on the CPU oracle)
        xq_o, xs_o = oracle_act_per_tensor(oracle, h.cpu(), oracle.SCALAR_CAST)
        wq_o, ws_o = oracle_weight(oracle, layer.weight.detach().cpu(), 8)
        xq, xs, _, _ = layer.input_quant(h)
        wq, ws, _, _ = layer.quant_weight()
        assert torch.equal(xs.cpu().reshape(-1), xs_o.reshape(-1))
        assert torch.equal(xq.cpu().view(torch.int16), xq_o.view(torch.int16))
        assert torch.equal(ws.cpu().reshape(-1).view(torch.int16), ws_o.reshape(-1).view(torch.int16))
        assert torch.equal(wq.cpu().view(torch.int16), wq_o.view(torch.int16))
        out = layer(h)
        ref = F.conv2d(xq_o.to(DEV), wq_o.to(DEV), None, layer.stride, layer.padding)
        assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
        h = torch.relu(out).detach()
    # one training step through the whole block: gradients reach every weight and the learned scales
    h = x.to(DEV).requires_grad_(True)
    y = h
    for layer in layers:
        y = torch.relu(layer(y))
    y.float().pow(2).mean().backward()
    for layer in layers:
        assert layer.weight.grad is not None and torch.isfinite(layer.weight.grad.float()).all()
        assert layer.input_quant.scaling_impl.value.grad is not None  # zero during collection (DDP workaround)
    assert torch.isfinite(h.grad.float()).all()


def test_config5_linear_int4_weight_int8_act(oracle):
    import brevitas_amd.quant as Q
    from brevitas_amd.nn import QuantLinear
    torch.manual_seed(123456)
    layer = QuantLinear(8192, 8192, bias=False, weight_quant=Q.Int4WeightPerChannelFloat,
                        input_quant=Q.Int8ActPerTensorFloat(scaling_stats_op='max'), dtype=torch.bfloat16, device=DEV)
    with torch.no_grad():
        layer.weight.copy_((torch.randn(8192, 8192) * 0.01).to(torch.bfloat16))
    x = torch.randn(64, 8192).to(torch.bfloat16)
    xq_o, xs_o = oracle_act_per_tensor(oracle, x, oracle.SCALAR_CAST)
    wq_o, ws_o = oracle_weight(oracle, layer.weight.detach().cpu(), 4)
    layer.train()
    xq, xs, _, _ = layer.input_quant(x.to(DEV))
    wq, ws, _, bw = layer.quant_weight()
    assert float(bw) == 4.0
    assert torch.equal(xq.cpu().view(torch.int16), xq_o.view(torch.int16)) and torch.equal(xs.cpu().reshape(-1), xs_o)
    assert torch.equal(wq.cpu().view(torch.int16), wq_o.view(torch.int16))
    codes = torch.round(wq.float() / ws.float())
    assert float(codes.min()) >= -7 and float(codes.max()) <= 7  # int4, narrow range
    out = layer(x.to(DEV))
    assert torch.equal(out.view(torch.int16), F.linear(xq_o.to(DEV), wq_o.to(DEV)).view(torch.int16))
    out.float().sum().backward()
    assert torch.isfinite(layer.weight.grad.float()).all()


@pytest.mark.parametrize('bias_bits', [8, 16, 32])
def test_bias_quantized_with_the_accumulator_scale(bias_bits):
    """QuantConv2d(bias_quant=IntNBias): the bias is quantized with quant_input.scale * quant_weight.scale, one
    scale per output channel (B/nn/quant_layer.py:316-326, B/quant/scaled_int.py:64-132)"""
    import torch.nn.functional as F

    import brevitas_amd.quant as Q
    from brevitas_amd.nn import QuantConv2d
    torch.manual_seed(123456)
    factory = {8: Q.Int8Bias, 16: Q.Int16Bias, 32: Q.Int32Bias}[bias_bits]
    layer = QuantConv2d(6, 10, 3, padding=1, bias=True, weight_quant=Q.Int8WeightPerChannelFloat,
                        input_quant=Q.Int8ActPerTensorFloat(scaling_impl_type='stats', scaling_stats_op='max'),
                        bias_quant=factory(), device=DEV)
    with torch.no_grad():
        layer.bias.mul_(3.0)
    x = torch.randn(4, 6, 9, 9, device=DEV, requires_grad=True)
    out = layer(x)
    xq, in_scale, _, _ = layer.input_quant(x.detach())
    wq, w_scale, _, _ = layer.quant_weight()
    scale = (w_scale.reshape(-1) * in_scale.reshape(-1)).reshape(-1)
    lo, hi = -(2.0 ** (bias_bits - 1)), 2.0 ** (bias_bits - 1) - 1
    codes = torch.clamp(torch.round(layer.bias / scale), lo, hi)
    bq = codes * scale
    want = F.conv2d(xq, wq, bq, padding=1)
    assert torch.equal(out, want)
    if bias_bits == 8:
        assert float(codes.abs().max()) == 128.0 or float(codes.max()) == 127.0  # the 8-bit range clips a bias this large
    out.sum().backward()
    assert layer.bias.grad is not None and x.grad is not None and layer.weight.grad is not None


def test_bias_quantizers_with_internal_scaling():
    import brevitas_amd.quant as Q
    from brevitas_amd.nn import QuantLinear
    torch.manual_seed(123456)
    for factory in (Q.Int8BiasPerTensorFloatInternalScaling, Q.Int8BiasPerTensorFixedPointInternalScaling):
        layer = QuantLinear(16, 12, bias=True, weight_quant=Q.Int8WeightPerTensorFloat, bias_quant=factory, device=DEV)
        y, scale, _, _ = layer.bias_quant(layer.bias)
        codes = y / scale
        assert float((codes - torch.round(codes)).abs().max()) < 1e-3 and float(codes.abs().max()) <= 128.001
        if factory is Q.Int8BiasPerTensorFixedPointInternalScaling:
            m, _ = torch.frexp(scale.detach())
            assert float(m) == 0.5
        out = layer(torch.randn(3, 16, device=DEV))
        out.sum().backward()
        assert layer.bias.grad is not None
