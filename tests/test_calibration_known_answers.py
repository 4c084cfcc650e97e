"""brevitas_amd.graph.calibrate against the KNOWN ANSWERS the reference's own test file holds
(/root/reference/tests/brevitas/graph/test_calibration.py): the closed-form scale of a calibrated fixed-point activation
quantizer (reference_implementation_scale_factors_po2, :18-32 -> test_scale_factors_ptq_calibration_po2, :35-54), the
training-state contract of calibration_mode (:57-74), and the bias-correction results and hook behaviour of
bias_correction_mode (:80-175), restated against this package's quantizers and thin layers.  The reference module itself
cannot be imported here (it sits on brevitas.nn / the injector stack, whose third-party dependency is absent), so these
closed forms -- written by the reference's authors -- are what pins calibration_mode and bias_correction_mode.  Every
test runs on CPU tensors (the package's pure-torch CPU route) and, marked gpu, on device tensors (the HIP kernels)."""
import math

import pytest
import torch

IN_CH, OUT_CH, BATCH = 8, 16, 1
DEVICES = [pytest.param('cpu', id='cpu'), pytest.param('cuda:0', id='gpu', marks=pytest.mark.gpu)]


def compute_quantile(x, q):
    k = int(math.floor(.01 * q * x.numel() + 0.5))
    return x.abs().view(-1).kthvalue(k).values


def reference_implementation_scale_factors_po2(x, q=99.999, min_val=1e-10, int_scale=128.):
    """the closed form of the reference's test (test_calibration.py:22-31)"""
    quant = compute_quantile(x, q)
    quant = torch.max(torch.tensor(min_val, device=x.device), quant)
    quant_float_to_int = torch.ceil(torch.log2(quant))   # float-to-int of a power-of-two scale
    return torch.pow(torch.tensor(2., device=x.device), quant_float_to_int) / int_scale


def _identity_model(dev):
    import brevitas_amd.quant as Q
    from brevitas_amd.nn import QuantIdentity

    class TestModel(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.act = QuantIdentity(act_quant=Q.Int8ActPerTensorFixedPoint())

        def forward(self, x):
            return self.act(x)
    return TestModel().to(dev)


@pytest.mark.parametrize('dev', DEVICES)
@pytest.mark.parametrize('shape,seed,gain', [((7,), 0, 1.0), ((3, 5), 1, 30.0), ((2, 3, 4, 5), 2, 1e-3), ((64, 33), 3, 1.0),
                                            ((1,), 4, 5.0), ((4, 1, 17), 5, 1e4), ((25000,), 6, 1.0)])
def test_scale_factors_ptq_calibration_po2(dev, shape, seed, gain):
    from brevitas_amd.graph.calibrate import calibration_mode
    g = torch.Generator().manual_seed(seed)
    inp = (torch.randn(shape, generator=g) * gain).to(dev)
    model = _identity_model(dev)
    model.eval()
    with torch.no_grad():
        with calibration_mode(model):
            model(inp)
    expected_scale = reference_implementation_scale_factors_po2(inp)
    with torch.no_grad():
        scale = model.act.act_quant(inp)[1]      # the reference's quant_act_scale(): the quantizer's scale in eval mode
    assert torch.allclose(expected_scale, scale.reshape(()).to(expected_scale.dtype))
    assert torch.equal(expected_scale, scale.reshape(()).to(expected_scale.dtype))   # powers of two: exact


@pytest.mark.parametrize('dev', DEVICES)
def test_calibration_training_state(dev):
    from brevitas_amd.graph.calibrate import calibration_mode
    model = _identity_model(dev)
    model.eval()
    with torch.no_grad():
        with calibration_mode(model):
            assert model.act.act_quant.training is True
            assert model.training is False
    assert model.act.act_quant.training is False
    assert model.training is False


def _models(dev):
    import brevitas_amd.quant as Q
    from brevitas_amd.nn import QuantLinear

    class MyModel(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.module_list = torch.nn.ModuleList([torch.nn.Linear(IN_CH, OUT_CH, bias=False),
                                                    torch.nn.Linear(OUT_CH, OUT_CH, bias=False)])

        def forward(self, inp):
            out_0 = self.module_list[0](inp)
            out_1 = self.module_list[1](out_0)
            return torch.cat((out_0, out_1))

    class MyQuantModel(torch.nn.Module):
        def __init__(self):
            super().__init__()
            # QuantLinear's default weight quantizer in the reference: Int8WeightPerTensorFloat, no input quantizer
            self.module_list = torch.nn.ModuleList([
                QuantLinear(IN_CH, OUT_CH, bias=False, weight_quant=Q.Int8WeightPerTensorFloat),
                QuantLinear(OUT_CH, OUT_CH, bias=False, weight_quant=Q.Int8WeightPerTensorFloat)])

        def forward(self, inp):
            out_0 = self.module_list[0](inp)
            out_1 = self.module_list[1](out_0)
            return torch.cat((out_0, out_1))
    torch.manual_seed(7)
    model, quant_model = MyModel().to(dev), MyQuantModel().to(dev)
    quant_model.module_list[0].weight.data = model.module_list[0].weight.data
    quant_model.module_list[1].weight.data = model.module_list[1].weight.data
    model.eval()
    quant_model.eval()
    return model, quant_model


@pytest.mark.parametrize('dev', DEVICES)
def test_bias_correction_results(dev):
    from brevitas_amd.graph.calibrate import bias_correction_mode
    fp_model, quant_model = _models(dev)
    num_layers = len(quant_model.module_list)
    inp_list = [torch.randn(BATCH, IN_CH, device=dev), torch.randn(BATCH, IN_CH, device=dev)]
    fp_outs = torch.zeros(len(inp_list), num_layers, OUT_CH, device=dev)
    quant_outs = torch.zeros(len(inp_list), num_layers, OUT_CH, device=dev)
    error = torch.zeros(num_layers, OUT_CH, device=dev)
    with torch.no_grad():
        # the reference's own restatement of bias correction (test_calibration.py:126-133)
        for b, inp in enumerate(inp_list):
            fp_outs[b, :, :] = fp_model(inp)
            quant_outs[b, 0, :] = quant_model.module_list[0](inp)
            quant_outs[b, 1, :] = quant_model.module_list[1](fp_outs[b, 0, :])  # fed the "corrected" output
            error += fp_outs[b] - quant_outs[b]
        with bias_correction_mode(quant_model):
            for inp in inp_list:
                quant_model(inp)
    assert quant_model.module_list[0].bias is not None
    assert quant_model.module_list[1].bias is not None
    assert torch.allclose(quant_model.module_list[0].bias, error[0] / len(inp_list), atol=1e-6)
    assert torch.allclose(quant_model.module_list[1].bias, error[1] / len(inp_list), atol=1e-6)


@pytest.mark.parametrize('dev', DEVICES)
def test_bias_correction_hook(dev):
    from brevitas_amd.graph.calibrate import bias_correction_mode
    fp_model, quant_model = _models(dev)
    num_layers = len(quant_model.module_list)
    inp_list = [torch.randn(BATCH, IN_CH, device=dev), torch.randn(BATCH, IN_CH, device=dev)]
    inputs, outputs = [], []

    def simple_hook(mod, inp, out):   # a user hook on the second layer: must fire once per call, on the corrected input
        inputs.append(*inp)
        outputs.append(*out)
    fp_outs = torch.zeros(len(inp_list), num_layers, OUT_CH, device=dev)
    with torch.no_grad():
        for b, inp in enumerate(inp_list):
            fp_outs[b, :, :] = fp_model(inp)
        quant_model.module_list[1].register_forward_hook(simple_hook)
        with bias_correction_mode(quant_model):
            for inp in inp_list:
                quant_model(inp)
    assert len(outputs) == 2   # once per input, although every layer ran three forwards per input
    # in bias-correction mode the input of a layer equals the float output of the previous one
    assert torch.allclose(inputs[0], fp_outs[0, 0, :], atol=1e-6)
    assert torch.allclose(inputs[1], fp_outs[1, 0, :], atol=1e-6)
