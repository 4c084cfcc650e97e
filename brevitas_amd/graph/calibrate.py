"""Post-training calibration: run float forward passes while the activation quantizers only collect
their statistics (drop-in for calibration_mode / DisableEnableQuantization / finalize_collect_stats,
B/graph/calibrate.py:46-66,99-166).

In the reference a calibration forward runs every activation quantizer in training mode in full --
statistic, scale, quantize, dequantize -- and a forward hook then throws the quantized tensor away and
hands the float activation on (B/graph/calibrate.py:115-127); parameter quantizers are switched off
(`disable_quant`).  Here the activation quantizers are put into *collect-only* mode instead: they run
the statistic kernels (one streaming read for abs-max / min-max, two or three for a percentile), update
their buffers and counters exactly as the full forward would, and skip the quantize/dequantize pass whose
result nobody reads.  State after calibration is identical to the reference procedure's; a calibration
step moves 1x (abs-max) instead of 3x the activation through HBM.
"""
import torch

from brevitas_amd.core.quant.int import RescalingIntQuant
from brevitas_amd.proxy import FusedActivationQuantProxy

__all__ = ['calibration_mode', 'bias_correction_mode', 'finalize_collect_stats', 'DisableEnableQuantization']


def finalize_collect_stats(module):
    """end the statistics collection phase of a ParameterFromRuntime* module now
    (B/graph/calibrate.py:46-48)"""
    if hasattr(module, 'collect_stats_steps') and hasattr(module, 'counter'):
        module.counter = module.collect_stats_steps


def _act_quantizers(model):
    """activation-side tensor_quant modules of the thin layers, each once"""
    from brevitas_amd.nn import QuantIdentity, _QuantWeightMixin
    seen = set()
    for m in model.modules():
        cands = []
        if isinstance(m, _QuantWeightMixin):
            cands.append(m.input_quant)
        elif isinstance(m, QuantIdentity):
            cands.append(m.act_quant)
        elif isinstance(m, FusedActivationQuantProxy):
            cands.append(m.tensor_quant)
        for q in cands:
            if isinstance(q, FusedActivationQuantProxy):
                q = q.tensor_quant
            if q is not None and id(q) not in seen:
                seen.add(id(q))
                yield q


def _weight_layers(model):
    from brevitas_amd.nn import _QuantWeightMixin
    return [m for m in model.modules() if isinstance(m, _QuantWeightMixin)]


class DisableEnableQuantization:
    """apply(model, is_training, quantization_enabled) (B/graph/calibrate.py:99-166)"""

    def disable_act_quantization(self, model, is_training):
        for q in _act_quantizers(model):
            q.train(is_training)
            if isinstance(q, RescalingIntQuant):
                q.bvq_collect_only = True
            else:
                raise NotImplementedError('calibration of %s' % type(q).__name__)

    def disable_param_quantization(self, model, is_training):
        for layer in _weight_layers(model):
            layer.bvq_disable_weight_quant = True
            if layer.weight_quant is not None:
                layer.weight_quant.train(is_training)

    def enable_act_quantization(self, model, is_training):
        for q in _act_quantizers(model):
            q.train(is_training)
            q.bvq_collect_only = False

    def enable_param_quantization(self, model, is_training):
        for layer in _weight_layers(model):
            layer.bvq_disable_weight_quant = False
            if layer.weight_quant is not None:
                layer.weight_quant.train(is_training)

    def apply(self, model, is_training, quantization_enabled):
        if not quantization_enabled:
            self.disable_act_quantization(model, is_training)
            self.disable_param_quantization(model, is_training)
        else:
            self.enable_act_quantization(model, is_training)
            self.enable_param_quantization(model, is_training)
        return model


class calibration_mode:
    """with calibration_mode(model): model(batch) ...   -- float forwards that collect activation statistics;
    on exit the collection phase is closed and the model returns to its previous training state."""

    def __init__(self, model: torch.nn.Module, enabled: bool = True):
        self.model = model
        self.previous_training_state = model.training
        self.disable_quant_inference = DisableEnableQuantization()
        self.enabled = enabled

    def __enter__(self):
        if self.enabled:
            self.disable_quant_inference.apply(self.model, is_training=True, quantization_enabled=False)
        return self

    def __exit__(self, exc_type, exc, tb):
        self.model.apply(finalize_collect_stats)
        self.disable_quant_inference.apply(self.model, is_training=self.previous_training_state,
                                           quantization_enabled=True)
        return False


class bias_correction_mode:
    """with bias_correction_mode(model): model(batch) ...   (B/graph/calibrate.py:68-80,166-276)

    Every quantized conv / linear layer is run twice more per call: once with its quantizers off and once with them on
    (plain `forward`, so hooks of the caller fire once per call); the difference of the two outputs' per-channel means
    is accumulated, and the layer hands on its quantized output PLUS that difference -- the next layer sees what
    the float layer would have produced on average.  On exit each layer's bias receives the mean difference over the
    calls (a bias parameter is created where the layer had none)."""

    def __init__(self, model: torch.nn.Module, enabled: bool = True):
        self.model = model
        self.enabled = enabled
        self.hooks = []
        self.iterations = {}
        self.correction_map = {}

    @staticmethod
    def _channel_dim(t, module):
        return 2 if t.dim() == 3 and isinstance(module, torch.nn.Linear) else 1   # B/graph/calibrate.py:183-188

    @staticmethod
    def _channel_mean(t, dim):
        t = t.transpose(0, dim)
        return t.reshape(t.shape[0], -1).mean(dim=1).detach()                     # compute_mean, :179-181

    def _hook(self, module, inp, output, name):
        saved = (getattr(module, 'bvq_disable_weight_quant', False), getattr(module, 'bvq_disable_input_quant', False))
        module.bvq_disable_weight_quant = module.bvq_disable_input_quant = True
        try:
            float_out = module.forward(*inp)        # forward, not __call__: no recursion into this hook
        finally:
            module.bvq_disable_weight_quant, module.bvq_disable_input_quant = saved
        quant_out = module.forward(*inp)
        dim = self._channel_dim(quant_out, module)
        error = self._channel_mean(float_out, dim) - self._channel_mean(quant_out, dim)
        if name in self.correction_map:
            self.correction_map[name] += error
        else:
            self.correction_map[name] = error
        self.iterations[name] += 1
        shape = [1] * quant_out.dim()
        shape[dim] = -1
        return quant_out + error.reshape(shape)

    def __enter__(self):
        if self.enabled:
            from functools import partial

            from brevitas_amd.nn import _QuantWeightMixin
            for name, module in self.model.named_modules():
                if isinstance(module, _QuantWeightMixin):
                    self.iterations[name] = 0
                    self.hooks.append(module.register_forward_hook(partial(self._hook, name=name)))
        return self

    def __exit__(self, exc_type, exc, tb):
        for name, module in self.model.named_modules():
            if name in self.correction_map:
                correction = self.correction_map[name] / self.iterations[name]
                if module.bias is not None:
                    module.bias.data += correction
                else:
                    module.register_parameter('bias', torch.nn.Parameter(correction).to(module.weight.device))
        for hook in self.hooks:
            hook.remove()
        self.hooks = []
        return False
