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

__all__ = ['calibration_mode', 'finalize_collect_stats', 'DisableEnableQuantization']


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
