"""Environment flags, read once at import (mirrors B/config.py:13-25 for the flags this path honours)."""
import os


def _env_flag(name, default):
    v = os.environ.get(name)
    if v is None:
        return default
    return v.strip().lower() not in ('0', 'false', 'no', '')


IGNORE_MISSING_KEYS = _env_flag('BREVITAS_IGNORE_MISSING_KEYS', False)
REINIT_ON_STATE_DICT_LOAD = _env_flag('BREVITAS_REINIT_ON_STATE_DICT_LOAD', True)
VERBOSE = _env_flag('BREVITAS_VERBOSE', False)

# How a 0-dim scale / zero-point WIDER than the compute dtype enters the kernels (include/bvq.h,
# bvq_scalar_mode).  'device' reproduces what torch's own device kernels do with such an operand
# (it is rounded to the compute dtype first); 'cpu' reproduces ATen's CPU reduced-float scalar path
# (the scalar keeps its float32 value).  Same-dtype and per-channel operands are unaffected.
SCALAR_OPERAND_MODE = os.environ.get('BREVITAS_AMD_SCALAR_OPERAND_MODE', 'device')
# fused fast paths of RescalingIntQuant (recognised quantizer graphs); 0 forces the generic composition
FUSED_PATHS = _env_flag('BREVITAS_AMD_FUSED', True)
# the weight quantizer's autograd node in C++ (brevitas_amd/_bvq_autograd.so, host glue over the same C-ABI calls);
# 0: always the Python torch.autograd.Function
CPP_AUTOGRAD = _env_flag('BREVITAS_AMD_CPP_AUTOGRAD', True)
# the batch-sharded form of the activation node, which issues its two collectives through c10d from C++; 0: sharded
# quantizers take the Python Function (bench.py switches it off for the run if its start-up probe of the node fails)
CPP_AUTOGRAD_SHARDED = _env_flag('BREVITAS_AMD_CPP_SHARDED', True)
