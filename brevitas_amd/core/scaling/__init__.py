"""Threshold ("scaling") modules: what a RescalingIntQuant divides by the integer threshold to obtain its scale
(mirror of B/core/scaling/)."""
from brevitas_amd.core.stats import SCALAR_SHAPE  # noqa: F401  re-exported like the reference does

from . import int_scaling, runtime, standalone

SCALING_STATS_REDUCE_DIM = 1  # per-channel statistics reduce along dim 1 of the [C, -1] view

IntScaling, PowerOfTwoIntScaling = int_scaling.IntScaling, int_scaling.PowerOfTwoIntScaling
StatsFromParameterScaling, RuntimeStatsScaling = runtime.StatsFromParameterScaling, runtime.RuntimeStatsScaling
_StatsScaling = runtime._StatsScaling
ConstScaling, ParameterScaling = standalone.ConstScaling, standalone.ParameterScaling
ParameterFromRuntimeStatsScaling = standalone.ParameterFromRuntimeStatsScaling
