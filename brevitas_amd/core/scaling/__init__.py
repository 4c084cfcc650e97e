from brevitas_amd.core.stats import SCALAR_SHAPE

from .int_scaling import IntScaling, PowerOfTwoIntScaling
from .runtime import RuntimeStatsScaling, StatsFromParameterScaling, _StatsScaling
from .standalone import ConstScaling, ParameterFromRuntimeStatsScaling, ParameterScaling

SCALING_STATS_REDUCE_DIM = 1
