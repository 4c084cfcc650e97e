from .stats_op import (AbsAve, AbsMax, AbsMaxAve, AbsMaxL2, AbsMinMax, AbsPercentile, KLMinimizerThreshold,
                       MeanLearnedSigmaStd, MeanSigmaStd,
                       NegativeMinOrZero, NegativePercentileOrZero, PercentileInterval)
from .stats_wrapper import DEFAULT_MOMENTUM, SCALAR_SHAPE, _ParameterListStats, _RuntimeStats, _Stats
from .view_wrapper import _ViewCatParameterWrapper, _ViewParameterWrapper
