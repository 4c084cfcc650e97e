from .clamp import ClampMin, ScalarClamp, TensorClamp
from .misc import Identity, InplaceLogTwo, LogTwo, PowerOfTwo
from .ops_ste import (CeilSte, DPURoundSte, FloorSte, InplaceTensorClampSte, RoundSte, RoundToZeroSte,
                      ScalarClampMinSte, TensorClampSte)
from .shape import (OverBatchOverOutputChannelView, OverBatchOverTensorView, OverOutputChannelView,
                    OverTensorView, PermuteDims, StatsInputViewShapeImpl)
