"""Module wrappers around the op namespace (mirror of B/core/function_wrapper/): rounding and clamping modules the
quantizers are parameterised with, and the views that bring a tensor into the shape a statistic reduces over."""
from . import clamp as _clamp
from . import misc as _misc
from . import ops_ste as _ste
from . import shape as _shape

# clamps without / with straight-through gradients
TensorClamp, ScalarClamp, ClampMin = _clamp.TensorClamp, _clamp.ScalarClamp, _clamp.ClampMin
TensorClampSte, InplaceTensorClampSte = _ste.TensorClampSte, _ste.InplaceTensorClampSte
ScalarClampMinSte = _ste.ScalarClampMinSte
# float -> integer maps (straight-through)
RoundSte, FloorSte, CeilSte = _ste.RoundSte, _ste.FloorSte, _ste.CeilSte
RoundToZeroSte, DPURoundSte = _ste.RoundToZeroSte, _ste.DPURoundSte
# scale-shaped helpers
Identity, PowerOfTwo, LogTwo, InplaceLogTwo = _misc.Identity, _misc.PowerOfTwo, _misc.LogTwo, _misc.InplaceLogTwo
# statistic input views
PermuteDims, StatsInputViewShapeImpl = _shape.PermuteDims, _shape.StatsInputViewShapeImpl
OverTensorView, OverOutputChannelView = _shape.OverTensorView, _shape.OverOutputChannelView
OverBatchOverTensorView = _shape.OverBatchOverTensorView
OverBatchOverOutputChannelView = _shape.OverBatchOverOutputChannelView
