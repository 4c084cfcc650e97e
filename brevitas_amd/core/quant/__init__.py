from .delay import DelayWrapper
from .int import PrescaledRestrictIntQuant, PrescaledRestrictIntQuantWithInputBitWidth, RescalingIntQuant
from .int_base import IntQuant
