from .delay import DelayWrapper
from .int import RescalingIntQuant
from .int_base import IntQuant
