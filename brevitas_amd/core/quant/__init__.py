from .binary import BinaryQuant, ClampedBinaryQuant
from .delay import DelayWrapper
from .int import (DecoupledRescalingIntQuant, PrescaledRestrictIntQuant, PrescaledRestrictIntQuantWithInputBitWidth,
                  RescalingIntQuant, TruncIntQuant)
from .int_base import DecoupledIntQuant, IntQuant
from .ternary import TernaryQuant
