from .const import BitWidthConst, MsbClampBitWidth
from .parameter import BitWidthParameter, RemoveBitwidthParameter
