from .const import BitWidthConst
