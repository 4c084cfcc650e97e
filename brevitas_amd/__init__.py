"""MI355X-native fake-quantization engine behind Brevitas' op / module surfaces.

Importing the package loads libbvq.so (the C-ABI HIP library); it fails loudly if that is missing.
The flag below mirrors `brevitas.NATIVE_STE_BACKEND_LOADED` (B/__init__.py:60-84): here the native
backend is the only backend.
"""
from . import _native  # noqa: F401
from . import config  # noqa: F401
from . import ops  # noqa: F401

NATIVE_STE_BACKEND_LOADED = True

__version__ = '0.1.0'
