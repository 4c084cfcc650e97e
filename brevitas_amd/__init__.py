"""MI355X-native fake-quantization engine behind Brevitas' op / module surfaces.

Importing the package loads libbvq.so (the C-ABI HIP library); it fails loudly if that is missing.
The flag below mirrors `brevitas.NATIVE_STE_BACKEND_LOADED` (B/__init__.py:60-84): here the native
backend is the only backend.
"""
import sys as _sys

# `python -m brevitas_amd.csrc.build` imports this package on its way to the build script, possibly before
# any library exists (or with a stale one): that one invocation gets the bare package.
_BUILDING = 'brevitas_amd.csrc.build' in getattr(_sys, 'orig_argv', ())

if not _BUILDING:
    from . import _native  # noqa: F401
    from . import config  # noqa: F401
    from . import ops  # noqa: F401

NATIVE_STE_BACKEND_LOADED = not _BUILDING

__version__ = '0.1.0'
