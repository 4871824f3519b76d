"""feature_extraction_amd — MI355X-native per-scan pole detector + 3DSC descriptor.

The product is `lib/libfx_hip.so` (hand-written gfx950 kernels behind the C-ABI of
include/fx.h) and the C++ host classes in csrc/.  This Python package only holds the build
helper and a ctypes binding used by tests/ and bench.py.
"""
from . import build as _build  # noqa: F401

__all__ = ["capi", "build"]
