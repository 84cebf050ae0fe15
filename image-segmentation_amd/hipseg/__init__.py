"""hipseg -- MI355X-native (gfx950) kernels and runtime for the U-Net / ClipUnet training hot path.

Importing this package loads libhipseg.so (the C ABI of include/hipseg.h); it raises if the
library has not been built -- there is no CPU fallback in the product path.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is missing)
from .ops import precision, precision_mode  # noqa: F401

__all__ = ["precision", "precision_mode"]
