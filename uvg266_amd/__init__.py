"""uvg266_amd -- host-side (Python/ctypes) mirror of libuvg266hip.so.

The product is the HIP library (uvg266_amd/csrc, C ABI in include/uvg266_hip.h).
This package only loads it and marshals torch device tensors into the plain
pointers the C ABI takes; it is the driver used by tests/ and bench.py.  There
is no CPU fallback: if the library or a gfx950 device is missing, calls raise.
"""
from .lib import load_library, LibraryMissing, DeviceMissing  # noqa: F401
from . import api  # noqa: F401

__all__ = ["load_library", "LibraryMissing", "DeviceMissing", "api"]
