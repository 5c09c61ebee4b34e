"""uvg266_amd -- host-side (Python/ctypes) mirror of libuvg266hip.so.

The product is the HIP library (uvg266_amd/csrc, C ABI in include/uvg266_hip.h).
This package only loads it and marshals torch device tensors into the plain
pointers the C ABI takes; it is the driver used by tests/ and bench.py.  There
is no CPU fallback: if the library or a gfx950 device is missing, calls raise.
"""
import os as _os

# A tiles plan (csrc/tiles.hip) runs one launch per tile size beside the others, each on its own HIP stream -- up to four for a uniform
# grid, more when the picture's edges cut CTUs.  The runtime carries GPU_MAX_HW_QUEUES (default 4) streams side by side; two streams on
# one hardware queue run their launches one after the other (measured: one 1080p picture in 4 x 2 tiles 296 ms instead of 155).  The
# variable is read when the HIP runtime initialises, i.e. at the first device call: a default here, before any, costs nothing else.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .lib import load_library, LibraryMissing, DeviceMissing  # noqa: F401
from . import api  # noqa: F401

__all__ = ["load_library", "LibraryMissing", "DeviceMissing", "api"]
