"""ntpoly_amd: MI355X-native engine for NTPoly's SpGEMM-driven matrix-function hot path.

The product is the shared library `libntpoly_amd.so` (hand-written gfx950 HIP kernels + C++ host
+ the reference's own C ABI, see include/*.h).  This package only loads it and mirrors the
reference's C++/SWIG class surface in Python (host.py).  There is no CPU fallback: importing
fails loudly if the library has not been built (`python -m ntpoly_amd._build`).
"""
from .capi import LIB_PATH, NativeLibraryMissing, lib  # noqa: F401
from .host import *  # noqa: F401,F403
