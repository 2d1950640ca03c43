"""ntpoly_amd: MI355X-native engine for NTPoly's SpGEMM-driven matrix-function hot path.

The product is the shared library `libntpoly_amd.so` (hand-written gfx950 HIP kernels + C++ host
+ the reference's own C ABI, see include/*.h).  This package only loads it and mirrors the
reference's C++/SWIG class surface in Python (host.py).  There is no CPU fallback: the first use of
anything but the build helper loads the library and fails loudly if it has not been built
(`python -m ntpoly_amd._build`).  The load is deferred to that first use so that the build helper itself
(`ntpoly_amd._build`) can be imported on a fresh checkout, before the library exists.
"""
import importlib

_CAPI_NAMES = ("LIB_PATH", "NativeLibraryMissing", "lib")
_SUBMODULES = ("_build", "capi", "host")


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    if name in _SUBMODULES:
        return importlib.import_module("." + name, __name__)
    if name in _CAPI_NAMES:
        return getattr(importlib.import_module(".capi", __name__), name)
    host = importlib.import_module(".host", __name__)   # raises NativeLibraryMissing if the .so is absent
    try:
        return getattr(host, name)
    except AttributeError:
        raise AttributeError("module 'ntpoly_amd' has no attribute %r" % name) from None
