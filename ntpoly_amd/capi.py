"""ctypes view of libntpoly_amd.so -- the same binding a maintainer of the reference would
write against its C ABI (Source/C/*_c.h): handles are caller-owned int[12] buffers, every
scalar is passed by reference, there are no status codes.

There is NO fallback: if the library is missing (or was not built for this machine) importing
this module raises.
"""
import ctypes as C
import os

SIZE_wrp = 12  # Source/C/Wrapper.h:4
HERE = os.path.dirname(os.path.abspath(__file__))
# NTPOLY_AMD_LIB: another build of the SAME library (the host-sanitizer build of ntpoly_amd/_build.py build_sanitized)
LIB_PATH = os.environ.get("NTPOLY_AMD_LIB") or os.path.join(HERE, "libntpoly_amd.so")


class NativeLibraryMissing(ImportError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            "ntpoly_amd/libntpoly_amd.so is missing: build it with `python -m ntpoly_amd._build` "
            "(hipcc --offload-arch=gfx950); the engine has no CPU fallback")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    lib.MatrixNorm_ps_wrp.restype = C.c_double
    lib.MeasureAsymmetry_ps_wrp.restype = C.c_double
    lib.GetGlobalIsRoot_wrp.restype = C.c_bool
    return lib


lib = _load()


def handle():
    return (C.c_int * SIZE_wrp)()


def i(v):
    return C.byref(C.c_int(int(v)))


def d(v):
    return C.byref(C.c_double(float(v)))


def b(v):
    return C.byref(C.c_bool(bool(v)))


def ll(v):
    return C.byref(C.c_longlong(int(v)))


def s(text):
    raw = text.encode()
    return raw, i(len(raw))


def exported_symbols():
    """names declared in include/*.h (used by the CPU test that checks the ABI is complete)"""
    import re
    inc = os.path.join(os.path.dirname(HERE), "include")
    names = []
    for f in sorted(os.listdir(inc)):
        if f.endswith(".h"):
            for m in re.finditer(r"^(?:void|int|double|bool)\s+(\w+)\s*\(", open(os.path.join(inc, f)).read(), re.M):
                names.append(m.group(1))
    return names
