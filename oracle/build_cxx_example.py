#!/usr/bin/env python3
"""Drop-in proof at the reference's own C++ layer (test infrastructure, like the rest of oracle/).

Compiles, FROM WHERE THEY LIE under /root/reference (nothing is copied into the repo), the reference's C++ class
layer (ALL of Source/CPlusPlus/*.cc, which binds the C ABI of Source/C/*_c.h) and its shipped C++ example drivers
(Examples/{PremadeMatrix,HydrogenAtom,GraphTheory,ComplexMatrix,MatrixMaps}/main.cc), and links them -- unchanged --
against ntpoly_amd/libntpoly_amd.so instead of libNTPolyWrapper + libNTPoly.  Output: oracle/_ref/*_cxx (git-ignored,
travel to the GPU box as built files; tests/test_gpu_extras.py runs them there).

    python oracle/build_cxx_example.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "oracle", "_ref")
MPI_INC, MPI_LIB = "/opt/conda/include", "/opt/conda/lib/libmpi.so"
# executable name -> the reference's unchanged example driver
EXAMPLES = {"premade_cxx": "PremadeMatrix", "hydrogen_cxx": "HydrogenAtom", "graph_cxx": "GraphTheory",
            "complex_cxx": "ComplexMatrix", "maps_cxx": "MatrixMaps"}


def build():
    if not os.path.isdir(REF):
        print("no /root/reference here: nothing to build")
        return None
    os.makedirs(os.path.join(OUT, "cxx_obj"), exist_ok=True)
    inc = ["-I%s/Source/CPlusPlus" % REF, "-I%s/Source/C" % REF, "-I" + MPI_INC]
    # the whole class layer: every class binds entry points the library exports
    classes = sorted(f[:-3] for f in os.listdir("%s/Source/CPlusPlus" % REF) if f.endswith(".cc"))
    layer = []
    for c in classes:
        obj = os.path.join(OUT, "cxx_obj", c + ".o")
        subprocess.run(["g++", "-O1", "-c", "%s/Source/CPlusPlus/%s.cc" % (REF, c), "-o", obj] + inc, check=True)
        layer.append(obj)
    libdir = os.path.join(ROOT, "ntpoly_amd")
    # MPICH lives in /opt/conda/lib next to an OLD libstdc++; putting that directory on the rpath would shadow the
    # system libstdc++ the ROCm libraries need.  A private directory with links to just the MPI libraries avoids it.
    mpidir = os.path.join(OUT, "mpilib")
    os.makedirs(mpidir, exist_ok=True)
    for lib in ("libmpi.so.12", "libgfortran.so.4", "libquadmath.so.0", "libgomp.so.1"):
        dst = os.path.join(mpidir, lib)
        if os.path.lexists(dst):
            os.unlink(dst)
        os.symlink(os.path.join("/opt/conda/lib", lib), dst)
    exe = None
    for name, example in EXAMPLES.items():
        obj = os.path.join(OUT, "cxx_obj", name + "_main.o")
        subprocess.run(["g++", "-O1", "-c", "%s/Examples/%s/main.cc" % (REF, example), "-o", obj] + inc, check=True)
        exe = os.path.join(OUT, name)
        subprocess.run(["g++", "-o", exe, obj] + layer + ["-L" + libdir, "-lntpoly_amd", MPI_LIB,
                                                          "-Wl,-rpath,$ORIGIN/../../ntpoly_amd", "-Wl,-rpath,$ORIGIN/mpilib",
                                                          "-Wl,-rpath,/opt/rocm/lib"], check=True)
        print("built", exe)
    return exe


if __name__ == "__main__":
    sys.exit(0 if build() or not os.path.isdir(REF) else 1)
