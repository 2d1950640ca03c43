#!/usr/bin/env python3
"""Drop-in proof at the reference's own C++ layer (test infrastructure, like the rest of oracle/).

Compiles, FROM WHERE THEY LIE under /root/reference (nothing is copied into the repo), the reference's C++ class
layer (Source/CPlusPlus/*.cc, which binds the C ABI of Source/C/*_c.h) and its shipped example driver
Examples/PremadeMatrix/main.cc, and links them -- unchanged -- against ntpoly_amd/libntpoly_amd.so instead of
libNTPolyWrapper + libNTPoly.  Output: oracle/_ref/premade_cxx (git-ignored, travels to the GPU box as a built
file; tests/test_gpu_extras.py runs it there on the reference's PremadeMatrix inputs).

    python oracle/build_cxx_example.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "oracle", "_ref")
# the classes the example touches (every other class of the layer binds solver families outside this engine's scope)
CLASSES = ["DensityMatrixSolvers", "Logging", "PSMatrix", "Permutation", "ProcessGrid", "SolverParameters",
           "SquareRootSolvers", "TripletList", "PMatrixMemoryPool", "SolverBase"]
MPI_INC, MPI_LIB = "/opt/conda/include", "/opt/conda/lib/libmpi.so"


def build():
    if not os.path.isdir(REF):
        print("no /root/reference here: nothing to build")
        return None
    os.makedirs(os.path.join(OUT, "cxx_obj"), exist_ok=True)
    inc = ["-I%s/Source/CPlusPlus" % REF, "-I%s/Source/C" % REF, "-I" + MPI_INC]
    objs = []
    for name, src in [(c, "%s/Source/CPlusPlus/%s.cc" % (REF, c)) for c in CLASSES] + [
            ("premade_main", "%s/Examples/PremadeMatrix/main.cc" % REF)]:
        obj = os.path.join(OUT, "cxx_obj", name + ".o")
        subprocess.run(["g++", "-O1", "-c", src, "-o", obj] + inc, check=True)
        objs.append(obj)
    exe = os.path.join(OUT, "premade_cxx")
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
    subprocess.run(["g++", "-o", exe] + objs + ["-L" + libdir, "-lntpoly_amd", MPI_LIB,
                                                "-Wl,-rpath,$ORIGIN/../../ntpoly_amd", "-Wl,-rpath,$ORIGIN/mpilib",
                                                "-Wl,-rpath,/opt/rocm/lib"], check=True)
    print("built", exe)
    return exe


if __name__ == "__main__":
    sys.exit(0 if build() or not os.path.isdir(REF) else 1)
