#!/usr/bin/env python3
"""Drop-in proof at the reference's Fortran API (test infrastructure, like the rest of oracle/).

fortran/ntpoly_amd_modules.f90 is the product's Fortran layer: the reference's module / type / procedure names as
ISO_C_BINDING wrappers over libntpoly_amd.so.  This script compiles it with flang, then compiles the reference's
shipped Fortran example Examples/PremadeMatrix/main.f90 -- FROM WHERE IT LIES, unchanged -- against those modules
and links it with libntpoly_amd.so (+ MPICH for MPI_Init; `USE MPI` is served by a three-line module that includes
the image's own mpif.h, because MPICH's mpi.mod was written by gfortran).  Output: oracle/_ref/premade_f90
(git-ignored, travels to the GPU box as a built file; tests/test_gpu_extras.py runs it there).

    python oracle/build_fortran_example.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "oracle", "_ref")
FLANG = "/opt/rocm/lib/llvm/bin/flang"
# executable name -> the reference's unchanged Fortran example driver
EXAMPLES = {"premade_f90": "PremadeMatrix", "hydrogen_f90": "HydrogenAtom", "graph_f90": "GraphTheory",
            "maps_f90": "MatrixMaps", "overlap_f90": "OverlapMatrix", "complex_f90": "ComplexMatrix"}


def build():
    if not os.path.isdir(REF):
        print("no /root/reference here: nothing to build")
        return None
    work = os.path.join(OUT, "f90_obj")
    os.makedirs(work, exist_ok=True)
    shim = os.path.join(work, "mpi_shim.f90")
    with open(shim, "w") as f:
        f.write("MODULE MPI\n  INCLUDE \"mpif.h\"\nEND MODULE MPI\n")
    common = []
    for name, src in (("mods", os.path.join(ROOT, "fortran", "ntpoly_amd_modules.f90")),
                      ("mods_more", os.path.join(ROOT, "fortran", "ntpoly_amd_modules_more.f90")), ("mpi_shim", shim)):
        obj = os.path.join(work, name + ".o")
        subprocess.run([FLANG, "-O1", "-cpp", "-c", src, "-o", obj, "-J", work, "-I", work, "-I/opt/conda/include"], check=True)
        common.append(obj)
    mpidir = os.path.join(OUT, "mpilib")
    os.makedirs(mpidir, exist_ok=True)
    for lib in ("libmpi.so.12", "libmpifort.so.12", "libgfortran.so.4", "libquadmath.so.0", "libgomp.so.1"):
        dst = os.path.join(mpidir, lib)
        if os.path.lexists(dst):
            os.unlink(dst)
        os.symlink(os.path.join("/opt/conda/lib", lib), dst)
    exe = None
    for name, example in EXAMPLES.items():
        obj = os.path.join(work, name + "_main.o")
        subprocess.run([FLANG, "-O1", "-cpp", "-c", "%s/Examples/%s/main.f90" % (REF, example), "-o", obj, "-J", work,
                        "-I", work, "-I/opt/conda/include"], check=True)
        exe = os.path.join(OUT, name)
        subprocess.run([FLANG, "-o", exe, obj] + common + ["-L" + os.path.join(ROOT, "ntpoly_amd"), "-lntpoly_amd",
                                                            "/opt/conda/lib/libmpifort.so", "/opt/conda/lib/libmpi.so",
                                                            "-Wl,-rpath,$ORIGIN/../../ntpoly_amd", "-Wl,-rpath,$ORIGIN/mpilib",
                                                            "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib/llvm/lib"],
                       check=True)
        print("built", exe)
    return exe


if __name__ == "__main__":
    sys.exit(0 if build() or not os.path.isdir(REF) else 1)
