#!/usr/bin/env python3
"""Build the REAL NTPoly reference (Fortran) from /root/reference into oracle/_ref/.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path uses this.

The reference sources are compiled *where they lie* (read-only) with the image's
flang (ROCm LLVM), the image's MPICH (/opt/conda: mpif.h + libmpi/libmpifort) and
the image's MKL (LAPACK/BLAS for the reference's dense branch).  No reference
source is copied into this repo; all outputs go to oracle/_ref/ (git-ignored).
We do not run the reference's CMake: module order is derived here from the
MODULE/USE statements.

Outputs:
  oracle/_ref/libNTPoly_ref.a     the reference library (Source/Fortran only)
  oracle/_ref/mod/*.mod           its Fortran modules
  oracle/_ref/ref_driver          our own driver (oracle/ref_driver.f90) linked to it
"""
import os, re, subprocess, sys, glob

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("NTPOLY_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "Source", "Fortran")
OUT = os.path.join(HERE, "_ref")
FLANG = os.environ.get("FLANG", "/opt/rocm/lib/llvm/bin/flang")
MPI_INC = os.environ.get("MPI_INC", "/opt/conda/include")
MPI_LIB = os.environ.get("MPI_LIB", "/opt/conda/lib")
FFLAGS = ["-O2", "-cpp", "-fopenmp", "-fPIC", "-DUSE_MPIH=1"]
LIBS = ["-L" + MPI_LIB, "-lmpifort", "-lmpi", "-lmkl_intel_lp64", "-lmkl_sequential",
        "-lmkl_core", "-lpthread", "-lm", "-ldl", "-Wl,-rpath," + MPI_LIB]


def available():
    return (os.path.isdir(SRC) and os.path.exists(FLANG)
            and os.path.exists(os.path.join(MPI_INC, "mpif.h")))


def scan(path):
    mods, uses = set(), set()
    for line in open(path, errors="replace"):
        m = re.match(r"\s*MODULE\s+(\w+)\s*$", line, re.I)
        if m and m.group(1).upper() != "PROCEDURE":
            mods.add(m.group(1).lower())
        m = re.match(r"\s*USE\s+(\w+)", line, re.I)
        if m:
            uses.add(m.group(1).lower())
    return mods, uses


def topo_order(files):
    info = {f: scan(f) for f in files}
    provider = {}
    for f, (mods, _) in info.items():
        for m in mods:
            provider[m] = f
    order, state = [], {}

    def visit(f):
        if state.get(f) == 2:
            return
        if state.get(f) == 1:
            raise RuntimeError("cycle at " + f)
        state[f] = 1
        for u in info[f][1]:
            p = provider.get(u)
            if p and p != f:
                visit(p)
        state[f] = 2
        order.append(f)

    for f in sorted(files):
        visit(f)
    return order


def run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout[-4000:] + "\n")
        raise SystemExit(1)
    return r.stdout


def build(force=False, fma=False):
    """fma=True: a second build under oracle/_ref/fma with floating-point contraction on (-ffp-contract=fast
    -march=haswell: `acc = acc + a*b` becomes one FMA), i.e. what the reference computes on targets where FMA is
    baseline (its Fugaku/aarch64 target) or with -march=native; pins the engine's spgemm_fma option."""
    global OUT, FFLAGS
    if not available():
        print("reference toolchain/sources not present: skipping oracle/_ref build")
        return False
    out0, flags0 = OUT, FFLAGS
    if fma:
        OUT = os.path.join(out0, "fma")
        FFLAGS = flags0 + ["-ffp-contract=fast", "-march=haswell"]
    try:
        return _build(force)
    finally:
        OUT, FFLAGS = out0, flags0


def _build(force):
    os.makedirs(os.path.join(OUT, "obj"), exist_ok=True)
    os.makedirs(os.path.join(OUT, "mod"), exist_ok=True)
    lib = os.path.join(OUT, "libNTPoly_ref.a")
    files = [f for f in glob.glob(os.path.join(SRC, "*.F90"))]
    if force or not os.path.exists(lib):
        objs = []
        for f in topo_order(files):
            o = os.path.join(OUT, "obj", os.path.basename(f)[:-4] + ".o")
            run([FLANG] + FFLAGS + ["-I" + MPI_INC, "-I" + SRC, "-module-dir",
                 os.path.join(OUT, "mod"), "-c", f, "-o", o])
            objs.append(o)
        if os.path.exists(lib):
            os.remove(lib)
        run(["ar", "rcs", lib] + objs)
    drv_src = os.path.join(HERE, "ref_driver.f90")
    drv = os.path.join(OUT, "ref_driver")
    if os.path.exists(drv_src) and (force or not os.path.exists(drv)
                                    or os.path.getmtime(drv) < os.path.getmtime(drv_src)):
        run([FLANG] + FFLAGS + ["-I" + MPI_INC, "-I" + os.path.join(OUT, "mod"),
             "-module-dir", os.path.join(OUT, "drvmod"), drv_src, lib, "-o", drv] + LIBS,
            env=dict(os.environ, TMPDIR="/tmp"))
    return True


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv, fma="--fma" in sys.argv)
    sys.exit(0 if ok else 2)
