"""Build the engine's shared library (HIP kernels + C++ host + C ABI) for gfx950 with hipcc.

    python -m ntpoly_amd._build        # -> ntpoly_amd/libntpoly_amd.so (in-tree, git-ignored)

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical contract
(DESIGN.md "Parity"): products and sums are rounded separately, as the reference does.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libntpoly_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["kernels.hip", "spgemm_tile.hip", "spgemm_tile_c.hip", "spgemm_thin.hip", "column_fused.hip", "spgemm_grouped.hip", "spgemm_block.hip", "relabel.hip", "slab_extra.hip", "dense.hip", "common.cpp", "comm.cpp", "psmatrix.cpp", "band_scope.cpp", "solvers.cpp", "solvers_poly.cpp", "solvers_func.cpp", "solvers_extra.cpp", "io.cpp", "wrp.cpp"]
# experiments kept out of the product build (csrc/experiments/): NTPOLY_AMD_WITH_TILE2=1 adds the two-block geometry of the
# MFMA kernel (measured slower than k_spgemm_tile in round 5, profiles/README.md 83; option tile2 is a no-op without it)
WITH_TILE2 = os.environ.get("NTPOLY_AMD_WITH_TILE2", "0") == "1"
if WITH_TILE2:
    SOURCES.insert(2, "experiments/spgemm_tile2.hip")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result"] + (["-DNTP_WITH_TILE2"] if WITH_TILE2 else []) + os.environ.get("NTPOLY_AMD_EXTRA_FLAGS", "").split()


# per-file flags.  spgemm_block.hip: its matrix instructions sit behind wave-uniform branches; with the accumulators in
# the AGPR half of the register file the compiler copies them to VGPRs and back around every branch target (with the
# 18 wait states a read of a fresh matrix result needs) -- in VGPR form they stay where they are
FILE_FLAGS = {"spgemm_block.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in os.listdir(CSRC):
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def _depfile_deps(depfile):
    """prerequisites listed in a -MMD dependency file (own sources only: system headers are not listed)"""
    try:
        text = open(depfile).read()
    except OSError:
        return None
    text = text.replace("\\\n", " ")
    if ":" not in text:
        return None
    return [t for t in text.split(":", 1)[1].split() if t]


def object_is_stale(obj, src):
    """True when `obj` must be recompiled: it is missing, or older than its source or than ANY file the source
    includes -- the compiler's own list (-MMD, kept beside the object) when there is one, otherwise every
    non-source file in csrc (*.hpp and the generated *.inc loops)."""
    if not os.path.exists(obj):
        return True
    deps = _depfile_deps(obj[:-2] + ".d")
    if deps is None or any(not os.path.exists(d) for d in deps):
        deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.endswith((".cpp", ".hip"))]
    deps = list(deps) + [src]
    return os.path.getmtime(obj) <= max(map(os.path.getmtime, deps))


ASAN_LIB = os.path.join(HERE, "libntpoly_amd_asan.so")
ASAN_FLAGS = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]


def build_sanitized(force=False, verbose=False):
    """The same library with AddressSanitizer + UndefinedBehaviorSanitizer on the HOST side (the analogue of the
    reference's -fcheck=all debug leg, Targets/Linux.cmake:20-22): ntpoly_amd/libntpoly_amd_asan.so.  Device code is
    compiled as usual (GPU sanitizers are not available on this pool); tests/test_abi_cpu.py drives the host-only
    entry points of this build under the sanitizer runtime."""
    return build(force=force, verbose=verbose, lib=ASAN_LIB, objdir=os.path.join(HERE, "build_asan"),
                 flags=[f for f in FLAGS if f != "-O3"] + ASAN_FLAGS, link_extra=["-fsanitize=address,undefined"])


def build(force=False, verbose=False, lib=None, objdir=None, flags=None, link_extra=()):
    custom = lib is not None
    lib = lib or LIB
    flags = flags or FLAGS
    if not force and not custom and not _stale():
        return LIB
    objdir = objdir or os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, os.path.basename(s).rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if not force and not object_is_stale(obj, src):
            continue
        cmd = [HIPCC] + flags + FILE_FLAGS.get(s, []) + ["-MMD", "-MF", obj[:-2] + ".d", "-x", "hip", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write("---- %s\n%s\n" % (s, out))
            failed = True
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    if custom and not procs and os.path.exists(lib) and os.path.getmtime(lib) >= max(map(os.path.getmtime, objs)):
        return lib
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + list(link_extra) + [
        "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("link failed")
    return lib


FLANG = os.environ.get("FLANG", "/opt/rocm/lib/llvm/bin/flang")
FORTRAN_SRC = os.path.join(os.path.dirname(HERE), "fortran", "ntpoly_amd_modules.f90")
FORTRAN_SRC2 = os.path.join(os.path.dirname(HERE), "fortran", "ntpoly_amd_modules_more.f90")  # tools/gen_fortran_more.py
FORTRAN_LIB = os.path.join(HERE, "libntpoly_amd_fortran.a")
FORTRAN_MOD = os.path.join(HERE, "fortran_mod")


def build_fortran(force=False):
    """The Fortran module layer (NTPoly's module / type / procedure names over the C ABI): flang ->
    ntpoly_amd/libntpoly_amd_fortran.a + ntpoly_amd/fortran_mod/*.mod.  A Fortran program written against NTPoly
    compiles with `-I ntpoly_amd/fortran_mod` and links `libntpoly_amd_fortran.a -lntpoly_amd`."""
    if not os.path.exists(FLANG):
        return None
    srcs = [FORTRAN_SRC, FORTRAN_SRC2]
    if not force and os.path.exists(FORTRAN_LIB) and os.path.getmtime(FORTRAN_LIB) > max(map(os.path.getmtime, srcs)):
        return FORTRAN_LIB
    os.makedirs(FORTRAN_MOD, exist_ok=True)
    objs = []
    for src in srcs:   # in order: part 2 uses the modules of part 1
        obj = os.path.join(HERE, "build", os.path.basename(src)[:-4] + ".o")
        os.makedirs(os.path.dirname(obj), exist_ok=True)
        r = subprocess.run([FLANG, "-O2", "-fPIC", "-c", src, "-o", obj, "-J", FORTRAN_MOD, "-I", FORTRAN_MOD],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout)
            raise RuntimeError("flang failed")
        objs.append(obj)
    if os.path.exists(FORTRAN_LIB):
        os.unlink(FORTRAN_LIB)
    subprocess.run(["ar", "rcs", FORTRAN_LIB] + objs, check=True)
    return FORTRAN_LIB


if __name__ == "__main__":
    if "--asan" in sys.argv:
        print(build_sanitized(force="--force" in sys.argv, verbose=True))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_fortran(force="--force" in sys.argv))
