"""CPU: the C-ABI library builds, loads without a GPU and exports every symbol that include/*.h
declares; the host-only entry points (triplet lists, permutations, solver parameters, panel
ranges) behave like the reference's.  No compute call is made (there is no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nt():
    from ntpoly_amd import _build
    _build.build()
    import ntpoly_amd
    return ntpoly_amd


def test_every_declared_symbol_is_exported(nt):
    names = nt.capi.exported_symbols()
    assert len(names) >= 160
    missing = [n for n in names if not hasattr(nt.lib, n)]
    assert not missing, missing


def test_headers_cite_the_reference_interface():
    inc = os.path.join(ROOT, "include")
    n_decl = n_cite = 0
    for f in os.listdir(inc):
        text = open(os.path.join(inc, f)).read()
        n_decl += len(re.findall(r"^\w[\w ]*\s\w+_wrp\s*\(", text, re.M))
        n_cite += len(re.findall(r"replaces Source/C/\w+\.h:\d+ \(wrapper Source/Wrapper/\w+\.F90:\d+\)", text))
    assert n_decl >= 130 and n_cite == n_decl


def test_headers_compile_as_c():
    src = '#include "ntpoly_amd.h"\nint main(void){int ih[NTPOLY_AMD_SIZE_WRP]; (void)ih; return 0;}\n'
    r = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c", "-"],
                       input=src, text=True, capture_output=True)
    assert r.returncode == 0, r.stderr


def test_triplet_list_semantics(nt):
    """TripletList_c.h: append / get (1-based at the ABI, 0-based in the class like TripletList.cc)"""
    t = nt.TripletList_r()
    t.Append(3, 1, 2.5)
    t.Append(1, 2, -1.0)
    assert t.GetSize() == 2
    assert tuple(t.GetTripletAt(0)) == (3, 1, 2.5)
    assert tuple(t.GetTripletAt(1)) == (1, 2, -1.0)
    srt = nt.capi.handle()
    nt.lib.SortTripletList_r_wrp(t.ih, nt.capi.i(3), srt)
    col, row, val = C.c_int(), C.c_int(), C.c_double()
    nt.lib.GetTripletAt_r_wrp(srt, nt.capi.i(1), C.byref(col), C.byref(row), C.byref(val))
    assert (col.value, row.value, val.value) == (1, 2, -1.0)
    nt.lib.DestructTripletList_r_wrp(srt)
    c = nt.TripletList_c(2)
    assert c.GetSize() == 2
    c.set_arrays([1, 2], [2, 1], np.array([1 + 2j, 3 - 4j]))
    assert tuple(c.GetTripletAt(1)) == (2, 1, 3 - 4j)
    # the SWIG classes' way: a triplet object in, a triplet object out
    tr = nt.Triplet_r()
    tr.index_column, tr.index_row, tr.point_value = 4, 5, 0.25
    t.Append(tr)
    back = t.GetTripletAt(2)
    assert (back.index_column, back.index_row, back.point_value) == (4, 5, 0.25)


def test_panel_ranges_cover_the_matrix(nt):
    for dim in (7, 33, 262144, 1048576 + 3):
        for nranks in (1, 2, 3, 8):
            prev = 0
            for r in range(nranks):
                a, b = C.c_int(), C.c_int()
                nt.lib.ntpoly_amd_panel_range(nt.capi.i(dim), nt.capi.i(nranks), nt.capi.i(r), C.byref(a), C.byref(b))
                assert a.value == prev and b.value >= a.value
                prev = b.value
            assert prev == dim


def test_parameters_and_permutation_handles(nt):
    p = nt.SolverParameters()
    p.SetThreshold(1e-6)
    p.SetConvergeDiff(1e-4)
    p.SetMaxIterations(17)
    p.SetVerbosity(False)
    p.SetMonitorConvergence(False)
    perm = nt.Permutation(10)
    perm.SetReversePermutation()
    p.SetLoadBalance(perm)
    del p, perm


def test_product_never_touches_the_oracle():
    """the shipped package must not import, link or call anything under oracle/"""
    pkg = os.path.join(ROOT, "ntpoly_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower() or f == "capi.py", (f, "mentions the oracle")
    out = subprocess.run(["ldd", os.path.join(pkg, "libntpoly_amd.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_fortran_module_layer_builds():
    """fortran/ntpoly_amd_modules.f90 (NTPoly's Fortran module names over the C ABI) compiles with flang and every
    BIND(C) name it declares is exported by the library."""
    import re
    import subprocess
    from ntpoly_amd import _build
    if not os.path.exists(_build.FLANG):
        pytest.skip("no flang in this image")
    assert _build.build_fortran() and os.path.exists(os.path.join(_build.FORTRAN_MOD, "psmatrixmodule.mod"))
    src = open(_build.FORTRAN_SRC).read()
    names = set(re.findall(r'BIND\(C, name="(\w+)"\)', src))
    syms = subprocess.run(["nm", "-D", "--defined-only", _build.LIB], capture_output=True, text=True).stdout
    have = set(ln.split()[-1] for ln in syms.splitlines() if ln.strip())
    assert len(names) > 70 and not sorted(names - have)


def test_multi_process_launch_without_communicator_is_refused():
    """P processes launched together (a process manager's environment says so) that construct the process
    grid without having given the engine a communicator -- no MPI initialised in the process, no ntpoly_amd_init_comm --
    must not silently run P identical single-rank solves: the grid constructor aborts with a message before any GPU call."""
    import subprocess
    import sys
    code = ("import ntpoly_amd as nt\n"
            "nt.ConstructGlobalProcessGrid()\n"
            "print('constructed')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PMI_SIZE="2", PMI_RANK="0", PYTHONPATH=root)
    env.pop("NTPOLY_AMD_ALLOW_REPLICAS", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode != 0 and "constructed" not in r.stdout
    assert "no communicator was given to the engine" in r.stdout
    # the same program alone (no launcher environment) is a legitimate single-rank run up to the point where it needs a GPU
    env2 = {k: v for k, v in env.items() if not k.startswith("PMI_")}
    r2 = subprocess.run([sys.executable, "-c", code], env=env2, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert "constructed" in r2.stdout
    # a SERIAL process inside a multi-task allocation (SLURM_NTASKS, a stray WORLD_SIZE) is not a rank of a
    # launch -- only a rank variable together with the size of its step says so
    env3 = dict(env2, SLURM_NTASKS="8", WORLD_SIZE="4")
    for k in ("RANK", "SLURM_PROCID", "PMIX_RANK", "OMPI_COMM_WORLD_RANK"):
        env3.pop(k, None)
    r3 = subprocess.run([sys.executable, "-c", code], env=env3, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert "constructed" in r3.stdout, r3.stdout
    env4 = dict(env3, SLURM_PROCID="1", SLURM_STEP_NUM_TASKS="8")
    r4 = subprocess.run([sys.executable, "-c", code], env=env4, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r4.returncode != 0 and "no communicator was given to the engine" in r4.stdout


def test_host_side_under_address_and_ub_sanitizers(tmp_path):
    """The host side of the library built with -fsanitize=address,undefined (ntpoly_amd._build.build_sanitized, the
    analogue of the reference's -fcheck=all leg, Targets/Linux.cmake:20-22) and driven through the host-only part of the
    C ABI: triplet lists (append / resize / set / get / sort / symmetrize through the Python mirror), permutations,
    solver parameters, the logger, the single-rank grid, and the fatal path of a multi-process launch without a
    communicator.  Any sanitizer report fails the test.  (GPU sanitizers are not available on this pool.)"""
    import glob
    import subprocess
    import sys
    from ntpoly_amd import _build
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rt:
        pytest.skip("no AddressSanitizer runtime in this image")
    lib = _build.build_sanitized()
    code = r'''
import numpy as np
import ntpoly_amd as nt
from ntpoly_amd.capi import lib
assert "asan" in nt.capi.LIB_PATH
tl = nt.TripletList_r()
rng = np.random.default_rng(1)
for k in range(300):
    tl.Append(int(rng.integers(1, 40)), int(rng.integers(1, 40)), float(rng.normal()))
assert tl.GetSize() == 300
c, r, v = tl.arrays()
tl2 = nt.TripletList_r(10)
tl2.set_arrays(c, r, v)
c2, r2, v2 = tl2.arrays()
assert np.array_equal(c, c2) and np.array_equal(v, v2)
t = tl.GetTripletAt(7)
tc = nt.TripletList_c()
tc.Append(3, 4, 1.5 - 2j)
assert tc.GetSize() == 1
for n in (1, 7, 64):
    p = nt.Permutation(n)
    p.SetReversePermutation()
    p.SetDefaultPermutation()
    p.set_lookup(np.arange(n, 0, -1))
sp = nt.SolverParameters()
sp.SetConvergeDiff(1e-7); sp.SetMaxIterations(11); sp.SetVerbosity(False); sp.SetThreshold(1e-9)
sp.SetStepThreshold(1e-3); sp.SetMonitorConvergence(True)
pp = nt.Permutation(12)
sp.SetLoadBalance(pp)
nt.init_comm()
nt.ConstructGlobalProcessGrid(1, 1, 1)
assert nt.GetGlobalIsRoot()
nt.ActivateLogger(True, r"%s")
nt.WriteGridInfo()
nt.DeactivateLogger()
print("HOST-OK")
''' % str(tmp_path / "log.yaml")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LD_PRELOAD=rt[-1], NTPOLY_AMD_LIB=lib, PYTHONPATH=root,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=600)
    assert "HOST-OK" in r.stdout, r.stdout[-4000:]
    assert "AddressSanitizer" not in r.stdout and "runtime error" not in r.stdout, r.stdout[-4000:]
    assert r.returncode == 0, r.stdout[-2000:]


def test_default_arithmetic_is_the_benched_one():
    """The arithmetic a drop-in caller gets without asking is the FMA chain that bench.py times:
    option spgemm_fma defaults to 1, NTPOLY_AMD_ARITHMETIC=unfused|fma selects the mode from the environment of an
    unmodified program.  Fresh processes: the test session itself pins the unfused mode (tests/conftest.py)."""
    import subprocess
    import sys
    code = "import ntpoly_amd as nt; print(nt.get_option('spgemm_fma'))"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for env_val, want in ((None, "1"), ("unfused", "0"), ("fma", "1")):
        env = dict(os.environ)
        env.pop("NTPOLY_AMD_SPGEMM_FMA", None)
        env.pop("NTPOLY_AMD_ARITHMETIC", None)
        if env_val:
            env["NTPOLY_AMD_ARITHMETIC"] = env_val
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
        assert r.returncode == 0, r.stdout
        assert r.stdout.strip().splitlines()[-1] == want, (env_val, r.stdout)


def test_round5_options_defaults_and_round_trip():
    """The options added in round 5 through the C ABI's option entry points (host only, no GPU): their defaults -- what a
    drop-in caller gets -- and a set / get round trip; the environment selectors of the two that have one."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    defaults = {"plan_fused": 1, "panel_sessions": 1, "block_scope": 1, "band_scope": 1, "exchange_ahead": 1, "block_unfused": 0,
                "block_match": 0, "tile2": 0,
                "tile_off32": 1, "tile_bbuf": 2, "ghash_mfma": 1}   # (round 6)
    code = ("import ntpoly_amd as nt\n"
            "names = %r\n"
            "print(' '.join(str(nt.get_option(k)) for k in names))\n"
            "for k in names: nt.set_option(k, 7)\n"
            "print(' '.join(str(nt.get_option(k)) for k in names))\n") % (sorted(defaults),)
    env = dict(os.environ)
    for k in ("NTPOLY_AMD_PANEL_SESSIONS", "NTPOLY_AMD_BLOCK_UNFUSED"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout
    lines = r.stdout.strip().splitlines()[-2:]
    assert lines[0].split() == [str(defaults[k]) for k in sorted(defaults)], r.stdout
    assert lines[1].split() == ["7"] * len(defaults), r.stdout
    env["NTPOLY_AMD_PANEL_SESSIONS"] = "0"
    env["NTPOLY_AMD_BLOCK_UNFUSED"] = "1"
    r = subprocess.run([sys.executable, "-c", "import ntpoly_amd as nt; print(nt.get_option('panel_sessions'), nt.get_option('block_unfused'))"],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].split() == ["0", "1"], r.stdout
