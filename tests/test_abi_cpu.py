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
