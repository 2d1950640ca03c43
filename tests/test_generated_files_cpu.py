"""CPU: the generated sources in the tree are what their generators produce (no hand edits, no stale output):
csrc/slab_loop.inc (tools/gen_slab_asm.py), fortran/ntpoly_amd_modules_more.f90 (tools/gen_fortran_more.py) and, where
/root/reference is present (the generator reads the reference's C headers for the citations), include/*.h
(tools/gen_headers.py)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _regenerates_identically(tool, outputs, tmp_path):
    saved, times = {}, {}
    for rel in outputs:
        p = os.path.join(ROOT, rel)
        saved[rel] = open(p, "rb").read()
        st = os.stat(p)
        times[rel] = (st.st_atime, st.st_mtime)
    try:
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)], check=True, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        for rel, old in saved.items():
            new = open(os.path.join(ROOT, rel), "rb").read()
            assert new == old, "%s is not what tools/%s generates" % (rel, tool)
    finally:
        for rel, old in saved.items():   # leave the tree as it was (also on failure), time stamps included
            with open(os.path.join(ROOT, rel), "wb") as f:
                f.write(old)
            os.utime(os.path.join(ROOT, rel), times[rel])


def test_slab_loop_is_generated(tmp_path):
    _regenerates_identically("gen_slab_asm.py", ["ntpoly_amd/csrc/slab_loop.inc"], tmp_path)


def test_fortran_part2_is_generated(tmp_path):
    _regenerates_identically("gen_fortran_more.py", ["fortran/ntpoly_amd_modules_more.f90"], tmp_path)


def test_headers_are_generated(tmp_path):
    if not os.path.isdir("/root/reference/Source/C"):
        pytest.skip("tools/gen_headers.py cites the reference's C headers; /root/reference is not here")
    inc = os.path.join(ROOT, "include")
    outputs = [os.path.join("include", f) for f in sorted(os.listdir(inc)) if f.endswith(".h")]
    _regenerates_identically("gen_headers.py", outputs, tmp_path)


def test_incremental_build_sees_included_files(tmp_path):
    """an object must be rebuilt when a file its source #includes changes -- in particular the generated
    slab loops (csrc/slab_loop.inc, included by kernels.hip), which are neither a source nor a header."""
    import time
    from ntpoly_amd import _build
    src = tmp_path / "k.hip"
    inc = tmp_path / "loop.inc"
    obj = tmp_path / "k.o"
    dep = tmp_path / "k.d"
    for f in (src, inc):
        f.write_text("// x\n")
    obj.write_text("o")
    dep.write_text("%s: %s \\\n  %s\n" % (obj, src, inc))
    now = time.time()
    os.utime(src, (now - 100, now - 100))
    os.utime(inc, (now - 100, now - 100))
    os.utime(obj, (now - 50, now - 50))
    assert not _build.object_is_stale(str(obj), str(src))
    os.utime(inc, (now, now))                       # the included file is regenerated
    assert _build.object_is_stale(str(obj), str(src))
    # without a dependency file every non-source file of csrc counts (so slab_loop.inc does)
    dep.unlink()
    real_inc = os.path.join(_build.CSRC, "slab_loop.inc")
    assert os.path.exists(real_inc)
    os.utime(obj, (0, 0))
    assert _build.object_is_stale(str(obj), str(src))
    # the real build tree: the compiler's own list for kernels.o names the generated loop file
    d = os.path.join(_build.HERE, "build", "kernels.d")
    if os.path.exists(d):
        assert any(p.endswith("slab_loop.inc") for p in _build._depfile_deps(d))
