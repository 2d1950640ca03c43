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
