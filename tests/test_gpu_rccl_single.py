"""GPU: run the distributed code paths (RCCL communicator, panel all-gather with size exchange +
grouped broadcasts + offset fix-up, scalar all-reduces, gathered transpose / permutation / triplet
fill) on ONE GPU by forcing a 1-rank RCCL communicator (NTPOLY_AMD_FORCE_RCCL=1).  The results must
still match the reference's golden vectors bit for bit.  Multi-GPU runs are the driver's; this is
the closest exercise of that code a single-GPU box allows."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forced_rccl_paths_match_golden():
    env = dict(os.environ, NTPOLY_AMD_FORCE_RCCL="1")
    sel = "test_ps_gemm_golden and -1--1 or test_ps_scalars_golden or test_solvers_golden or test_premade_fixture or " \
          "test_load_balanced_solver_matches or test_ps_increment_golden or test_trs2_fused_steps_match"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-x", "-q",
                        "-m", "gpu", "-k", sel], env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "passed" in r.stdout


def test_forced_rccl_overlapped_halo_matches_golden():
    """the same with every distributed multiply split into interior / boundary columns and its send / recv group
    enqueued on the communication stream (NTPOLY_AMD_HALO_OVERLAP=3): event ordering between the two streams and the
    RCCL group on a second stream run for real, results still bit-identical to the reference's."""
    env = dict(os.environ, NTPOLY_AMD_FORCE_RCCL="1", NTPOLY_AMD_HALO_OVERLAP="3")
    sel = "test_ps_gemm_golden and -1--1 or test_solvers_golden or test_premade_fixture"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-x", "-q",
                        "-m", "gpu", "-k", sel], env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "passed" in r.stdout


def test_forced_rccl_split_communicator():
    """ncclCommSplit for real: the process grid of a 1-rank RCCL communicator is split (SplitProcessGrid,
    ProcessGridModule.F90:430-515; csrc/comm.cpp comm_split) and a matrix hosted on the new grid is multiplied and reduced --
    every collective of it an RCCL call on the SPLIT communicator -- with the results of the grid of all processes."""
    code = r'''
import ctypes as C, numpy as np, sys
sys.path.insert(0, "tests")
import ntpoly_amd as nt
from ntpoly_amd.capi import handle, i
from gen import banded_triplets
nt.init_comm()
nt.ConstructGlobalProcessGrid(1, 1, 1)
n = 3000
A = nt.Matrix_ps.from_triplets(n, *banded_triplets(n, 20))
g, g2 = handle(), handle()
nt.lib.GetMatrixProcessGrid_ps_wrp(A.ih, g)
color, ss = C.c_int(-1), C.c_bool(False)
nt.lib.ntpoly_amd_split_process_grid(g, g2, C.byref(color), C.byref(ss))
info = (C.c_int * 3)()
nt.lib.ntpoly_amd_grid_comm_info(g2, info)
assert (color.value, info[0], info[1], info[2]) == (0, 0, 1, 1), (color.value, list(info))
S = nt.Matrix_ps.__new__(nt.Matrix_ps)
S.ih = handle()
nt.lib.ConstructEmptyMatrixPG_ps_wrp(S.ih, i(n), g2)
t = nt.TripletList_r()
t.set_arrays(*banded_triplets(n, 20))
S.FillFromTripletList(t)
assert S.grid_comm_info() == (0, 1, True)
C1, C2 = nt.Matrix_ps(n), nt.Matrix_ps(n)
C1.Gemm(A, A, None, 1.0, 0.0, 1e-9)
C2.Gemm(S, S, None, 1.0, 0.0, 1e-9)
assert C2.grid_comm_info()[2] is True and C1.grid_comm_info()[2] is False
assert all(np.array_equal(x, y) for x, y in zip(C1.triplets(), C2.triplets()))
assert C1.Trace() == C2.Trace() and C1.Norm() == C2.Norm()
print("split ok")
'''
    env = dict(os.environ, NTPOLY_AMD_FORCE_RCCL="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and "split ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
