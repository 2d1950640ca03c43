#!/usr/bin/env python3
"""Time per iteration of TRS4 / SignFunction on the headline operand (N = 262 144, h = 100, threshold 1e-8) by
differencing solves capped at 4 and 14 iterations; host synchronisations per solve are printed as well.
    SOLVER=trs4|sign|isq|pm|hpcp ARITH=fma|unfused [CPLX=1] NTPOLY_AMD_SLAB_ALGEBRA=0|1 python3 tools/solver_iterations.py
Under rocprofv3 --kernel-trace --stats this gives profiles/r03_trs4_kernel_stats_*.csv (tools/prof_summary.py)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import ntpoly_amd as nt
from gen import banded_triplets
nt.init_comm(); nt.ConstructGlobalProcessGrid(1, 1, 1)
nt.set_option("spgemm_fma", 1 if os.environ.get("ARITH", "fma") == "fma" else 0)
n, h, thr = 262144, 100, 1e-8
if os.environ.get("LATTICE"):   # LATTICE=64: the 64^3 lattice Hamiltonian (block path; NTPOLY_AMD_SLAB_ALGEBRA also switches the block algebra)
    from gen import lattice_triplets
    L = int(os.environ["LATTICE"])
    n = L ** 3
    col, row, val = lattice_triplets(L)
elif os.environ.get("CPLX"):    # CPLX=1: the configs[4] operand (Hermitian complex, N = 131 072, h = 50); SOLVER=sign (H) | isq (H + 2 I)
    n, h = 131072, 50
    col, row, val = banded_triplets(n, h, complex_=True, shift=2.0 if os.environ.get("SOLVER") == "isq" else 0.0)
else:
    col, row, val = banded_triplets(n, h, shift=2.0 if os.environ.get("SOLVER") == "isq" else 0.0)   # (isq: H + 2 I, positive definite)
H = nt.Matrix_ps.from_triplets(n, col, row, val)
I = nt.Matrix_ps(n); I.FillIdentity()
which = os.environ.get("SOLVER", "trs4")
for iters in (4, 14, 4, 14):
    K = nt.Matrix_ps(n)
    p = nt.SolverParameters(); p.SetThreshold(thr); p.SetConvergeDiff(1e-30); p.SetMaxIterations(iters); p.SetMonitorConvergence(False)
    nt.synchronize(); s0 = nt.exchange_stats()[2]; t0 = time.perf_counter()
    if which == "trs4":
        e, _ = nt.DensityMatrixSolvers.TRS4(H, I, n / 2.0, K, p)
    elif which == "sign":
        nt.SignSolvers.ComputeSign(H, K, p); e = 0
    elif which == "isq":
        nt.SquareRootSolvers.InverseSquareRoot(H, K, p); e = 0
    elif which == "pm":
        e, _ = nt.DensityMatrixSolvers.PM(H, I, n / 2.0, K, p)
    elif which == "hpcp":
        e, _ = nt.DensityMatrixSolvers.HPCP(H, I, n / 2.0, K, p)
    nt.synchronize()
    walls = globals().setdefault("walls", {})
    walls[iters] = time.perf_counter() - t0
    print(which, "iters", iters, "wall", time.perf_counter() - t0, "syncs", nt.exchange_stats()[2] - s0, "e", e, "nnz", K.GetSize(), flush=True)
    del K
print("%s: %.2f ms per iteration (arithmetic %s, slab_algebra %s, lattice %s, block algebra operations %s)" % (which, 1e3 * (walls[14] - walls[4]) / 10,
      os.environ.get("ARITH", "fma"), os.environ.get("NTPOLY_AMD_SLAB_ALGEBRA", "1"), os.environ.get("LATTICE"), nt.block_algebra_counts()))
