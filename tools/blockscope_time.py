#!/usr/bin/env python3
"""TRS2 on the L^3 lattice through the solver entry point; time per iteration by differencing solves of 4 and 12 iterations.
With NTPOLY_AMD_FORCE_RCCL=1 (a 1-rank RCCL communicator: every collective a real RCCL call) the solve takes the several-rank
path -- block scope (band_scope.cpp) with its panel products on the block path; BLOCK_SCOPE=0: the path it replaces.
    [NTPOLY_AMD_FORCE_RCCL=1] [BLOCK_SCOPE=0] LATTICE=64 python3 tools/blockscope_time.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import ntpoly_amd as nt
from gen import lattice_triplets
nt.init_comm(); nt.ConstructGlobalProcessGrid(1, 1, 1)
nt.set_option("block_scope", int(os.environ.get("BLOCK_SCOPE", "1")))
L = int(os.environ.get("LATTICE", "48")); n = L ** 3
H = nt.Matrix_ps.from_triplets(n, *lattice_triplets(L))
I = nt.Matrix_ps(n); I.FillIdentity()
walls = {}
for iters in (4, 12, 4, 12):
    K = nt.Matrix_ps(n)
    p = nt.SolverParameters(); p.SetThreshold(1e-8); p.SetConvergeDiff(1e-30); p.SetMaxIterations(iters); p.SetMonitorConvergence(False)
    nt.synchronize(); t0 = time.perf_counter()
    e, mu = nt.DensityMatrixSolvers.TRS2(H, I, n / 2.0, K, p)
    nt.synchronize(); walls[iters] = time.perf_counter() - t0
    print("iters", iters, "wall", walls[iters], "e", e, "scope", nt.block_scope_counts(), "block", nt.last_block_stats().get("used"), flush=True)
print("ms per iteration: %.2f" % (1e3 * (walls[12] - walls[4]) / 8))
