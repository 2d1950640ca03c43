#!/usr/bin/env python3
"""Wall time of whole converged TRS2 solves through TRS2_wrp at the headline size (N = 262 144, h = 100, threshold 1e-8,
convergence 1e-6): the banded operand, and the same band under a random symmetric relabelling -- first solve (the
bandwidth-reducing order is searched), then the next cycles of a self-consistent-field loop: a NEW matrix with the same
sparsity pattern and other values (the order is reused: RelabelCache::fingerprint), and the relabelled operand on the
grouped-hash path (label_order = 0).  Table: profiles/r03_solve_time.txt, DESIGN.md section 7c.
    python3 tools/solve_time.py [fma|unfused]"""
import sys, time, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from gen import banded_triplets, permuted_banded_triplets
import ntpoly_amd as nt
nt.init_comm(); nt.ConstructGlobalProcessGrid(1, 1, 1)
arith = sys.argv[1] if len(sys.argv) > 1 else "fma"
nt.set_option("spgemm_fma", 1 if arith == "fma" else 0)
n, h, thr = 262144, 100, 1e-8
print("arithmetic", arith)
for tag, gen in (("banded", lambda: banded_triplets(n, h)), ("relabelled", lambda: permuted_banded_triplets(n, h, 42))):
    col, row, val = gen()
    ISQ = nt.Matrix_ps(n); ISQ.FillIdentity()
    for lo in ((1, 0) if tag == "relabelled" else (1,)):
        nt.set_option("label_order", lo)
        for cycle in range(3 if lo else 1):
            v = val + 0.01 * cycle * (col == row)      # (the next SCF cycle: same pattern, other values, a new matrix)
            H = nt.Matrix_ps.from_triplets(n, col, row, v)
            K = nt.Matrix_ps(n)
            p = nt.SolverParameters(); p.SetThreshold(thr); p.SetConvergeDiff(1e-6)
            s0 = nt.band_searches()
            nt.synchronize(); t0 = time.time()
            e, mu = nt.DensityMatrixSolvers.TRS2(H, ISQ, n / 2.0, K, p)
            nt.synchronize(); dt = time.time() - t0
            tr = nt.solver_trace()
            print("%-10s label_order %d cycle %d: %.3f s, %d iterations (%.2f ms each), energy %.9f, band searches %d" % (
                tag, lo, cycle, dt, tr["iterations"], 1e3 * dt / tr["iterations"], e, nt.band_searches() - s0), flush=True)
            del H, K
nt.set_option("label_order", 1)
