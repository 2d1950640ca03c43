"""Wall time of whole converged TRS2 solves through TRS2_wrp at the headline size: banded, relabelled (label-ordered slab
steps) and relabelled on the grouped-hash path (label_order = 0); numbers in DESIGN.md section 7b / profiles/README.md 35."""
import sys, time, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from gen import banded_triplets, permuted_banded_triplets
import ntpoly_amd as nt
nt.init_comm(); nt.ConstructGlobalProcessGrid(1, 1, 1)
n, h, thr = 262144, 100, 1e-8
for tag, gen in (("banded", lambda: banded_triplets(n, h)), ("relabelled", lambda: permuted_banded_triplets(n, h, 42))):
    col, row, val = gen()
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    del col, row, val
    ISQ = nt.Matrix_ps(n); ISQ.FillIdentity()
    for lo in ((1, 0) if tag == "relabelled" else (1,)):
        nt.set_option("label_order", lo)
        for rep in range(2):
            K = nt.Matrix_ps(n)
            p = nt.SolverParameters(); p.SetThreshold(thr); p.SetConvergeDiff(1e-6)
            nt.synchronize(); t0 = time.time()
            e, mu = nt.DensityMatrixSolvers.TRS2(H, ISQ, n / 2.0, K, p)
            nt.synchronize(); t1 = time.time()
            tr = nt.solver_trace()
            print(tag, "label_order", lo, "rep", rep, "iterations", tr["iterations"], "solve %.3f s" % (t1 - t0), "setup %.1f ms loop %.1f ms" % (tr["setup_ms"], tr["loop_ms"]), "energy %.10f" % e, "nnz", K.GetSize(), flush=True)
    nt.set_option("label_order", 1)
