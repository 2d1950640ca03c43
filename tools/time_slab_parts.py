import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ntpoly_amd as nt
from gen import banded_triplets
from oracle import oracle_py as O
nt.init_comm(); nt.ConstructGlobalProcessGrid(1, 1, 1)
nt.set_option("spgemm_fma", 1); O.set_fma(True)
for solver, n, h, thr, shift in (("sign", 2560, 20, 1e-8, 0.0), ("sign", 2048, 8, 1e-7, 0.3), ("inverse_square_root", 2560, 20, 1e-8, 2.0)):
    col, row, val = banded_triplets(n, h, shift=shift)
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    for sa in (0, 1):
        nt.set_option("slab_algebra", sa)
        p = nt.SolverParameters(); p.SetThreshold(thr); p.SetConvergeDiff(1e-8)
        Out = nt.Matrix_ps(n)
        nt.synchronize(); t0 = time.perf_counter()
        if solver == "sign": nt.SignSolvers.ComputeSign(H, Out, p)
        else: nt.SquareRootSolvers.InverseSquareRoot(H, Out, p)
        nt.synchronize(); t1 = time.perf_counter()
        tr = nt.solver_trace()
        print(solver, n, "slab_algebra", sa, "gpu solve %.2f s" % (t1 - t0), "iterations", tr["iterations"], "nnz out", len(Out.triplets()[2]))
    Ho = O.Mat.from_triplets(n, n, col, row, val)
    t0 = time.perf_counter()
    Oo, tro = O.matrix_function(solver, Ho, O.params(converge_diff=1e-8, threshold=thr))
    print(solver, n, "oracle %.2f s" % (time.perf_counter() - t0), "iterations", tro["iterations"])
