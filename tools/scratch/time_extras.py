import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ntpoly_amd as nt
from golden_util import Golden
nt.init_comm(); nt.ConstructGlobalProcessGrid(1, 1, 1)
g = Golden("extras")
def pm(name):
    t = g.tri(None, "M_" + name); return nt.Matrix_ps.from_triplets(t[0], t[2], t[3], t[4])
Hs, Hp, D, I, ISQ = pm("Hs"), pm("Hp"), pm("D"), pm("I"), pm("ISQ")
n = 96
p = nt.SolverParameters(); p.SetThreshold(1e-10); p.SetConvergeDiff(1e-8)
def timed(label, f):
    t0 = time.time(); r = f(); nt.synchronize(); print("%-12s %8.3f s" % (label, time.time() - t0), flush=True); return r
O = nt.Matrix_ps(n); O2 = nt.Matrix_ps(n)
timed("cg", lambda: nt.LinearSolvers.CGSolver(Hp, O, Hs, p))
timed("pade", lambda: nt.ExponentialSolvers.ComputeExponentialPade(Hs, O, p))
timed("eig first", lambda: nt.EigenSolvers.EigenDecomposition(Hs, O, n, O2, p))
timed("eig again", lambda: nt.EigenSolvers.EigenDecomposition(Hs, O, n, O2, p))
timed("dsqrt", lambda: nt.DenseSolvers.SquareRoot(Hp, O, p))
timed("foe", lambda: nt.FermiOperator.ComputeDenseFOE(Hs, I, 30.0, O, 20.0, p))
p.SetThreshold(1e-8)
timed("womgc", lambda: nt.FermiOperator.WOM_GC(Hs, I, O, -0.365, 4.0, p))
timed("womc", lambda: nt.FermiOperator.WOM_C(Hs, ISQ, O, 30.0, 4.0, p))
timed("chol", lambda: nt.LinearSolvers.CholeskyDecomposition(Hp, O, p))
timed("pchol", lambda: nt.Analysis.PivotedCholeskyDecomposition(D, O, 30, p))
R = nt.Matrix_ps(30)
timed("reduce", lambda: nt.Analysis.ReduceDimension(Hs, 30, R, p))
timed("purify", lambda: nt.GeometryOptimization.PurificationExtrapolate(D, pm("S_old"), 30.0, O, p))
timed("lowdin", lambda: nt.GeometryOptimization.LowdinExtrapolate(D, pm("S_old"), pm("S_new"), O, p))
timed("svd", lambda: nt.EigenSolvers.SingularValueDecomposition(pm("R"), O, O2, nt.Matrix_ps(n), p))
