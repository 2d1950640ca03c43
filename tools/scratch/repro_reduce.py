import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import ntpoly_amd as nt
from golden_util import Golden
nt.init_comm(); nt.ConstructGlobalProcessGrid(1, 1, 1)
g = Golden("extras")
t = g.tri(None, "M_Hs")
A = nt.Matrix_ps.from_triplets(t[0], t[2], t[3], t[4])
n = 96; dim = 30
p = nt.SolverParameters(); p.SetThreshold(1e-9); p.SetConvergeDiff(1e-8)
I = nt.Matrix_ps(n); I.FillIdentity()
P = nt.Matrix_ps(n)
def say(*a):
    print(*a, flush=True)
nt.DensityMatrixSolvers.TRS4(A, I, float(dim), P, p); nt.synchronize(); say("trs4 ok", P.GetSize())
L = nt.Matrix_ps(n)
nt.Analysis.PivotedCholeskyDecomposition(P, L, dim, p); nt.synchronize(); say("pchol ok", L.GetSize())
LT = nt.Matrix_ps(n); LT.Transpose(L); nt.synchronize(); say("transpose ok", LT.GetSize())
say("is identity", LT.IsIdentity())
T = nt.Matrix_ps(n); T.Gemm(LT, A, None, 1.0, 0.0, 1e-9); nt.synchronize(); say("gemm1 ok", T.GetSize(), nt.last_spgemm_stats())
V = nt.Matrix_ps(n); V.Gemm(T, L, None, 1.0, 0.0, 1e-9); nt.synchronize(); say("gemm2 ok", V.GetSize(), nt.last_spgemm_stats())
R = nt.Matrix_ps(dim)
nt.Analysis.ReduceDimension(A, dim, R, p); nt.synchronize(); say("reduce ok", R.GetSize())
