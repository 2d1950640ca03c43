import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ntpoly_amd as nt
from gen import lattice_triplets
from oracle import oracle_py as O
nt.init_comm(); nt.ConstructGlobalProcessGrid(1, 1, 1)
nt.set_option("time_kernels", 1)
def srt(t):
    c, r, v = t; o = np.lexsort((r, c)); return c[o], r[o], v[o]
for arith in (0, 1):
    nt.set_option("spgemm_fma", arith); O.set_fma(bool(arith))
    L = 24; n = L ** 3; thr = 1e-8
    col, row, val = lattice_triplets(L)
    A = nt.Matrix_ps.from_triplets(n, col, row, val)
    Ao = O.Mat.from_triplets(n, n, col, row, val)
    X2o = O.ps_multiply(Ao, Ao, None, 1.0, 0.0, thr)
    X3o = O.ps_multiply(X2o, Ao, None, 1.0, 0.0, thr)
    C = nt.Matrix_ps(n); C.Gemm(A, A, None, 1.0, 0.0, thr)
    print("arith", arith, "H*H stats", nt.last_spgemm_stats(), nt.last_grouped_stats())
    g, w = srt(C.triplets()), srt(X2o.triplets())
    print("  H*H equal:", len(g[2]) == len(w[2]) and all(np.array_equal(a, b) for a, b in zip(g, w)), len(g[2]) / n)
    D = nt.Matrix_ps(n); D.Gemm(C, A, None, 1.0, 0.0, thr)
    print("  (H*H)*H stats", nt.last_spgemm_stats())
    g, w = srt(D.triplets()), srt(X3o.triplets())
    print("  (H*H)*H equal:", len(g[2]) == len(w[2]) and all(np.array_equal(a, b) for a, b in zip(g, w)), len(g[2]) / n)
