import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ntpoly_amd as nt
from gen import banded_triplets
from oracle import oracle_py as O
nt.init_comm(); nt.ConstructGlobalProcessGrid(1, 1, 1)
nt.set_option("spgemm_fma", 1); O.set_fma(True)
def srt(t):
    c, r, v = t; o = np.lexsort((r, c)); return c[o], r[o], v[o]
def cmp(got, want, tag):
    g, w = srt(got), srt(want)
    if len(g[2]) != len(w[2]) or not np.array_equal(g[0], w[0]) or not np.array_equal(g[1], w[1]):
        print(tag, "PATTERN differs", len(g[2]), len(w[2])); return False
    d = np.nonzero(g[2] != w[2])[0]
    print(tag, "entries", len(g[2]), "value diffs", len(d), "max", np.abs(g[2] - w[2]).max() if len(d) else 0)
    return len(d) == 0
n, h, thr = 4099, 25, 1e-6
col, row, val = banded_triplets(n, h)
H = nt.Matrix_ps.from_triplets(n, col, row, val)
Ho = O.Mat.from_triplets(n, n, col, row, val)
e_min, e_max = nt.EigenBounds.GershgorinBounds(H)
eo = O.gershgorin(Ho)
print("gersh", (e_min, e_max), eo)
I = nt.Matrix_ps(n); I.FillIdentity()
X = nt.Matrix_ps(H); X.Scale(-1.0); X.Increment(I, e_max, 0.0); X.Scale(1.0 / (e_max - e_min))
Xo = Ho.copy(); O.scale(Xo, -1.0); Xo = O.increment(O.Mat.identity(n), Xo, e_max, 0.0); O.scale(Xo, 1.0 / (e_max - e_min))
cmp(X.triplets(), Xo.triplets(), "X0")
X2 = nt.Matrix_ps(n); X2.Gemm(X, X, None, 1.0, 0.0, thr)
print(nt.last_spgemm_stats())
X2o = O.ps_multiply(Xo, Xo, None, 1.0, 0.0, thr)
cmp(X2.triplets(), X2o.triplets(), "X0*X0")
X.Scale(2.0); X.Increment(X2, -1.0, thr)
O.scale(Xo, 2.0); Xo = O.increment(X2o, Xo, -1.0, thr)
cmp(X.triplets(), Xo.triplets(), "X1")
