#!/usr/bin/env python3
"""tile_triple = 1 against tile_triple = 0: the same TRS2 solve (banded N, fixed iterations), densities and energies must be
bit-identical (the option changes which workgroup computes a block, not what is computed)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ntpoly_amd as nt
from gen import banded_triplets
nt.init_comm(); nt.ConstructGlobalProcessGrid(1, 1, 1)
nt.set_option("spgemm_fma", 1)
for n, h, iters in ((65536, 100, 12), (40000, 37, 9), (4099, 25, 8)):
    col, row, val = banded_triplets(n, h)
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    I = nt.Matrix_ps(n); I.FillIdentity()
    res = []
    for tri in (0, 1):
        nt.set_option("tile_triple", tri)
        p = nt.SolverParameters(); p.SetThreshold(1e-8); p.SetConvergeDiff(1e-30); p.SetMaxIterations(iters); p.SetMonitorConvergence(False)
        K = nt.Matrix_ps(n)
        f0 = nt.fusion_counts()
        e, mu = nt.DensityMatrixSolvers.TRS2(H, I, n / 2.0, K, p)
        f1 = nt.fusion_counts()
        tr = nt.solver_trace()
        res.append((e, mu, np.array(tr["energy"]), np.array(tr["nnz"]), K.triplets(), [f1[k] - f0[k] for k in f1]))
    a, b = res
    same = a[0] == b[0] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and all(np.array_equal(x, y) for x, y in zip(a[4], b[4]))
    print("n", n, "h", h, "identical:", same, "fused", a[5], b[5], "energy", a[0], b[0])
    assert same
nt.set_option("tile_triple", 0)
# a product in slab sessions too (EPI 0): TRS4 few iterations
n, h = 16384, 60
H = nt.Matrix_ps.from_triplets(n, *banded_triplets(n, h)); I = nt.Matrix_ps(n); I.FillIdentity()
out = []
for tri in (0, 1):
    nt.set_option("tile_triple", tri)
    p = nt.SolverParameters(); p.SetThreshold(1e-8); p.SetConvergeDiff(1e-30); p.SetMaxIterations(6); p.SetMonitorConvergence(False)
    K = nt.Matrix_ps(n)
    nt.DensityMatrixSolvers.TRS4(H, I, n / 2.0, K, p)
    out.append(K.triplets())
print("trs4 identical:", all(np.array_equal(x, y) for x, y in zip(*out)))
assert all(np.array_equal(x, y) for x, y in zip(*out))
nt.set_option("tile_triple", 0)
print("triple ok")
