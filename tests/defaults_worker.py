#!/usr/bin/env python3
"""A drop-in caller with the library's DEFAULT configuration (tests/test_gpu_defaults.py): a fresh process, no option set,
no environment override -- what an unmodified program written against the reference's C ABI runs with.  Exercises the
vocabulary of the C ABI (a McWeeny loop spelled with MatrixMultiply / Copy / Scale / Increment on a lattice operand:
block path, results kept in block form across calls), TRS2_wrp on a banded operand and SignFunction_wrp on a complex one,
and writes the results to <out>.npz.

    python tests/defaults_worker.py <out-prefix>
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out = sys.argv[1]
    for k in list(os.environ):      # (nothing of the engine's environment knobs)
        if k.startswith("NTPOLY_AMD_") and k not in ("NTPOLY_AMD_COMM",):
            del os.environ[k]
    import ntpoly_amd as nt
    from gen import banded_triplets, lattice_triplets
    nt.init_comm()
    nt.ConstructGlobalProcessGrid(1, 1, 1)
    res = {}
    res["options"] = np.array([nt.get_option(k) for k in ("spgemm_fma", "block_path", "complex_tile", "complex_sessions", "thin_left",
                                                          "slab_algebra", "tile2")])

    def keep(tag, M):
        c, r, v = M.triplets()
        res[tag + "_col"], res[tag + "_row"], res[tag + "_val"] = c, r, np.asarray(v)

    # ---- 1. TRS2_wrp, banded
    n, h = 8192, 40
    H = nt.Matrix_ps.from_triplets(n, *banded_triplets(n, h))
    I = nt.Matrix_ps(n)
    I.FillIdentity()
    p = nt.SolverParameters()
    p.SetThreshold(1e-7)
    p.SetConvergeDiff(1e-30)
    p.SetMaxIterations(12)
    p.SetMonitorConvergence(False)
    K = nt.Matrix_ps(n)
    f0 = nt.fusion_counts()
    e, mu = nt.DensityMatrixSolvers.TRS2(H, I, n / 2.0, K, p)
    f1 = nt.fusion_counts()
    tr = nt.solver_trace()
    res["trs2_scal"] = np.array([e, mu])
    res["trs2_nnz"] = np.array(tr["nnz"])
    res["trs2_fused"] = np.array([f1[k] - f0[k] for k in ("square", "update", "repeated")])
    keep("trs2_K", K)

    # ---- 2. a caller's own loop over the C ABI on a lattice operand (no band: the block path)
    L, thr = 24, 1e-9
    m = L ** 3
    X = nt.Matrix_ps.from_triplets(m, *lattice_triplets(L, shift=2.0))
    X.Scale(0.2)
    X2, X3, T = nt.Matrix_ps(m), nt.Matrix_ps(m), nt.Matrix_ps(m)
    used = []
    for it in range(3):
        X2.Gemm(X, X, None, 1.0, 0.0, thr)
        used.append(int(nt.last_block_stats().get("used", 0)))
        X3.Gemm(X2, X, None, 1.0, 0.0, thr)
        used.append(int(nt.last_block_stats().get("used", 0)))
        nt.lib.CopyMatrix_ps_wrp(X2.ih, T.ih)
        T.Scale(3.0)
        T.Increment(X3, -2.0, thr)
        nt.lib.CopyMatrix_ps_wrp(T.ih, X.ih)
    res["mcweeny_block_used"] = np.array(used)
    res["mcweeny_scal"] = np.array([X.Trace(), X.Norm(), float(np.real(X.Dot(X2))), X.GetSize()])
    keep("mcweeny_X", X)

    # ---- 3. SignFunction_wrp on a complex Hermitian operand (complex tile kernel, complex session)
    nc, hc = 2048, 24
    Hc = nt.Matrix_ps.from_triplets(nc, *banded_triplets(nc, hc, complex_=True))
    ps = nt.SolverParameters()
    ps.SetThreshold(1e-8)
    ps.SetConvergeDiff(1e-6)
    S = nt.Matrix_ps(nc)
    nt.SignSolvers.ComputeSign(Hc, S, ps)
    res["sign_iters"] = np.array([nt.solver_trace()["iterations"]])
    keep("sign_S", S)
    np.savez(out + ".npz", **res)
    nt.DestructGlobalProcessGrid()


if __name__ == "__main__":
    main()
