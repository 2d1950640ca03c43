#!/usr/bin/env python3
"""One rank of the slab-session-across-ranks test (tests/test_gpu_panel_sessions.py): the loops of TRS4, the sign function,
the inverse square root and the inverse on a banded operand, with their matrices kept as column panels in slab form and the
products exchanging runs (psmatrix.cpp panel_slab_multiply).  RANK / WORLD_SIZE / NTPOLY_AMD_COMM come from the environment;
the ranks share ONE GPU and exchange through the shared-memory test transport.

    python tests/panel_session_worker.py <out-prefix>
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out = sys.argv[1]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    n = int(os.environ.get("NTPOLY_AMD_PANEL_N", "16384"))
    import ntpoly_amd as nt
    from gen import banded_triplets
    nt.init_comm(nt.get_unique_id(), rank, world)
    nt.ConstructGlobalProcessGrid(1, world, 1)
    res = {}

    def banded(h, shift=0.0):
        M = nt.Matrix_ps(n)
        c0, c1 = M.local_columns()
        t = nt.TripletList_r()
        t.set_arrays(*banded_triplets(n, h, shift=shift, c0=c0, c1=c1))
        M.FillFromTripletList(t, prepartitioned=True)
        return M

    def keep(tag, M):
        c, r, v = M.triplets()
        res[tag + "_col"], res[tag + "_row"], res[tag + "_val"] = c, r, v

    def counted(tag, fn):
        p0, s0, e0 = nt.panel_product_counts(), nt.slab_algebra_counts(), nt.exchange_stats()
        fn()
        p1, s1, e1 = nt.panel_product_counts(), nt.slab_algebra_counts(), nt.exchange_stats()
        res[tag + "_panel"] = np.array([p1["slab"] - p0["slab"], p1["declined"] - p0["declined"], p1["host_syncs"] - p0["host_syncs"]])
        res[tag + "_slab"] = np.array([s1[k] - s0[k] for k in ("products", "merges", "others", "refusals")])
        res[tag + "_exchanges"] = np.array([e1[0] - e0[0]])
        tr = nt.solver_trace()
        res[tag + "_iters"] = np.array([tr["iterations"]])

    H = banded(20)
    S = banded(12, shift=3.0)          # diagonally dominant, positive definite
    Ident = nt.Matrix_ps(n)
    Ident.FillIdentity()
    p = nt.SolverParameters()
    p.SetThreshold(1e-8)
    p.SetConvergeDiff(1e-7)

    # (TRS4 with a fixed number of iterations: at this threshold its energy differences stall around the convergence limit and
    # the count of a converged solve depends on the last bits of the reductions, one rank or many, sessions or not)
    p4 = nt.SolverParameters()
    p4.SetThreshold(1e-8)
    p4.SetConvergeDiff(1e-30)
    p4.SetMaxIterations(14)
    p4.SetMonitorConvergence(False)
    K = nt.Matrix_ps(n)
    out4 = {}
    counted("trs4", lambda: out4.update(r=nt.DensityMatrixSolvers.TRS4(H, Ident, n / 2.0, K, p4)))
    res["trs4_scal"] = np.array(out4["r"])
    res["trs4_log"] = np.array(nt.solver_trace()["energy"])
    keep("trs4_K", K)

    Sg = nt.Matrix_ps(n)
    counted("sign", lambda: nt.SignSolvers.ComputeSign(H, Sg, p))
    keep("sign", Sg)

    Z = nt.Matrix_ps(n)
    counted("isq", lambda: nt.SquareRootSolvers.InverseSquareRoot(S, Z, p))
    keep("isq", Z)

    Iv = nt.Matrix_ps(n)
    counted("inv", lambda: nt.InverseSolvers.Invert(S, Iv, p))
    keep("inv", Iv)

    # ---- TRS4 on a RELABELLED band: on several ranks the solver recovers the band (band_scope.cpp) and its loop then runs as a
    # session of column panels in the recovered order
    # (opt-in, NTPOLY_AMD_PANEL_PERM=1: two processes time-slicing ONE GPU over the test transport wait ~64 ms per step of a
    # solve inside the band scope -- profiles/README.md 86 -- which makes this case minutes long on the one-GPU box)
    from gen import permuted_banded_triplets
    if os.environ.get("NTPOLY_AMD_PANEL_PERM", "0") != "1":
        permuted_banded_triplets = None
    if permuted_banded_triplets is not None:
        Hp = nt.Matrix_ps(n)
        c0, c1 = Hp.local_columns()
        t = nt.TripletList_r()
        t.set_arrays(*permuted_banded_triplets(n, 20, 42, c0=c0, c1=c1))
        Hp.FillFromTripletList(t, prepartitioned=True)
        del t
        Kp = nt.Matrix_ps(n)
        b0 = nt.band_scope_counts()
        counted("trs4p", lambda: out4.update(p=nt.DensityMatrixSolvers.TRS4(Hp, Ident, n / 2.0, Kp, p4)))
        b1 = nt.band_scope_counts()
        res["trs4p_scope"] = np.array([b1["solves"] - b0["solves"]])
        res["trs4p_scal"] = np.array(out4["p"])
        res["trs4p_log"] = np.array(nt.solver_trace()["energy"])
        kc, kr, kv = Kp.triplets()
        res["trs4p_sums"] = np.array([float(len(kv)), float(np.sum(kv)), float(np.sum(kv * kv)), float(np.sum(np.abs(kv)))])
        del Hp, Kp

    # ---- the other loop families that open a session (polynomials, functions, density solvers beside TRS4) on a small
    # operand: every collective of their vocabulary must be entered by every rank whatever form its panel is in
    m = int(os.environ.get("NTPOLY_AMD_PANEL_M", "4096"))
    n_big, n = n, m

    def banded_m(h, shift=0.0, scale=1.0):
        M = nt.Matrix_ps(m)
        c0, c1 = M.local_columns()
        t = nt.TripletList_r()
        col, row, val = banded_triplets(m, h, shift=shift, c0=c0, c1=c1)
        t.set_arrays(col, row, val * scale)
        M.FillFromTripletList(t, prepartitioned=True)
        return M

    Hm = banded_m(10, scale=0.4)
    Sm = banded_m(8, shift=3.0)
    Im = nt.Matrix_ps(m)
    Im.FillIdentity()
    pm = nt.SolverParameters()
    pm.SetThreshold(1e-9)
    pm.SetConvergeDiff(1e-8)
    poly = nt.Polynomial(5)
    for k, cf in enumerate((0.3, -0.7, 0.25, 0.11, -0.05)):
        poly.SetCoefficient(k, cf)
    for tag, fn in (("horner", lambda O: poly.HornerCompute(Hm, O, pm)),
                    ("paterson", lambda O: poly.PatersonStockmeyerCompute(Hm, O, pm)),
                    ("exp", lambda O: nt.ExponentialSolvers.ComputeExponential(Hm, O, pm)),
                    ("log", lambda O: nt.ExponentialSolvers.ComputeLogarithm(Sm, O, pm)),
                    ("sine", lambda O: nt.TrigonometrySolvers.Sine(Hm, O, pm)),
                    ("root3", lambda O: nt.RootSolvers.ComputeRoot(Sm, O, 3, pm)),
                    ("sqrt", lambda O: nt.SquareRootSolvers.SquareRoot(Sm, O, pm)),
                    ("hpcp", lambda O: nt.DensityMatrixSolvers.HPCP(Hm, Im, m / 2.0, O, pm)),
                    ("pm", lambda O: nt.DensityMatrixSolvers.PM(Hm, Im, m / 2.0, O, pm))):
        O = nt.Matrix_ps(m)
        counted("m_" + tag, lambda: fn(O))
        keep("m_" + tag, O)
        del O

    np.savez(out + ".%d.npz" % rank, **res)
    nt.DestructGlobalProcessGrid()


if __name__ == "__main__":
    main()
