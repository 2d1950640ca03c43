#!/usr/bin/env python3
"""One rank of the multi-process engine test (tests/test_gpu_multirank.py): RANK / WORLD_SIZE / NTPOLY_AMD_COMM
come from the environment, the ranks are processes sharing ONE GPU and exchange through the shared-memory test
transport (csrc/comm.cpp).  Writes this rank's results to <out>.<rank>.npz.

    python tests/multirank_worker.py <out-prefix>
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
    import ntpoly_amd as nt
    from gen import banded_triplets
    nt.init_comm(nt.get_unique_id(), rank, world)
    # (NTPOLY_AMD_TEST_GRID=rows,columns,slices: another shape over the same ranks -- the CommSplitMatrix test splits along slices)
    shape = tuple(int(x) for x in os.environ.get("NTPOLY_AMD_TEST_GRID", "1,%d,1" % world).split(","))
    nt.ConstructGlobalProcessGrid(*shape)
    res = {}

    def keep(tag, M):
        c, r, v = M.triplets()
        res[tag + "_col"], res[tag + "_row"], res[tag + "_val"] = c, r, v

    # ---- banded operands (halo exchange): A*B with A != B, then the same product with a threshold
    n, h = 2500, 30
    A = nt.Matrix_ps(n)
    c0, c1 = A.local_columns()
    res["c0"], res["c1"] = c0, c1
    t = nt.TripletList_r()
    t.set_arrays(*banded_triplets(n, h, c0=c0, c1=c1))
    A.FillFromTripletList(t, prepartitioned=True)
    B = nt.Matrix_ps(n)
    t2 = nt.TripletList_r()
    col, row, val = banded_triplets(n, h + 7, shift=0.3, c0=c0, c1=c1)
    t2.set_arrays(col, row, val * 1.01)
    B.FillFromTripletList(t2, prepartitioned=True)
    C = nt.Matrix_ps(n)
    x0 = nt.exchange_stats()
    C.Gemm(A, B, None, 0.5, 0.0, 1e-7)
    x1 = nt.exchange_stats()
    res["exchanges"], res["exchange_host_syncs"], res["gemm_host_syncs"] = x1[0] - x0[0], x1[1] - x0[1], x1[2] - x0[2]
    keep("AB", C)
    res["AB_trace"], res["AB_norm"], res["AB_dot"] = C.Trace(), C.Norm(), float(np.real(C.Dot(A)))
    AT = nt.Matrix_ps(n)
    AT.Transpose(C)
    keep("ABT", AT)
    # GatherMatrixToProcess (PSMatrixModule.F90:1704-1808): the whole product as a local matrix on every rank, and on rank 1
    # of the slice only
    Lall = C.GatherMatrixToProcess()
    gc, gr, gv = Lall.triplets()
    res["gather_all"] = np.array([len(gv), float(np.sum(gv)), float(np.sum(gv * gr)), float(np.sum(gv * gc))])
    Lone = C.GatherMatrixToProcess(min(1, world - 1))
    res["gather_one"] = np.array([-1.0 if Lone is None else float(len(Lone.triplets()[2]))])
    # the same product with the halo exchange overlapped: interior columns multiplied while the halo travels on the
    # communication stream, boundary columns afterwards (forced; by default only when the halo is large)
    nt.set_option("halo_overlap", 2)
    C2 = nt.Matrix_ps(n)
    C2.Gemm(A, B, None, 0.5, 0.0, 1e-7)
    keep("AB_ov", C2)
    C3 = nt.Matrix_ps(n)
    C3.Gemm(A, A, None, 1.0, 0.0, 0.0)
    nt.set_option("halo_overlap", 0)
    C4 = nt.Matrix_ps(n)
    C4.Gemm(A, A, None, 1.0, 0.0, 0.0)
    nt.set_option("halo_overlap", 1)
    keep("AA_ov", C3)
    keep("AA", C4)

    # ---- TRS2 on the banded Hamiltonian (ISQ = I): energies per iteration, chemical potential, density
    Ident = nt.Matrix_ps(n)
    Ident.FillIdentity()
    p = nt.SolverParameters()
    p.SetThreshold(1e-7)
    p.SetConvergeDiff(1e-9)
    K = nt.Matrix_ps(n)
    f0, e0 = nt.fusion_counts(), nt.exchange_stats()
    nt.set_option("time_kernels", 1)     # (the multiplies' statistics are kept only with the timers on)
    nt.reset_spgemm_accum()
    energy, mu = nt.DensityMatrixSolvers.TRS2(A, Ident, n / 2.0, K, p)
    acc = nt.spgemm_accum()
    nt.set_option("time_kernels", 0)
    res["trs2_products"], res["trs2_nnz_c"] = acc["products"], acc["nnz_c"]
    f1, e1 = nt.fusion_counts(), nt.exchange_stats()
    res["trs2_fused"] = np.array([f1[k] - f0[k] for k in ("square", "update", "repeated")])
    res["trs2_exchanges"] = np.array([e1[0] - e0[0], e1[1] - e0[1], e1[2] - e0[2]])
    tr = nt.solver_trace()
    res["trs2_energy"], res["trs2_mu"], res["trs2_iters"] = energy, mu, tr["iterations"]
    res["trs2_log"] = np.array(tr["energy"])
    keep("K", K)

    # ---- operands without band structure (the halo degenerates to a full gather; general SpGEMM path)
    rng = np.random.default_rng(1234)   # same seed on every rank: every rank generates the whole matrix ...
    m = 900
    dense = rng.random((m, m))
    mask = rng.random((m, m)) < 0.02
    R = np.where(mask, dense - 0.5, 0.0)
    rr, cc = np.nonzero(R.T)            # column-major triplets
    colg, rowg, valg = rr + 1, cc + 1, R.T[rr, cc]
    G = nt.Matrix_ps(m)
    g0, g1 = G.local_columns()
    sel = (colg > g0) & (colg <= g1)    # ... and passes its own columns
    tg = nt.TripletList_r()
    tg.set_arrays(colg[sel].astype(np.int32), rowg[sel].astype(np.int32), valg[sel])
    G.FillFromTripletList(tg, prepartitioned=True)
    GG = nt.Matrix_ps(m)
    GG.Gemm(G, G, None, 1.0, 0.0, 1e-4)
    keep("GG", GG)
    res["g0"], res["g1"] = g0, g1

    # ---- the gather-based solver families: eigendecomposition (every rank factors the gathered matrix and keeps its
    # panel), dense matrix function, Cholesky, pivoted Cholesky, CG, dense FOE, pattern snap
    ms = 160
    S = nt.Matrix_ps(ms)
    s0, s1 = S.local_columns()
    ts = nt.TripletList_r()
    ts.set_arrays(*banded_triplets(ms, 6, shift=2.5, c0=s0, c1=s1))
    S.FillFromTripletList(ts, prepartitioned=True)
    res["s0"], res["s1"] = s0, s1
    q = nt.SolverParameters()
    q.SetThreshold(1e-10)
    q.SetConvergeDiff(1e-9)
    W, V = nt.Matrix_ps(ms), nt.Matrix_ps(ms)
    nt.EigenSolvers.EigenDecomposition(S, W, ms, V, q)
    keep("eigW", W)
    keep("eigV", V)
    F = nt.Matrix_ps(ms)
    nt.DenseSolvers.InverseSquareRoot(S, F, q)
    keep("disq", F)
    L = nt.Matrix_ps(ms)
    nt.LinearSolvers.CholeskyDecomposition(S, L, q)
    keep("chol", L)
    L2 = nt.Matrix_ps(ms)
    nt.Analysis.PivotedCholeskyDecomposition(S, L2, 40, q)
    keep("pchol", L2)
    q0 = nt.SolverParameters()
    q0.SetThreshold(0.0)
    q0.SetConvergeDiff(1e-9)
    X = nt.Matrix_ps(ms)
    nt.LinearSolvers.CGSolver(S, X, L, q0)
    keep("cg", X)
    Is = nt.Matrix_ps(ms)
    Is.FillIdentity()
    Kf = nt.Matrix_ps(ms)
    res["foe_energy"], res["foe_mu"] = nt.FermiOperator.ComputeDenseFOE(S, Is, 50.0, Kf, 25.0, q)
    keep("foe", Kf)
    Sn = nt.Matrix_ps(F)
    nt.MatrixConversion.SnapMatrixToSparsityPattern(Sn, S)
    keep("snap", Sn)

    # ---- CommSplitMatrix (PSMatrixModule.F90:1489-1541; SplitProcessGrid, ProcessGridModule.F90:430-515): a copy of the WHOLE
    # matrix on each half of the grid, on a sub-communicator; a product, reductions and a TRS2 solve INSIDE the half
    # (the halves work at the same time and must not wait for each other), then the grid of all processes again
    Sp, color, split_slice = A.CommSplit()
    sub_rank, sub_size, is_sub = Sp.grid_comm_info()
    res["split_info"] = np.array([color, int(split_slice), sub_rank, sub_size, int(is_sub)])
    keep("split", Sp)
    SS = nt.Matrix_ps(n)      # (a matrix of the default grid; the result of a product takes the grid of its operands)
    SS.Gemm(Sp, Sp, None, 1.0, 0.0, 1e-7)
    keep("split_SS", SS)
    res["split_scal"] = np.array([Sp.Trace(), Sp.Norm(), float(np.real(SS.Dot(Sp)))])
    Isub = nt.Matrix_ps(Sp)
    Isub.FillIdentity()
    ps = nt.SolverParameters()
    ps.SetThreshold(1e-7)
    ps.SetConvergeDiff(1e-30)
    ps.SetMaxIterations(4 + color)      # (the halves do different amounts of work: no hidden coupling between them)
    ps.SetMonitorConvergence(False)
    Ksub = nt.Matrix_ps(Sp)
    e_sub, mu_sub = nt.DensityMatrixSolvers.TRS2(Sp, Isub, n / 2.0, Ksub, ps)
    res["split_trs2"] = np.array([e_sub, mu_sub])
    keep("split_K", Ksub)
    res["after_split"] = np.array([A.Trace(), A.Norm()])      # (the grid of all processes again)
    nt.barrier()

    np.savez(out + ".%d.npz" % rank, **res)
    nt.DestructGlobalProcessGrid()


if __name__ == "__main__":
    main()
