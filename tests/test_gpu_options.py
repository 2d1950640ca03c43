"""GPU: options of round 5 that must not change a single bit -- plan_fused (the plan of a slab step in two launches instead of
six, kernels.hip k_slab_offsets) and block_match (the matches of the block path's candidates found by a kernel of their own,
spgemm_block.hip k_bs_match): the same TRS2 solve / the same product with the option on and off."""
import numpy as np
import pytest

from gen import banded_triplets, lattice_triplets

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nt():
    import ntpoly_amd as nt
    nt.init_comm()
    nt.ConstructGlobalProcessGrid(1, 1, 1)
    return nt


def srt(t):
    c, r, v = t
    o = np.lexsort((r, c))
    return c[o], r[o], v[o]


def trs2(nt, n, trip, iters):
    H = nt.Matrix_ps.from_triplets(n, *trip)
    I = nt.Matrix_ps(n)
    I.FillIdentity()
    p = nt.SolverParameters()
    p.SetThreshold(1e-8)
    p.SetConvergeDiff(1e-30)
    p.SetMaxIterations(iters)
    p.SetMonitorConvergence(False)
    K = nt.Matrix_ps(n)
    f0 = nt.fusion_counts()
    e, mu = nt.DensityMatrixSolvers.TRS2(H, I, n / 2.0, K, p)
    f1 = nt.fusion_counts()
    return srt(K.triplets()), np.array(nt.solver_trace()["energy"]), sum(f1[k] - f0[k] for k in ("square", "update"))


@pytest.mark.parametrize("arith", ["fma", "unfused"])
def test_plan_fused_changes_no_bit(nt, arith):
    nt.set_option("spgemm_fma", 1 if arith == "fma" else 0)
    try:
        n = 20000          # (1 250 blocks of 16 columns: more than one part of k_slab_offsets is busy, ragged last part)
        trip = banded_triplets(n, 30)
        out = {}
        for pf in (1, 0):
            nt.set_option("plan_fused", pf)
            out[pf] = trs2(nt, n, trip, 8)
        assert out[1][2] >= 6 and out[0][2] >= 6, (out[1][2], out[0][2])       # (the steps ran inside the SpGEMM kernel: the plan is used)
        assert np.array_equal(out[1][1], out[0][1])
        for a, b in zip(out[1][0], out[0][0]):
            assert np.array_equal(a, b)
    finally:
        nt.set_option("plan_fused", 1)
        nt.set_option("spgemm_fma", 0)


def test_block_match_changes_no_bit(nt):
    nt.set_option("spgemm_fma", 1)
    nt.set_option("block_path", 2)
    nt.set_option("slab_algebra", 0)
    nt.drop_block_caches()
    try:
        L = 16
        n = L ** 3
        A = nt.Matrix_ps.from_triplets(n, *lattice_triplets(L))
        out = {}
        for m in (0, 1):
            nt.set_option("block_match", m)
            C = nt.Matrix_ps(n)
            C.Gemm(A, A, None, 1.0, 0.0, 1e-8)
            assert nt.last_block_stats()["used"] == 1
            out[m] = srt(C.triplets())
        for a, b in zip(out[0], out[1]):
            assert np.array_equal(a, b)
    finally:
        nt.set_option("block_match", 0)
        nt.set_option("block_path", 1)
        nt.set_option("slab_algebra", 1)
        nt.set_option("spgemm_fma", 0)
