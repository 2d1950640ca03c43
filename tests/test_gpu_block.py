"""GPU: the block path (csrc/spgemm_block.hip) -- real square operands WITHOUT run structure (3-D lattice Hamiltonians,
relabelled bands) multiplied as 16 x 16 tiles of a clustered index order on the FP64 matrix cores, FMA arithmetic.

Parity statement.  The engine picks a relabelling `pos` of the index set (as the reference's own load balancer does,
LoadBalancerModule.F90:14-52) and computes every product entry as the chain of fma() over ascending POSITION.  So

  (1) BIT FOR BIT: engine(A, B) == un-relabel( oracle_fma( relabel(A), relabel(B) ) ) -- the oracle (CPU restatement of
      MultiplyBlock.f90:9-36 + PruneList.f90:8-38, FMA mode pinned to the contracted reference build) run on exactly
      the relabelled matrices, order of positions = order of its k loop;
  (2) TOLERANCE against the oracle on the caller's labels (chain over ascending label): every entry within 1e-13
      relative of the product's scale, the same pattern except entries within roundoff of the threshold
      (the contract of label-ordered operands).
"""
import numpy as np
import pytest

from gen import banded_triplets, lattice_triplets, permuted_banded_triplets

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nt():
    import ntpoly_amd as nt
    nt.init_comm()
    nt.ConstructGlobalProcessGrid(1, 1, 1)
    return nt


@pytest.fixture()
def fma(nt):
    from oracle import oracle_py as O
    nt.set_option("spgemm_fma", 1)
    # (2 = whatever the row windows: left to itself the engine keeps operands whose windows fit the direct-mapped LDS kernels
    # -- 4096 rows, i.e. every lattice up to 16^3 -- away from the block path; the 64^3 test runs with the automatic rule)
    nt.set_option("block_path", 2)
    O.set_fma(True)
    nt.drop_block_caches()
    yield O
    O.set_fma(False)
    nt.set_option("spgemm_fma", 0)
    nt.set_option("block_path", 1)
    nt.set_option("slab_algebra", 1)


@pytest.fixture()
def unfused_block(nt):
    """the block path in UNFUSED arithmetic (option block_unfused): every product rounded, then added, in ascending position"""
    from oracle import oracle_py as O
    nt.set_option("spgemm_fma", 0)
    nt.set_option("block_unfused", 1)
    nt.set_option("block_path", 2)
    O.set_fma(False)
    nt.drop_block_caches()
    yield O
    nt.set_option("block_unfused", 0)
    nt.set_option("block_path", 1)


def srt(t):
    c, r, v = t
    o = np.lexsort((r, c))
    return c[o], r[o], v[o]


def relabel(trip, rank):
    """entries (col, row, val) 1-based moved to (rank[col], rank[row])"""
    c, r, v = trip
    return srt(((rank[c - 1] + 1).astype(np.int32), (rank[r - 1] + 1).astype(np.int32), v))


def oracle_product_in_engine_order(O, n, ta, tb, pos, alpha, thr):
    """oracle FMA product of the matrices relabelled by the engine's positions, mapped back to the caller's labels"""
    order = np.argsort(pos, kind="stable")          # index at every rank
    rank = np.empty(n, dtype=np.int64)
    rank[order] = np.arange(n)
    Ao = O.Mat.from_triplets(n, n, *relabel(ta, rank))
    Bo = Ao if tb is ta else O.Mat.from_triplets(n, n, *relabel(tb, rank))
    c, r, v = O.ps_multiply(Ao, Bo, None, alpha, 0.0, thr).triplets()
    return srt(((order[c - 1] + 1).astype(np.int32), (order[r - 1] + 1).astype(np.int32), v))


def exact(got, want, what):
    g, w = srt(got), srt(want)
    assert len(g[2]) == len(w[2]), "%s: %d vs %d entries" % (what, len(g[2]), len(w[2]))
    assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1]), what + ": pattern differs"
    assert np.array_equal(g[2], w[2]), "%s: values differ, max |d| = %g" % (what, np.abs(g[2] - w[2]).max())


def close(got, want, n, thr, what, rel=1e-13):
    import scipy.sparse as sp
    G = sp.csr_matrix((got[2], (got[1] - 1, got[0] - 1)), shape=(n, n))
    W = sp.csr_matrix((want[2], (want[1] - 1, want[0] - 1)), shape=(n, n))
    scale = max(1.0, np.abs(want[2]).max())
    D = (G - W).tocoo()
    bad = np.abs(D.data) > rel * scale
    # entries present on one side only must sit at the threshold
    assert np.all(np.abs(D.data[bad]) <= thr * (1 + 1e-9) + rel * scale), "%s: max |d| = %g" % (what, np.abs(D.data).max())
    assert abs(G.nnz - W.nnz) <= max(8, 1e-5 * W.nnz), "%s: %d vs %d entries" % (what, G.nnz, W.nnz)


def lattice_case(L, r2=13):
    return lattice_triplets(L, r2=r2)


CASES = [("lattice16", 1e-8, 1.0), ("lattice20", 1e-6, 0.5), ("lattice12", 0.0, -0.75), ("permuted_band", 1e-8, 1.0), ("lattice_ab", 1e-7, 1.0),
         ("lattice_asym", 1e-9, 1.0)]


def block_product_case(nt, O, kind, thr, alpha):
    if kind.startswith("lattice") and kind not in ("lattice_ab", "lattice_asym"):
        L = int(kind[7:])
        n = L ** 3
        ta = lattice_case(L)
        tb = ta
    elif kind == "lattice_ab":
        L = 16
        n = L ** 3
        ta = lattice_case(L)
        c, r, v = lattice_triplets(L, r2=6, shift=0.3)
        tb = (c, r, v * 1.25)
    elif kind == "lattice_asym":      # patterns that are NOT symmetric, operands that differ, columns left empty
        L = 16
        n = L ** 3
        c, r, v = lattice_case(L)
        keep = ((r.astype(np.int64) * 7 + c.astype(np.int64) * 13) % 5 != 0) & (c % 97 != 0)
        ta = (c[keep], r[keep], v[keep] * (1.0 + 0.001 * (r[keep] % 11)))
        keep = ((r.astype(np.int64) * 3 + c.astype(np.int64) * 17) % 7 != 0) & (r % 89 != 0)
        tb = (c[keep], r[keep], v[keep] * (1.0 - 0.002 * (c[keep] % 5)))
    else:
        n = 6000
        ta = permuted_banded_triplets(n, 40, 7)
        tb = ta
    A = nt.Matrix_ps.from_triplets(n, *ta)
    B = A if tb is ta else nt.Matrix_ps.from_triplets(n, *tb)
    nt.set_option("slab_algebra", 0)     # (the C ABI's session would try the slab form first; the refusal is tested elsewhere)
    C = nt.Matrix_ps(n)
    C.Gemm(A, B, None, alpha, 0.0, thr)
    bs = nt.last_block_stats()
    assert bs["used"] == 1, (kind, bs, nt.last_spgemm_stats())
    assert bs["fill"] >= 0.08
    got = srt(C.triplets())
    pos = nt.block_order(A)
    assert pos is not None and len(np.unique(pos)) == n
    # (1) bit for bit against the oracle on the relabelled matrices
    want_rel = oracle_product_in_engine_order(O, n, ta, tb, pos, alpha, thr)
    exact(got, want_rel, kind + " (engine order)")
    # (2) tolerance against the oracle on the caller's labels
    Ao = O.Mat.from_triplets(n, n, *ta)
    Bo = Ao if tb is ta else O.Mat.from_triplets(n, n, *tb)
    want = srt(O.ps_multiply(Ao, Bo, None, alpha, 0.0, thr).triplets())
    close(got, want, n, thr, kind + " (caller's labels)")


@pytest.mark.parametrize("kind,thr,alpha", CASES)
def test_block_product_vs_oracle(nt, fma, kind, thr, alpha):
    block_product_case(nt, fma, kind, thr, alpha)


@pytest.mark.parametrize("kind,thr,alpha", [CASES[0], CASES[2], CASES[4], CASES[5]])
def test_block_product_vs_oracle_unfused(nt, unfused_block, kind, thr, alpha):
    """option block_unfused: the same parity statement in UNFUSED arithmetic (MultiplyBlock.f90:33 as the reference's default
    x86-64 build computes it: product rounded, then added) -- bit for bit the oracle's unfused mode on the matrices relabelled
    by the engine's positions, 1e-13 against the sums over ascending labels"""
    block_product_case(nt, unfused_block, kind, thr, alpha)


def test_unfused_arithmetic_keeps_the_label_order_by_default(nt):
    """without the option a lattice product in unfused arithmetic stays off the block path: bit for bit the oracle on the caller's labels"""
    from oracle import oracle_py as O
    nt.set_option("spgemm_fma", 0)
    nt.set_option("block_path", 2)
    O.set_fma(False)
    nt.drop_block_caches()
    try:
        L = 12
        n = L ** 3
        ta = lattice_case(L)
        A = nt.Matrix_ps.from_triplets(n, *ta)
        C = nt.Matrix_ps(n)
        C.Gemm(A, A, None, 1.0, 0.0, 1e-8)
        assert nt.last_block_stats()["used"] == 0
        Ao = O.Mat.from_triplets(n, n, *ta)
        exact(C.triplets(), O.ps_multiply(Ao, Ao, None, 1.0, 0.0, 1e-8).triplets(), "unfused lattice on the caller's labels")
    finally:
        nt.set_option("block_path", 1)


def test_block_path_is_declined_for_unstructured_operands(nt, fma):
    """a random sparse matrix has no blocks: the clustering is tried once, the product takes the LDS-hash path and is
    bit-exact against the oracle on the caller's labels"""
    O = fma
    n, per = 8192, 12      # (row windows beyond the direct-mapped LDS kernels: the block path is asked, and declines on the fill)
    rng = np.random.default_rng(5)
    col = np.repeat(np.arange(1, n + 1, dtype=np.int32), per)
    row = rng.integers(1, n + 1, size=n * per).astype(np.int32)
    key = np.unique(col.astype(np.int64) * (n + 1) + row)
    col, row = (key // (n + 1)).astype(np.int32), (key % (n + 1)).astype(np.int32)
    val = rng.standard_normal(len(col))
    A = nt.Matrix_ps.from_triplets(n, col, row, val)
    nt.set_option("slab_algebra", 0)
    nt.set_option("block_path", 1)
    C = nt.Matrix_ps(n)
    C.Gemm(A, A, None, 1.0, 0.0, 1e-9)
    assert nt.last_block_stats()["used"] == 0
    Ao = O.Mat.from_triplets(n, n, col, row, val)
    exact(C.triplets(), O.ps_multiply(Ao, Ao, None, 1.0, 0.0, 1e-9).triplets(), "random sparse")


def test_lattice_trs2_through_the_block_path(nt, fma):
    """TRS2 on a 16^3 lattice: every product of the loop on the block path; sigma sequence and entry counts as the oracle's
    FMA mode on the caller's labels, energies 1e-11, density 1e-10 (the chain order differs: tolerance contract)"""
    O = fma
    L = 16
    n = L ** 3
    col, row, val = lattice_triplets(L)
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    I = nt.Matrix_ps(n)
    I.FillIdentity()
    p = nt.SolverParameters()
    p.SetThreshold(1e-8)
    p.SetConvergeDiff(1e-30)
    p.SetMaxIterations(10)
    p.SetMonitorConvergence(False)
    K = nt.Matrix_ps(n)
    energy, mu = nt.DensityMatrixSolvers.TRS2(H, I, n / 2.0, K, p)
    assert nt.last_block_stats()["used"] == 1
    tr = nt.solver_trace()
    Ho = O.Mat.from_triplets(n, n, col, row, val)
    Ko, e_o, mu_o, tro = O.density("trs2", Ho, O.Mat.identity(n), n / 2.0,
                                   O.params(converge_diff=1e-30, max_iterations=10, threshold=1e-8, monitor_convergence=False))
    assert list(tr["sigma"]) == list(tro["sigma"])
    assert abs(energy - e_o) <= 1e-11 * abs(e_o)
    close(srt(K.triplets()), srt(Ko.triplets()), n, 1e-8, "density", rel=1e-10)


def test_products_stay_in_block_form_across_c_abi_calls(nt, fma):
    """A caller's loop over MatrixMultiply_ps_wrp on a lattice operand: from the second product on the result stays in block
    form (DevMat::blk) and is multiplied as it is; every other entry point packs on access.  (A A) A through the
    sessions equals the oracle on the relabelled matrices bit for bit, and Dot / Norm / Increment / Scale / Copy on a
    block-form matrix give what they give on compressed columns."""
    import scipy.sparse as sp
    O = fma
    L = 16
    n = L ** 3
    ta = lattice_triplets(L)
    A = nt.Matrix_ps.from_triplets(n, *ta)
    nt.set_option("slab_algebra", 1)
    thr = 1e-7
    C1 = nt.Matrix_ps(n)
    C1.Gemm(A, A, None, 1.0, 0.0, thr)        # (first product of the dimension: compressed columns)
    assert nt.last_block_stats()["used"] == 1
    C2 = nt.Matrix_ps(n)
    C2.Gemm(A, A, None, 1.0, 0.0, thr)        # block form from here on
    C3 = nt.Matrix_ps(n)
    C3.Gemm(C2, A, None, 0.5, 0.0, thr)       # a block-form operand
    assert nt.last_block_stats()["used"] == 1
    pos = nt.block_order(A)
    order = np.argsort(pos, kind="stable")
    rank = np.empty(n, dtype=np.int64)
    rank[order] = np.arange(n)
    Ao = O.Mat.from_triplets(n, n, *relabel(ta, rank))
    P2 = O.ps_multiply(Ao, Ao, None, 1.0, 0.0, thr)
    P3 = O.ps_multiply(P2, Ao, None, 0.5, 0.0, thr)
    back = lambda M: srt(((order[M.triplets()[0] - 1] + 1).astype(np.int32), (order[M.triplets()[1] - 1] + 1).astype(np.int32), M.triplets()[2]))
    exact(C1.triplets(), back(P2), "first product")
    exact(C3.triplets(), back(P3), "(A A) A, block-form operand")
    exact(C2.triplets(), back(P2), "second product (packed on access)")
    # the other vocabulary calls on a block-form matrix
    C4 = nt.Matrix_ps(n)
    C4.Gemm(A, A, None, 1.0, 0.0, thr)
    want = sp.csr_matrix((back(P2)[2], (back(P2)[1] - 1, back(P2)[0] - 1)), shape=(n, n))
    assert abs(C4.Norm() - abs(want).sum(axis=0).max()) <= 1e-12 * abs(want).sum(axis=0).max()
    C5 = nt.Matrix_ps(n)
    C5.Gemm(A, A, None, 1.0, 0.0, thr)
    d = C5.Dot(A)
    As = sp.csr_matrix((ta[2], (ta[1] - 1, ta[0] - 1)), shape=(n, n))
    assert abs(d - want.multiply(As).sum()) <= 1e-10 * max(1.0, abs(want.multiply(As).sum()))
    C6 = nt.Matrix_ps(n)
    C6.Gemm(A, A, None, 1.0, 0.0, thr)
    C6.Scale(2.0)
    C6.Increment(A, -1.0, 0.0)
    G = sp.csr_matrix((srt(C6.triplets())[2], (srt(C6.triplets())[1] - 1, srt(C6.triplets())[0] - 1)), shape=(n, n))
    assert abs(G - (2.0 * want - As)).max() <= 1e-12
    C7 = nt.Matrix_ps(n)
    C7.Gemm(A, A, None, 1.0, 0.0, thr)
    C8 = nt.Matrix_ps(C7)                      # CopyMatrix of a block-form matrix
    exact(C8.triplets(), back(P2), "copy")


@pytest.mark.parametrize("L,thr", [(16, 1e-8), (12, 1e-5)])
def test_block_form_trs2_steps_equal_the_separate_passes(nt, fma, L, thr):
    """The TRS2 loop with the iterate kept in block form (product, AddSparseVectors merge with the tail rule in the caller's
    labels, energy, trace: spgemm_block.hip block_trs2_step) against the same loop with the option fused_update off --
    the block path's product followed by the merge / dot passes on compressed columns: the same arithmetic element by
    element, so the density must be IDENTICAL (pattern and bits), the sigma sequence and the entry counts of every
    iteration equal, energies to summation order."""
    n = L ** 3
    col, row, val = lattice_triplets(L)
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    I = nt.Matrix_ps(n)
    I.FillIdentity()
    res = []
    for fused in (0, 1):
        nt.set_option("fused_update", fused)
        try:
            p = nt.SolverParameters()
            p.SetThreshold(thr)
            p.SetConvergeDiff(1e-30)
            p.SetMaxIterations(14)
            p.SetMonitorConvergence(False)
            K = nt.Matrix_ps(n)
            f0 = nt.fusion_counts()
            energy, mu = nt.DensityMatrixSolvers.TRS2(H, I, n / 2.0, K, p)
            f1 = nt.fusion_counts()
        finally:
            nt.set_option("fused_update", 1)
        tr = nt.solver_trace()
        res.append((srt(K.triplets()), energy, list(tr["sigma"]), list(tr["nnz"]), f1["square"] + f1["update"] - f0["square"] - f0["update"]))
    sep, blk = res
    assert sep[4] == 0 and blk[4] >= 12, (sep[4], blk[4])      # the steps after the first ran in block form
    assert sep[2] == blk[2] and sep[3] == blk[3]
    assert abs(sep[1] - blk[1]) <= 1e-12 * abs(sep[1])
    exact(blk[0], sep[0], "density, block-form steps vs separate passes")


def _digest(c, r, v):
    """order-independent digest of a set of entries (column, row, value bits): two 64-bit sums of mixed keys"""
    k = (c.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ (r.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F))
    b = np.ascontiguousarray(v, dtype=np.float64).view(np.uint64)
    with np.errstate(over="ignore"):
        h = (k ^ (b * np.uint64(0xD6E8FEB86659FD93))) * np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(31)
        return int(h.sum(dtype=np.uint64)), int((h * np.uint64(0x94D049BB133111EB)).sum(dtype=np.uint64)), len(v)


def test_lattice64_product_bit_exact_at_the_benched_size(nt, fma):
    """ONE product H * H of the 64^3 lattice Hamiltonian (N = 262 144, 50 M entries, 9.7e9 intermediate products, 140 M
    entries out at threshold 1e-8 -- the operand `bench.py --lattice 64` runs on) through the block path against the oracle's
    FMA mode on the relabelled matrix (about 12 s of oracle on the host cores): the same 140 M entries, bit for bit
    (compared through an order-independent digest of (column, row, value bits))."""
    O = fma
    L = 64
    n = L ** 3
    thr = 1e-8
    col, row, val = lattice_triplets(L)
    A = nt.Matrix_ps.from_triplets(n, col, row, val)
    nt.set_option("slab_algebra", 0)
    nt.set_option("block_path", 1)      # (the engine's own choice at this size)
    C = nt.Matrix_ps(n)
    C.Gemm(A, A, None, 1.0, 0.0, thr)
    assert nt.last_block_stats()["used"] == 1
    gc, gr, gv = C.triplets()
    del C
    pos = nt.block_order(A)
    del A
    order = np.argsort(pos, kind="stable")
    rank = np.empty(n, dtype=np.int64)
    rank[order] = np.arange(n)
    rc, rr, rv = relabel((col, row, val), rank)
    del col, row, val
    Ao = O.Mat.from_triplets(n, n, rc, rr, rv)
    del rc, rr, rv
    Co = O.ps_multiply(Ao, Ao, None, 1.0, 0.0, thr)
    oc, orow, ov = Co.triplets()
    del Co, Ao
    assert len(ov) == len(gv), (len(ov), len(gv))
    want = _digest(order[oc - 1] + 1, order[orow - 1] + 1, ov)
    got = _digest(gc, gr, gv)
    assert got == want


@pytest.mark.parametrize("solver", ["sign", "invert", "trs4", "inverse_square_root"])
def test_solver_loops_in_block_form(nt, fma, solver):
    """The loops of SignFunction, Invert, TRS4 and InverseSquareRoot on a LATTICE operand (no runs: the slab algebra
    declines): with the session on, products stay in block form and the loop's merges, scalings, copies, dots, traces and
    norms run on tiles (block algebra, spgemm_block.hpp); with it off, every product is converted back and the merges run
    on compressed columns.  Both use the block path's products and the same element rules, so sign / inverse / square
    root agree to roundoff (same iteration count, values 1e-12; TRS4, whose sigma is a quotient of dots summed in another
    order, to 1e-10), and both agree with the oracle on the caller's labels within the solver tolerances."""
    import scipy.sparse as sp
    O = fma
    L, thr = 16, 1e-6      # (12^3 is too small: its iterates fill in until the tile kernel takes them)
    n = L ** 3
    shift = 0.0 if solver in ("sign", "trs4") else 2.5
    col, row, val = lattice_triplets(L, shift=shift)
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    res = []
    for on in (0, 1):
        nt.set_option("slab_algebra", on)
        p = nt.SolverParameters()
        p.SetThreshold(thr)
        p.SetConvergeDiff(1e-30 if solver == "trs4" else 1e-7)
        if solver == "trs4":
            p.SetMaxIterations(10)
            p.SetMonitorConvergence(False)
        Out = nt.Matrix_ps(n)
        c0 = nt.block_algebra_counts()
        if solver == "sign":
            nt.SignSolvers.ComputeSign(H, Out, p)
        elif solver == "invert":
            nt.InverseSolvers.Invert(H, Out, p)
        elif solver == "trs4":
            I = nt.Matrix_ps(n)
            I.FillIdentity()
            nt.DensityMatrixSolvers.TRS4(H, I, n / 2.0, Out, p)
        else:
            nt.SquareRootSolvers.InverseSquareRoot(H, Out, p)
        c1 = nt.block_algebra_counts()
        res.append((srt(Out.triplets()), nt.solver_trace()["iterations"], c1["operations"] - c0["operations"], c1["fallbacks"] - c0["fallbacks"]))
    off, on = res
    assert off[2] == 0 and on[2] >= 2 * on[1], (off[2], on[2], on[1])     # the loop's vocabulary ran on tiles
    assert off[1] == on[1]
    G = sp.csr_matrix((on[0][2], (on[0][1] - 1, on[0][0] - 1)), shape=(n, n))
    W = sp.csr_matrix((off[0][2], (off[0][1] - 1, off[0][0] - 1)), shape=(n, n))
    # (with the session off some products of a loop take the LDS-hash kernels -- an operand like 3 I - X^2 is too sparse for
    # tiles when it arrives in compressed columns -- whose chain runs over labels: the two runs agree to roundoff, not bit
    # for bit; the element rules themselves are pinned bit for bit by test_block_form_trs2_steps_equal_the_separate_passes)
    if solver == "trs4":
        assert abs(G - W).max() <= 1e-10
    else:
        close(on[0], off[0], n, thr, solver + ": block algebra vs compressed columns", rel=1e-12)
    # the oracle on the caller's labels (tolerance: the chains run over positions)
    Ho = O.Mat.from_triplets(n, n, col, row, val)
    if solver == "trs4":
        Ko, e_o, mu_o, tro = O.density("trs4", Ho, O.Mat.identity(n), n / 2.0,
                                       O.params(converge_diff=1e-30, max_iterations=10, threshold=thr, monitor_convergence=False))
        w = srt(Ko.triplets())
        tol = 1e-6
    else:
        Oo, tro = O.matrix_function(solver, Ho, O.params(converge_diff=1e-7, threshold=thr))
        assert tro["iterations"] == on[1]
        w = srt(Oo.triplets())
        tol = 1e-9
    Wo = sp.csr_matrix((w[2], (w[1] - 1, w[0] - 1)), shape=(n, n))
    assert abs(G - Wo).max() <= tol * max(1.0, abs(Wo).max())


def test_a_second_pattern_of_the_same_dimension_gets_an_order_of_its_own(nt, fma):
    """The block order is keyed on (dimension, sparsity pattern), not on the dimension alone: a lattice in natural
    order, then the SAME lattice under a random relabelling -- another graph on the same index set -- multiplied in one
    process.  The second pattern does not inherit the clustering made for the first: it tiles as well as it does in a
    fresh process, and its product has the same bits whatever was multiplied before it."""
    from gen import random_permutation
    L, thr = 16, 1e-8
    n = L ** 3
    t1 = lattice_case(L)
    rank = random_permutation(n, 11)
    t2 = relabel(t1, rank)
    nt.set_option("slab_algebra", 0)

    def product(trip):
        A = nt.Matrix_ps.from_triplets(n, *trip)
        C = nt.Matrix_ps(n)
        C.Gemm(A, A, None, 1.0, 0.0, thr)
        bs = nt.last_block_stats()
        assert bs["used"] == 1, bs
        return srt(C.triplets()), bs["fill"], nt.block_order(A).copy()

    nt.drop_block_caches()
    alone, fill_alone, pos_alone = product(t2)          # the relabelled lattice in a fresh cache
    nt.drop_block_caches()
    first, fill_first, pos_first = product(t1)          # history: the natural lattice first ...
    after, fill_after, pos_after = product(t2)          # ... then the relabelled one
    assert fill_after >= 0.95 * fill_alone, (fill_after, fill_alone, fill_first)
    assert np.array_equal(pos_after, pos_alone)
    exact(after, alone, "relabelled lattice after the natural one vs alone")
    again, fill_again, pos_again = product(t1)          # and the first pattern finds ITS order again
    assert np.array_equal(pos_again, pos_first)
    exact(again, first, "natural lattice again")
