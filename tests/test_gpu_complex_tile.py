"""Complex run-like operands on the FP64 matrix cores (ntpoly_amd/csrc/spgemm_tile_c.hip; option complex_tile = 1, the
default under FMA arithmetic) against the oracle's complex multiply (MultiplyBlock.f90:9-36, PruneList.f90:8-38; the
reference's complex multiply-add rounds every product and sum on its own).

The kernel is a TOLERANCE mode (two FMA chains per part of an entry): entries within 1e-13 of the largest entry, the
pattern identical except where |C(i, j)| lies within that distance of the threshold.  complex_tile = 0 in the same
arithmetic mode is the bit-for-bit register-slab kernel: both are run and compared with each other as well, so a
silently unchanged kernel choice cannot pass."""
import numpy as np
import pytest

from gen import banded_triplets

pytestmark = pytest.mark.gpu
REL = 1e-13


@pytest.fixture(scope="module")
def nt():
    import ntpoly_amd as nt
    nt.init_comm()
    nt.ConstructGlobalProcessGrid(1, 1, 1)
    return nt


@pytest.fixture()
def fma(nt):
    nt.set_option("spgemm_fma", 1)
    nt.set_option("complex_tile", 1)
    yield
    nt.set_option("complex_tile", 1)
    nt.set_option("spgemm_fma", 0)


def srt(t):
    c, r, v = (np.asarray(x) for x in t)
    o = np.lexsort((r, c))
    return c[o], r[o], v[o]


def close(got, want, n, thr, what):
    import scipy.sparse as sp
    G = sp.csr_matrix((got[2], (got[1] - 1, got[0] - 1)), shape=(n, n))
    W = sp.csr_matrix((want[2], (want[1] - 1, want[0] - 1)), shape=(n, n))
    scale = max(1.0, np.abs(want[2]).max())
    D = (G - W).tocoo()
    bad = np.abs(D.data) > REL * scale
    # entries present on one side only must sit at the threshold
    assert np.all(np.abs(D.data[bad]) <= thr * (1 + 1e-9) + REL * scale), "%s: max |d| = %g" % (what, np.abs(D.data).max())
    assert abs(G.nnz - W.nnz) <= max(8, 1e-5 * W.nnz), "%s: %d vs %d entries" % (what, G.nnz, W.nnz)


def holes_pair(n, h, holes, seed):
    rng = np.random.default_rng(seed)
    mats = []
    for t in range(2):
        col, row, val = banded_triplets(n, h, shift=0.1 * t, complex_=True)
        keep = (rng.random(len(val)) >= holes) | (col == row)
        mats.append((col[keep], row[keep], val[keep] * (1.0 + 0.01 * t)))
    return mats


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("n,h,holes,thr,alpha", [(4096, 100, 0.0, 1e-8, 1.0), (3001, 37, 0.4, 0.0, 0.5), (5000, 180, 0.05, 1e-7, -0.75),
                                                 (777, 3, 0.3, 1e-3, 2.0), (4500, 240, 0.1, 1e-8, 1.0), (5000, 320, 0.0, 1e-8, 1.0),
                                                 (6144, 450, 0.05, 1e-7, 1.0)])
def test_complex_tile_kernel_vs_oracle(nt, fma, waves, n, h, holes, thr, alpha):
    """A * B with A != B and A * A on banded Hermitian-like complex operands with random holes (zero padding of the
    expanded runs, tiles without survivors, windows beyond the 1024 rows of the register-slab kernel)."""
    from oracle import oracle_py as O
    mats = holes_pair(n, h, holes, n + h)
    A = nt.Matrix_ps.from_triplets(n, *mats[0])
    B = nt.Matrix_ps.from_triplets(n, *mats[1])
    Ao = O.Mat.from_triplets(n, n, *mats[0])
    Bo = O.Mat.from_triplets(n, n, *mats[1])
    nt.set_option("tile_waves", waves)
    nt.set_option("spgemm_variant", 400)   # (force the run-based path whatever the hole density)
    try:
        for (X, Y, Xo, Yo, tag) in ((A, B, Ao, Bo, "A*B"), (A, A, Ao, Ao, "A*A")):
            C = nt.Matrix_ps(n)
            C.Gemm(X, Y, None, alpha, 0.0, thr)
            assert nt.last_spgemm_stats()["slab"] == 1
            got = srt(C.triplets())
            oc, orow, ov = O.ps_multiply(Xo, Yo, None, alpha, 0.0, thr).triplets()
            want = srt((oc, orow, ov))
            close(got, want, n, thr, "complex tile %s n=%d h=%d holes=%g" % (tag, n, h, holes))
            if 4 * h + 2 + 16 <= 1024:   # the bit-for-bit kernel takes this window: it must agree with the oracle exactly, and
                nt.set_option("complex_tile", 0)   # differ from the matrix-core kernel in the last bits of some entry
                try:
                    C0 = nt.Matrix_ps(n)
                    C0.Gemm(X, Y, None, alpha, 0.0, thr)
                finally:
                    nt.set_option("complex_tile", 1)
                g0 = srt(C0.triplets())
                assert len(g0[2]) == len(want[2]) and np.array_equal(g0[2], want[2]), "register-slab kernel vs oracle " + tag
                if h >= 30 and len(got[2]) == len(g0[2]):
                    assert not np.array_equal(got[2], g0[2]), "the two kernels returned identical bits: was the tile kernel taken?"
    finally:
        nt.set_option("spgemm_variant", -1)
        nt.set_option("tile_waves", 0)


def test_complex_tile_solver_products(nt, fma):
    """What the complex solver loops multiply (SquareRootSolversModule.F90:342-531): powers of a Hermitian operand, alpha and
    beta in play, C = alpha A B + beta C -- the Gemm vocabulary around the kernel stays what it was."""
    from oracle import oracle_py as O
    n, h, thr = 6000, 60, 1e-9
    col, row, val = banded_triplets(n, h, complex_=True)
    A = nt.Matrix_ps.from_triplets(n, col, row, val)
    Ao = O.Mat.from_triplets(n, n, col, row, val)
    A2 = nt.Matrix_ps(n)
    A2.Gemm(A, A, None, 1.0, 0.0, thr)
    A2o = O.ps_multiply(Ao, Ao, None, 1.0, 0.0, thr)
    close(srt(A2.triplets()), srt(A2o.triplets()), n, thr, "A^2")
    A3 = nt.Matrix_ps(A)
    A3.Gemm(A2, A, None, 0.25, -1.5, thr)
    A3o = O.ps_multiply(A2o, Ao, Ao, 0.25, -1.5, thr)
    close(srt(A3.triplets()), srt(A3o.triplets()), n, thr, "0.25 A^2 A - 1.5 A")


def test_complex_sign_session_keeps_the_iterates_in_slab_form(nt, fma):
    """SignFunction on a complex Hermitian operand (SignSolversModule.F90:150-258) with the loop's iterates kept in the complex
    tile kernel's operand form between products (option complex_sessions; no expansion, no pack) against the same loop on
    compressed columns.  Both are the FMA mode's tolerance arithmetic; they are not bit-identical, because the
    compressed-column loop hands the products whose right operand has become sparse inside wide extents (3 I - X^2 near
    convergence) to the general kernels with the reference's complex multiply-add: same iteration count, result within 1e-13
    of the largest entry, convergence norms equal to that accuracy, S^2 = I."""
    n, h, thr = 8000, 40, 1e-8
    col, row, val = banded_triplets(n, h, complex_=True)
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    res = {}
    for opt in (1, 0):
        nt.set_option("complex_sessions", opt)
        try:
            p = nt.SolverParameters()
            p.SetThreshold(thr)
            p.SetConvergeDiff(1e-9)
            S = nt.Matrix_ps(n)
            c0 = nt.slab_algebra_counts()
            nt.SignSolvers.ComputeSign(H, S, p)
            c1 = nt.slab_algebra_counts()
            tr = nt.solver_trace()
            res[opt] = (srt(S.triplets()), tr["iterations"], np.asarray(tr["value"]), {k: c1[k] - c0[k] for k in c0}, S)
        finally:
            nt.set_option("complex_sessions", 1)
    it = res[1][1]
    assert it == res[0][1] and it >= 5
    # (the generated H stores a few exact zeros, which a slab form would read as "no entry": the first iteration's two
    # products are refused and run on compressed columns; every later operand is a product, zero-free by construction)
    assert res[1][3]["products"] >= 2 * (it - 1) and res[1][3]["refusals"] <= 2, res[1][3]
    assert res[1][3]["merges"] >= it - 1 and res[1][3]["others"] >= it - 1, res[1][3]
    assert res[0][3]["products"] == 0
    # the norms are sums of differences of nearly equal iterates: roundoff-sized terms at convergence
    assert np.allclose(res[1][2], res[0][2], rtol=1e-9, atol=1e-12 * n)
    close(res[1][0], res[0][0], n, thr, "sign with / without the complex session")
    # and it is the sign function: S^2 = I
    S = res[1][4]
    S2 = nt.Matrix_ps(n)
    S2.Gemm(S, S, None, 1.0, 0.0, thr)
    Ident = nt.Matrix_ps(n)
    Ident.FillIdentity()
    S2.Increment(Ident, -1.0, 0.0)
    assert S2.Norm() <= 1e-5


@pytest.mark.parametrize("solver", ["inverse_square_root", "square_root", "invert"])
def test_complex_sessions_of_the_inverse_and_square_root_loops(nt, fma, solver):
    """InverseSquareRoot / SquareRoot (order 5: SquareRootSolversModule.F90:342-531) and Invert (InverseSolversModule.F90:29-149)
    on a complex Hermitian positive definite operand with the loop's matrices kept in the complex tile kernel's operand form
    between the operations -- products, merges (AddSparseVectors rules, threshold on the modulus), scalings, copies, identity
    increments, norms on runs of (re, im) pairs -- against the same loop on compressed columns: the same iteration count,
    convergence values to roundoff, the result within 1e-13 of the largest entry, and F(A) what it should be."""
    n, h, thr = 8000, 30, 1e-8
    col, row, val = banded_triplets(n, h, complex_=True, shift=2.5)
    A = nt.Matrix_ps.from_triplets(n, col, row, val)
    res = {}
    for opt in (1, 0):
        nt.set_option("complex_sessions", opt)
        try:
            p = nt.SolverParameters()
            p.SetThreshold(thr)
            p.SetConvergeDiff(1e-8)
            Out = nt.Matrix_ps(n)
            c0 = nt.slab_algebra_counts()
            if solver == "inverse_square_root":
                nt.SquareRootSolvers.InverseSquareRoot(A, Out, p)
            elif solver == "square_root":
                nt.SquareRootSolvers.SquareRoot(A, Out, p)
            else:
                nt.InverseSolvers.Invert(A, Out, p)
            c1 = nt.slab_algebra_counts()
            tr = nt.solver_trace()
            res[opt] = (srt(Out.triplets()), tr["iterations"], np.asarray(tr["value"]), {k: c1[k] - c0[k] for k in c0}, Out)
        finally:
            nt.set_option("complex_sessions", 1)
    it = res[1][1]
    assert it == res[0][1] and it >= 3, (res[1][1], res[0][1])
    on, off = res[1][3], res[0][3]
    assert off["products"] == 0
    assert on["products"] >= 2 * (it - 1) and on["merges"] >= it - 1 and on["refusals"] <= 3, on
    assert np.allclose(res[1][2], res[0][2], rtol=1e-8, atol=1e-12 * n), (res[1][2], res[0][2])
    close(res[1][0], res[0][0], n, thr, solver + " with / without the complex session")
    # what it is: Out A Out = I, Out Out = A, Out A = I
    F = res[1][4]
    T, U = nt.Matrix_ps(n), nt.Matrix_ps(n)
    Ident = nt.Matrix_ps(n)
    Ident.FillIdentity()
    if solver == "inverse_square_root":
        T.Gemm(F, A, None, 1.0, 0.0, thr)
        U.Gemm(T, F, None, 1.0, 0.0, thr)
        U.Increment(Ident, -1.0, 0.0)
    elif solver == "square_root":
        U.Gemm(F, F, None, 1.0, 0.0, thr)
        U.Increment(A, -1.0, 0.0)
    else:
        U.Gemm(F, A, None, 1.0, 0.0, thr)
        U.Increment(Ident, -1.0, 0.0)
    assert U.Norm() <= 1e-4, U.Norm()
