"""IncrementMatrix(Identity, B, alpha) in place and the norm of a difference without forming it, on compressed columns
(ntpoly_amd/csrc/column_fused.hip) against the merge path of the same engine (itself pinned to the reference's
AddSparseVectors.f90 rules by tests/test_gpu_parity.py): the in-place increment BIT-EXACT including every case that must
fall back, the norm to 1e-14 relative (another summation order), the complex solver loops unchanged by the option."""
import numpy as np
import pytest

from gen import banded_triplets

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nt():
    import ntpoly_amd as nt
    nt.init_comm()
    nt.ConstructGlobalProcessGrid(1, 1, 1)
    return nt


def srt(t):
    c, r, v = (np.asarray(x) for x in t)
    o = np.lexsort((r, c))
    return c[o], r[o], v[o]


def same(a, b, what):
    a, b = srt(a), srt(b)
    assert len(a[2]) == len(b[2]), "%s: %d vs %d entries" % (what, len(a[2]), len(b[2]))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), what + ": pattern"
    assert np.array_equal(a[2], b[2]), what + ": values"


def cases(n, h, cplx):
    col, row, val = banded_triplets(n, h, complex_=cplx)
    rng = np.random.default_rng(n)
    keep = rng.random(len(val)) > 0.2
    keep |= col == row
    full = (col[keep], row[keep], val[keep])
    yield "every diagonal stored", full, 3.0, True
    yield "negative alpha", full, -0.375, True
    nodiag = keep & ~((col == row) & (col % 97 == 5))
    yield "columns without diagonal", (col[nodiag], row[nodiag], val[nodiag]), 3.0, False
    v0 = val[keep].copy()
    c0, r0 = col[keep], row[keep]
    above = np.flatnonzero((r0 < c0) & (c0 % 211 == 7))
    v0[above[:3]] = 0.0
    yield "stored zeros above the diagonal", (c0, r0, v0), 3.0, False
    v1 = val[keep].copy()
    below = np.flatnonzero((r0 > c0) & (c0 % 211 == 9))
    v1[below[:3]] = 0.0
    yield "stored zeros below the diagonal (kept by the tail rule)", (c0, r0, v1), 3.0, True
    v2 = val[keep].copy()
    d = np.flatnonzero(c0 == r0)
    v2[d[10]] = -3.0
    yield "a diagonal sum of exactly zero", (c0, r0, v2), 3.0, False


@pytest.mark.parametrize("cplx", [False, True])
def test_increment_identity_in_place(nt, cplx):
    n, h = 3000, 40
    nt.set_option("slab_algebra", 0)   # (real operands: compressed columns, as the complex ones always are)
    try:
        Ident = nt.Matrix_ps(n)
        Ident.FillIdentity()
        for what, tri, alpha, in_place in cases(n, h, cplx):
            B = nt.Matrix_ps.from_triplets(n, *tri)
            Bref = nt.Matrix_ps(B)
            before = nt.column_fused_counts()["identity_in_place"]
            nt.increment_identity(Ident, B, alpha)
            assert (nt.column_fused_counts()["identity_in_place"] - before == 1) == in_place, what
            Bref.Increment(Ident, alpha, 0.0)            # the merge kernel (AddSparseVectors rules)
            same(B.triplets(), Bref.triplets(), "%s (complex %d)" % (what, cplx))
    finally:
        nt.set_option("slab_algebra", 1)


@pytest.mark.parametrize("cplx", [False, True])
def test_norm_of_a_difference(nt, cplx):
    n = 5000
    nt.set_option("slab_algebra", 0)
    try:
        rng = np.random.default_rng(5)
        mats = []
        for t, h in enumerate((60, 45)):
            col, row, val = banded_triplets(n, h, complex_=cplx, shift=0.2 * t)
            keep = rng.random(len(val)) > 0.3
            keep &= ~np.isin(col, [11, 12, n - 1] if t == 0 else [12, 40])   # empty columns on either side
            mats.append(nt.Matrix_ps.from_triplets(n, col[keep], row[keep], val[keep] * (1 + 0.1 * t)))
        A, B = mats
        for alpha in (-1.0, 0.5):
            got = nt.norm_axpby(A, B, alpha, 1.0)
            assert got is not None
            D = nt.Matrix_ps(B)
            D.Increment(A, alpha, 0.0)
            want = D.Norm()
            assert got == pytest.approx(want, rel=1e-14)
        assert nt.norm_axpby(A, B, -1.0, 2.0) is None     # (beta != 1: the caller forms the combination)
    finally:
        nt.set_option("slab_algebra", 1)


@pytest.mark.parametrize("solver", ["sign", "isq"])
def test_complex_loops_unchanged_by_the_option(nt, solver):
    """SignFunction / InverseSquareRoot on a complex Hermitian operand: the same iterates bit for bit with the fused column
    operations on and off (the in-place increment is exact; the norms differ in their last bits and only feed the
    convergence test), the same iteration count, norms to 1e-13."""
    n, h, thr = 6000, 30, 1e-8
    col, row, val = banded_triplets(n, h, complex_=True, shift=2.0 if solver == "isq" else 0.0)
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    res = {}
    for opt in (1, 0):
        nt.set_option("column_fused", opt)
        try:
            p = nt.SolverParameters()
            p.SetThreshold(thr)
            p.SetConvergeDiff(1e-9)
            K = nt.Matrix_ps(n)
            c0 = nt.column_fused_counts()
            if solver == "sign":
                nt.SignSolvers.ComputeSign(H, K, p)
            else:
                nt.SquareRootSolvers.InverseSquareRoot(H, K, p)
            c1 = nt.column_fused_counts()
            tr = nt.solver_trace()
            res[opt] = (K.triplets(), tr["iterations"], np.asarray(tr["value"]), {k: c1[k] - c0[k] for k in c0})
        finally:
            nt.set_option("column_fused", 1)
    assert res[1][3]["identity_in_place"] >= res[1][1] and res[0][3]["identity_in_place"] == 0
    if solver == "sign":
        assert res[1][3]["norms_of_differences"] >= res[1][1] - 1
    assert res[1][1] == res[0][1] and res[1][1] >= 5
    assert np.allclose(res[1][2], res[0][2], rtol=1e-13, atol=0)
    same(res[1][0], res[0][0], solver + " result")
