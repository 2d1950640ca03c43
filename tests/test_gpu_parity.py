"""GPU parity tests: the HIP engine, called through its C ABI (ntpoly_amd.host is a thin ctypes
mirror of the reference's class surface), against
  (1) the golden vectors produced by the REAL reference (tests/golden/*.npz), and
  (2) the C oracle (oracle/) on seeded inputs at sizes it finishes in seconds.
Sparse-branch SpGEMM and increment must be BIT-EXACT (same ascending-k, unfused arithmetic);
reductions (dot/trace/norm) are compared to 1e-13 relative (different summation tree);
the reference's BLAS dense branch to 1e-13 relative."""
import numpy as np
import pytest

from golden_util import Golden, same_pattern, to_dense
from gen import banded_triplets

pytestmark = pytest.mark.gpu
DENSE_RTOL = 1e-13


@pytest.fixture(scope="module")
def nt():
    import ntpoly_amd as nt
    nt.init_comm()
    nt.ConstructGlobalProcessGrid(1, 1, 1)
    return nt


def exact(got, want, what):
    assert same_pattern((0, 0) + tuple(got), want), "%s: pattern differs (%d vs %d nnz)" % (what, len(got[0]), len(want[2]))
    assert np.array_equal(got[2], want[4]), "%s: values differ, max |d| = %g" % (what, np.abs(got[2] - want[4]).max())


def close(got, want, rtol, what):
    g = to_dense((want[0], want[1]) + tuple(got))
    w = to_dense(want)
    assert np.abs(g - w).max() <= rtol * max(1.0, np.abs(w).max()), "%s: max |d| = %g" % (what, np.abs(g - w).max())


def lmat(nt, t):
    cls = nt.Matrix_lsc if np.iscomplexobj(t[4]) else nt.Matrix_lsr
    return cls.from_triplets(t[0], t[1], t[2], t[3], t[4])


def pmat(nt, t):
    return nt.Matrix_ps.from_triplets(t[0], t[2], t[3], t[4])


def test_local_gemm_golden(nt):
    """test_matrix.py:224-362 through MatrixMultiply_ls{r,c}_wrp"""
    g = Golden("local_gemm")
    n_exact = 0
    for i, c in enumerate(g.cases):
        A, B = g.tri(i, "A"), g.tri(i, "B")
        want = g.tri(i, "C")
        is_c = c["complex"]
        cls = nt.Matrix_lsc if is_c else nt.Matrix_lsr
        mA = cls.from_triplets(A[0], A[1], A[2], A[3], A[4].astype(complex) if is_c else A[4])
        mB = cls.from_triplets(B[0], B[1], B[2], B[3], B[4].astype(complex) if is_c else B[4])
        if g.has(i, "Cin"):
            Ci = g.tri(i, "Cin")
            mC = cls.from_triplets(Ci[0], Ci[1], Ci[2], Ci[3], Ci[4].astype(complex) if is_c else Ci[4])
        else:
            mC = cls(want[0], want[1])
        mC.Gemm(mA, mB, bool(c["tA"]), bool(c["tB"]), c["alpha"], 0.0 if c["beta"] is None else c["beta"], c["thr"])
        got = mC.triplets()
        sp_a = len(A[2]) / float(max(1, A[0] * A[1]))
        sp_b = len(B[2]) / float(max(1, B[0] * B[1]))
        if min(sp_a, sp_b) > 0.1:
            if c["thr"] == 0.0:
                close(got, want, DENSE_RTOL, "case %d (dense branch)" % i)
        else:
            exact(got, want, "case %d %s" % (i, c))
            n_exact += 1
    assert n_exact >= 40


def test_local_increment_golden(nt):
    g = Golden("local_increment")
    for i, c in enumerate(g.cases):
        mA, mB = lmat(nt, g.tri(i, "A")), lmat(nt, g.tri(i, "B"))
        if c["complex"] and not np.iscomplexobj(g.tri(i, "A")[4]):
            continue
        mB.Increment(mA, c["alpha"], c["thr"])
        exact(mB.triplets(), g.tri(i, "C"), "case %d %s" % (i, c))


@pytest.mark.parametrize("force_bin,variant", [(-1, -1), (-1, 0), (2, 321), (3, 361), (-1, 351), (4, -1), (5, -1), (6, -1), (-1, 400),
                                               (-1, 500), (-1, 501)])
def test_ps_gemm_golden(nt, force_bin, variant):
    """test_psmatrixalgebra.py:193-218 through MatrixMultiply_ps_wrp; every kernel path (column-pair
    kernel, first-generation window kernel, other generations, LDS window sizes, LDS hash, HBM
    accumulator, grouped LDS hash) must give the same bits."""
    nt.set_option("spgemm_force_bin", force_bin)
    nt.set_option("spgemm_variant", variant)
    try:
        g = Golden("ps_gemm")
        n_exact = 0
        for i, c in enumerate(g.cases):
            mA, mB = pmat(nt, g.tri(i, "A")), pmat(nt, g.tri(i, "B"))
            want = g.tri(i, "C")
            mC = pmat(nt, g.tri(i, "Cin")) if g.has(i, "Cin") else nt.Matrix_ps(want[0])
            mC.Gemm(mA, mB, None, c["alpha"], c["beta"], c["thr"])
            got = mC.triplets()
            if c["dense_branch"]:
                close(got, want, DENSE_RTOL, "case %d %s" % (i, c["tag"]))
            else:
                exact(got, want, "case %d %s (force_bin %d)" % (i, c["tag"], force_bin))
                n_exact += 1
        assert n_exact >= 20
    finally:
        nt.set_option("spgemm_force_bin", -1)
        nt.set_option("spgemm_variant", -1)


@pytest.mark.parametrize("force_seq", [0, 1, 2])   # automatic / sequential merge / rank merge
def test_ps_increment_golden(nt, force_seq):
    nt.set_option("increment_force_seq", force_seq)
    try:
        g = Golden("ps_increment")
        for i, c in enumerate(g.cases):
            mA, mB = pmat(nt, g.tri(i, "A")), pmat(nt, g.tri(i, "B"))
            mB.Increment(mA, c["alpha"], c["thr"])
            exact(mB.triplets(), g.tri(i, "C"), "case %d" % i)
    finally:
        nt.set_option("increment_force_seq", 0)


def test_ps_scalars_golden(nt):
    g = Golden("ps_scalars")
    for i, c in enumerate(g.cases):
        A, B = pmat(nt, g.tri(i, "A")), pmat(nt, g.tri(i, "B"))
        assert A.Trace() == pytest.approx(c["trace"], rel=1e-13, abs=1e-13)
        assert A.Norm() == pytest.approx(c["norm"], rel=1e-13)
        assert np.real(A.Dot(B)) == pytest.approx(c["dot_real"], rel=1e-12, abs=1e-12)
        emin, emax = nt.EigenBounds.GershgorinBounds(A)
        assert emin == pytest.approx(c["gersh_min"], rel=1e-13)
        assert emax == pytest.approx(c["gersh_max"], rel=1e-13)
        assert A.GetSize() == c["nnz"]


def _params(nt, c):
    p = nt.SolverParameters()
    p.SetConvergeDiff(c["conv"])
    p.SetMaxIterations(c["maxit"])
    p.SetThreshold(c["thr"])
    p.SetMonitorConvergence(c["monitor"])
    return p


def test_solvers_golden(nt):
    """test_chemistry.py:193-233,266-276 and test_solvers.py:139-162,211-234,364-386 with seeded
    inputs and the reference's own outputs / per-iteration log as the expected values."""
    g = Golden("solvers")
    for i, c in enumerate(g.cases):
        H = pmat(nt, g.tri(i, "H"))
        p = _params(nt, c)
        want = g.tri(i, "K")
        K = nt.Matrix_ps(want[0])
        if c["solver"] in ("trs2", "trs4"):
            if c["isq"] == "identity":
                ISQ = nt.Matrix_ps(want[0])
                ISQ.FillIdentity()
                if H.IsComplex():
                    pass
            else:
                ISQ = pmat(nt, g.tri(i, "ISQ"))
            fn = nt.DensityMatrixSolvers.TRS2 if c["solver"] == "trs2" else nt.DensityMatrixSolvers.TRS4
            energy, mu = fn(H, ISQ, c["nel"], K, p)
            tr = nt.solver_trace()
            log_e = g.arr(i, "log_energy")
            n = len(log_e)
            assert tr["iterations"] in (n, n + 1), (c["tag"], tr["iterations"], n)
            assert np.allclose(tr["energy"][:n], log_e, rtol=1e-11, atol=1e-11), c["tag"]
            assert energy == pytest.approx(c["energy"], rel=1e-11), c["tag"]
            mu_tol = 1e-9 if c["solver"] == "trs2" else 1e-5
            assert mu == pytest.approx(c["mu"], rel=mu_tol, abs=1e-11), c["tag"]
        else:
            fn = dict(sign=nt.SignSolvers.ComputeSign, invert=nt.InverseSolvers.Invert,
                      isq=nt.SquareRootSolvers.InverseSquareRoot, sqrt=nt.SquareRootSolvers.SquareRoot)[c["solver"]]
            fn(H, K, p)
            tr = nt.solver_trace()
            log_c = g.arr(i, "log_convergence")
            if c["solver"] == "invert":
                log_c = log_c[::2]
            assert tr["iterations"] == len(log_c), (c["tag"], tr["iterations"], len(log_c))
            assert np.allclose(tr["value"], log_c, rtol=1e-9, atol=1e-13), c["tag"]
        got = K.triplets()
        gd = to_dense((want[0], want[1]) + tuple(got))
        wd = to_dense(want)
        tol = max(10 * c["thr"], 1e-10) * max(1.0, np.abs(wd).max())
        assert np.abs(gd - wd).max() <= tol, "%s: max |d| = %g" % (c["tag"], np.abs(gd - wd).max())
        assert abs(K.GetSize() - c["nnz"]) <= max(2, 0.01 * c["nnz"]), (c["tag"], K.GetSize(), c["nnz"])


def test_premade_fixture(nt):
    """the reference's shipped Examples/PremadeMatrix output (nel = 5)"""
    g = Golden("solvers")
    idx = [i for i, c in enumerate(g.cases) if c["tag"] == "premade_trs2_nel5"][0]
    c = g.cases[idx]
    H, ISQ = pmat(nt, g.tri(idx, "H")), pmat(nt, g.tri(idx, "ISQ"))
    K = nt.Matrix_ps(7)
    energy, mu = nt.DensityMatrixSolvers.TRS2(H, ISQ, 5.0, K, _params(nt, c))
    Dref = to_dense(g.tri(None, "premade_density_reference"))
    assert np.linalg.norm(K.to_scipy().toarray() - Dref) <= 5e-5
    assert energy == pytest.approx(-22.971963210096895, rel=1e-10)
    assert mu == pytest.approx(0.1491684406929008, rel=1e-8)
    assert nt.solver_trace()["iterations"] == 19


def test_load_balanced_solver_matches(nt):
    """test_chemistry.py:203-205: a random permutation must not change the result"""
    g = Golden("solvers")
    idx = [i for i, c in enumerate(g.cases) if c["tag"] == "banded512_trs2_conv"][0]
    c = g.cases[idx]
    H = pmat(nt, g.tri(idx, "H"))
    ISQ = nt.Matrix_ps(512)
    ISQ.FillIdentity()
    p = _params(nt, c)
    perm = nt.Permutation(512)
    perm.SetRandomPermutation()
    p.SetLoadBalance(perm)
    K = nt.Matrix_ps(512)
    energy, mu = nt.DensityMatrixSolvers.TRS2(H, ISQ, c["nel"], K, p)
    assert energy == pytest.approx(c["energy"], rel=1e-10)
    wd = to_dense(g.tri(idx, "K"))
    assert np.abs(K.to_scipy().toarray() - wd).max() <= 10 * c["thr"]
    # engine option load_balance = 0: the permutation is not applied (the solve stays in the caller's ordering and on
    # the run-based kernels); the result is the same up to summation order
    nt.set_option("load_balance", 0)
    try:
        p2 = _params(nt, c)
        p2.SetLoadBalance(perm)
        K2 = nt.Matrix_ps(512)
        e2, mu2 = nt.DensityMatrixSolvers.TRS2(H, ISQ, c["nel"], K2, p2)
    finally:
        nt.set_option("load_balance", 1)
    assert e2 == pytest.approx(c["energy"], rel=1e-10)
    assert np.abs(K2.to_scipy().toarray() - wd).max() <= 10 * c["thr"]


def test_spgemm_vs_oracle_banded_4096(nt):
    """config-2 family at a size the oracle does in well under a second: bit-exact"""
    from oracle import oracle_py as O
    n, h = 4096, 50
    col, row, val = banded_triplets(n, h)
    A = nt.Matrix_ps.from_triplets(n, col, row, val)
    Co = O.ps_multiply(O.Mat.from_triplets(n, n, col, row, val), O.Mat.from_triplets(n, n, col, row, val), None, 1.0, 0.0, 1e-8)
    C = nt.Matrix_ps(n)
    nt.set_option("time_kernels", 1)   # statistics (product count) on
    try:
        C.Gemm(A, A, None, 1.0, 0.0, 1e-8)
        st = nt.last_spgemm_stats()
    finally:
        nt.set_option("time_kernels", 0)
    oc, orow, ov = Co.triplets()
    exact(C.triplets(), (n, n, oc, orow, ov), "banded 4096")
    assert st["nnz_c"] == len(ov) and st["products"] > 4e7


def test_trs2_vs_oracle_banded_4096(nt):
    """BASELINE.md golden scalars of the reference itself (N=4096, h=50, thr 1e-8, ISQ = I):
    energy after 2 / 8 iterations, nnz(K) after 8."""
    n, h = 4096, 50
    col, row, val = banded_triplets(n, h)
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    ISQ = nt.Matrix_ps(n)
    ISQ.FillIdentity()
    for iters, e_ref in ((2, -5.80031452295105E+02), (8, -1.06326987302440E+03)):
        p = nt.SolverParameters()
        p.SetConvergeDiff(1e-30)
        p.SetThreshold(1e-8)
        p.SetMaxIterations(iters)
        p.SetMonitorConvergence(False)
        K = nt.Matrix_ps(n)
        energy, mu = nt.DensityMatrixSolvers.TRS2(H, ISQ, n / 2.0, K, p)
        assert energy == pytest.approx(e_ref, rel=1e-12)
    assert K.GetSize() == 940848


def test_full_size_properties_config2(nt):
    """BASELINE config 2 (N=65536, 101 nnz/row): size-independent checks at full size --
    symmetry of A*A for symmetric A, trace(A*A) == dot(A, A^T) and exact agreement of two kernel
    paths (LDS window vs LDS hash) -- plus bit-exact agreement with the oracle."""
    from oracle import oracle_py as O
    n, h = 65536, 50
    col, row, val = banded_triplets(n, h)
    A = nt.Matrix_ps.from_triplets(n, col, row, val)
    C = nt.Matrix_ps(n)
    nt.set_option("time_kernels", 1)   # statistics (product count) on
    try:
        C.Gemm(A, A, None, 1.0, 0.0, 1e-8)
        st = nt.last_spgemm_stats()
    finally:
        nt.set_option("time_kernels", 0)
    assert st["products"] > 6.6e8
    AT = nt.Matrix_ps(n)
    AT.Transpose(A)
    assert C.Trace() == pytest.approx(A.Dot(AT), rel=1e-12)
    assert C.MeasureAsymmetry() <= 1e-15
    got = C.triplets()
    assert st["slab"] == 1   # run-like columns: the register-slab kernel is the default here
    for (fb, var) in ((5, -1), (-1, 0), (-1, 351)):   # LDS hash path, first-generation window kernel, column-pair kernel
        nt.set_option("spgemm_force_bin", fb)
        nt.set_option("spgemm_variant", var)
        try:
            C2 = nt.Matrix_ps(n)
            C2.Gemm(A, A, None, 1.0, 0.0, 1e-8)
        finally:
            nt.set_option("spgemm_force_bin", -1)
            nt.set_option("spgemm_variant", -1)
        g2 = C2.triplets()
        assert np.array_equal(got[0], g2[0]) and np.array_equal(got[1], g2[1]) and np.array_equal(got[2], g2[2])
    Ao = O.Mat.from_triplets(n, n, col, row, val)
    oc, orow, ov = O.ps_multiply(Ao, Ao, None, 1.0, 0.0, 1e-8).triplets()
    exact(got, (n, n, oc, orow, ov), "config 2 vs oracle")


@pytest.mark.parametrize("n,h,holes,thr,cplx", [(4096, 100, 0.0, 1e-8, False), (4096, 140, 0.2, 1e-6, False),
                                                  (3000, 30, 0.5, 0.0, False), (5000, 250, 0.05, 1e-7, False),
                                                  (4096, 100, 0.0, 1e-8, True), (3001, 37, 0.4, 0.0, True),
                                                  (5000, 180, 0.05, 1e-7, True), (777, 3, 0.3, 1e-3, True),
                                                  (4500, 240, 0.0, 1e-8, False), (4500, 230, 0.1, 1e-7, False),
                                                  (4500, 240, 0.1, 1e-8, True), (5000, 320, 0.0, 1e-8, False),
                                                  (5000, 250, 0.0, 1e-8, True)])
def test_slab_kernel_vs_oracle(nt, n, h, holes, thr, cplx):
    """register-slab SpGEMM kernels (forced; real and complex) on banded operands with random holes punched into
    the band, A*B with A != B, against the oracle: bit-exact.  Holes exercise the zero padding of the expanded runs."""
    from oracle import oracle_py as O
    rng = np.random.default_rng(n + h)
    mats = []
    for t in range(2):
        col, row, val = banded_triplets(n, h, shift=0.1 * t, complex_=cplx)
        keep = (rng.random(len(val)) >= holes) | (col == row)
        mats.append((col[keep], row[keep], val[keep] * (1.0 + 0.01 * t)))
    A = nt.Matrix_ps.from_triplets(n, *mats[0])
    B = nt.Matrix_ps.from_triplets(n, *mats[1])
    nt.set_option("spgemm_variant", 400)
    try:
        C = nt.Matrix_ps(n)
        C.Gemm(A, B, None, 0.5, 0.0, thr)
        used = nt.last_spgemm_stats()["slab"]
    finally:
        nt.set_option("spgemm_variant", -1)
    if 4 * h + 2 + 16 <= (1024 if cplx else 1536) and 2 * h + 1 + 16 <= 1023:   # row window and multiplier-tile limits
        assert used == 1
    Ao = O.Mat.from_triplets(n, n, *mats[0])
    Bo = O.Mat.from_triplets(n, n, *mats[1])
    oc, orow, ov = O.ps_multiply(Ao, Bo, None, 0.5, 0.0, thr).triplets()
    exact(C.triplets(), (n, n, oc, orow, ov), "slab vs oracle n=%d h=%d holes=%g cplx=%d" % (n, h, holes, cplx))


def test_slab_kernel_fma_option(nt):
    """option spgemm_fma = 1: products accumulated with v_fma_f64 (one rounding per product, what a reference
    built with floating-point contraction computes).  Not bit-identical to the unfused reference build:
    same sparsity pattern away from the threshold, values within 1e-13 relative of the oracle."""
    from oracle import oracle_py as O
    n, h, thr = 4096, 100, 1e-8
    col, row, val = banded_triplets(n, h)
    A = nt.Matrix_ps.from_triplets(n, col, row, val)
    nt.set_option("spgemm_fma", 1)
    try:
        C = nt.Matrix_ps(n)
        C.Gemm(A, A, None, 1.0, 0.0, thr)
        assert nt.last_spgemm_stats()["slab"] == 1
    finally:
        nt.set_option("spgemm_fma", 0)
    Ao = O.Mat.from_triplets(n, n, col, row, val)
    oc, orow, ov = O.ps_multiply(Ao, Ao, None, 1.0, 0.0, thr).triplets()
    got = C.to_scipy()
    import scipy.sparse as sp
    want = sp.csc_matrix((ov, (orow - 1, oc - 1)), shape=(n, n))
    d = abs(got - want)
    scale = abs(want).max()
    assert d.max() <= max(1e-13 * scale, 2 * thr * 1e-6) or d.max() <= 1.000001 * thr  # entries straddling the threshold
    big = abs(want) > 10 * thr
    assert abs((got - want).multiply(big)).max() <= 1e-13 * scale


def test_full_size_properties_config3_panel(nt):
    """BASELINE configs[3] operand (N = 1 048 576, 201 nnz/row; the 8-GPU config) as ONE A*A on one GPU:
    size-independent checks at full size -- exact symmetry of A*A for symmetric A, trace(A*A) == dot(A, A^T),
    bit-exact agreement of the register-slab kernel with the column-pair kernel, and the analytic product
    count of a band."""
    n, h = 1048576, 100
    col, row, val = banded_triplets(n, h)
    A = nt.Matrix_ps.from_triplets(n, col, row, val)
    nnz_a = len(val)
    del col, row, val
    C = nt.Matrix_ps(n)
    nt.set_option("time_kernels", 1)
    try:
        C.Gemm(A, A, None, 1.0, 0.0, 1e-8)
        st = nt.last_spgemm_stats()
    finally:
        nt.set_option("time_kernels", 0)
    assert st["slab"] == 1 and st["nnz_a"] == nnz_a
    w = 2 * h + 1
    ip_interior = n * w * w   # minus the band truncation at the two ends
    assert 0.999 * ip_interior < st["products"] <= ip_interior
    assert C.MeasureAsymmetry() == 0.0
    AT = nt.Matrix_ps(n)
    AT.Transpose(A)
    assert C.Trace() == pytest.approx(A.Dot(AT), rel=1e-12)
    got = C.triplets()
    nt.set_option("spgemm_variant", 351)
    try:
        C2 = nt.Matrix_ps(n)
        C2.Gemm(A, A, None, 1.0, 0.0, 1e-8)
    finally:
        nt.set_option("spgemm_variant", -1)
    g2 = C2.triplets()
    assert all(np.array_equal(u, v) for u, v in zip(got, g2))


def test_full_size_properties_config4_complex(nt):
    """BASELINE configs[4] (Hermitian complex N = 131 072, 101 nnz/row, H + 2I): SignFunction and
    InverseSquareRoot at full size, checked through what they must satisfy: sign(H)^2 = I and
    Z H Z = I up to the solver threshold, Hermitian results."""
    n, h, thr = 131072, 50, 1e-8
    col, row, val = banded_triplets(n, h, complex_=True, shift=2.0)
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    del col, row, val
    p = nt.SolverParameters()
    p.SetThreshold(thr)
    p.SetConvergeDiff(1e-10)
    Ident = nt.Matrix_ps(n)
    Ident.FillIdentity()

    S = nt.Matrix_ps(n)
    nt.SignSolvers.ComputeSign(H, S, p)
    S2 = nt.Matrix_ps(n)
    S2.Gemm(S, S, None, 1.0, 0.0, thr)
    S2.Increment(Ident, -1.0, 0.0)
    assert S2.Norm() <= 1e-5
    assert S.MeasureAsymmetry() <= 1e-5

    Z = nt.Matrix_ps(n)
    nt.SquareRootSolvers.InverseSquareRoot(H, Z, p)
    T = nt.Matrix_ps(n)
    T.Gemm(Z, H, None, 1.0, 0.0, thr)
    R = nt.Matrix_ps(n)
    R.Gemm(T, Z, None, 1.0, 0.0, thr)
    R.Increment(Ident, -1.0, 0.0)
    assert R.Norm() <= 1e-5
    assert Z.MeasureAsymmetry() <= 1e-5   # Newton-Schulz with threshold 1e-8 is Hermitian to the pruning error only


def test_fma_option_matches_contracted_reference_build(nt):
    """spgemm_fma = 1 against the reference compiled WITH floating-point contraction (tests/golden/ps_gemm_fma.npz,
    make_golden.py ps_gemm_fma; oracle/build_ref.py --fma): bit-identical.  So each arithmetic mode of the engine
    equals a build of the reference: the default its Linux/x86-64 build (no FMA), the option its FMA builds."""
    g = Golden("ps_gemm_fma")
    nt.set_option("spgemm_fma", 1)
    nt.set_option("spgemm_variant", 400)   # the option lives in the register-slab kernel: take it whenever it fits
    try:
        for i, c in enumerate(g.cases):
            A = pmat(nt, g.tri(i, "A"))
            B = A if c["same"] else pmat(nt, g.tri(i, "B"))
            want = g.tri(i, "C")
            C = nt.Matrix_ps(want[0])
            C.Gemm(A, B, None, c["alpha"], 0.0, c["thr"])
            assert nt.last_spgemm_stats()["slab"] == 1, c["tag"]
            exact(C.triplets(), want, "fma " + c["tag"])
    finally:
        nt.set_option("spgemm_fma", 0)
        nt.set_option("spgemm_variant", -1)
    # and the default mode must NOT equal it (the two builds of the reference differ in the last bits)
    i = 1
    A = pmat(nt, g.tri(i, "A"))
    C = nt.Matrix_ps(g.tri(i, "C")[0])
    C.Gemm(A, A, None, 1.0, 0.0, g.cases[i]["thr"])
    got, want = C.triplets(), g.tri(i, "C")
    assert not (len(got[2]) == len(want[4]) and np.array_equal(got[2], want[4]))


def test_full_size_properties_config2_trs2(nt):
    """BASELINE configs[2] (the bench workload: N = 262 144, 201 nnz/row, threshold 1e-8, ISQ = I, trace = N/2) as a
    whole converged TRS2 solve, checked through what does not depend on the size: the density is symmetric,
    idempotent up to the threshold, has the requested trace, energy = dot(K, H); and two independent kernel
    implementations (register-slab with the product handed uncompacted to the update, vs the LDS column-pair kernel
    with the unfused call sequence) give the same energies at every iteration and the same density, bit for bit."""
    n, h, thr = 262144, 100, 1e-8
    col, row, val = banded_triplets(n, h)
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    del col, row, val
    ISQ = nt.Matrix_ps(n)
    ISQ.FillIdentity()
    p = nt.SolverParameters()
    p.SetThreshold(thr)
    p.SetConvergeDiff(1e-6)
    K = nt.Matrix_ps(n)
    energy, mu = nt.DensityMatrixSolvers.TRS2(H, ISQ, n / 2.0, K, p)
    tr = nt.solver_trace()
    assert 10 <= tr["iterations"] <= 60
    assert K.Trace() == pytest.approx(n / 2.0, abs=5e-2)
    assert K.MeasureAsymmetry() <= 1e-12
    assert energy == pytest.approx(float(np.real(K.Dot(H))), rel=1e-12)
    K2 = nt.Matrix_ps(n)
    K2.Gemm(K, K, None, 1.0, 0.0, thr)
    K2.Increment(K, -1.0, 0.0)
    assert K2.Norm() <= 1e-4            # idempotent to the purification's own convergence
    nt.set_option("spgemm_variant", 351)
    try:
        Kb = nt.Matrix_ps(n)
        energy_b, mu_b = nt.DensityMatrixSolvers.TRS2(H, ISQ, n / 2.0, Kb, p)
        tr_b = nt.solver_trace()
    finally:
        nt.set_option("spgemm_variant", -1)
    assert tr_b["iterations"] == tr["iterations"]
    assert np.array_equal(tr_b["sigma"], tr["sigma"])
    assert np.allclose(tr_b["energy"], tr["energy"], rtol=1e-13, atol=0)
    assert energy_b == pytest.approx(energy, rel=1e-13) and mu_b == pytest.approx(mu, rel=1e-12)
    ka, kb = K.triplets(), Kb.triplets()
    assert all(np.array_equal(u, v) for u, v in zip(ka, kb))


def test_spgemm_blocks_of_empty_columns(nt):
    """B with whole 16-column blocks empty (a rank-30 factor stored in an N x N matrix, as the pivoted Cholesky
    returns it): the slab planner must not walk a k range for such a block.  Bit-exact against the oracle."""
    from oracle import oracle_py as O
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    n = 200
    A = sp.random(n, n, 0.2, random_state=rng, format="csc")
    B = sp.random(n, n, 0.3, random_state=rng, format="csc").tolil()
    B[:, 30:170] = 0
    B = sp.csc_matrix(B)
    B.eliminate_zeros()
    for X, Y in ((A, B), (B, A), (B, B)):
        mA, mB = nt.Matrix_ps.from_scipy(X), nt.Matrix_ps.from_scipy(Y)
        C = nt.Matrix_ps(n)
        C.Gemm(mA, mB, None, 1.0, 0.0, 1e-9)
        tx, ty = mA.triplets(), mB.triplets()
        oc, orow, ov = O.ps_multiply(O.Mat.from_triplets(n, n, *tx), O.Mat.from_triplets(n, n, *ty), None, 1.0, 0.0, 1e-9).triplets()
        exact(C.triplets(), (n, n, oc, orow, ov), "empty column blocks")


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_spgemm_vs_oracle(nt, seed):
    """seeded random operands (sizes 1..400, densities 0..40 %, empty rows / columns, real and complex, alpha, beta with
    an existing C, thresholds), every kernel family forced in turn: the product must equal the oracle's bit for bit on
    the sparse branch and to 1e-13 where the reference would take its dense branch."""
    from oracle import oracle_py as O
    import scipy.sparse as sp
    rng = np.random.default_rng(1000 + seed)
    for case in range(8):
        n = int(rng.choice([1, 2, 7, 33, 64, 65, 130, 257, 400]))
        dens_a, dens_b = float(rng.choice([0.0, 0.02, 0.08, 0.4])), float(rng.choice([0.01, 0.05, 0.3]))
        cplx = bool(rng.integers(0, 2))
        thr = float(rng.choice([0.0, 1e-9, 1e-3]))
        alpha = float(rng.choice([1.0, -0.5, 2.0]))
        beta = float(rng.choice([0.0, 0.0, 0.7]))

        def rnd(d):
            m = sp.random(n, n, d, random_state=rng, format="csc", data_rvs=lambda k: rng.uniform(-1, 1, k))
            if cplx:
                m = m + 1j * sp.random(n, n, d, random_state=rng, format="csc", data_rvs=lambda k: rng.uniform(-1, 1, k))
            m = sp.csc_matrix(m)
            if n > 4:   # a few empty columns and rows
                m = m.tolil()
                m[:, n // 3] = 0
                m[n // 2, :] = 0
                m = sp.csc_matrix(m)
            m.eliminate_zeros()
            m.sort_indices()
            return m

        A, B, C0 = rnd(dens_a), rnd(dens_b), rnd(0.05)
        dense_branch = min(A.nnz, B.nnz) / float(n * n) > 0.1
        Ao, Bo, Co = (O.Mat.from_triplets(n, n, *nt.Matrix_ps.from_scipy(M).triplets()) for M in (A, B, C0))
        want = O.ps_multiply(Ao, Bo, Co if beta != 0.0 else None, alpha, beta, thr).triplets()
        for fb, var in ((-1, -1), (5, -1), (6, -1), (-1, 0), (-1, 400), (-1, 500)):
            nt.set_option("spgemm_force_bin", fb)
            nt.set_option("spgemm_variant", var)
            try:
                mA, mB = nt.Matrix_ps.from_scipy(A), nt.Matrix_ps.from_scipy(B)
                mC = nt.Matrix_ps.from_scipy(C0) if beta != 0.0 else nt.Matrix_ps(n)
                mC.Gemm(mA, mB, None, alpha, beta, thr)
                got = mC.triplets()
            finally:
                nt.set_option("spgemm_force_bin", -1)
                nt.set_option("spgemm_variant", -1)
            tag = "seed %d case %d n=%d cplx=%d fb=%d var=%d" % (seed, case, n, cplx, fb, var)
            if dense_branch or beta != 0.0:
                gd = to_dense((n, n) + tuple(got))
                wd = to_dense((n, n) + tuple(want))
                assert np.abs(gd - wd).max() <= 1e-13 * max(1.0, np.abs(wd).max()) + 2 * thr * (1 if dense_branch else 0), tag
            else:
                exact(got, (n, n) + tuple(want), tag)


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_elementwise_vs_oracle(nt, seed):
    """seeded random operands through the element-wise entry points against the oracle: Increment (AddSparseVectors
    rules incl. threshold and the unfiltered tails; both merge kernels) and PairwiseMultiply bit for bit; Dot, Trace,
    Norm and the Gershgorin bounds to 1e-13."""
    from oracle import oracle_py as O
    import scipy.sparse as sp
    rng = np.random.default_rng(2000 + seed)
    for case in range(10):
        n = int(rng.choice([1, 3, 40, 64, 129, 300, 700]))
        cplx = bool(rng.integers(0, 2))
        thr = float(rng.choice([0.0, 1e-6, 0.3]))
        alpha = float(rng.choice([1.0, -1.0, 0.37]))

        def rnd(d):
            m = sp.random(n, n, d, random_state=rng, format="csc", data_rvs=lambda k: rng.uniform(-1, 1, k))
            if cplx:
                m = m + 1j * sp.random(n, n, d, random_state=rng, format="csc", data_rvs=lambda k: rng.uniform(-1, 1, k))
            m = sp.csc_matrix(m)
            m.eliminate_zeros()
            m.sort_indices()
            return m

        A, B = rnd(float(rng.choice([0.0, 0.05, 0.5]))), rnd(float(rng.choice([0.02, 0.3])))
        mA, mB = nt.Matrix_ps.from_scipy(A), nt.Matrix_ps.from_scipy(B)
        Ao, Bo = O.Mat.from_triplets(n, n, *mA.triplets()), O.Mat.from_triplets(n, n, *mB.triplets())
        tag = "seed %d case %d n=%d cplx=%d thr=%g" % (seed, case, n, cplx, thr)
        want = O.increment(Ao, Bo, alpha, thr).triplets()
        for force_seq in (0, 1, 2):
            nt.set_option("increment_force_seq", force_seq)
            try:
                R = nt.Matrix_ps(mB)
                R.Increment(mA, alpha, thr)
            finally:
                nt.set_option("increment_force_seq", 0)
            exact(R.triplets(), (n, n) + tuple(want), tag + " increment seq=%d" % force_seq)
        Pm = nt.Matrix_ps(n)
        Pm.PairwiseMultiply(mA, mB)
        exact(Pm.triplets(), (n, n) + tuple(O.pairwise(Ao, Bo).triplets()), tag + " pairwise")
        d_want = O.dot(Ao, Bo)
        assert abs(mA.Dot(mB) - d_want) <= 1e-13 * max(1.0, abs(d_want)), tag + " dot"
        assert mB.Trace() == pytest.approx(O.trace(Bo), rel=1e-13, abs=1e-13), tag + " trace"
        assert mB.Norm() == pytest.approx(O.norm(Bo), rel=1e-13, abs=1e-13), tag + " norm"
        if not cplx:
            lo, hi = nt.EigenBounds.GershgorinBounds(mB)
            wlo, whi = O.gershgorin(Bo)
            assert lo == pytest.approx(wlo, rel=1e-13, abs=1e-13) and hi == pytest.approx(whi, rel=1e-13, abs=1e-13), tag


@pytest.mark.parametrize("density,expect_overflow", [(0.006, False), (0.02, True)])
def test_hash_table_classes_vs_oracle(nt, density, expect_overflow):
    """unstructured operand whose output columns outgrow the small hash table: N = 6000, ~36 / ~120 entries per column,
    so a column of A*A has ~1200 distinct rows (1024-bucket table -> 4096-bucket pass) or ~5400 (-> HBM accumulator
    as well).  Bit-exact against the oracle; the statistics show which passes ran."""
    from oracle import oracle_py as O
    import scipy.sparse as sp
    rng = np.random.default_rng(7)
    n = 6000
    A = sp.random(n, n, density, random_state=rng, format="csc", data_rvs=lambda k: rng.uniform(-1, 1, k))
    A.sort_indices()
    mA = nt.Matrix_ps.from_scipy(A)
    C = nt.Matrix_ps(n)
    C.Gemm(mA, mA, None, 1.0, 0.0, 1e-6)
    st = nt.last_spgemm_stats()
    assert st["bins"][5] == n and st["slab"] == 0          # every column went to the hash bin
    assert (st["overflow"] > 0) == expect_overflow          # columns handed on to the HBM accumulator
    t = mA.triplets()
    Ao = O.Mat.from_triplets(n, n, *t)
    oc, orow, ov = O.ps_multiply(Ao, Ao, None, 1.0, 0.0, 1e-6).triplets()
    dense_branch = A.nnz / float(n * n) > 0.1
    assert not dense_branch
    exact(C.triplets(), (n, n, oc, orow, ov), "hash classes density %g" % density)


def _grouped_case(nt, A, B, thr, alpha=1.0):
    """C = alpha*A*B through the grouped LDS-hash kernel (forced) against the oracle, bit for bit; returns its statistics"""
    from oracle import oracle_py as O
    mA, mB = nt.Matrix_ps.from_scipy(A), (nt.Matrix_ps.from_scipy(B) if B is not A else None)
    n = A.shape[0]
    C = nt.Matrix_ps(n)
    nt.set_option("spgemm_variant", 500)
    try:
        C.Gemm(mA, mB if mB is not None else mA, None, alpha, 0.0, thr)
        st = nt.last_spgemm_stats()
        gs = nt.last_grouped_stats()
    finally:
        nt.set_option("spgemm_variant", -1)
    Ao = O.Mat.from_triplets(n, n, *mA.triplets())
    Bo = O.Mat.from_triplets(n, n, *mB.triplets()) if mB is not None else Ao
    oc, orow, ov = O.ps_multiply(Ao, Bo, None, alpha, 0.0, thr).triplets()
    assert st["slab"] == 0
    exact(C.triplets(), (n, n, oc, orow, ov), "grouped hash")
    return st, gs


@pytest.mark.parametrize("n,h,holes,thr,cplx,same", [(8192, 50, 0.0, 1e-8, False, True), (6000, 100, 0.15, 1e-7, False, False),
                                                       (5000, 30, 0.3, 0.0, False, False), (7000, 160, 0.0, 1e-8, False, True),
                                                       (4096, 50, 0.0, 1e-8, True, True), (3001, 37, 0.3, 0.0, True, False),
                                                       (4097, 300, 0.0, 1e-6, False, True)])
def test_grouped_hash_permuted_vs_oracle(nt, n, h, holes, thr, cplx, same):
    """banded operands under a random symmetric relabelling (what LoadBalancerModule.F90:14-52 hands the multiply): no
    run structure, every row window spans the matrix.  The grouped LDS-hash kernel must find the similar columns
    (min-hash clustering), compute every group without falling back, and equal the oracle bit for bit."""
    import scipy.sparse as sp
    from gen import permuted_banded_triplets
    rng = np.random.default_rng(n + h)
    mats = []
    for t in range(1 if same else 2):
        col, row, val = permuted_banded_triplets(n, h, 42, shift=0.1 * t, complex_=cplx)
        keep = (rng.random(len(val)) >= holes) | (col == row)
        mats.append(sp.csc_matrix((val[keep], (row[keep] - 1, col[keep] - 1)), shape=(n, n)))
    A = mats[0]
    B = A if same else mats[1]
    st, gs = _grouped_case(nt, A, B, thr, alpha=1.0 if same else -0.5)
    assert gs["used"] == 1 and gs["minhash"] == 1, gs
    assert gs["failed_cols"] <= n // 20, gs    # (a group that straddles two clusters may outgrow the largest table)
    # union ratio = steps / (nnz / 16): holes thin the columns (fewer shared rows) and narrow bands make small clusters
    assert gs["union_ratio"] < (2.5 if holes == 0.0 and h >= 50 else 6.0), gs


def test_grouped_hash_natural_order_and_fallback(nt):
    """(1) a 3-D lattice operator in its natural order (adjacent columns are similar, no clustering needed); (2) an
    operand whose columns have nothing in common (uniformly random pattern): the row unions outgrow every table class,
    all groups are handed back to the per-column LDS hash, and the result is still the oracle's."""
    import scipy.sparse as sp
    L = 18
    idx = np.arange(L ** 3).reshape(L, L, L)
    rows, cols, vals = [], [], []
    rng = np.random.default_rng(5)
    for dx in range(-2, 3):
        for dy in range(-2, 3):
            for dz in range(-2, 3):
                if dx * dx + dy * dy + dz * dz > 5:
                    continue
                src = idx[max(0, -dx):L - max(0, dx), max(0, -dy):L - max(0, dy), max(0, -dz):L - max(0, dz)].ravel()
                dst = idx[max(0, dx):L - max(0, -dx), max(0, dy):L - max(0, -dy), max(0, dz):L - max(0, -dz)].ravel()
                rows.append(dst); cols.append(src)
                vals.append(rng.uniform(-1, 1, len(src)) * np.exp(-0.5 * (dx * dx + dy * dy + dz * dz)))
    A = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(L ** 3, L ** 3))
    A.sort_indices()
    st, gs = _grouped_case(nt, A, A, 1e-9)
    assert gs["used"] == 1 and gs["failed_cols"] == 0, gs
    n = 6000
    R = sp.random(n, n, 0.006, random_state=np.random.default_rng(7), format="csc", data_rvs=lambda k: rng.uniform(-1, 1, k))
    R.sort_indices()
    st, gs = _grouped_case(nt, R, R, 1e-6)
    assert gs["used"] == 1 and gs["failed_cols"] > n // 2, gs    # forced: the automatic rule declines such operands
    C = nt.Matrix_ps(n)
    mR = nt.Matrix_ps.from_scipy(R)
    C.Gemm(mR, mR, None, 1.0, 0.0, 1e-6)
    assert nt.last_grouped_stats()["used"] == 0 and nt.last_grouped_stats()["union_ratio"] > 6.0


def test_grouped_hash_full_size_permuted_config2(nt):
    """the permuted configs[1]-family operand (N = 65 536, 201 entries per row, seed 42) through the
    DEFAULT dispatch -- the grouped LDS-hash kernel -- against the oracle at full size, bit for bit; the one-column-per
    -wave hash (variant 501) gives the same bits."""
    from oracle import oracle_py as O
    from gen import permuted_banded_triplets
    n, h = 65536, 100
    col, row, val = permuted_banded_triplets(n, h, 42)
    A = nt.Matrix_ps.from_triplets(n, col, row, val)
    C = nt.Matrix_ps(n)
    nt.set_option("time_kernels", 1)
    try:
        C.Gemm(A, A, None, 1.0, 0.0, 1e-8)
        st, gs = nt.last_spgemm_stats(), nt.last_grouped_stats()
    finally:
        nt.set_option("time_kernels", 0)
    assert st["slab"] == 0 and gs["used"] == 1 and gs["minhash"] == 1 and gs["failed_cols"] == 0, (st, gs)
    assert st["products"] > 2.6e9
    got = C.triplets()
    Ao = O.Mat.from_triplets(n, n, col, row, val)
    oc, orow, ov = O.ps_multiply(Ao, Ao, None, 1.0, 0.0, 1e-8).triplets()
    exact(got, (n, n, oc, orow, ov), "permuted config 2 (h = 100) vs oracle")
    nt.set_option("spgemm_variant", 501)
    try:
        C2 = nt.Matrix_ps(n)
        C2.Gemm(A, A, None, 1.0, 0.0, 1e-8)
    finally:
        nt.set_option("spgemm_variant", -1)
    g2 = C2.triplets()
    assert np.array_equal(got[0], g2[0]) and np.array_equal(got[1], g2[1]) and np.array_equal(got[2], g2[2])


def test_process_slices_semantics_vs_reference(nt):
    """VERDICT r1 'missing' 3: grids with process slices.  The reference's multiply then sums per-slice partial products
    (threshold / (1000 slices) each, the caller's threshold only in the last addition, MatrixMultiply.f90:25-29, 230-267,
    ReduceAndSumMatrixCleanup.f90:11-32), so its result depends on the grid.  tests/golden/ps_gemm_slices.npz holds the
    reference's own results on 2..8 ranks for six grids, two thresholds, real and complex, a dimension that needs
    padding; the engine (its grid shape only selects the summation semantics) must reproduce every one of them bit for
    bit -- and the slice-free grid must differ from the sliced ones where the reference's do."""
    g = Golden("ps_gemm_slices")
    nt.set_option("virtual_grid", 1)
    try:
        seen = {}
        for c in g.cases:
            pr, pc, ps = c["grid"]
            nt.ConstructGlobalProcessGrid(pr, pc, ps)
            A = pmat(nt, g.tri(None, "m%d_A" % c["matrix"]))
            B = pmat(nt, g.tri(None, "m%d_B" % c["matrix"]))
            C = nt.Matrix_ps(c["n"])
            C.Gemm(A, B, None, c["alpha"], 0.0, c["thr"])
            want = g.tri(None, c["key"])
            tA, tB = g.tri(None, "m%d_A" % c["matrix"]), g.tri(None, "m%d_B" % c["matrix"])
            if min(len(tA[4]), len(tB[4])) / float(c["n"] ** 2) > 0.1:
                # the reference's blocks take its BLAS branch here (GemmMatrix.f90:59): summation order not restated
                close(C.triplets(), want, DENSE_RTOL, "slices %s (dense branch)" % c["key"])
            else:
                exact(C.triplets(), want, "slices %s" % c["key"])
            seen[(c["matrix"], c["thr"], tuple(c["grid"]))] = len(want[4])
        assert seen[(0, 2e-2, (1, 1, 1))] != seen[(0, 2e-2, (1, 1, 2))] != seen[(0, 2e-2, (1, 1, 4))]
    finally:
        nt.ConstructGlobalProcessGrid(1, 1, 1)
        nt.set_option("virtual_grid", 0)


# ------------------------------------------------------------------ fused purification steps / loose iterates
def _trs2_run(nt, n, col, row, val, thr, iters, fused, loose, label_order=1):
    nt.set_option("fused_update", fused)
    nt.set_option("loose_iterates", loose)
    nt.set_option("label_order", label_order)
    nt.set_option("time_kernels", 1)     # (the statistics below are kept only with the timers on)
    try:
        H = nt.Matrix_ps.from_triplets(n, col, row, val)
        ISQ = nt.Matrix_ps(n)
        ISQ.FillIdentity()
        K = nt.Matrix_ps(n)
        p = nt.SolverParameters()
        p.SetConvergeDiff(1e-30)
        p.SetThreshold(thr)
        p.SetMaxIterations(iters)
        p.SetMonitorConvergence(False)
        before = nt.fusion_counts()
        nt.reset_spgemm_accum()
        energy, mu = nt.DensityMatrixSolvers.TRS2(H, ISQ, n / 2.0, K, p)
        acc = nt.spgemm_accum()
        after = nt.fusion_counts()
        tr = nt.solver_trace()
        return dict(energy=energy, mu=mu, tr=tr, K=K.triplets(), acc=acc,
                    counts={k: after[k] - before[k] for k in after})
    finally:
        nt.set_option("fused_update", 1)
        nt.set_option("loose_iterates", 1)
        nt.set_option("label_order", 1)
        nt.set_option("time_kernels", 0)


@pytest.mark.parametrize("case", ["banded", "holes", "ragged", "threshold0", "tall", "wide_d", "stored_zeros"])
def test_trs2_fused_steps_match_the_separate_passes(nt, case):
    """The TRS2 step computed inside the SpGEMM kernel's epilogue (kernels.hpp SlabFusion: X*X or 2X - X*X, energy and
    trace in one kernel, the iterate left in its slots) against the same solve with the product, the merge and the
    reductions as separate passes (the path the goldens and the oracle pin): identical sigma sequence, identical
    density bit for bit, energies to reduction-order roundoff.  'stored_zeros': X holds stored zeros (threshold 0 and
    explicit zeros in H), which the merge inside the kernel cannot tell from holes -- it must step aside.  'wide_d':
    the columns of WH do not fit the kernel's LDS tile."""
    rng = np.random.default_rng(11)
    if case == "tall":
        n, h, thr, iters = 6144, 110, 1e-11, 10     # row windows beyond 768 rows: the six / eight wave geometries
    elif case == "wide_d":
        n, h, thr, iters = 6144, 300, 1e-6, 8       # sixteen columns of WH exceed the LDS tile: no fusion, same results
    elif case == "stored_zeros":
        n, h, thr, iters = 768, 6, 0.0, 4
    elif case == "ragged":
        n, h, thr, iters = 4099, 25, 1e-6, 12       # last block of 16 columns incomplete
    elif case == "threshold0":
        n, h, thr, iters = 1024, 10, 0.0, 5         # nothing is ever dropped
    else:
        n, h, thr, iters = 8192, 40, 1e-7, 14
    col, row, val = banded_triplets(n, h)
    if case == "holes":      # symmetric holes: columns with several runs, ragged block windows
        a, b = np.minimum(col, row).astype(np.int64), np.maximum(col, row).astype(np.int64)
        keep = (((a * 2654435761 + b * 40503) >> 7) % 10 >= 3) | (col == row)
        col, row, val = col[keep], row[keep], val[keep]
    if case == "stored_zeros":
        off = np.abs(col.astype(np.int64) - row) == 3
        val = np.where(off, 0.0, val)
    ref = _trs2_run(nt, n, col, row, val, thr, iters, fused=0, loose=0)
    assert ref["counts"]["square"] == ref["counts"]["update"] == 0
    for fused, loose in ((1, 1), (0, 1)):
        got = _trs2_run(nt, n, col, row, val, thr, iters, fused=fused, loose=loose)
        tag = (case, fused, loose)
        assert got["tr"]["iterations"] == ref["tr"]["iterations"] == iters, tag
        assert np.array_equal(got["tr"]["sigma"], ref["tr"]["sigma"]), tag
        assert np.array_equal(got["tr"]["nnz"], ref["tr"]["nnz"]), tag
        assert np.allclose(got["tr"]["energy"], ref["tr"]["energy"], rtol=1e-12, atol=1e-12), tag
        for q in range(2):
            assert np.array_equal(got["K"][q], ref["K"][q]), tag
        assert np.array_equal(got["K"][2], ref["K"][2]), tag
        # the statistics of the multiplies: the same intermediate products and product entries on every path
        assert got["acc"]["calls"] == ref["acc"]["calls"], tag
        assert got["acc"]["products"] == ref["acc"]["products"], tag
        assert got["acc"]["nnz_c"] == ref["acc"]["nnz_c"], tag
        c = got["counts"]
        if fused and case == "wide_d":
            assert c["square"] == c["update"] == c["repeated"] == 0, (tag, c)
        elif fused and case != "stored_zeros":
            # (an operand full of holes may take another SpGEMM path in the first iterations: nothing to fuse into)
            assert c["repeated"] == 0 and iters - 2 <= c["square"] + c["update"] <= iters, (tag, c)
            assert c["update"] > 0 and c["square"] > 0, (tag, c)
        # ('stored_zeros': a merge that would read the stored zeros is not fused; once a step has produced the iterate
        # inside the kernel every entry is non-zero and the following steps may be -- the equalities above are the test)
        if not fused:
            assert c["square"] == c["update"] == 0, (tag, c)


@pytest.mark.parametrize("n,h,thr,iters,holes", [(8192, 40, 1e-7, 14, False), (6000, 25, 1e-6, 10, True), (4099, 30, 1e-7, 8, False)])
def test_trs2_relabelled_operand_label_ordered_steps(nt, n, h, thr, iters, holes):
    """A band hidden under a random symmetric relabelling (what the load balancer hands the solver): the engine finds a
    bandwidth-reducing order (relabel.hip), runs the loop in it with the fused slab kernel and lets the arithmetic
    follow the ORIGINAL labels (order of the k steps, last-row tests of the merge) -- the density must be the one the
    grouped LDS-hash path computes on the relabelled matrix itself (the path pinned to the oracle), bit for bit."""
    from gen import permuted_banded_triplets
    col, row, val = permuted_banded_triplets(n, h, 7)
    if holes:
        a, b = np.minimum(col, row).astype(np.int64), np.maximum(col, row).astype(np.int64)
        keep = (((a * 2654435761 + b * 40503) >> 7) % 10 >= 2) | (col == row)
        col, row, val = col[keep], row[keep], val[keep]
    ref = _trs2_run(nt, n, col, row, val, thr, iters, fused=0, loose=0, label_order=0)
    assert ref["counts"]["square"] == ref["counts"]["update"] == 0
    got = _trs2_run(nt, n, col, row, val, thr, iters, fused=1, loose=1, label_order=1)
    # (the same with the tiles copied into step order before every launch instead of the loop variant that takes a
    # step's multiplier row from its run record)
    nt.set_option("label_rowoff", 0)
    try:
        got2 = _trs2_run(nt, n, col, row, val, thr, iters, fused=1, loose=1, label_order=1)
    finally:
        nt.set_option("label_rowoff", 1)
    for q in range(3):
        assert np.array_equal(got2["K"][q], ref["K"][q]), q
    assert got2["counts"]["square"] + got2["counts"]["update"] == iters
    assert got["tr"]["iterations"] == ref["tr"]["iterations"] == iters
    assert np.array_equal(got["tr"]["sigma"], ref["tr"]["sigma"])
    assert np.array_equal(got["tr"]["nnz"], ref["tr"]["nnz"])
    assert np.allclose(got["tr"]["energy"], ref["tr"]["energy"], rtol=1e-12, atol=1e-12)
    for q in range(3):
        assert np.array_equal(got["K"][q], ref["K"][q]), q
    c = got["counts"]
    assert c["repeated"] == 0 and c["square"] + c["update"] == iters, c      # every step inside the slab kernel
    assert got["acc"]["products"] == ref["acc"]["products"] and got["acc"]["nnz_c"] == ref["acc"]["nnz_c"]


def test_band_order_recovers_hidden_bands(nt):
    """csrc/relabel.hip on its own: a band hidden under a random relabelling is recovered exactly (bandwidth = the
    half band width), also when the graph has several components; a pattern without a band gets an order too (and a
    bandwidth that tells the caller not to bother); the result is always a permutation."""
    import scipy.sparse as sp
    from gen import permuted_banded_triplets
    rng = np.random.default_rng(3)
    # one band
    n, h = 12000, 20
    col, row, val = permuted_banded_triplets(n, h, 11)
    pos, bw = nt.band_order(nt.Matrix_ps.from_triplets(n, col, row, val))
    assert np.array_equal(np.sort(pos), np.arange(n)) and bw == h
    # three bands of different widths, disconnected, shuffled together
    blocks = []
    for m, hb in ((3000, 8), (5000, 15), (2500, 4)):
        i = np.arange(m)
        d = [np.ones(m - abs(o)) for o in range(-hb, hb + 1)]
        blocks.append(sp.diags(d, list(range(-hb, hb + 1)), shape=(m, m), format="csc"))
    A = sp.block_diag(blocks, format="csc")
    n = A.shape[0]
    perm = rng.permutation(n)
    P = sp.csc_matrix((np.ones(n), (perm, np.arange(n))), shape=(n, n))
    Ap = (P @ A @ P.T).tocsc()
    pos, bw = nt.band_order(nt.Matrix_ps.from_scipy(Ap))
    assert np.array_equal(np.sort(pos), np.arange(n)) and bw == 15
    # every entry within the reported bandwidth under the new order
    c = Ap.tocoo()
    assert np.abs(pos[c.row] - pos[c.col]).max() == bw
    # no band at all: a random sparse symmetric pattern
    n = 4000
    R = sp.random(n, n, density=0.003, random_state=5, format="csc")
    R = (R + R.T + sp.identity(n)).tocsc()
    pos, bw = nt.band_order(nt.Matrix_ps.from_scipy(R))
    assert np.array_equal(np.sort(pos), np.arange(n)) and bw > 200
