"""CPU: pin the C oracle (oracle/ntpoly_oracle.c) against the golden vectors produced by the
REAL reference (tests/golden/make_golden.py).  Sparse-branch SpGEMM, increment and pattern
results must be BIT-EXACT; the reference's dense branch goes through MKL DGEMM/ZGEMM whose
summation order is not restated, so those cases are compared to 1e-13 relative."""
import numpy as np
import pytest

from golden_util import Golden, same_pattern, to_dense
from oracle import oracle_py as O

DENSE_RTOL = 1e-13


def mat(t):
    return O.Mat.from_triplets(t[0], t[1], t[2], t[3], t[4])


def assert_bitexact(got, want, what):
    assert same_pattern((0, 0) + got, want), "%s: pattern differs (%d vs %d nnz)" % (
        what, len(got[0]), len(want[2]))
    assert np.array_equal(got[2], want[4]), "%s: values differ, max |d| = %g" % (
        what, np.abs(got[2] - want[4]).max())


def assert_close(got_mat, want, rtol, what):
    g = got_mat.to_scipy().toarray()
    w = to_dense(want)
    scale = max(1.0, np.abs(w).max())
    assert np.abs(g - w).max() <= rtol * scale, "%s: max |d| = %g" % (what, np.abs(g - w).max())


def test_local_gemm_vs_reference():
    g = Golden("local_gemm")
    n_exact = 0
    for i, c in enumerate(g.cases):
        A, B = g.tri(i, "A"), g.tri(i, "B")
        Cin = mat(g.tri(i, "Cin")) if g.has(i, "Cin") else None
        want = g.tri(i, "C")
        got = O.gemm(mat(A), mat(B), Cin, tA=c["tA"], tB=c["tB"], alpha=c["alpha"], beta=c["beta"],
                     threshold=c["thr"])
        sp_a = len(A[2]) / float(max(1, A[0] * A[1]))
        sp_b = len(B[2]) / float(max(1, B[0] * B[1]))
        if min(sp_a, sp_b) > 0.1:  # reference took the BLAS dense branch (GemmMatrix.f90:59)
            if c["thr"] == 0.0:
                assert_close(got, want, DENSE_RTOL, "case %d (dense branch)" % i)
            else:  # entries within roundoff of the threshold may flip; compare values only
                gd, wd = got.to_scipy().toarray(), to_dense(want)
                bad = np.abs(gd - wd) > DENSE_RTOL * max(1.0, np.abs(wd).max())
                assert np.all(np.abs(np.where(bad, np.maximum(np.abs(gd), np.abs(wd)), 0.0))
                              <= c["thr"] * (1 + 1e-10) * max(1.0, abs(c["alpha"])))
        else:
            assert_bitexact(got.triplets(), want, "case %d %s" % (i, c))
            n_exact += 1
    assert n_exact >= 40


def test_fma_mode_vs_contracted_reference_build():
    """The oracle's second arithmetic mode (oracle_py.set_fma: fma(a, b, acc) in the real multiply kernel) against the
    products of the REAL reference built with FP contraction (oracle/build_ref.py --fma, make_golden.py ps_gemm_fma):
    bit-exact.  This pins what the engine's FMA arithmetic (option spgemm_fma, the MFMA tile kernel) is compared with
    in tests/test_gpu_fma.py.  The default mode must NOT reproduce these products."""
    g = Golden("ps_gemm_fma")
    assert not O.get_fma()
    O.set_fma(True)
    try:
        for i, c in enumerate(g.cases):
            A = mat(g.tri(i, "A"))
            B = A if c["same"] else mat(g.tri(i, "B"))
            got = O.ps_multiply(A, B, None, c["alpha"], 0.0, c["thr"])
            assert_bitexact(got.triplets(), g.tri(i, "C"), "fma " + c["tag"])
    finally:
        O.set_fma(False)
    A = mat(g.tri(1, "A"))
    got = O.ps_multiply(A, A, None, 1.0, 0.0, g.cases[1]["thr"]).triplets()
    want = g.tri(1, "C")
    assert not (len(got[2]) == len(want[4]) and np.array_equal(got[2], want[4]))


def test_local_increment_vs_reference():
    g = Golden("local_increment")
    for i, c in enumerate(g.cases):
        got = O.increment(mat(g.tri(i, "A")), mat(g.tri(i, "B")), c["alpha"], c["thr"])
        assert_bitexact(got.triplets(), g.tri(i, "C"), "case %d %s" % (i, c))


def test_ps_gemm_vs_reference():
    g = Golden("ps_gemm")
    n_exact = 0
    for i, c in enumerate(g.cases):
        Cin = mat(g.tri(i, "Cin")) if g.has(i, "Cin") else None
        got = O.ps_multiply(mat(g.tri(i, "A")), mat(g.tri(i, "B")), Cin, c["alpha"], c["beta"], c["thr"])
        want = g.tri(i, "C")
        if c["dense_branch"]:
            assert_close(got, want, DENSE_RTOL, "case %d %s" % (i, c["tag"]))
        else:
            assert_bitexact(got.triplets(), want, "case %d %s" % (i, c["tag"]))
            n_exact += 1
    assert n_exact >= 20


def test_ps_gemm_grid_independence_recorded_by_reference():
    """slices == 1 grids give identical values in the reference; our oracle (1x1x1) matches them
    bit for bit.  The 2x2x2 grid keeps sub-threshold entries: values agree, nnz not."""
    g = Golden("ps_gemm_grids")
    A = g.tri(None, "A")
    got = O.ps_multiply(mat(A), mat(A), None, 1.0, 0.0, g.meta["thr"])
    for grid in ("111", "221", "141"):
        assert_bitexact(got.triplets(), g.tri(None, "C_" + grid), "grid " + grid)
    w = to_dense(g.tri(None, "C_222"))
    assert np.abs(got.to_scipy().toarray() - w).max() <= 2 * g.meta["thr"]


def test_process_slices_vs_reference():
    """grids with process slices: the oracle's restatement of the reference's K-split sums (oracle_py.ps_multiply_sliced)
    against what the real reference produced on 2..8 ranks (tests/golden/ps_gemm_slices.npz) -- bit for bit on the
    sparse branch, 1e-13 where the reference's blocks take its BLAS branch"""
    g = Golden("ps_gemm_slices")
    n_exact = 0
    for c in g.cases:
        pr, pc, ps = c["grid"]
        tA, tB = g.tri(None, "m%d_A" % c["matrix"]), g.tri(None, "m%d_B" % c["matrix"])
        if ps == 1:
            got = O.ps_multiply(mat(tA), mat(tB), None, c["alpha"], 0.0, c["thr"])
        else:
            got = O.ps_multiply_sliced(mat(tA), mat(tB), c["alpha"], c["thr"], pr, pc, ps)
        want = g.tri(None, c["key"])
        if min(len(tA[4]), len(tB[4])) / float(c["n"] ** 2) > 0.1:
            w = to_dense(want)
            assert np.abs(got.to_scipy().toarray() - w).max() <= 1e-5 + 1e-13 * np.abs(w).max(), c["key"]
        else:
            assert_bitexact(got.triplets(), want, c["key"])
            n_exact += 1
    assert n_exact >= 12


def test_ps_increment_vs_reference():
    g = Golden("ps_increment")
    for i, c in enumerate(g.cases):
        got = O.increment(mat(g.tri(i, "A")), mat(g.tri(i, "B")), c["alpha"], c["thr"])
        assert_bitexact(got.triplets(), g.tri(i, "C"), "case %d" % i)


def test_ps_scalars_vs_reference():
    g = Golden("ps_scalars")
    for i, c in enumerate(g.cases):
        A, B = mat(g.tri(i, "A")), mat(g.tri(i, "B"))
        assert O.trace(A) == pytest.approx(c["trace"], rel=1e-14, abs=1e-14)
        assert O.norm(A) == pytest.approx(c["norm"], rel=1e-14)
        d = O.dot(A, B)
        assert np.real(d) == pytest.approx(c["dot_real"], rel=1e-13, abs=1e-13)
        emin, emax = O.gershgorin(A)
        assert emin == pytest.approx(c["gersh_min"], rel=1e-14)
        assert emax == pytest.approx(c["gersh_max"], rel=1e-14)
        assert A.nnz == c["nnz"]


SOLVER_FN = dict(sign="sign", invert="invert", isq="inverse_square_root", sqrt="square_root")


def test_solvers_vs_reference():
    g = Golden("solvers")
    for i, c in enumerate(g.cases):
        H = mat(g.tri(i, "H"))
        p = O.params(converge_diff=c["conv"], max_iterations=c["maxit"], threshold=c["thr"],
                     monitor_convergence=c["monitor"])
        want = g.tri(i, "K")
        if c["solver"] in ("trs2", "trs4"):
            ISQ = O.Mat.identity(H.rows, H.is_complex) if c["isq"] == "identity" else mat(g.tri(i, "ISQ"))
            K, energy, mu, tr = O.density(c["solver"], H, ISQ, c["nel"], p)
            log_e = g.arr(i, "log_energy")
            # the reference logs the energy of every iteration except the converged one
            n = len(log_e)
            assert tr["iterations"] in (n, n + 1), (c["tag"], tr["iterations"], n)
            assert np.allclose(tr["energy"][:n], log_e, rtol=1e-12, atol=1e-12), c["tag"]
            assert energy == pytest.approx(c["energy"], rel=1e-12), c["tag"]
            # TRS4's sigma = (trace - tr_fx)/tr_gx is a 0/0 quotient near convergence, so its
            # bisection-derived mu amplifies summation-order roundoff (1e-7 observed)
            mu_tol = 1e-10 if c["solver"] == "trs2" else 1e-5
            assert mu == pytest.approx(c["mu"], rel=mu_tol, abs=1e-12), c["tag"]
        else:
            K, tr = O.matrix_function(SOLVER_FN[c["solver"]], H, p)
            log_c = g.arr(i, "log_convergence")
            if c["solver"] == "invert":  # logged twice per iteration (InverseSolversModule.F90:102-104)
                log_c = log_c[::2]
            assert tr["iterations"] == len(log_c), (c["tag"], tr["iterations"], len(log_c))
            assert np.allclose(tr["value"], log_c, rtol=1e-10, atol=1e-14), c["tag"]
        gd, wd = K.to_scipy().toarray(), to_dense(want)
        tol = max(10 * c["thr"], 1e-11) * max(1.0, np.abs(wd).max())
        assert np.abs(gd - wd).max() <= tol, "%s: max |d| = %g" % (c["tag"], np.abs(gd - wd).max())
        assert abs(K.nnz - c["nnz"]) <= max(2, 0.01 * c["nnz"]), (c["tag"], K.nnz, c["nnz"])


def test_premade_fixture_of_the_reference():
    """Examples/PremadeMatrix/Density-Reference.mtx (the reference's own shipped output):
    matches TRS2 with nel = 5 to 5e-5 Frobenius."""
    g = Golden("solvers")
    idx = [i for i, c in enumerate(g.cases) if c["tag"] == "premade_trs2_nel5"][0]
    c = g.cases[idx]
    H, ISQ = mat(g.tri(idx, "H")), mat(g.tri(idx, "ISQ"))
    p = O.params(converge_diff=c["conv"], max_iterations=c["maxit"], threshold=c["thr"])
    K, energy, mu, tr = O.density("trs2", H, ISQ, 5.0, p)
    Dref = to_dense(g.tri(None, "premade_density_reference"))
    assert np.linalg.norm(K.to_scipy().toarray() - Dref) <= 5e-5
    assert energy == pytest.approx(-22.971963210096895, rel=1e-10)
    assert mu == pytest.approx(0.1491684406929008, rel=1e-8)
    assert tr["iterations"] == 19


def test_convergence_monitor_rules():
    """ConvergenceMonitorModule.F90:122-191."""
    import ctypes as C
    L = O.lib()
    m = O.OMonitor()
    L.omonitor_init(C.byref(m), 1, 1e-8)
    seq = [1.0, 0.5, 0.1, 1e-3, 1e-4, 1e-4, 1.2e-4]
    flags = []
    for v in seq:
        L.omonitor_append(C.byref(m), v)
        flags.append(bool(L.omonitor_converged(C.byref(m))))
    assert flags[:5] == [False] * 5          # fewer than 6 values
    L.omonitor_init(C.byref(m), 0, 1e-2)     # basic mode: only the tight cutoff
    L.omonitor_append(C.byref(m), 5e-3)
    assert L.omonitor_converged(C.byref(m))
    L.omonitor_append(C.byref(m), -0.5)
    assert not L.omonitor_converged(C.byref(m))
