"""GPU: the solver loops on matrices that stay in slab form (option slab_algebra, one rank, real operands, both arithmetic
modes: psmatrix.cpp SlabSession + kernels.hip "slab algebra") -- TRS4, SignFunction, Invert, InverseSquareRoot / SquareRoot,
HPCP, ScaleAndFold, the matrix polynomials and functions, a caller's own loop over the C ABI.

Three checks per solver: (1) the slab session really ran (ntpoly_amd_slab_algebra_counts: the loop's products were done
in slab form, and the number of refusals is what the operand explains); (2) the result equals the
one computed with the session off -- products, merges and scalings are the same arithmetic in the same order, so sign,
inverse and square roots are equal BIT FOR BIT with the same pattern; TRS4's sigma is a quotient of two dots whose
summation order differs, so its density agrees to 1e-10; (3) the result equals the oracle's FMA mode (the CPU
restatement of the reference, pinned to the contracted reference build) within the solver tolerances of
tests/test_gpu_parity.py.  Operands include a Hamiltonian with STORED ZEROS on its diagonal (the generator's h_ii = 0
where 7919 i = 500 mod 1000): the slab form of such a matrix is a read-only view that keeps its compressed columns."""
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


@pytest.fixture(params=[1, 0], ids=["fma", "unfused"])
def fma(nt, request):
    """both arithmetic modes: FMA (products on the MFMA tile kernel) and unfused, the library's default (products on the
    register-slab kernel, runs packed back to back); the oracle is switched to the same mode"""
    from oracle import oracle_py as O
    nt.set_option("spgemm_fma", request.param)
    O.set_fma(bool(request.param))
    # (unfused arithmetic: a product whose operands have become sparse inside wide extents is handed to the general kernels
    # -- psmatrix.cpp runs_dense -- so HOW MANY operations ran in slab form is only asserted exactly in FMA arithmetic)
    nt.slab_counts_exact = request.param == 1
    yield O
    O.set_fma(False)
    nt.set_option("spgemm_fma", 0)
    nt.set_option("slab_algebra", 1)


def srt(t):
    c, r, v = t
    o = np.lexsort((r, c))
    return c[o], r[o], v[o]


def same_pattern(a, b):
    return len(a[2]) == len(b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def run(nt, solver, H, n, thr, conv, iters=None):
    p = nt.SolverParameters()
    p.SetThreshold(thr)
    p.SetConvergeDiff(conv)
    if iters:
        p.SetMaxIterations(iters)
        p.SetMonitorConvergence(False)
    Out = nt.Matrix_ps(n)
    extra = None
    if solver == "trs4":
        I = nt.Matrix_ps(n)
        I.FillIdentity()
        extra = nt.DensityMatrixSolvers.TRS4(H, I, n / 2.0, Out, p)
    elif solver == "sign":
        nt.SignSolvers.ComputeSign(H, Out, p)
    elif solver == "invert":
        nt.InverseSolvers.Invert(H, Out, p)
    elif solver == "inverse_square_root":
        nt.SquareRootSolvers.InverseSquareRoot(H, Out, p)
    else:
        nt.SquareRootSolvers.SquareRoot(H, Out, p)
    tr = nt.solver_trace()
    return srt(Out.triplets()), tr, extra


ORDER2 = [("inverse_square_root", 4096, 20, 1e-8, 2.0)]
# (sizes: what these cases cost is the ORACLE's solve on the host cores -- the iterates of the sign and root loops fill in -- so the
# loops run at N = 2048 .. 2560, every case in both arithmetic modes, instead of at N = 4096 with most unfused cases left out)
CASES = [("trs4", 4096, 20, 1e-8, 0.0), ("trs4", 3000, 12, 1e-6, 0.0), ("sign", 2560, 20, 1e-8, 0.0), ("sign", 2048, 8, 1e-7, 0.3),
         ("invert", 2560, 20, 1e-8, 2.0), ("inverse_square_root", 2560, 20, 1e-8, 2.0), ("square_root", 2048, 12, 1e-7, 2.0)]


@pytest.mark.parametrize("solver,n,h,thr,shift", CASES)
def test_slab_session_equals_compressed_columns_and_oracle(nt, fma, solver, n, h, thr, shift):
    O = fma
    col, row, val = banded_triplets(n, h, shift=shift)
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    # (TRS4: a fixed number of iterations -- its sigma is a 0/0 quotient near convergence, where the stopping iteration is noise)
    conv, iters = (1e-30, 14) if solver == "trs4" else (1e-8, None)
    nt.set_option("slab_algebra", 0)
    c0 = nt.slab_algebra_counts()
    want, tr0, ex0 = run(nt, solver, H, n, thr, conv, iters)
    c1 = nt.slab_algebra_counts()
    assert c1 == c0   # (session off: nothing in slab form)
    nt.set_option("slab_algebra", 1)
    got, tr1, ex1 = run(nt, solver, H, n, thr, conv, iters)
    c2 = nt.slab_algebra_counts()
    # the loop's products ran in slab form (two per iteration, three for the square roots; TRS4: one or two), and at most
    # the first merge on an operand with stored zeros was refused
    per = {"trs4": 1, "sign": 2, "invert": 2, "inverse_square_root": 3, "square_root": 3}[solver]
    if nt.slab_counts_exact:
        assert c2["products"] - c1["products"] >= per * (tr1["iterations"] - 1), (c1, c2, tr1["iterations"])
        assert c2["refusals"] - c1["refusals"] <= 1
    else:
        assert c2["products"] - c1["products"] >= per
    assert tr0["iterations"] == tr1["iterations"]
    assert same_pattern(got, want), "%s: pattern differs (%d vs %d entries)" % (solver, len(got[2]), len(want[2]))
    if solver == "trs4":
        assert np.abs(got[2] - want[2]).max() <= 1e-10
        assert abs(ex0[0] - ex1[0]) <= 1e-9 * abs(ex0[0]) and abs(ex0[1] - ex1[1]) <= 1e-6
    else:
        assert np.array_equal(got[2], want[2]), "%s: max |d| = %g" % (solver, np.abs(got[2] - want[2]).max())
        assert np.array_equal(np.asarray(tr0["value"]), np.asarray(tr1["value"])) or np.allclose(tr0["value"], tr1["value"], rtol=1e-12, atol=1e-14)
    # the oracle's FMA mode (CPU restatement of the reference)
    Ho = O.Mat.from_triplets(n, n, col, row, val)
    if solver == "trs4":
        po = O.params(converge_diff=conv, threshold=thr, max_iterations=iters, monitor_convergence=False)
        Ko, e_o, mu_o, tro = O.density("trs4", Ho, O.Mat.identity(n), n / 2.0, po)
        assert tr1["iterations"] == tro["iterations"] == iters
        assert np.allclose(tr1["energy"], tro["energy"], rtol=1e-10, atol=1e-10)
        assert np.allclose(tr1["sigma"], tro["sigma"], rtol=1e-6, atol=1e-8)
        assert abs(e_o - ex1[0]) <= 1e-10 * abs(e_o)
        w = srt(Ko.triplets())
        # (TRS4's sigma is a 0/0 quotient near convergence: densities agree to the convergence level, not to roundoff)
        import scipy.sparse as sp
        G = sp.csr_matrix((got[2], (got[1] - 1, got[0] - 1)), shape=(n, n))
        W = sp.csr_matrix((w[2], (w[1] - 1, w[0] - 1)), shape=(n, n))
        assert abs(G - W).max() <= 1e-6
    else:
        po = O.params(converge_diff=conv, threshold=thr)
        Oo, tro = O.matrix_function(solver, Ho, po)
        assert tr1["iterations"] == tro["iterations"]
        w = srt(Oo.triplets())
        assert same_pattern(got, w)
        assert np.abs(got[2] - w[2]).max() <= 1e-12 * max(1.0, np.abs(w[2]).max())


def test_slab_session_runs_and_counts_its_refusals(nt, fma):
    """full path check on a mid-size operand: with the session on the products of the loop come from slab_multiply, and
    the sign of an operand with STORED ZEROS needs no refusal at all -- the operand enters as a read-only view (products
    and the norm of the first difference read it; nothing has to merge with it)"""
    n, h, thr = 16384, 40, 1e-8
    col, row, val = banded_triplets(n, h)
    assert ((val == 0) & (col == row)).sum() > 0   # (the generator's zero diagonals)
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    nt.set_option("slab_algebra", 1)
    c0 = nt.slab_algebra_counts()
    got, tr, _ = run(nt, "sign", H, n, thr, 1e-8)
    c1 = nt.slab_algebra_counts()
    if nt.slab_counts_exact:
        assert c1["products"] - c0["products"] == 2 * tr["iterations"]
        assert c1["refusals"] - c0["refusals"] == 0
    else:
        assert c1["products"] - c0["products"] >= 2
    nt.set_option("slab_algebra", 0)
    want, tr0, _ = run(nt, "sign", H, n, thr, 1e-8)
    assert tr0["iterations"] == tr["iterations"] and same_pattern(got, want) and np.array_equal(got[2], want[2])


@pytest.mark.parametrize("solver,n,h,thr,shift", ORDER2)
def test_second_order_square_root_in_slab_form(nt, fma, solver, n, h, thr, shift):
    """NewtonSchultzISROrder2 (SquareRootSolversModule.F90: order 2) with the session on and off: bit for bit"""
    col, row, val = banded_triplets(n, h, shift=shift)
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    p = nt.SolverParameters()
    p.SetThreshold(thr)
    p.SetConvergeDiff(1e-8)
    res = []
    for on in (0, 1):
        nt.set_option("slab_algebra", on)
        c0 = nt.slab_algebra_counts()
        Out = nt.Matrix_ps(n)
        nt.SquareRootSolvers.with_order(H, Out, p, True, 2)
        c1 = nt.slab_algebra_counts()
        res.append((srt(Out.triplets()), nt.solver_trace()["iterations"], c1["products"] - c0["products"]))
    assert res[0][2] == 0 and res[1][2] >= (3 * (res[1][1] - 1) if nt.slab_counts_exact else 3)
    assert res[0][1] == res[1][1] and same_pattern(res[0][0], res[1][0]) and np.array_equal(res[0][0][2], res[1][0][2])


def test_callers_own_loop_over_the_c_abi_stays_in_slab_form(nt, fma):
    """A caller's own loop written against the C ABI -- here McWeeny steps X <- 3 X^2 - 2 X^3 spelled with
    MatrixMultiply / CopyMatrix / ScaleMatrix / IncrementMatrix, then DotMatrix, MatrixNorm, GetMatrixSize -- runs its
    vocabulary calls in slab form (every call is a slab session of its own: wrp.cpp ApiSession), and whatever reads the
    matrices afterwards sees compressed columns: results, scalars and entry counts equal the ones with the option off
    bit for bit (the dot to reduction order)."""
    n, h, thr = 8192, 30, 1e-8
    col, row, val = banded_triplets(n, h, shift=2.0)
    pool = None
    out = []
    for on in (0, 1):
        nt.set_option("slab_algebra", on)
        c0 = nt.slab_algebra_counts()
        X = nt.Matrix_ps.from_triplets(n, col, row, val)
        X.Scale(0.2)
        X2, X3, T = nt.Matrix_ps(n), nt.Matrix_ps(n), nt.Matrix_ps(n)
        sizes = []
        for it in range(4):
            X2.Gemm(X, X, pool, 1.0, 0.0, thr)
            X3.Gemm(X2, X, pool, 1.0, 0.0, thr)
            nt.lib.CopyMatrix_ps_wrp(X2.ih, T.ih)
            T.Scale(3.0)
            T.Increment(X3, -2.0, thr)
            nt.lib.CopyMatrix_ps_wrp(T.ih, X.ih)
            sizes.append(X.GetSize())
        d = float(np.real(X.Dot(X2)))
        nrm = X.Norm()
        tr = X.Trace()
        c1 = nt.slab_algebra_counts()
        out.append((srt(X.triplets()), srt(X3.triplets()), sizes, d, nrm, tr, c1["products"] - c0["products"],
                    c1["merges"] - c0["merges"], c1["refusals"] - c0["refusals"]))
    off, on = out
    if nt.slab_counts_exact:
        assert off[6] == 0 and on[6] == 8 and on[7] >= 12 and on[8] == 0, (off[6:], on[6:])
    else:
        assert off[6] == 0 and on[6] >= 1, (off[6:], on[6:])
    assert off[2] == on[2]
    for a, b in ((off[0], on[0]), (off[1], on[1])):
        assert same_pattern(a, b) and np.array_equal(a[2], b[2])
    assert abs(off[3] - on[3]) <= 1e-12 * abs(off[3]) and off[4] == pytest.approx(on[4], rel=1e-13) and off[5] == pytest.approx(on[5], rel=1e-13)


def test_trs2_step_on_an_iterate_the_slab_algebra_left_behind(nt, fma):
    """the step API (ntpoly_amd_trs2_step, what bench.py times) on an iterate that vocabulary calls have just touched --
    scaled, copied, multiplied: slab form without multiplier tiles -- gives what it gives on compressed columns"""
    n, h, thr = 8192, 30, 1e-8
    col, row, val = banded_triplets(n, h)
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    e_min, e_max = nt.EigenBounds.GershgorinBounds(H)
    I = nt.Matrix_ps(n)
    I.FillIdentity()
    res = []
    for on in (0, 1):
        nt.set_option("slab_algebra", on)
        X = nt.Matrix_ps(H)
        X.Scale(-1.0)
        X.Increment(I, e_max, 0.0)
        X.Scale(1.0 / (e_max - e_min))
        X2 = nt.Matrix_ps(n)
        tr, log = None, []
        for it in range(6):
            if it == 3:   # (vocabulary calls between the steps: X <- (X * I) scaled by one)
                T = nt.Matrix_ps(n)
                T.Gemm(X, I, None, 1.0, 0.0, 0.0)
                T.Scale(1.0)
                nt.lib.CopyMatrix_ps_wrp(T.ih, X.ih)
                tr = None
            sigma, energy, tr = nt.trs2_step(X, X2, H, n / 2.0, thr, tr)
            log.append((sigma, energy, tr))
        res.append((log, srt(X.triplets())))
    assert [l[0] for l in res[0][0]] == [l[0] for l in res[1][0]]
    assert np.allclose([l[1] for l in res[0][0]], [l[1] for l in res[1][0]], rtol=1e-12, atol=1e-12)
    assert same_pattern(res[0][1], res[1][1]) and np.abs(res[0][1][2] - res[1][1][2]).max() <= 1e-13


@pytest.mark.parametrize("solver", ["trs4", "sign", "inverse_square_root"])
def test_load_balanced_solves_fall_back_cleanly(nt, fma, solver):
    """under the load balancer's random permutation the loop's matrices are not run-like: every operation of the session
    is refused (operands back to compressed columns, the session gives up after a few) and the results are those of the
    session-less run bit for bit"""
    n, h, thr = 4096, 16, 1e-8
    shift = 0.0 if solver != "inverse_square_root" else 2.0
    col, row, val = banded_triplets(n, h, shift=shift)
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    perm = nt.Permutation(n)
    perm.SetRandomPermutation()
    out = []
    for on in (0, 1):
        nt.set_option("slab_algebra", on)
        p = nt.SolverParameters()
        p.SetThreshold(thr)
        p.SetConvergeDiff(1e-30 if solver == "trs4" else 1e-8)
        p.SetLoadBalance(perm)
        if solver == "trs4":
            p.SetMaxIterations(10)
            p.SetMonitorConvergence(False)
        K = nt.Matrix_ps(n)
        c0 = nt.slab_algebra_counts()
        if solver == "trs4":
            I = nt.Matrix_ps(n)
            I.FillIdentity()
            nt.DensityMatrixSolvers.TRS4(H, I, n / 2.0, K, p)
        elif solver == "sign":
            nt.SignSolvers.ComputeSign(H, K, p)
        else:
            nt.SquareRootSolvers.InverseSquareRoot(H, K, p)
        c1 = nt.slab_algebra_counts()
        out.append((srt(K.triplets()), nt.solver_trace()["iterations"], c1["products"] - c0["products"]))
    assert out[1][2] == 0   # (nothing could run in slab form)
    assert out[0][1] == out[1][1] and same_pattern(out[0][0], out[1][0]) and np.array_equal(out[0][0][2], out[1][0][2])


@pytest.mark.parametrize("solver", ["hpcp", "scale_and_fold"])
def test_other_purification_loops_in_slab_form(nt, fma, solver):
    """HPCP and ScaleAndFold (DensityMatrixSolversModule.F90) open the same session (PM does not: its update scales the
    iterate by zero in half its iterations, which leaves stored zeros no slab can hold): their loops run in slab form
    (products counted) and give the density of the session-less run -- same pattern, values to 1e-10 (their step sizes
    are quotients of traces / dots whose summation order differs), energies to 1e-10 relative"""
    n, h, thr, iters = 4096, 20, 1e-8, 12
    col, row, val = banded_triplets(n, h)
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    I = nt.Matrix_ps(n)
    I.FillIdentity()
    out = []
    for on in (0, 1):
        nt.set_option("slab_algebra", on)
        p = nt.SolverParameters()
        p.SetThreshold(thr)
        p.SetConvergeDiff(1e-30)
        p.SetMaxIterations(iters)
        p.SetMonitorConvergence(False)
        K = nt.Matrix_ps(n)
        c0 = nt.slab_algebra_counts()
        if solver == "pm":
            e, mu = nt.DensityMatrixSolvers.PM(H, I, n / 2.0, K, p)
        elif solver == "hpcp":
            e, mu = nt.DensityMatrixSolvers.HPCP(H, I, n / 2.0, K, p)
        else:
            import ctypes as C
            ev = C.c_double()
            dd = lambda x: C.byref(C.c_double(x))
            nt.lib.ScaleAndFold_wrp(H.ih, I.ih, dd(n / 2.0), K.ih, dd(-0.05), dd(0.05), C.byref(ev), p.ih)
            e = ev.value
        c1 = nt.slab_algebra_counts()
        out.append((srt(K.triplets()), e if not isinstance(e, tuple) else e[0], nt.solver_trace()["iterations"], c1["products"] - c0["products"]))
    off, on = out
    assert off[3] == 0 and on[3] >= (iters - 1 if nt.slab_counts_exact else 1), (off[3], on[3])
    assert off[2] == on[2]
    assert abs(off[1] - on[1]) <= 1e-10 * abs(off[1])
    import scipy.sparse as sp
    G = sp.csr_matrix((on[0][2], (on[0][1] - 1, on[0][0] - 1)), shape=(n, n))
    W = sp.csr_matrix((off[0][2], (off[0][1] - 1, off[0][0] - 1)), shape=(n, n))
    assert abs(G - W).max() <= 1e-9


@pytest.mark.parametrize("kind", ["horner", "paterson_stockmeyer", "chebyshev", "chebyshev_factorized", "hermite", "exponential", "sine", "inverse_root3", "root3"])
def test_polynomial_evaluations_in_slab_form(nt, fma, kind):
    """the matrix polynomials (Horner, Paterson-Stockmeyer, Chebyshev standard / recursive, Hermite) and two of the
    functions built on them: their products, merges and scalings run in slab form under the session and give the
    session-less result bit for bit (a zero coefficient is the one merge that goes back to compressed columns)"""
    n, h, thr = 4096, 12, 1e-9
    col, row, val = banded_triplets(n, h)
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    H.Scale(0.25)   # (spectrum inside [-1, 1] for the orthogonal-polynomial recurrences)
    coeffs = [0.3, -0.7, 0.0, 0.45, 0.2, -0.15, 0.05, 0.11, -0.02]
    out = []
    for on in (0, 1):
        nt.set_option("slab_algebra", on)
        p = nt.SolverParameters()
        p.SetThreshold(thr)
        Out = nt.Matrix_ps(n)
        c0 = nt.slab_algebra_counts()
        if kind in ("horner", "paterson_stockmeyer"):
            P = nt.Polynomial(len(coeffs))
            for k, c in enumerate(coeffs):
                P.SetCoefficient(k, c)
            (P.HornerCompute if kind == "horner" else P.PatersonStockmeyerCompute)(H, Out, p)
        elif kind in ("chebyshev", "chebyshev_factorized"):
            P = nt.ChebyshevPolynomial(len(coeffs))
            for k, c in enumerate(coeffs):
                P.SetCoefficient(k, c)
            (P.Compute if kind == "chebyshev" else P.ComputeFactorized)(H, Out, p)
        elif kind == "hermite":
            P = nt.HermitePolynomial(len(coeffs))
            for k, c in enumerate(coeffs):
                P.SetCoefficient(k, c * 1e-2)
            P.Compute(H, Out, p)
        elif kind in ("inverse_root3", "root3"):
            H3 = nt.Matrix_ps(H)
            I3 = nt.Matrix_ps(n)
            I3.FillIdentity()
            H3.Increment(I3, 1.5, 0.0)   # (positive definite)
            p.SetConvergeDiff(1e-8)
            (nt.RootSolvers.ComputeInverseRoot if kind == "inverse_root3" else nt.RootSolvers.ComputeRoot)(H3, Out, 3, p)
        elif kind == "exponential":
            nt.ExponentialSolvers.ComputeExponential(H, Out, p)
        else:
            nt.TrigonometrySolvers.Sine(H, Out, p)
        c1 = nt.slab_algebra_counts()
        out.append((srt(Out.triplets()), c1["products"] - c0["products"]))
    off, on = out
    assert off[1] == 0 and on[1] >= 2, (off[1], on[1])
    assert same_pattern(off[0], on[0]) and np.array_equal(off[0][2], on[0][2]), np.abs(off[0][2] - on[0][2]).max() if same_pattern(off[0], on[0]) else "pattern"


def _far_band(n, off, h):
    """entries (row, col) with row - col in [off - h, off + h] (mod-free: only inside the matrix): one run per column
    that lies `off` rows away from the diagonal"""
    j = np.arange(n, dtype=np.int64)
    o = np.arange(off - h, off + h + 1, dtype=np.int64)
    col = np.repeat(j, len(o))
    row = col + np.tile(o, n)
    ok = (row >= 0) & (row < n)
    col, row = col[ok], row[ok]
    val = 0.05 + 0.9 * (((row * 7919 + col * 104729) % 1000) / 1000.0)
    return (col + 1).astype(np.int32), (row + 1).astype(np.int32), val


def test_merge_of_slab_operands_with_disjoint_runs_is_refused_not_overflowed(nt, fma):
    """a merge of two slab-form operands whose runs lie far apart in every column (an identity and a
    product whose entries sit n / 2 rows from the diagonal) has a union extent of about n^2 / 2 rows against operands of
    about 20 n slots.  The merge kernel must refuse such a column (the host then merges on compressed columns) instead
    of writing past its output buffer.  Through the C ABI's own vocabulary calls, in both arithmetic modes."""
    import scipy.sparse as sp
    n, h = 4096, 6
    col, row, val = _far_band(n, n // 4, h)
    A = nt.Matrix_ps.from_triplets(n, col, row, val)
    nt.set_option("slab_algebra", 1)
    C1 = nt.Matrix_ps(n)
    C1.Gemm(A, A, threshold=0.0)           # entries n / 2 below the diagonal, left in slab form by the session
    I = nt.Matrix_ps(n)
    I.FillIdentity()
    C1.Increment(I, 0.5, 0.0)              # identity run [j, j] + run around j + n / 2
    B2 = nt.Matrix_ps(A)
    B2.Increment(C1, -1.0, 0.0)            # two far-apart bands + the diagonal
    got = srt(B2.triplets())
    As = sp.csr_matrix((val, (row - 1, col - 1)), shape=(n, n))
    want = (As - (As @ As + 0.5 * sp.identity(n))).tocsc()
    want.sort_indices()
    G = sp.csr_matrix((got[2], (got[1] - 1, got[0] - 1)), shape=(n, n))
    assert G.nnz == want.nnz
    assert abs(G - want).max() <= 1e-12
    # and the TRS4 operand pass on the same kind of operand (X and X^2 far apart) goes back to compressed columns too
    X = nt.Matrix_ps(A)
    X2 = nt.Matrix_ps(n)
    X2.Gemm(X, X, threshold=0.0)
    assert abs(X2.Norm() - abs(As @ As).sum(axis=0).max()) <= 1e-9


def test_mixed_real_complex_product_after_a_slab_form_result(nt, fma):
    """a real product left in slab form by the C ABI's session, then multiplied by a COMPLEX matrix
    (PSMatrixAlgebraModule.F90:171-188 up-casts): the up-cast must pack the slab-form operand first."""
    import scipy.sparse as sp
    n, h = 2048, 10
    col, row, val = banded_triplets(n, h)
    A = nt.Matrix_ps.from_triplets(n, col, row, val)
    nt.set_option("slab_algebra", 1)
    P = nt.Matrix_ps(n)
    P.Gemm(A, A, threshold=0.0)            # real, slab form
    zc, zr, zv = banded_triplets(n, 4, complex_=True)
    Z = nt.Matrix_ps.from_triplets(n, zc, zr, zv)
    Out = nt.Matrix_ps(n)
    Out.Gemm(P, Z, threshold=0.0)
    got = srt(Out.triplets())
    As = sp.csr_matrix((val, (row - 1, col - 1)), shape=(n, n))
    Zs = sp.csr_matrix((zv, (zr - 1, zc - 1)), shape=(n, n))
    want = (As @ As @ Zs).tocsc()
    G = sp.csr_matrix((got[2], (got[1] - 1, got[0] - 1)), shape=(n, n))
    assert abs(G - want).max() <= 1e-11
