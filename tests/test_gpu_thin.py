"""Products with a thin left operand (ntpoly_amd/csrc/spgemm_thin.hip: identities, near-diagonal factors of the square-root
loops) against the oracle's multiply (MultiplyBlock.f90:9-36, PruneList.f90:8-38): BIT-EXACT in both arithmetic modes, real
and complex -- the kernel accumulates every entry over ascending k with the reference's own multiply-add."""
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


@pytest.fixture(params=["unfused", "fma"])
def arith(nt, request):
    from oracle import oracle_py as O
    fma = request.param == "fma"
    nt.set_option("spgemm_fma", 1 if fma else 0)
    O.set_fma(fma)
    yield request.param
    O.set_fma(False)
    nt.set_option("spgemm_fma", 0)


def srt(t):
    c, r, v = (np.asarray(x) for x in t)
    o = np.lexsort((r, c))
    return c[o], r[o], v[o]


def exact(got, want, what):
    g, w = srt(got), srt(want)
    assert len(g[2]) == len(w[2]), "%s: %d vs %d entries" % (what, len(g[2]), len(w[2]))
    assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1]), what + ": pattern differs"
    assert np.array_equal(g[2], w[2]), "%s: values differ, max |d| = %g" % (what, np.abs(g[2] - w[2]).max())


def thin_triplets(n, reach, extra, cplx, seed, drop_diag=0.0):
    """diagonal (some of it missing) + `extra` entries per hundred rows within `reach` of the diagonal; 1-based (col, row, val)"""
    rng = np.random.default_rng(seed)
    d = np.arange(n)
    keep = rng.random(n) >= drop_diag
    rows = [d[keep]]
    cols = [d[keep]]
    ne = int(n * extra / 100)
    r = rng.integers(0, n, ne)
    c = np.clip(r + rng.integers(-reach, reach + 1, ne), 0, n - 1)
    rows.append(r)
    cols.append(c)
    row = np.concatenate(rows)
    col = np.concatenate(cols)
    key = col.astype(np.int64) * n + row
    _, first = np.unique(key, return_index=True)
    row, col = row[first], col[first]
    val = 1.0 + 0.25 * rng.standard_normal(len(row))
    if cplx:
        val = val + 0.3j * rng.standard_normal(len(row))
    return (col + 1).astype(np.int32), (row + 1).astype(np.int32), val


@pytest.mark.parametrize("n,h,holes,reach,extra,cplx,thr,alpha", [
    (6000, 120, 0.0, 0, 0, False, 1e-8, 1.0),          # identity-like (diagonal only)
    (6000, 120, 0.1, 40, 30, False, 1e-7, -0.5),       # near-diagonal factor
    (5003, 200, 0.3, 300, 150, False, 0.0, 2.0),       # several entries per row, far reach, threshold 0
    (6000, 60, 0.0, 0, 0, True, 1e-8, 1.0),
    (6000, 100, 0.2, 50, 60, True, 1e-7, 0.75),
    (4099, 240, 0.05, 200, 200, True, 1e-9, 1.0),
])
def test_thin_left_vs_oracle(nt, arith, n, h, holes, reach, extra, cplx, thr, alpha):
    from oracle import oracle_py as O
    rng = np.random.default_rng(n + h)
    col, row, val = banded_triplets(n, h, complex_=cplx)
    keep = (rng.random(len(val)) >= holes)
    keep &= ~np.isin(col, [7, 8, 9, n - 3])          # a few empty columns of B
    Bt = (col[keep], row[keep], val[keep])
    At = thin_triplets(n, reach, extra, cplx, seed=n + reach, drop_diag=0.02)
    A = nt.Matrix_ps.from_triplets(n, *At)
    B = nt.Matrix_ps.from_triplets(n, *Bt)
    C = nt.Matrix_ps(n)
    C.Gemm(A, B, None, alpha, 0.0, thr)
    # (real operands: a diagonal is run-like -- runs of one row -- and the slab session of the call multiplies it in slab form;
    # under FMA arithmetic the tile kernel's sessions take near-diagonal operands as well)
    if cplx or reach > 0:
        assert nt.last_spgemm_thin() == 1
    Ao = O.Mat.from_triplets(n, n, *At)
    Bo = O.Mat.from_triplets(n, n, *Bt)
    want = O.ps_multiply(Ao, Bo, None, alpha, 0.0, thr).triplets()
    exact(C.triplets(), want, "thin A * B n=%d cplx=%d %s" % (n, cplx, arith))
    # the same product with the kernel switched off: the general kernels agree bit for bit as well
    nt.set_option("thin_left", 0)
    try:
        C0 = nt.Matrix_ps(n)
        C0.Gemm(A, B, None, alpha, 0.0, thr)
        assert nt.last_spgemm_thin() == 0
    finally:
        nt.set_option("thin_left", 1)
    exact(C0.triplets(), want, "general kernels, thin A * B")
    # the thin operand on the right (real operands under FMA arithmetic: the slab session's gather kernel, spgemm_thin.hip
    # k_thin_slab_right; otherwise the column-driven kernels, which walk few entries per column there)
    C1 = nt.Matrix_ps(n)
    C1.Gemm(B, A, None, alpha, 0.0, thr)
    exact(C1.triplets(), O.ps_multiply(Bo, Ao, None, alpha, 0.0, thr).triplets(), "B * thin A")
    # thin * thin
    C2 = nt.Matrix_ps(n)
    C2.Gemm(A, A, None, alpha, 0.0, thr)
    exact(C2.triplets(), O.ps_multiply(Ao, Ao, None, alpha, 0.0, thr).triplets(), "thin A * A")


def test_thin_left_declines_far_entries(nt):
    """a left operand with entries far from its diagonal (a permuted identity) is not this kernel's: the general kernels take it"""
    from oracle import oracle_py as O
    n, h = 5000, 50
    rng = np.random.default_rng(3)
    perm = rng.permutation(n)
    At = ((np.arange(n) + 1).astype(np.int32), (perm + 1).astype(np.int32), np.ones(n))
    col, row, val = banded_triplets(n, h)
    A = nt.Matrix_ps.from_triplets(n, *At)
    B = nt.Matrix_ps.from_triplets(n, col, row, val)
    C = nt.Matrix_ps(n)
    C.Gemm(A, B, None, 1.0, 0.0, 1e-9)
    assert nt.last_spgemm_thin() == 0
    want = O.ps_multiply(O.Mat.from_triplets(n, n, *At), O.Mat.from_triplets(n, n, col, row, val), None, 1.0, 0.0, 1e-9).triplets()
    exact(C.triplets(), want, "permutation * B")


def test_kept_transposes_are_released_with_the_operand_caches(nt):
    """the transposes the thin-left path keeps per left operand (two slots) are device memory held between calls:
    ntpoly_amd_release_cache() returns it with the other operand caches (the engine's in-use figure goes back)"""
    n = 4096
    At = thin_triplets(n, 2, 3, True, seed=5, drop_diag=0.0)
    col, row, val = banded_triplets(n, 40, complex_=True)
    nt.set_option("spgemm_fma", 0)
    nt.release_cache()
    A = nt.Matrix_ps.from_triplets(n, *At)
    B = nt.Matrix_ps.from_triplets(n, col, row, val)
    C = nt.Matrix_ps(n)
    nt.synchronize()
    before = nt.memory()[0]
    C.Gemm(A, B, None, 1.0, 0.0, 1e-9)
    assert nt.last_spgemm_thin() == 1
    del C
    nt.synchronize()
    held = nt.memory()[0]
    assert held > before, "the product left nothing behind: the kept transpose is gone, update this test"
    nt.release_cache()
    nt.synchronize()
    assert nt.memory()[0] <= before, (before, held, nt.memory()[0])
