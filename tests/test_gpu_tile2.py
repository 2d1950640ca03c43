"""GPU, EXPERIMENT BUILD ONLY (NTPOLY_AMD_WITH_TILE2=1 python -m ntpoly_amd._build; skipped on the product build, which does not
carry the kernel: measured slower than k_spgemm_tile in round 5, profiles/README.md 83).
The two-block geometry of the MFMA kernel (csrc/experiments/spgemm_tile2.hip: pairs of 16-column blocks share every fragment
of A, the multiplier rows stream through LDS in chunks of 32 k) against k_spgemm_tile (option tile2 = 0) and the
oracle's FMA mode.  Same FMA chain over ascending k (MultiplyBlock.f90:9-36 in the reference's FP-contracted build), same
prune (PruneList.f90:8-38), same fused TRS2 update (AddSparseVectors.f90:21-70): results must agree BIT FOR BIT; where a
pair of blocks does not fit the geometry (window beyond 1024 rows, a wave's two slabs in progress together) the launch is
repeated on k_spgemm_tile and nothing changes."""
import numpy as np
import pytest

from gen import banded_triplets

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nt():
    import ntpoly_amd as nt
    nt.init_comm()
    nt.ConstructGlobalProcessGrid(1, 1, 1)
    # is the experiment kernel in this build?  One small product with the option on: the product build's stub declines
    nt.set_option("spgemm_fma", 1)
    nt.set_option("tile2", 1)
    try:
        c0 = nt.tile2_counts()
        A = nt.Matrix_ps.from_triplets(4096, *banded_triplets(4096, 60))
        C = nt.Matrix_ps(4096)
        C.Gemm(A, A, None, 1.0, 0.0, 1e-8)
        c1 = nt.tile2_counts()
    finally:
        nt.set_option("tile2", 0)
        nt.set_option("spgemm_fma", 0)
    if sum(c1.values()) == sum(c0.values()):
        pytest.skip("spgemm_tile2.hip is not part of this build (experiment: NTPOLY_AMD_WITH_TILE2=1)")
    return nt


@pytest.fixture()
def fma(nt):
    from oracle import oracle_py as O
    nt.set_option("spgemm_fma", 1)
    O.set_fma(True)
    yield O
    O.set_fma(False)
    nt.set_option("spgemm_fma", 0)
    nt.set_option("tile2", 0)
    nt.set_option("slab_algebra", 1)


def srt(t):
    c, r, v = t
    o = np.lexsort((r, c))
    return c[o], r[o], v[o]


def exact(got, want, what):
    g, w = srt(got), srt(want)
    assert len(g[2]) == len(w[2]), "%s: %d vs %d entries" % (what, len(g[2]), len(w[2]))
    assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1]), what + ": pattern differs"
    assert np.array_equal(g[2], w[2]), "%s: values differ, max |d| = %g" % (what, np.abs(g[2] - w[2]).max())


def holes(col, row, val, frac, seed):
    rng = np.random.default_rng(seed)
    keep = (rng.random(len(val)) >= frac) | (col == row)
    return col[keep], row[keep], val[keep]


def solve(nt, n, trip, thr, iters):
    H = nt.Matrix_ps.from_triplets(n, *trip)
    I = nt.Matrix_ps(n)
    I.FillIdentity()
    p = nt.SolverParameters()
    p.SetThreshold(thr)
    p.SetConvergeDiff(1e-30)
    p.SetMaxIterations(iters)
    p.SetMonitorConvergence(False)
    K = nt.Matrix_ps(n)
    f0, t0 = nt.fusion_counts(), nt.tile2_counts()
    e, mu = nt.DensityMatrixSolvers.TRS2(H, I, n / 2.0, K, p)
    f1, t1 = nt.fusion_counts(), nt.tile2_counts()
    tr = nt.solver_trace()
    return dict(K=K.triplets(), e=e, mu=mu, log=np.array(tr["energy"]), nnz=np.array(tr["nnz"]), sigma=np.array(tr["sigma"]),
                fused=f1["square"] + f1["update"] - f0["square"] - f0["update"], rep=f1["repeated"] - f0["repeated"],
                t2=t1["done"] - t0["done"], t2rep=t1["repeated"] - t0["repeated"])


# (n, halfband, holes, threshold, iterations): a ragged last pair (odd number of column blocks), holes in the runs, a
# threshold of zero, bands of 60 .. 460 rows (windows of 300 .. 1000 rows: one and two slabs per wave)
@pytest.mark.parametrize("n,h,hl,thr,iters", [(8192, 40, 0.0, 1e-7, 12), (4099, 25, 0.0, 1e-6, 10), (6000, 25, 0.3, 1e-6, 9),
                                               (4112, 100, 0.0, 1e-8, 8), (1024, 10, 0.0, 0.0, 5), (6144, 150, 0.1, 1e-8, 7)])
def test_fused_trs2_steps_two_block_geometry_equals_the_tile_kernel_and_the_oracle(nt, fma, n, h, hl, thr, iters):
    O = fma
    trip = banded_triplets(n, h)
    if hl:
        trip = holes(*trip, hl, n + h)
    nt.set_option("tile2", 0)
    ref = solve(nt, n, trip, thr, iters)
    nt.set_option("tile2", 1)
    got = solve(nt, n, trip, thr, iters)
    assert ref["t2"] == 0 and got["t2"] >= iters - 2, (ref["t2"], got["t2"], got["t2rep"], got["fused"], got["rep"])
    assert got["t2rep"] == 0 and got["rep"] == ref["rep"] and got["fused"] == ref["fused"]
    exact(got["K"], ref["K"], "density, tile2 vs tile")
    assert np.array_equal(got["nnz"], ref["nnz"]) and np.array_equal(got["sigma"], ref["sigma"])
    # (energy and trace are sums over a block's slabs: the two geometries add the same terms in a different -- fixed -- order)
    assert np.allclose(got["log"], ref["log"], rtol=1e-12, atol=1e-10), np.abs(got["log"] - ref["log"]).max()
    # ... and the oracle's FMA mode (pattern equal, values 1e-13: the spectral bounds are reductions)
    Ho = O.Mat.from_triplets(n, n, *trip)
    po = O.params(converge_diff=1e-30, max_iterations=iters, threshold=thr, monitor_convergence=False)
    Ko, eo, muo, tro = O.density("trs2", Ho, O.Mat.identity(n), n / 2.0, po)
    g, w = srt(got["K"]), srt(Ko.triplets())
    assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1])
    assert np.allclose(g[2], w[2], rtol=0, atol=1e-13)
    assert np.array_equal(got["nnz"], np.array(tro["nnz"]))


@pytest.mark.parametrize("n,ha,hb,hl,thr,alpha", [(8192, 60, 40, 0.0, 1e-8, 1.0), (4099, 100, 120, 0.2, 1e-7, -0.5),
                                                  (5000, 12, 200, 0.0, 0.0, 2.0), (3000, 200, 30, 0.4, 1e-6, 1.0)])
def test_products_in_slab_form_two_block_geometry(nt, fma, n, ha, hb, hl, thr, alpha):
    """C = alpha A B over the C ABI (every call a slab session: slab_multiply, the right operand by its runs), A != B"""
    O = fma
    ta, tb = banded_triplets(n, ha), banded_triplets(n, hb, shift=0.2)
    if hl:
        ta, tb = holes(*ta, hl, 1), holes(*tb, hl, 2)
    out = {}
    for on in (0, 1):
        nt.set_option("tile2", on)
        A, B, C = nt.Matrix_ps.from_triplets(n, *ta), nt.Matrix_ps.from_triplets(n, *tb), nt.Matrix_ps(n)
        t0 = nt.tile2_counts()
        C.Gemm(A, B, None, alpha, 0.0, thr)
        t1 = nt.tile2_counts()
        out[on] = (C.triplets(), t1["done"] - t0["done"], t1["repeated"] - t0["repeated"], nt.last_spgemm_stats()["slab"])
    assert out[0][1] == 0 and out[0][3] == 1 and out[1][3] == 1
    assert out[1][1] == 1 and out[1][2] == 0, out[1][1:]
    exact(out[1][0], out[0][0], "tile2 vs tile")
    Ao, Bo = O.Mat.from_triplets(n, n, *ta), O.Mat.from_triplets(n, n, *tb)
    exact(out[1][0], O.ps_multiply(Ao, Bo, None, alpha, 0.0, thr).triplets(), "tile2 vs oracle")


def test_wide_runs_against_a_narrow_right_operand(nt, fma):
    """A wide left operand (runs of ~640 rows) against a narrow right one: slabs 16 apart are reached by the same k groups
    -- no constraint in this geometry (the waves take their slabs one after the other, the multipliers stay in LDS)"""
    O = fma
    n, thr = 6144, 1e-9
    ta, tb = banded_triplets(n, 320), banded_triplets(n, 20, shift=0.2)
    A, B, C = nt.Matrix_ps.from_triplets(n, *ta), nt.Matrix_ps.from_triplets(n, *tb), nt.Matrix_ps(n)
    t0 = nt.tile2_counts()
    C.Gemm(A, B, None, 1.0, 0.0, thr)
    t1 = nt.tile2_counts()
    assert nt.last_spgemm_stats()["slab"] == 1
    assert t1["done"] - t0["done"] == 1 and t1["repeated"] == t0["repeated"], (t0, t1)
    Ao, Bo = O.Mat.from_triplets(n, n, *ta), O.Mat.from_triplets(n, n, *tb)
    exact(C.triplets(), O.ps_multiply(Ao, Bo, None, 1.0, 0.0, thr).triplets(), "product vs oracle")


def test_a_pair_that_does_not_fit_is_repeated_on_the_tile_kernel(nt, fma):
    """Neighbouring column blocks whose k ranges lie 4000 rows apart (every other block of the right operand shifted): each
    block alone is narrow, the pair's union is not -- the kernel says so, the product is repeated on k_spgemm_tile, the
    result is the oracle's."""
    O = fma
    n, thr, hw = 8192, 1e-9, 30
    j = np.arange(n, dtype=np.int64)
    centre = np.where((j // 16) % 2 == 0, j, (j + 4000) % n)
    offs = np.arange(-hw, hw + 1, dtype=np.int64)
    rows = centre[:, None] + offs[None, :]
    cols = np.repeat(j[:, None], len(offs), axis=1)
    ok = (rows >= 0) & (rows < n)
    vals = 0.3 * np.cos(0.37 * rows + 0.11 * cols) / (1.0 + np.abs(offs)[None, :])
    tb = ((cols[ok] + 1).astype(np.int32), (rows[ok] + 1).astype(np.int32), vals[ok])
    ta = banded_triplets(n, 30)
    A, B, C = nt.Matrix_ps.from_triplets(n, *ta), nt.Matrix_ps.from_triplets(n, *tb), nt.Matrix_ps(n)
    t0 = nt.tile2_counts()
    C.Gemm(A, B, None, 1.0, 0.0, thr)
    t1 = nt.tile2_counts()
    Ao, Bo = O.Mat.from_triplets(n, n, *ta), O.Mat.from_triplets(n, n, *tb)
    exact(C.triplets(), O.ps_multiply(Ao, Bo, None, 1.0, 0.0, thr).triplets(), "repeated product vs oracle")
    if nt.last_spgemm_stats()["slab"] == 1:   # (the operands were taken as run-like: the two-block launch must have been refused)
        assert t1["repeated"] - t0["repeated"] == 1 and t1["done"] == t0["done"], (t0, t1)
    # asked again with the same left operand: the result does not change
    C2 = nt.Matrix_ps(n)
    C2.Gemm(A, B, None, 1.0, 0.0, thr)
    exact(C2.triplets(), C.triplets(), "second product")
