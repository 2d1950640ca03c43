"""GPU: the engine's second arithmetic mode (option spgemm_fma = 1) -- every product entry is the chain of fma() over
ascending k, one rounding per product, which is what the reference computes when it is built with FP contraction
(`oracle/build_ref.py --fma`; MultiplyBlock.f90:33 becomes one FMA).  For real run-like operands the chain runs on the
FP64 matrix cores (spgemm_tile.hip: v_mfma_f64_16x16x4_f64 accumulates its four products exactly like four fma()s in
ascending k); the other real kernels (LDS window, LDS hash, grouped hash, HBM accumulator) use v_fma_f64.

Pinning: the oracle's FMA mode equals the contracted reference build bit for bit (tests/test_oracle_golden.py::
test_fma_mode_vs_contracted_reference_build, tests/golden/ps_gemm_fma.npz); here the engine must equal the oracle's FMA
mode BIT FOR BIT on every kernel path, and whole TRS2 solves must give the same sigma sequence, entry counts and
density (same pattern, values to 1e-13: the spectral bounds are reductions) with energies to reduction-order roundoff.
The tolerance contract against the reference's default (unfused) build -- same iteration counts as its logs, energies
1e-11 -- is asserted by tests/test_gpu_scale.py, which runs in both arithmetic modes."""
import numpy as np
import pytest

from gen import banded_triplets, permuted_banded_triplets

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
    O.set_fma(True)
    yield O
    O.set_fma(False)
    nt.set_option("spgemm_fma", 0)
    nt.set_option("tile_rows", 2)
    nt.set_option("tile_waves", 0)
    nt.set_option("spgemm_variant", -1)
    nt.set_option("spgemm_force_bin", -1)
    nt.set_option("block_path", 1)


def srt(t):
    c, r, v = t
    o = np.lexsort((r, c))
    return c[o], r[o], v[o]


def exact(got, want, what):
    g, w = srt(got), srt(want)
    assert len(g[2]) == len(w[2]), "%s: %d vs %d entries" % (what, len(g[2]), len(w[2]))
    assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1]), what + ": pattern differs"
    assert np.array_equal(g[2], w[2]), "%s: values differ, max |d| = %g" % (what, np.abs(g[2] - w[2]).max())


def holes_pair(n, h, holes, seed):
    rng = np.random.default_rng(seed)
    mats = []
    for t in range(2):
        col, row, val = banded_triplets(n, h, shift=0.1 * t)
        keep = (rng.random(len(val)) >= holes) | (col == row)
        mats.append((col[keep], row[keep], val[keep] * (1.0 + 0.01 * t)))
    return mats


@pytest.mark.parametrize("rows,waves", [(1, 4), (2, 4), (2, 8), (4, 4)])   # (suite budget: the two remaining geometries differ in occupancy only)
@pytest.mark.parametrize("n,h,holes,thr,alpha", [(4096, 100, 0.0, 1e-8, 1.0), (4099, 140, 0.2, 1e-6, 0.5),
                                                 (3000, 30, 0.5, 0.0, -0.75), (5000, 320, 0.0, 1e-8, 1.0),
                                                 (777, 3, 0.3, 1e-3, 2.0), (6144, 450, 0.05, 1e-7, 1.0)])
def test_tile_kernel_vs_oracle(nt, fma, rows, waves, n, h, holes, thr, alpha):
    """MFMA tile kernel, A * B with A != B and A * A, banded operands with random holes (zero padding of the expanded
    runs, ragged last block, k ranges up to ~900): bit-exact against the oracle's FMA mode for 1 / 2 / 4 rows per lane
    and both workgroup sizes."""
    O = fma
    nt.set_option("tile_rows", rows)
    nt.set_option("tile_waves", waves)
    mats = holes_pair(n, h, holes, n + h)
    A = nt.Matrix_ps.from_triplets(n, *mats[0])
    B = nt.Matrix_ps.from_triplets(n, *mats[1])
    Ao = O.Mat.from_triplets(n, n, *mats[0])
    Bo = O.Mat.from_triplets(n, n, *mats[1])
    for (X, Y, Xo, Yo, tag) in ((A, B, Ao, Bo, "A*B"), (A, A, Ao, Ao, "A*A")):
        C = nt.Matrix_ps(n)
        C.Gemm(X, Y, None, alpha, 0.0, thr)
        if holes == 0.0:
            assert nt.last_spgemm_stats()["slab"] == 1, tag
        exact(C.triplets(), O.ps_multiply(Xo, Yo, None, alpha, 0.0, thr).triplets(), "%s n=%d h=%d" % (tag, n, h))


def test_tile_kernel_equals_the_fma_slab_loop(nt, fma):
    """the same FMA chain through the v_fma_f64 loop of the register-slab kernel (spgemm_fma = 3): identical bits"""
    n, h, thr = 8192, 100, 1e-8
    mats = holes_pair(n, h, 0.1, 5)
    A = nt.Matrix_ps.from_triplets(n, *mats[0])
    B = nt.Matrix_ps.from_triplets(n, *mats[1])
    out = {}
    for mode in (1, 3):
        nt.set_option("spgemm_fma", mode)
        C = nt.Matrix_ps(n)
        C.Gemm(A, B, None, 1.0, 0.0, thr)
        assert nt.last_spgemm_stats()["slab"] == 1
        out[mode] = C.triplets()
    exact(out[1], out[3], "tile vs fma loop")


@pytest.mark.parametrize("force_bin,variant", [(-1, -1), (-1, 501), (1, -1), (2, -1), (5, -1), (6, -1)])
def test_other_real_kernels_fma_vs_oracle(nt, fma, force_bin, variant):
    """operands without run structure (a seeded relabelling of a band) and forced kernel paths -- grouped LDS hash,
    per-column LDS hash, LDS windows, HBM accumulator: the same FMA chain, bit-exact against the oracle's FMA mode"""
    O = fma
    n, h, thr = 4096, 40, 1e-8
    col, row, val = permuted_banded_triplets(n, h, 7) if force_bin < 0 else banded_triplets(n, h)
    A = nt.Matrix_ps.from_triplets(n, col, row, val)
    nt.set_option("spgemm_force_bin", force_bin)
    nt.set_option("spgemm_variant", variant)
    # (this test is about the LDS kernels, whose chain runs over ascending LABEL; left to itself the engine multiplies a
    # relabelled band through the block path -- chain over ascending position, tests/test_gpu_block.py)
    nt.set_option("block_path", 0)
    C = nt.Matrix_ps(n)
    C.Gemm(A, A, None, 1.0, 0.0, thr)
    st = nt.last_spgemm_stats()
    if force_bin > 0:
        assert st["slab"] == 0
    Ao = O.Mat.from_triplets(n, n, col, row, val)
    exact(C.triplets(), O.ps_multiply(Ao, Ao, None, 1.0, 0.0, thr).triplets(), "bin %d variant %d" % (force_bin, variant))


@pytest.mark.parametrize("n,h,thr,iters,holes", [(8192, 40, 1e-7, 14, False), (4099, 25, 1e-6, 12, False),
                                                 (6000, 25, 1e-6, 10, True), (6144, 300, 1e-6, 8, False),
                                                 (1024, 10, 0.0, 5, False)])
def test_trs2_fma_vs_oracle(nt, fma, n, h, thr, iters, holes):
    """whole TRS2 solves in FMA arithmetic -- the first step from compressed columns, the following ones on the slab
    form through the fused epilogues of the tile kernel -- against the oracle's FMA mode: sigma and entry count of
    every iteration, energies, and the density (pattern equal, values 1e-13)."""
    O = fma
    col, row, val = banded_triplets(n, h)
    if holes:
        a, b = np.minimum(col, row).astype(np.int64), np.maximum(col, row).astype(np.int64)
        keep = (((a * 2654435761 + b * 40503) >> 7) % 10 >= 3) | (col == row)
        col, row, val = col[keep], row[keep], val[keep]
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    ISQ = nt.Matrix_ps(n)
    ISQ.FillIdentity()
    p = nt.SolverParameters()
    p.SetConvergeDiff(1e-30)
    p.SetThreshold(thr)
    p.SetMaxIterations(iters)
    p.SetMonitorConvergence(False)
    K = nt.Matrix_ps(n)
    f0 = nt.fusion_counts()
    energy, mu = nt.DensityMatrixSolvers.TRS2(H, ISQ, n / 2.0, K, p)
    f1 = nt.fusion_counts()
    tr = nt.solver_trace()
    if not holes and thr > 0.0:
        # (threshold 0: stored zeros, the merge inside the kernel steps aside; holes: another SpGEMM path may come first)
        assert f1["square"] + f1["update"] - f0["square"] - f0["update"] >= iters - 2
    Ho = O.Mat.from_triplets(n, n, col, row, val)
    Ko, e_o, mu_o, tro = O.density("trs2", Ho, O.Mat.identity(n), n / 2.0,
                                   O.params(converge_diff=1e-30, max_iterations=iters, threshold=thr,
                                            monitor_convergence=False))
    assert tr["iterations"] == tro["iterations"] == iters
    assert np.array_equal(np.asarray(tr["sigma"]), np.asarray(tro["sigma"]))
    assert np.allclose(tr["energy"], tro["energy"], rtol=1e-11, atol=1e-11)
    # (the spectral bounds that scale H come from reductions whose summation order differs between the engine and the
    # oracle: the iterates may differ in their last bits from the first step on; products themselves are bit-exact,
    # test_tile_kernel_vs_oracle)
    g, w = srt(K.triplets()), srt(Ko.triplets())
    assert len(g[2]) == len(w[2]) and np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1])
    assert np.abs(g[2] - w[2]).max() <= 1e-13


def test_fma_and_unfused_modes_agree_to_roundoff(nt, fma):
    """the two arithmetic modes are two builds of the reference: entry by entry within 1e-13 relative, and not equal"""
    O = fma
    n, h, thr = 4096, 100, 0.0
    col, row, val = banded_triplets(n, h)
    A = nt.Matrix_ps.from_triplets(n, col, row, val)
    C1 = nt.Matrix_ps(n)
    C1.Gemm(A, A, None, 1.0, 0.0, thr)
    nt.set_option("spgemm_fma", 0)
    C0 = nt.Matrix_ps(n)
    C0.Gemm(A, A, None, 1.0, 0.0, thr)
    g1, g0 = srt(C1.triplets()), srt(C0.triplets())
    assert np.array_equal(g1[0], g0[0]) and np.array_equal(g1[1], g0[1])
    assert not np.array_equal(g1[2], g0[2])
    assert np.abs(g1[2] - g0[2]).max() <= 1e-13 * np.abs(g0[2]).max()


@pytest.mark.parametrize("n,h,holes,thr,same", [(8192, 50, 0.0, 1e-8, True), (6000, 100, 0.15, 1e-7, False), (7000, 160, 0.0, 1e-8, True),
                                                 (5000, 30, 0.3, 0.0, False), (4097, 300, 0.0, 1e-6, True)])
def test_grouped_hash_products_on_the_matrix_cores(nt, fma, n, h, holes, thr, same):
    """the grouped LDS-hash SpGEMM (csrc/spgemm_grouped.hip: operands WITHOUT runs as they stand -- `bench.py --random`) in FMA
    arithmetic with the products of a phase on the FP64 matrix cores (option ghash_mfma: four steps = the k dimension of one
    v_mfma_f64_16x16x4_f64 per 16 slots x 16 columns): bit for bit the oracle's FMA mode and the vector-unit path, every group
    computed by the kernel (no fallback beyond what the vector path hands back)."""
    import scipy.sparse as sp
    O = fma
    rng = np.random.default_rng(n + h)
    mats = []
    for t in range(1 if same else 2):
        col, row, val = permuted_banded_triplets(n, h, 42, shift=0.1 * t)
        keep = (rng.random(len(val)) >= holes) | (col == row)
        mats.append((col[keep], row[keep], val[keep]))
    A = nt.Matrix_ps.from_triplets(n, *mats[0])
    B = A if same else nt.Matrix_ps.from_triplets(n, *mats[1])
    alpha = 1.0 if same else -0.5
    nt.set_option("block_path", 0)
    res = []
    for mf in (1, 0):
        nt.set_option("ghash_mfma", mf)
        nt.set_option("spgemm_variant", 500)     # (the grouped kernel, forced)
        try:
            C = nt.Matrix_ps(n)
            C.Gemm(A, B, None, alpha, 0.0, thr)
            st, gs = nt.last_spgemm_stats(), nt.last_grouped_stats()
        finally:
            nt.set_option("spgemm_variant", -1)
            nt.set_option("ghash_mfma", 1)
        assert st["slab"] == 0 and gs["used"] == 1 and gs["minhash"] == 1, (mf, st, gs)
        res.append((srt(C.triplets()), gs))
    Ao = O.Mat.from_triplets(n, n, *mats[0])
    Bo = Ao if same else O.Mat.from_triplets(n, n, *mats[1])
    want = O.ps_multiply(Ao, Bo, None, alpha, 0.0, thr).triplets()
    exact(res[0][0], want, "grouped hash, matrix cores")
    exact(res[1][0], want, "grouped hash, vector units")
    assert res[0][1]["failed_cols"] <= res[1][1]["failed_cols"] + n // 50, (res[0][1], res[1][1])
