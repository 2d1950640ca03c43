"""GPU: the configuration an unmodified caller of the C ABI gets (the in-process tests state their arithmetic
themselves, so the shipped defaults -- FMA chain, block path, complex tile kernel and sessions, thin-operand kernels --
were only exercised where a test opted in).  A fresh process with no option and no environment override runs the
vocabulary and two solvers (tests/defaults_worker.py); its results are checked against the oracle in the arithmetic the
defaults promise: the FMA mode of the oracle -- bit for bit on the label-ordered paths, 1e-13 where
the engine multiplies in an order of its own (block path), the stated tolerance for complex operands."""
import os
import subprocess
import sys

import numpy as np
import pytest

from gen import banded_triplets, lattice_triplets

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def srt(c, r, v):
    o = np.lexsort((r, c))
    return c[o], r[o], v[o]


@pytest.fixture(scope="module")
def defaults(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("defaults") / "d")
    env = {k: v for k, v in os.environ.items() if not k.startswith("NTPOLY_AMD_")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "defaults_worker.py"), out], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    return dict(np.load(out + ".npz"))


def test_the_defaults_are_what_the_documents_say(defaults):
    fma, block, ctile, csess, thin, slab, tile2 = (int(x) for x in defaults["options"])
    assert (fma, block, ctile, csess, thin, slab, tile2) == (1, 1, 1, 1, 1, 1, 0)


def test_trs2_wrp_with_defaults_vs_oracle_fma(defaults):
    from oracle import oracle_py as O
    n, h = 8192, 40
    O.set_fma(True)
    try:
        Ho = O.Mat.from_triplets(n, n, *banded_triplets(n, h))
        po = O.params(converge_diff=1e-30, max_iterations=12, threshold=1e-7, monitor_convergence=False)
        Ko, eo, muo, tro = O.density("trs2", Ho, O.Mat.identity(n), n / 2.0, po)
    finally:
        O.set_fma(False)
    g = srt(defaults["trs2_K_col"], defaults["trs2_K_row"], defaults["trs2_K_val"])
    w = srt(*Ko.triplets())
    assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1])
    assert np.allclose(g[2], w[2], rtol=0, atol=1e-13)
    assert np.array_equal(defaults["trs2_nnz"], np.array(tro["nnz"]))
    assert defaults["trs2_scal"][0] == pytest.approx(eo, rel=1e-11)
    sq, up, rep = defaults["trs2_fused"]
    assert rep == 0 and sq + up >= 10      # the steps ran inside the SpGEMM kernel


def test_callers_loop_on_a_lattice_with_defaults(defaults):
    """three McWeeny steps over the C ABI: every product on the block path, the iterate within 1e-12 of the oracle's"""
    from oracle import oracle_py as O
    L, thr = 24, 1e-9
    m = L ** 3
    assert np.all(defaults["mcweeny_block_used"] == 1), defaults["mcweeny_block_used"]
    O.set_fma(True)
    try:
        X = O.Mat.from_triplets(m, m, *lattice_triplets(L, shift=2.0))
        O.scale(X, 0.2)
        for it in range(3):
            X2 = O.ps_multiply(X, X, None, 1.0, 0.0, thr)
            X3 = O.ps_multiply(X2, X, None, 1.0, 0.0, thr)
            T = X2.copy()
            O.scale(T, 3.0)
            T = O.increment(X3, T, -2.0, thr)
            X = T
    finally:
        O.set_fma(False)
    import scipy.sparse as sp
    c, r, v = defaults["mcweeny_X_col"], defaults["mcweeny_X_row"], defaults["mcweeny_X_val"]
    G = sp.csr_matrix((v, (r - 1, c - 1)), shape=(m, m))
    W = X.to_scipy().tocsr()
    D = (G - W).tocoo()
    scale = np.abs(W.data).max()
    big = np.abs(D.data) > 1e-12 * scale
    assert np.all(np.abs(D.data[big]) <= thr * (1 + 1e-6)), np.abs(D.data).max()   # (only entries at the threshold may differ)
    assert abs(G.nnz - W.nnz) <= max(8, 1e-5 * W.nnz)


def test_complex_sign_with_defaults_vs_oracle(defaults):
    from oracle import oracle_py as O
    nc, hc = 2048, 24
    Ho = O.Mat.from_triplets(nc, nc, *banded_triplets(nc, hc, complex_=True))
    po = O.params(converge_diff=1e-6, threshold=1e-8)
    So, tro = O.matrix_function("sign", Ho, po)
    assert int(defaults["sign_iters"][0]) == tro["iterations"]
    import scipy.sparse as sp
    c, r, v = defaults["sign_S_col"], defaults["sign_S_row"], defaults["sign_S_val"]
    G = sp.csr_matrix((v, (r - 1, c - 1)), shape=(nc, nc))
    W = So.to_scipy().tocsr()
    D = (G - W).tocoo()
    assert np.abs(D.data).max() <= 1e-8 * (1 + 1e-3) + 1e-10, np.abs(D.data).max()
