"""GPU: BASELINE configs[3] -- ONE distributed A * A at N = 1 048 576, 201 per row (the 8-GPU config of the reference's
2.5-D multiply, distributed_algebra_includes/MatrixMultiply.f90:92-267) -- as far as a one-GPU box can take it:

  (a) the product in the library's DEFAULT arithmetic (the FMA chain on the matrix cores: k_spgemm_tile in natural order,
      the block path under a random relabelling) against the oracle's FMA mode, BIT FOR BIT, on a seeded sample of 4 096
      columns (sixteen clusters of 256, the two ends of the matrix among them).  The oracle multiplies exactly what those
      columns need -- C(:, S) = A(:, K) * A(K, S), K = the rows of A(:, S) in ascending order, so every entry is the same
      chain over ascending k as in the full product (MultiplyBlock.f90:9-36, PruneList.f90:8-38);
  (b) the same product on 2 and 4 ranks (processes sharing the GPU over the shared-memory test transport): the additive
      digests of the panels sum to the digest of the one-rank product -- the same entries with the same bits.

What stays untested here is the 8 x RCCL run itself (no multi-GPU hardware)."""
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

from gen import banded_triplets, permuted_banded_triplets

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, H, THR = 1048576, 100, 1e-8
_ONE_RANK_DIGEST = {}


@pytest.fixture(scope="module")
def nt():
    import ntpoly_amd as nt
    nt.init_comm()
    nt.ConstructGlobalProcessGrid(1, 1, 1)
    return nt


@pytest.fixture()
def fma(nt):
    from oracle import oracle_py as O
    nt.set_option("spgemm_fma", 1)     # (the library's default; the in-process tests start from unfused, conftest.py)
    O.set_fma(True)
    yield O
    O.set_fma(False)
    nt.set_option("spgemm_fma", 0)


def sample_columns(n, clusters=16, width=256, seed=3):
    """0-based columns: `clusters` runs of `width`, the first and the last columns among them"""
    rng = np.random.default_rng(seed)
    starts = np.concatenate(([0, n - width], rng.integers(width, n - 2 * width, clusters - 2)))
    cols = np.unique(np.concatenate([np.arange(s, s + width) for s in starts]))
    return cols


def srt(c, r, v):
    o = np.lexsort((r, c))
    return c[o], r[o], v[o]


def oracle_sampled_product(O, n, col, row, val, S, thr):
    """C(:, S) of A * A by the oracle, A given by 1-based triplets sorted by column; S ascending 0-based columns.
    Returns 1-based (col, row, val) in A's labels, sorted."""
    in_s = np.zeros(n + 1, dtype=bool)
    in_s[S + 1] = True
    sel = in_s[col]
    bc, br, bv = col[sel], row[sel], val[sel]
    K = np.unique(br) - 1                               # ascending: the k order of the chains is kept
    in_k = np.zeros(n + 1, dtype=bool)
    in_k[K + 1] = True
    sel = in_k[col]
    ac, ar, av = col[sel], row[sel], val[sel]
    kidx = np.zeros(n + 1, dtype=np.int32)
    kidx[K + 1] = np.arange(1, len(K) + 1, dtype=np.int32)
    sidx = np.zeros(n + 1, dtype=np.int32)
    sidx[S + 1] = np.arange(1, len(S) + 1, dtype=np.int32)
    Ao = O.Mat.from_triplets(n, len(K), kidx[ac], ar, av)            # A(:, K)
    Bo = O.Mat.from_triplets(len(K), len(S), sidx[bc], kidx[br], bv)   # A(K, S)
    oc, orow, ov = O.gemm(Ao, Bo, None, False, False, 1.0, None, thr).triplets()
    return srt((S[oc - 1] + 1).astype(np.int32), orow, ov)


def test_config3_default_arithmetic_vs_oracle_natural_order(nt, fma):
    """(a), natural order: the tile kernel's product, 4 096 sampled columns bit for bit; the digest of the whole product is kept
    for the several-rank runs below."""
    O = fma
    col, row, val = banded_triplets(N, H)
    A = nt.Matrix_ps.from_triplets(N, col, row, val)
    C = nt.Matrix_ps(N)
    C.Gemm(A, A, None, 1.0, 0.0, THR)
    st = nt.last_spgemm_stats()
    assert st["slab"] == 1 and nt.last_block_stats()["used"] == 0, st      # (run-like operand: the MFMA tile kernel)
    del A
    gc, gr, gv = C.triplets()
    del C
    from multirank_big_worker import digest
    _ONE_RANK_DIGEST["natural"] = digest(gc, gr, gv)
    S = sample_columns(N)
    in_s = np.zeros(N + 1, dtype=bool)
    in_s[S + 1] = True
    sel = in_s[gc]
    got = srt(gc[sel], gr[sel], gv[sel])
    del gc, gr, gv
    want = oracle_sampled_product(O, N, col, row, val, S, THR)
    assert len(got[2]) == len(want[2]) and len(want[2]) > 201 * len(S)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(got[2], want[2]), np.abs(got[2] - want[2]).max()


def test_config3_default_arithmetic_vs_oracle_relabelled(nt, fma):
    """(a), `bench.py --config 3 --permute 42`: the operand under a random symmetric relabelling (what the reference's load
    balancer hands its multiply, LoadBalancerModule.F90:14-52) takes the block path; statement (1) of DESIGN.md section 4:
    engine(A, A) == un-relabel(oracle_fma(relabel(A), relabel(A))) with the engine's own positions, bit for bit, on 4 096
    sampled columns of the relabelled index space."""
    O = fma
    col, row, val = permuted_banded_triplets(N, H, 42)
    A = nt.Matrix_ps.from_triplets(N, col, row, val)
    nt.set_option("slab_algebra", 0)
    try:
        C = nt.Matrix_ps(N)
        C.Gemm(A, A, None, 1.0, 0.0, THR)
    finally:
        nt.set_option("slab_algebra", 1)
    bs = nt.last_block_stats()
    assert bs["used"] == 1, (bs, nt.last_spgemm_stats())
    pos = nt.block_order(A)
    del A
    assert pos is not None
    order = np.argsort(pos, kind="stable").astype(np.int64)      # index at every rank
    rank = np.empty(N, dtype=np.int64)
    rank[order] = np.arange(N)
    # the sample: clusters of consecutive RANKS; the operand's entries that matter, moved to (rank[col], rank[row])
    S = sample_columns(N)
    rcol = (rank[col - 1] + 1).astype(np.int32)
    in_s = np.zeros(N + 1, dtype=bool)
    in_s[S + 1] = True
    sel = in_s[rcol]
    K = np.unique(rank[row[sel] - 1])
    in_k = np.zeros(N + 1, dtype=bool)
    in_k[K + 1] = True
    sel = in_k[rcol]
    sc, sr, sv = srt(rcol[sel], (rank[row[sel] - 1] + 1).astype(np.int32), val[sel])     # relabel(A)(:, K), sorted
    del rcol, col, row, val
    want = oracle_sampled_product(O, N, sc, sr, sv, S, THR)
    wc, wr, wv = srt((order[want[0] - 1] + 1).astype(np.int32), (order[want[1] - 1] + 1).astype(np.int32), want[2])
    gc, gr, gv = C.triplets()
    del C
    in_c = np.zeros(N + 1, dtype=bool)
    in_c[order[S] + 1] = True
    sel = in_c[gc]
    got = srt(gc[sel], gr[sel], gv[sel])
    del gc, gr, gv
    assert len(got[2]) == len(wv), (len(got[2]), len(wv))
    assert np.array_equal(got[0], wc) and np.array_equal(got[1], wr)
    assert np.array_equal(got[2], wv), np.abs(got[2] - wv).max()


def run_world(world, tmp_path):
    out = str(tmp_path / ("c3_%d" % world))
    name = "c%s" % uuid.uuid4().hex[:12]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", NTPOLY_AMD_COMM="shm:" + name,
                   NTPOLY_AMD_SHM_MB="512", NTPOLY_AMD_SPGEMM_FMA="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "config3_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=600)
            logs.append(o)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        try:
            os.unlink("/dev/shm/ntpoly_amd_" + name)
        except OSError:
            pass
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d of %d failed:\n%s" % (r, world, logs[r][-3000:])
    return [dict(np.load(out + ".%d.npz" % r)) for r in range(world)]


@pytest.mark.parametrize("world", [2, 4])
def test_config3_product_on_several_ranks_equals_one_rank(world, nt, fma, tmp_path):
    """(b): the N = 1 048 576 product in default arithmetic on 2 and 4 ranks -- column panels, halo of a bandwidth, the tile
    kernel on every rank -- has the entries and the bits of the one-rank product (additive digests)."""
    if "natural" not in _ONE_RANK_DIGEST:      # (run alone: the one-rank product here)
        col, row, val = banded_triplets(N, H)
        A = nt.Matrix_ps.from_triplets(N, col, row, val)
        del col, row, val
        C = nt.Matrix_ps(N)
        C.Gemm(A, A, None, 1.0, 0.0, THR)
        del A
        from multirank_big_worker import digest
        _ONE_RANK_DIGEST["natural"] = digest(*C.triplets())
        del C
    parts = run_world(world, tmp_path)
    with np.errstate(over="ignore"):
        got = np.sum(np.stack([p["AA"] for p in parts]), axis=0, dtype=np.uint64)
    assert np.array_equal(got, _ONE_RANK_DIGEST["natural"]), (got, _ONE_RANK_DIGEST["natural"])
    for p in parts:
        assert p["kernel"][0] == 1, p["kernel"]       # (slab = the tile kernel on this rank's panel)
