"""GPU: the multi-rank code paths of the engine (halo exchange of the distributed multiply, panel gathers of the
transposes, all-reduced scalars, a whole TRS2 solve) with 2, 3 and 4 ranks.  The GPU box has ONE MI355X, and RCCL
needs one GPU per rank, so here the ranks are processes sharing the GPU and exchanging through the engine's
shared-memory TEST transport (NTPOLY_AMD_COMM=shm:<name>, csrc/comm.cpp) -- every line of the distributed
algorithm except the RCCL calls themselves runs as in production.  Results must match the one-rank run: products
bit for bit (a column's arithmetic does not depend on who owns it), reductions to 1e-12."""
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_world(world, tmp_path, arith="unfused", grid=None):
    out = str(tmp_path / ("w%d%s%s" % (world, arith, "" if grid is None else grid.replace(",", "x"))))
    name = "t%s" % uuid.uuid4().hex[:12]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", NTPOLY_AMD_COMM="shm:" + name,
                   NTPOLY_AMD_SHM_MB="8", NTPOLY_AMD_SPGEMM_FMA="1" if arith == "fma" else "0")
        if grid is not None:
            env["NTPOLY_AMD_TEST_GRID"] = grid
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), out], env=env,
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


def cat(parts, tag):
    return tuple(np.concatenate([p[tag + s] for p in parts]) for s in ("_col", "_row", "_val"))


@pytest.fixture(scope="module")
def references(tmp_path_factory):
    """the one-rank run in either arithmetic mode (made on first use)"""
    cache = {}

    def get(arith):
        if arith not in cache:
            cache[arith] = run_world(1, tmp_path_factory.mktemp("ref_" + arith), arith)[0]
        return cache[arith]
    return get


# (FMA arithmetic, option spgemm_fma = 1: the panel steps run on the MFMA tile kernel -- one row per lane, the halo
# runs sit packed in the receive buffer -- and must reproduce the one-rank FMA run the same way)
@pytest.mark.parametrize("world,arith", [(2, "unfused"), (3, "unfused"), (4, "unfused"), (2, "fma"), (4, "fma")])
def test_multirank_equals_single_rank(world, arith, references, tmp_path):
    reference = references(arith)
    parts = run_world(world, tmp_path, arith)
    # panels tile the columns in rank order
    assert parts[0]["c0"] == 0 and all(parts[r]["c1"] == parts[r + 1]["c0"] for r in range(world - 1))
    for tag in ("AB", "ABT", "GG", "K"):
        got = cat(parts, tag)
        want = tuple(reference[tag + s] for s in ("_col", "_row", "_val"))
        if tag == "K":   # solver result: same iterations, entries equal up to reduction-order effects on sigma
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
            assert np.allclose(got[2], want[2], rtol=0, atol=1e-10)
        else:
            assert all(np.array_equal(g, w) for g, w in zip(got, want)), tag
    # GatherMatrixToProcess: every rank holds the whole product; the "to one process" variant leaves the others empty-handed
    for r in range(world):
        assert np.allclose(parts[r]["gather_all"], reference["gather_all"], rtol=1e-13, atol=0), (r, parts[r]["gather_all"])
        want_one = float(reference["gather_all"][0]) if r == min(1, world - 1) else -1.0
        assert float(parts[r]["gather_one"][0]) == want_one, (r, parts[r]["gather_one"])
    # the TRS2 steps ran inside the SpGEMM kernel on every rank (fused epilogue, panels in slab form, halo exchanged as
    # dense column runs: psmatrix.cpp dist_fused_step), with one host synchronisation per exchange
    for r in range(world):
        sq, up, rep = parts[r]["trs2_fused"]
        iters = int(parts[r]["trs2_iters"])
        assert rep == 0 and iters - 1 <= sq + up <= iters, (world, r, sq, up, rep, iters)
        # host synchronisations MEASURED (counter in sync_stream): a panel step costs ONE -- the layout of its exchange and its
        # plan were prepared by the step before it, from that step's result, and came back on that step's read-back of its
        # totals (psmatrix.cpp PanelExchange, option exchange_ahead); only the first exchanges of the solve wait for their own
        # layout.  Over the whole solve: one per panel step plus what the solver does around its loop (bounds, the first step
        # from compressed columns, the chemical potential)
        ex, ex_syncs, syncs = parts[r]["trs2_exchanges"]
        print("world", world, "rank", r, "exchanges", ex, "syncs inside", ex_syncs, "syncs of the solve", syncs)
        assert ex >= iters and ex_syncs <= 3 and ex <= syncs <= ex + 35, (world, r, ex, ex_syncs, syncs)
    # ... and counted the same intermediate products and product entries as the one-rank solve
    assert sum(int(parts[r]["trs2_products"]) for r in range(world)) == int(reference["trs2_products"])
    assert sum(int(parts[r]["trs2_nnz_c"]) for r in range(world)) == int(reference["trs2_nnz_c"])
    # overlapped halo exchange (interior / boundary split): bit-identical to the plain product and to one rank
    for tag, base in (("AB_ov", "AB"), ("AA_ov", "AA")):
        got = cat(parts, tag)
        assert all(np.array_equal(g, w) for g, w in zip(got, cat(parts, base))), tag
        assert all(np.array_equal(g, reference[base + s]) for g, s in zip(got, ("_col", "_row", "_val"))), tag
    # gather-based families: every rank computes from the same gathered matrix -> identical panels
    for tag in ("eigW", "eigV", "disq", "chol", "pchol", "foe", "snap"):
        got = cat(parts, tag)
        want = tuple(reference[tag + s] for s in ("_col", "_row", "_val"))
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), tag
        assert np.allclose(got[2], want[2], rtol=0, atol=1e-11), tag
    got, want = cat(parts, "cg"), tuple(reference["cg" + s] for s in ("_col", "_row", "_val"))
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.allclose(got[2], want[2], rtol=0, atol=1e-9)
    for r in range(world):
        assert parts[r]["foe_energy"] == pytest.approx(float(reference["foe_energy"]), rel=1e-12)
        assert parts[r]["foe_mu"] == pytest.approx(float(reference["foe_mu"]), rel=1e-12)
    # one distributed multiply = one halo exchange with ONE host synchronisation inside the exchange (both MEASURED: the
    # counter lives in sync_stream), and a bounded number for the whole call (exchange, plan, nnz read-back)
    for r in range(world):
        assert int(parts[r]["exchanges"]) == 1 and int(parts[r]["exchange_host_syncs"]) == 1, (r, parts[r]["exchanges"],
                                                                                              parts[r]["exchange_host_syncs"])
        assert 1 <= int(parts[r]["gemm_host_syncs"]) <= 6, (r, parts[r]["gemm_host_syncs"])
    for r in range(world):
        for s in ("AB_trace", "AB_norm", "AB_dot", "trs2_energy", "trs2_mu"):
            assert parts[r][s] == pytest.approx(float(reference[s]), rel=1e-12, abs=1e-12), (s, r)
        assert parts[r]["trs2_iters"] == reference["trs2_iters"]
        assert np.allclose(parts[r]["trs2_log"], reference["trs2_log"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("world,grid", [(2, None), (3, None), (4, None), (4, "1,2,2"), (4, "4,1,1")])
def test_comm_split_matrix_on_several_ranks(world, grid, references, tmp_path):
    """CommSplitMatrix (PSMatrixModule.F90:1489-1541) on more than one process: the grid is split as SplitProcessGrid does
    (ProcessGridModule.F90:430-515: along the slices where there are several, else along the longer of rows / columns; the
    processes of a half keep the order of their old ranks), every half holds a copy of the WHOLE matrix on a sub-communicator,
    and what is done with the copy -- a product, reductions, a TRS2 solve of its own length -- happens inside the half and
    equals the one-rank results."""
    reference = references("unfused")
    parts = run_world(world, tmp_path, "unfused", grid)
    rows, cols, slices = (1, world, 1) if grid is None else tuple(int(x) for x in grid.split(","))
    # what SplitProcessGrid decides for every old rank (rank -> slice, row, column as ProcessGridModule.F90:180-183)
    want_color, want_slice = [], slices > 1
    for r in range(world):
        sl, rem = divmod(r, rows * cols)
        row, col = divmod(rem, cols)
        if slices > 1:
            want_color.append(0 if sl < slices // 2 else 1)
        elif rows > cols:
            want_color.append(0 if row < rows // 2 else 1)
        else:
            want_color.append(0 if col < cols // 2 else 1)
    for color in (0, 1):
        members = [r for r in range(world) if want_color[r] == color]
        assert members, (world, grid, color)
        for i, r in enumerate(members):
            c, ss, sub_rank, sub_size, is_sub = (int(x) for x in parts[r]["split_info"])
            assert (c, bool(ss), sub_rank, sub_size, bool(is_sub)) == (color, want_slice, i, len(members), True), (r, parts[r]["split_info"])
        half = [parts[r] for r in members]
        # the copy: the whole matrix, panels in the order of the half's ranks
        got = cat(half, "split")
        one = tuple(reference["split" + s] for s in ("_col", "_row", "_val"))
        assert all(np.array_equal(g, w) for g, w in zip(got, one)), ("copy", color)
        # the product on the half: bit for bit the one-rank product
        got = cat(half, "split_SS")
        assert all(np.array_equal(g, reference["split_SS" + s]) for g, s in zip(got, ("_col", "_row", "_val"))), ("product", color)
        for p in half:
            assert np.allclose(p["split_scal"], reference["split_scal"], rtol=1e-12, atol=1e-12)
            assert np.allclose(p["after_split"], reference["after_split"], rtol=1e-12, atol=1e-12)
        # the solve on the half (4 + colour iterations): the one-rank solve of the same length
        if color == 0:
            for p in half:
                assert np.allclose(p["split_trs2"], reference["split_trs2"], rtol=1e-11, atol=1e-11), (p["split_trs2"], reference["split_trs2"])
            got = cat(half, "split_K")
            want = tuple(reference["split_K" + s] for s in ("_col", "_row", "_val"))
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
            assert np.allclose(got[2], want[2], rtol=0, atol=1e-10)
        else:
            e = [float(p["split_trs2"][0]) for p in half]
            assert max(e) - min(e) <= 1e-11 * abs(e[0])      # (one value inside the half; five iterations, no one-rank twin)
