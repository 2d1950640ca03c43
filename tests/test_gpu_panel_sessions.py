"""GPU: slab sessions on more than one rank (option panel_sessions; psmatrix.cpp panel_slab_multiply).  The loops of TRS4
(DensityMatrixSolversModule.F90:586-638), the sign function (SignSolversModule.F90), the inverse square root
(SquareRootSolversModule.F90:342-531) and the inverse (InverseSolversModule.F90:29-149) keep their matrices in the tile
kernel's operand form as column panels; a product exchanges the runs of its left operand's halo.  Ranks are processes
sharing the box's GPU over the shared-memory test transport (see test_gpu_multirank.py).  Against the one-rank solve: the
same iteration counts, the same patterns, values to 1e-10 (a column's products have the same bits whoever owns it; the
reductions that steer the loops are summed in another order)."""
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAGS = ("trs4_K", "sign", "isq", "inv")
SMALL = ("horner", "paterson", "exp", "log", "sine", "root3", "sqrt", "hpcp", "pm")   # the other families, small operand


def run_world(world, tmp_path, extra=None):
    out = str(tmp_path / ("ps%d" % world))
    name = "p%s" % uuid.uuid4().hex[:12]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", NTPOLY_AMD_COMM="shm:" + name,
                   NTPOLY_AMD_SHM_MB="64", NTPOLY_AMD_SPGEMM_FMA="1")
        env.update(extra or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "panel_session_worker.py"), out], env=env,
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
def reference(tmp_path_factory):
    return run_world(1, tmp_path_factory.mktemp("psref"))[0]


@pytest.mark.parametrize("world", [2, 4])
def test_panel_sessions_equal_single_rank(world, reference, tmp_path):
    parts = run_world(world, tmp_path)
    for tag in TAGS:
        got = cat(parts, tag)
        want = tuple(reference[tag + s] for s in ("_col", "_row", "_val"))
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), tag
        assert np.allclose(got[2], want[2], rtol=0, atol=1e-10), (tag, float(np.max(np.abs(got[2] - want[2]))))
    for r in range(world):
        for loop in ("trs4", "sign", "isq", "inv"):
            slab, declined, syncs = parts[r][loop + "_panel"]
            iters = int(parts[r][loop + "_iters"][0])
            print("world", world, "rank", r, loop, "iterations", iters, "panel products", slab, "declined", declined, "host syncs inside", syncs, "slab ops",
                  parts[r][loop + "_slab"], "exchanges", parts[r][loop + "_exchanges"])
            assert iters == int(reference[loop + "_iters"][0]), (loop, r)
            # every product of the loop ran on the tile kernel with its operands in slab form, on every rank
            assert slab >= iters and declined <= 1, (loop, r, slab, declined, iters)
            # ... at TWO host round trips each, as on one rank (MEASURED in sync_stream): the exchange layout with the product's
            # plan, then the entry count with "every rank's kernel took its panel"; operands entering slab form add a few
            assert syncs <= 2 * slab + 8, (loop, r, syncs, slab)
        assert np.allclose(parts[r]["trs4_scal"], reference["trs4_scal"], rtol=1e-11, atol=1e-9)
        assert np.allclose(parts[r]["trs4_log"], reference["trs4_log"], rtol=1e-11, atol=1e-9)
    # TRS4 on a relabelled band: the band scope around a session of column panels (contract of the band scope: energies 1e-8,
    # entry counts 1e-4 against the one-rank solve on the caller's labels)
    have_perm = "trs4p_sums" in parts[0] and "trs4p_sums" in reference    # (opt-in: NTPOLY_AMD_PANEL_PERM=1, see the worker)
    sums = np.sum(np.stack([p["trs4p_sums"] for p in parts]), axis=0) if have_perm else None
    for r in range(world if have_perm else 0):
        assert int(parts[r]["trs4p_scope"][0]) == 1, (r, parts[r]["trs4p_scope"])
        slab, declined, syncs = parts[r]["trs4p_panel"]
        assert slab >= 14 and declined <= 1, (r, slab, declined)
        assert np.allclose(parts[r]["trs4p_log"], reference["trs4p_log"], rtol=1e-8, atol=1e-7), r
    if have_perm:
        assert abs(sums[0] - reference["trs4p_sums"][0]) <= 1e-4 * reference["trs4p_sums"][0] + 8, (sums, reference["trs4p_sums"])
        assert np.allclose(sums[1:], reference["trs4p_sums"][1:], rtol=1e-8, atol=1e-6), (sums, reference["trs4p_sums"])
    # polynomials, functions, the other density solvers: every collective of their loops is entered by every rank (the run
    # ends), same iteration counts, results to 1e-8 (HPCP divides by differences of traces: the reduction order shows at 3e-9)
    for tag in SMALL:
        got = cat(parts, "m_" + tag)
        want = tuple(reference["m_" + tag + s] for s in ("_col", "_row", "_val"))
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), tag
        assert np.allclose(got[2], want[2], rtol=0, atol=1e-8), (tag, float(np.max(np.abs(got[2] - want[2]))))
        for r in range(world):
            assert int(parts[r]["m_" + tag + "_iters"][0]) == int(reference["m_" + tag + "_iters"][0]), (tag, r)
        print("world", world, tag, "panel products (slab, declined, syncs) per rank:", [parts[r]["m_" + tag + "_panel"].tolist() for r in range(world)])


def test_panel_sessions_off_is_the_old_path(reference, tmp_path):
    """option panel_sessions = 0: compressed columns across ranks, same results"""
    parts = run_world(2, tmp_path, {"NTPOLY_AMD_PANEL_SESSIONS": "0"})
    for tag in TAGS + tuple("m_" + t for t in SMALL):
        got = cat(parts, tag)
        want = tuple(reference[tag + s] for s in ("_col", "_row", "_val"))
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), tag
        assert np.allclose(got[2], want[2], rtol=0, atol=1e-8), tag
    assert all(int(parts[r]["trs4_panel"][0]) == 0 for r in range(2))
