"""GPU: run the distributed code paths (RCCL communicator, panel all-gather with size exchange +
grouped broadcasts + offset fix-up, scalar all-reduces, gathered transpose / permutation / triplet
fill) on ONE GPU by forcing a 1-rank RCCL communicator (NTPOLY_AMD_FORCE_RCCL=1).  The results must
still match the reference's golden vectors bit for bit.  Multi-GPU runs are the driver's; this is
the closest exercise of that code a single-GPU box allows."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forced_rccl_paths_match_golden():
    env = dict(os.environ, NTPOLY_AMD_FORCE_RCCL="1")
    sel = "test_ps_gemm_golden and -1--1 or test_ps_scalars_golden or test_solvers_golden or test_premade_fixture or " \
          "test_load_balanced_solver_matches or test_ps_increment_golden or test_trs2_fused_steps_match"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-x", "-q",
                        "-m", "gpu", "-k", sel], env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "passed" in r.stdout


def test_forced_rccl_overlapped_halo_matches_golden():
    """the same with every distributed multiply split into interior / boundary columns and its send / recv group
    enqueued on the communication stream (NTPOLY_AMD_HALO_OVERLAP=3): event ordering between the two streams and the
    RCCL group on a second stream run for real, results still bit-identical to the reference's."""
    env = dict(os.environ, NTPOLY_AMD_FORCE_RCCL="1", NTPOLY_AMD_HALO_OVERLAP="3")
    sel = "test_ps_gemm_golden and -1--1 or test_solvers_golden or test_premade_fixture"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-x", "-q",
                        "-m", "gpu", "-k", sel], env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "passed" in r.stdout
