import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    # ask the engine's own HIP runtime (importing torch would load a second, bundled HIP runtime
    # into the process)
    try:
        import ntpoly_amd
        return int(ntpoly_amd.lib.ntpoly_amd_device_count()) > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True, scope="session")
def _unfused_baseline_for_in_process_tests():
    """The library's default arithmetic is the FMA chain (option spgemm_fma = 1) -- what a drop-in
    caller gets, what bench.py times and what the reference's own example programs run with in
    tests/test_gpu_extras.py.  The in-process GPU tests state their arithmetic themselves: they start from the
    unfused mode (bit-identical to the reference's default build, which most goldens come from) and switch to FMA
    through their `arith` fixtures / explicit set_option calls, restoring 0 afterwards."""
    try:
        import ntpoly_amd as nt
        nt.set_option("spgemm_fma", 0)
    except Exception:
        pass
    yield
