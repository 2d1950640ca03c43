#!/bin/bash
# after the plain launch of the band search's kernel: the tests that search for a band, then the headline's profiles on the final sources
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06aw; mkdir -p $o
timeout 120 python3 -m pytest tests/test_gpu_fma.py tests/test_gpu_multirank_big.py -q -m gpu -x -k "relabel or perm or other_real or label" > $o/tests.log 2>&1; echo "tests rc=$?"; tail -2 $o/tests.log
timeout 200 bash tools/collect_profiles.sh r06_v6; echo "collect rc=$?"
