#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06ax; mkdir -p $o
timeout 170 python3 -m pytest tests/test_gpu_multirank_big.py tests/test_gpu_scale.py -q -m gpu -x -k "big_multirank or relabelled_trs2" > $o/tests.log 2>&1; echo "tests rc=$?"; tail -3 $o/tests.log
