#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06d; mkdir -p $o
timeout 1200 python3 -m pytest tests/test_gpu_multirank.py tests/test_gpu_panel_sessions.py tests/test_gpu_rccl_single.py tests/test_gpu_extras.py -q -m gpu -x --durations=10 > $o/tests.log 2>&1; echo "tests rc=$?"
tail -25 $o/tests.log
