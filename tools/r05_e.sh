#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05e; mkdir -p $o
timeout 1500 python3 -m pytest tests/test_gpu_defaults.py tests/test_gpu_multirank_big.py tests/test_gpu_block.py tests/test_gpu_thin.py tests/test_gpu_column_fused.py -q --durations=15 > $o/sel.log 2>&1; echo "rc=$?"
tail -60 $o/sel.log
