#!/bin/bash
# complex paths with the tree's library (bounded: a hang costs two minutes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06ae; mkdir -p $o
timeout 240 python3 -m pytest tests/test_gpu_complex_tile.py -q -m gpu -x > $o/tests.log 2>&1; echo "tests rc=$?"; tail -2 $o/tests.log
export CPLX=1
for s in sign isq; do
  SOLVER=$s timeout 120 python3 tools/solver_iterations.py > $o/${s}_default.log 2>&1; echo "default $s rc=$? $(tail -1 $o/${s}_default.log | cut -c1-200)"
done
