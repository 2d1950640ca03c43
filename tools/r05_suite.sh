#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05suite; mkdir -p $o
timeout 2600 python3 -m pytest tests -q -m gpu --durations=70 > $o/suite.log 2>&1; echo "suite rc=$?"
grep -v "^tests/.*PASSED" $o/suite.log | tail -110
