#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05suite; mkdir -p $o
timeout 2400 python3 -m pytest tests -q -m gpu --durations=60 -x > $o/suite.log 2>&1; echo "suite rc=$?"
tail -80 $o/suite.log
