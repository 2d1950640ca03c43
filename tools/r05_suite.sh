#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05suite; mkdir -p $o
timeout 900 python3 tools/rank_share.py > gpurun_out/r05_rank_share.json 2> gpurun_out/r05_rank_share.err
timeout 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05_v4_bench_with_traffic.json 2> /dev/null
timeout 2600 python3 -m pytest tests -q -m gpu --durations=25 > $o/suite.log 2>&1; echo "suite rc=$?"
tail -40 $o/suite.log
