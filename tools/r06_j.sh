#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06j; mkdir -p $o
timeout 600 python3 tools/scratch_time_slab.py > $o/time_slab.log 2>&1; cat $o/time_slab.log | tail -12
timeout 300 tools/micro/l2_fragment_bw > $o/l2_fragment_bw.txt 2>&1; cat $o/l2_fragment_bw.txt
