#!/bin/bash
# usage: tools/pmc_passes.sh <outdir-under-gpurun_out> "<python script and its arguments>" "<counters pass 1>" "<counters pass 2>" ...
# Each pass is a separate rocprofv3 --pmc run (kernel trace only); summarise with tools/pmc_summary.py.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; prog=$2; shift 2
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d gpurun_out/$out/p$i -o run --output-format csv -- \
    python3 $prog > gpurun_out/$out.p$i.log 2>&1
  echo "pass $i ($ctrs): rc=$?"
done
