#!/bin/bash
# usage: tools/pmc_passes.sh <outdir-under-gpurun_out> <variant> "<counters pass 1>" "<counters pass 2>" ...
# Each pass is a separate rocprofv3 --pmc run (kernel trace only) of tools/bench_spgemm.py on one SpGEMM variant.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; var=$2; shift 2
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d gpurun_out/$out/p$i -o run --output-format csv -- \
    python3 tools/bench_spgemm.py --iters 6 --reps 2 --variants $var $EXTRA > gpurun_out/$out.p$i.log 2>&1
  echo "pass $i ($ctrs): rc=$?"
done
