#!/bin/bash
# usage: tools/pmc_bench_passes.sh <outdir-under-gpurun_out> "<bench.py arguments>" "<counters pass 1>" "<counters pass 2>" ...
# Each pass is a separate rocprofv3 --pmc run (kernel trace only) of bench.py; summarise with tools/pmc_summary.py.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; bargs=$2; shift 2
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d gpurun_out/$out/p$i -o run --output-format csv -- \
    python3 bench.py --gpus 1 --steps 4 --warmup 3 --no-cpu-baseline --no-wrp-check $bargs > gpurun_out/$out.p$i.log 2>&1
  echo "pass $i ($ctrs): rc=$?"
done
