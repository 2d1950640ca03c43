#!/bin/bash
# kernel statistics of the real solver loops on the headline operand under rocprofv3: tools/prof_real.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
for s in trs4 isq; do
  export SOLVER=$s
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/$tag/stats_$s -o run -- python3 tools/solver_iterations.py > gpurun_out/$tag/$s.log 2>&1
  python3 tools/prof_summary.py gpurun_out/$tag/stats_$s/run_results.db > gpurun_out/$tag/${s}_kernel_stats.csv
  rm -rf gpurun_out/$tag/stats_$s
  tail -1 gpurun_out/$tag/$s.log
  head -24 gpurun_out/$tag/${s}_kernel_stats.csv | cut -c1-120
done
