#!/bin/bash
# kernel statistics of the complex solver loops (configs[4]) under rocprofv3: tools/prof_complex.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
export CPLX=1
for s in sign isq; do
  export SOLVER=$s
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/$tag/stats_$s -o run -- python3 tools/solver_iterations.py > gpurun_out/$tag/$s.log 2>&1
  python3 tools/prof_summary.py gpurun_out/$tag/stats_$s/run_results.db > gpurun_out/$tag/${s}_kernel_stats.csv
  rm -rf gpurun_out/$tag/stats_$s
  tail -1 gpurun_out/$tag/$s.log
  head -14 gpurun_out/$tag/${s}_kernel_stats.csv | cut -c1-110
done
