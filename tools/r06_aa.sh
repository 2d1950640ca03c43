#!/bin/bash
# relabelled operand: which kernel is slower in the tree's library than in the commit before's
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06aa; mkdir -p $o
for v in default base; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  NTPOLY_AMD_LIB=$PWD/$lib timeout 300 rocprofv3 --kernel-trace --stats -d $o/prof_$v -o run -- python3 bench.py --gpus 1 --permute 42 --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/perm_$v.log 2>&1; echo "perm $v rc=$?"
  timeout 120 python3 tools/prof_summary.py $o/prof_$v/run_results.db > $o/kernel_stats_$v.csv; head -14 $o/kernel_stats_$v.csv | cut -c1-160
  rm -rf $o/prof_$v
done
