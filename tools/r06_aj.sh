#!/bin/bash
# the rule for eight waves per workgroup: fewer than four workgroups of four fit (tree) against fewer than three (libntpoly_amd_wide3.so)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06aj; mkdir -p $o
for rep in 1 2; do
for v in default wide3; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  export NTPOLY_AMD_LIB=$PWD/$lib
  timeout 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/bench.json 2> $o/err; python3 -c "import json;d=json.load(open('$o/bench.json'));print('$v headline',d['value'],d['roofline']['ms_per_launch'])"
  for s in trs4 sign isq; do
    SOLVER=$s timeout 150 python3 tools/solver_iterations.py > $o/${s}.log 2>&1; echo "$v $s $(tail -1 $o/${s}.log | cut -c1-40)"
  done
done
done
