#!/bin/bash
# complex tile kernel: dynamic tiles (counted loop) + records first (libntpoly_amd_cnew.so) against the tree's library; eight waves forced as well
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06ao; mkdir -p $o
NTPOLY_AMD_LIB=$PWD/ntpoly_amd/libntpoly_amd_cnew.so timeout 200 python3 -m pytest tests/test_gpu_complex_tile.py -q -m gpu -x > $o/tests.log 2>&1; echo "cnew tests rc=$? $(tail -1 $o/tests.log)"
export CPLX=1
for rep in 1 2; do
for v in cnew default; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  for tw in 0; do
  for s in sign isq; do
    NTPOLY_AMD_TILE_WAVES=$tw NTPOLY_AMD_LIB=$PWD/$lib SOLVER=$s timeout 100 python3 tools/solver_iterations.py > $o/${s}_$v.log 2>&1; echo "$v tw$tw $s rc=$? $(tail -1 $o/${s}_$v.log | cut -c1-32)"
  done
  done
done
done
