#!/bin/bash
# complex tile kernel experiments (bounded: a hang costs two minutes each): records first (crec), dynamic tiles (cdyn)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06af; mkdir -p $o
export CPLX=1
for v in crec cdyn default; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  NTPOLY_AMD_LIB=$PWD/$lib timeout 150 python3 -m pytest tests/test_gpu_complex_tile.py -q -m gpu -x > $o/tests_$v.log 2>&1; echo "$v tests rc=$? $(tail -1 $o/tests_$v.log)"
  for s in sign isq; do
    NTPOLY_AMD_LIB=$PWD/$lib SOLVER=$s timeout 100 python3 tools/solver_iterations.py > $o/${s}_$v.log 2>&1; echo "$v $s rc=$? $(tail -1 $o/${s}_$v.log | cut -c1-60)"
  done
done
