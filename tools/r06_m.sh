#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06m; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_slab_algebra.py tests/test_gpu_parity.py -q -m gpu -x -k "second_order or scalars or gershgorin or solvers_golden or extras" > $o/tests.log 2>&1; echo "tests rc=$?"; tail -5 $o/tests.log
for rep in 1 2; do
for v in default vhalf; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  NTPOLY_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --config 3 --n 262144 --halfband 157 --steps 10 --warmup 3 --blocks 3 > $o/c3_$v.json 2> $o/c3.err; echo "c3 $v rc=$?"
  python3 -c "import json;d=json.load(open('$o/c3_$v.json'));print('$v',d['ms_per_step'],d['roofline']['ms_per_launch'],d['config']['blocks_ms'])"
done
done
