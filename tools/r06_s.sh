#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06s; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_fma.py -q -m gpu -x > $o/tests.log 2>&1; echo "tests rc=$?"; tail -4 $o/tests.log
for rep in 1 2; do
for v in pre1 pre0 base; do
  lib=ntpoly_amd/libntpoly_amd.so; extra="--set tile_prefetch=1"
  [ $v = pre0 ] && extra="--set tile_prefetch=0"
  [ $v = base ] && lib=ntpoly_amd/libntpoly_amd_base.so && extra=""
  NTPOLY_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check $extra > $o/bench_$v.json 2> $o/bench.err; echo "bench $v rc=$?"
  python3 -c "import json;d=json.load(open('$o/bench_$v.json'));print('$v',d['value'],d['roofline']['ms_per_launch'],d['config']['blocks_ms'],d['config']['energy_end'])"
done
done
