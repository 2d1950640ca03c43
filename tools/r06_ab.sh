#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06ab; mkdir -p $o
timeout 1200 python3 -m pytest tests/test_gpu_fma.py tests/test_gpu_scale.py tests/test_gpu_multirank_big.py -q -m gpu -x > $o/tests.log 2>&1; echo "tests rc=$?"; tail -3 $o/tests.log
for rep in 1 2; do
for v in base default; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  NTPOLY_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --permute 42 --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/perm_$v.json 2> $o/perm.err; echo "perm $v rc=$?"
  python3 -c "import json;d=json.load(open('$o/perm_$v.json'));print('$v',d['value'],d['ms_per_step'],d['roofline']['ms_per_launch'],d['config']['blocks_ms'])"
  NTPOLY_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/bench_$v.json 2> $o/bench.err; echo "bench $v rc=$?"
  python3 -c "import json;d=json.load(open('$o/bench_$v.json'));print('$v',d['value'],d['roofline']['ms_per_launch'],d['config']['energy_end'])"
done
done
