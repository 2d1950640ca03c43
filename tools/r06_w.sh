#!/bin/bash
# tile kernel: run records requested before the multiplier tile's values (recfirst); ablation: no loads for the multiplier tile
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06w; mkdir -p $o
NTPOLY_AMD_LIB=$PWD/ntpoly_amd/libntpoly_amd_recfirst.so timeout 900 python3 -m pytest tests/test_gpu_fma.py -q -m gpu -x > $o/tests.log 2>&1; echo "tests rc=$?"; tail -3 $o/tests.log
for rep in 1 2 3; do
for v in default recfirst; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  NTPOLY_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/bench_$v.json 2> $o/bench.err; echo "bench $v rc=$?"
  python3 -c "import json;d=json.load(open('$o/bench_$v.json'));print('$v',d['value'],d['roofline']['ms_per_launch'],d['config']['blocks_ms'],d['config']['energy_end'])"
done
done
for rep in 1 2; do
for v in default nobload recfirst; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  NTPOLY_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --config 3 --n 262144 --halfband 157 --steps 5 --warmup 2 --blocks 3 --no-cpu-baseline > $o/c3_$v.json 2> $o/c3.err; echo "c3 $v rc=$?"
  python3 -c "import json;d=json.load(open('$o/c3_$v.json'));print('$v',d['value'],d['ms_per_step'],d['roofline']['ms_per_launch'])"
done
done
