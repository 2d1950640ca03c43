#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06o; mkdir -p $o
for rep in 1 2; do
for v in default vwpe3 vpf9 vpf12; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  NTPOLY_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/bench_$v.json 2> $o/bench.err; echo "bench $v rc=$?"
  python3 -c "import json;d=json.load(open('$o/bench_$v.json'));print('$v',d['value'],d['roofline']['ms_per_launch'],d['config']['blocks_ms'],d['config']['energy_end'])"
done
done
