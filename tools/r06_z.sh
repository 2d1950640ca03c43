#!/bin/bash
# A/B: the library of the commit before (libntpoly_amd_base.so) against the tree's (pairs of rows for the multiplier tile)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06z; mkdir -p $o
for rep in 1 2 3; do
for v in default base; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  NTPOLY_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/bench_$v.json 2> $o/bench.err; echo "bench $v rc=$?"
  python3 -c "import json;d=json.load(open('$o/bench_$v.json'));print('$v',d['value'],d['roofline']['ms_per_launch'],d['config']['blocks_ms'],d['config']['energy_end'],d.get('spgemm_products_per_s'))"
done
done
for v in default base; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  NTPOLY_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --config 3 --steps 5 --warmup 2 --blocks 3 --no-cpu-baseline > $o/c3_$v.json 2> $o/c3.err; echo "c3 $v rc=$?"
  python3 -c "import json;d=json.load(open('$o/c3_$v.json'));print('$v',d['value'],d['ms_per_step'],d['roofline']['ms_per_launch'])"
  NTPOLY_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --permute 42 --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/perm_$v.json 2> $o/perm.err; echo "perm $v rc=$?"
  python3 -c "import json;d=json.load(open('$o/perm_$v.json'));print('$v',d['value'],d['roofline']['ms_per_launch'])"
done
