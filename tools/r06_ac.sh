#!/bin/bash
# tile kernel: tiles taken dynamically (default) against the fixed deal (libntpoly_amd_static.so)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06ac; mkdir -p $o
timeout 1200 python3 -m pytest tests/test_gpu_fma.py tests/test_gpu_slab_algebra.py tests/test_gpu_panel_sessions.py -q -m gpu -x > $o/tests.log 2>&1; echo "tests rc=$?"; tail -3 $o/tests.log
for rep in 1 2 3; do
for v in default static; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  NTPOLY_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/bench_$v.json 2> $o/bench.err; echo "bench $v rc=$?"
  python3 -c "import json;d=json.load(open('$o/bench_$v.json'));print('$v',d['value'],d['roofline']['ms_per_launch'],d['config']['blocks_ms'],repr(d['config']['energy_end']))"
done
done
for v in default static; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  NTPOLY_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --permute 42 --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/perm_$v.json 2> $o/perm.err; echo "perm $v rc=$?"
  python3 -c "import json;d=json.load(open('$o/perm_$v.json'));print('$v',d['value'],d['roofline']['ms_per_launch'])"
  NTPOLY_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --config 3 --steps 5 --warmup 2 --blocks 3 --no-cpu-baseline > $o/c3_$v.json 2> $o/c3.err; echo "c3 $v rc=$?"
  python3 -c "import json;d=json.load(open('$o/c3_$v.json'));print('$v',d['value'],d['ms_per_step'],d['roofline']['ms_per_launch'])"
done
