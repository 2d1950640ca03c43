#!/bin/bash
# tile kernel: entry counts through LDS (one request per 64 rows and block), the epilogue's column constants in two requests: against the commit before
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06al; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_fma.py tests/test_gpu_slab_algebra.py tests/test_gpu_panel_sessions.py -q -m gpu -x > $o/tests.log 2>&1; echo "tests rc=$?"; tail -1 $o/tests.log
for rep in 1 2 3; do
for v in default base; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  export NTPOLY_AMD_LIB=$PWD/$lib
  timeout 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/bench.json 2> $o/err; python3 -c "import json;d=json.load(open('$o/bench.json'));print('$v headline',d['value'],d['roofline']['ms_per_launch'],repr(d['config']['energy_end']),d.get('spgemm_products_per_s'))"
done
done
for v in default base; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  export NTPOLY_AMD_LIB=$PWD/$lib
  timeout 200 python3 bench.py --permute 42 --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/perm.json 2> $o/err; python3 -c "import json;d=json.load(open('$o/perm.json'));print('$v perm',d['value'],d['roofline']['ms_per_launch'])"
done
