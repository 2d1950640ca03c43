#!/bin/bash
# label-aware variants with the pair path (tree) against libntpoly_amd_wide3.so (pitch 16, element-wise; old rule for eight waves)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06ak; mkdir -p $o
timeout 1500 python3 -m pytest tests/test_gpu_fma.py tests/test_gpu_scale.py tests/test_gpu_multirank_big.py tests/test_gpu_config3.py -q -m gpu -x > $o/tests.log 2>&1; echo "tests rc=$?"; tail -1 $o/tests.log
for rep in 1 2; do
for v in default wide3; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  export NTPOLY_AMD_LIB=$PWD/$lib
  timeout 200 python3 bench.py --permute 42 --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/perm.json 2> $o/err; python3 -c "import json;d=json.load(open('$o/perm.json'));print('$v perm',d['value'],d['roofline']['ms_per_launch'],repr(d['config']['energy_end']))"
done
done
