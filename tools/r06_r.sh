#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06r; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_fma.py tests/test_gpu_multirank.py -q -m gpu -x > $o/tests.log 2>&1; echo "tests rc=$?"; tail -4 $o/tests.log
timeout 900 python3 -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -q -m gpu -x -k "full_size_properties_config2 or fused_steps or trs2_fma" > $o/tests2.log 2>&1; echo "tests2 rc=$?"; tail -4 $o/tests2.log
for rep in 1 2; do
for v in default base; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  NTPOLY_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/bench_$v.json 2> $o/bench.err; echo "bench $v rc=$?"
  python3 -c "import json;d=json.load(open('$o/bench_$v.json'));print('$v',d['value'],d['roofline']['ms_per_launch'],d['config']['blocks_ms'],d['config']['energy_end'],d['spgemm_products_per_s'])"
done
done
