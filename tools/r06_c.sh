#!/bin/bash
# round 6: A/B of the tile kernel on ONE box -- the round-5 library (base), this round's without the full-tile epilogue, this round's
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06c; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_fma.py tests/test_gpu_options.py -q -m gpu -x > $o/tests.log 2>&1; echo "tests rc=$?"
tail -4 $o/tests.log
timeout 900 python3 -m pytest tests/test_gpu_scale.py -q -m gpu -x -k "fused_steps or headline_config2_vs or trs2" > $o/tests2.log 2>&1; echo "tests2 rc=$?"
tail -4 $o/tests2.log
for rep in 1 2; do
for v in base vnofull default; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  NTPOLY_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/bench_$v.json 2> $o/bench.err; echo "bench $v rc=$?"
  python3 -c "import json;d=json.load(open('$o/bench_$v.json'));print('$v',d['value'],d['roofline']['ms_per_launch'],d['config']['blocks_ms'])"
done
done
