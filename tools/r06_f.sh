#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06f; mkdir -p $o
timeout 600 python3 -m pytest tests/test_gpu_fma.py -q -m gpu -x -k "grouped" > $o/tests.log 2>&1; echo "tests rc=$?"
tail -5 $o/tests.log
for mf in 1 0 1; do
timeout 600 python3 bench.py --random 42 --steps 5 --warmup 2 --blocks 3 --no-cpu-baseline --no-wrp-check --set ghash_mfma=$mf > $o/random_mf$mf.json 2> $o/random.err; echo "random mf=$mf rc=$?"
python3 -c "import json;d=json.load(open('$o/random_mf$mf.json'));print(d['value'],d['ms_per_step'],d['roofline']['ms_per_launch'],d['config']['blocks_ms'])"
done
timeout 600 python3 bench.py --random 42 --steps 3 --warmup 2 --blocks 1 --no-cpu-baseline --no-wrp-check --set spgemm_variant=518 > /dev/null 2> $o/stamps_mf1.err; grep "ghash stamps" $o/stamps_mf1.err | tail -2
timeout 300 python3 bench.py --config 3 --random 42 --steps 3 --warmup 1 --blocks 3 > $o/c3_random.json 2> $o/c3_random.err; echo "c3 random rc=$?"
python3 -c "import json;d=json.load(open('$o/c3_random.json'));print(d['value'],d['ms_per_step'],d['roofline']['ms_per_launch'])"
for rep in 1 2; do
for v in default vprio1 vprio3; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  NTPOLY_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/bench_$v.json 2> $o/bench.err; echo "bench $v rc=$?"
  python3 -c "import json;d=json.load(open('$o/bench_$v.json'));print('$v',d['value'],d['roofline']['ms_per_launch'],d['config']['blocks_ms'])"
done
done
