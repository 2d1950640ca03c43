#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06g; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_multirank_big.py tests/test_gpu_multirank.py -q -m gpu --durations=6 > $o/tests.log 2>&1; echo "tests rc=$?"
tail -30 $o/tests.log
timeout 900 python3 -m pytest tests/test_gpu_fma.py tests/test_gpu_parity.py -q -m gpu -x -k "grouped or scalars or gershgorin or solvers_golden" > $o/tests2.log 2>&1; echo "tests2 rc=$?"
tail -5 $o/tests2.log
timeout 900 python3 -m pytest tests/test_gpu_scale.py -q -m gpu -x -k "relabelled or vs_reference or config2_vs" > $o/tests3.log 2>&1; echo "tests3 rc=$?"
tail -5 $o/tests3.log
for mf in 1 0; do
timeout 600 python3 bench.py --random 42 --steps 5 --warmup 2 --blocks 3 --no-cpu-baseline --no-wrp-check --set ghash_mfma=$mf > $o/random_mf$mf.json 2> $o/random.err; echo "random mf=$mf rc=$?"
python3 -c "import json;d=json.load(open('$o/random_mf$mf.json'));print(d['value'],d['ms_per_step'],d['roofline']['ms_per_launch'],d['config']['blocks_ms'])"
done
timeout 600 python3 bench.py --random 42 --steps 3 --warmup 2 --blocks 1 --no-cpu-baseline --no-wrp-check --set spgemm_variant=518 > /dev/null 2> $o/stamps_mf1.err; grep "ghash stamps" $o/stamps_mf1.err | tail -2
