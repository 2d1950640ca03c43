#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06e; mkdir -p $o
timeout 600 python3 -m pytest tests/test_gpu_fma.py tests/test_gpu_rccl_single.py -q -m gpu -x -k "grouped or split" > $o/tests.log 2>&1; echo "tests rc=$?"
tail -15 $o/tests.log
for mf in 1 0; do
timeout 600 python3 bench.py --random 42 --steps 5 --warmup 2 --blocks 1 --no-cpu-baseline --no-wrp-check --set ghash_mfma=$mf > $o/random_mf$mf.json 2> $o/random.err; echo "random mf=$mf rc=$?"
python3 -c "import json;d=json.load(open('$o/random_mf$mf.json'));print(d['value'],d['roofline']['ms_per_launch'],d.get('grouped_hash'))"
timeout 600 python3 bench.py --random 42 --steps 3 --warmup 2 --blocks 1 --no-cpu-baseline --no-wrp-check --set ghash_mfma=$mf --set spgemm_variant=518 > /dev/null 2> $o/stamps_mf$mf.err; grep "ghash stamps" $o/stamps_mf$mf.err | tail -3
done
timeout 300 python3 bench.py --config 3 --random 42 --steps 3 --warmup 1 --blocks 1 > $o/c3_random.json 2> $o/c3_random.err; echo "c3 random rc=$?"
python3 -c "import json;d=json.load(open('$o/c3_random.json'));print(d['value'],d['ms_per_step'],d['roofline']['ms_per_launch'])"
