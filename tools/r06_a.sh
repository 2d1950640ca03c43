#!/bin/bash
# round 6, first GPU call: the new configs[3] tests, the changed multi-rank tests, a baseline bench line and the unstructured workload
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06a; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_config3.py tests/test_gpu_multirank_big.py tests/test_gpu_multirank.py tests/test_gpu_tile2.py -q -m gpu --durations=15 > $o/tests.log 2>&1; echo "tests rc=$?"
tail -25 $o/tests.log
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"
python3 -c "import json;d=json.load(open('$o/bench.json'));print(d['value'],d['roofline']['ms_per_launch'],d['config']['blocks_ms'])"
timeout 600 python3 bench.py --random 42 --steps 5 --warmup 2 --blocks 1 --no-cpu-baseline --no-wrp-check > $o/random.json 2> $o/random.err; echo "random rc=$?"
python3 -c "import json;d=json.load(open('$o/random.json'));print(d['value'],d['roofline'])"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_random -- python3 bench.py --random 42 --steps 3 --warmup 1 --blocks 1 --no-cpu-baseline --no-wrp-check > $o/random_prof.json 2> $o/random_prof.err
f=$(find $o/prof_random -name '*kernel_stats.csv' | head -1); head -12 "$f"
timeout 300 python3 bench.py --config 3 --random 42 --steps 3 --warmup 1 --blocks 1 > $o/c3_random.json 2> $o/c3_random.err; echo "c3 random rc=$?"
python3 -c "import json;d=json.load(open('$o/c3_random.json'));print(d['value'],d['ms_per_step'],d['roofline'])"
