#!/bin/bash
# round 6: the tile kernel with 32-bit buffer offsets / constant LDS offsets -- bit-exactness tests, then A/B bench lines
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06b; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_fma.py tests/test_gpu_tile2.py tests/test_gpu_options.py -q -m gpu -x --durations=8 > $o/tests.log 2>&1; echo "tests rc=$?"
tail -15 $o/tests.log
for v in 1 0 1 0; do
  timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check --set tile_off32=$v > $o/bench_off$v.json 2> $o/bench.err; echo "bench off32=$v rc=$?"
  python3 -c "import json;d=json.load(open('$o/bench_off$v.json'));print(d['value'],d['roofline']['ms_per_launch'],d['config']['blocks_ms'])"
done
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check --permute 42 > $o/bench_perm.json 2> $o/bench.err; echo "perm rc=$?"
python3 -c "import json;d=json.load(open('$o/bench_perm.json'));print(d['value'],d['roofline']['ms_per_launch'],d['config']['blocks_ms'])"
timeout 300 python3 bench.py --config 3 --steps 5 --warmup 2 --blocks 3 > $o/c3.json 2> $o/bench.err; echo "c3 rc=$?"
python3 -c "import json;d=json.load(open('$o/c3.json'));print(d['value'],d['ms_per_step'],d['roofline']['ms_per_launch'])"
