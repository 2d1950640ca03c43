#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05e; mkdir -p $o
timeout 1500 python3 -m pytest tests/test_gpu_multirank.py tests/test_gpu_multirank_big.py tests/test_gpu_complex_tile.py tests/test_gpu_defaults.py tests/test_gpu_extras.py -q -x --durations=8 > $o/sel.log 2>&1; echo "rc=$?"
tail -45 $o/sel.log
timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-wrp-check > $o/bench2.json 2> $o/bench2.err; python3 -c "
import json; l=json.loads(open('$o/bench2.json').read().strip().splitlines()[-1]); print('2 ranks shm', l['value'], l['ms_per_step'])"
