#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05b; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_tile2.py -x -q > $o/tile2_tests.log 2>&1; echo "tile2 tests rc=$?"; tail -12 $o/tile2_tests.log
timeout 600 python3 tools/t2_ablate.py
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_tile2.json 2> $o/bench_tile2.err; echo "bench rc=$?"
python3 - <<PY
import json
try:
    l=json.loads(open("$o/bench_tile2.json").read().strip().splitlines()[-1])
    print(l["value"], l["ms_per_step"], l["roofline"]["ms_per_launch"], l["config"]["blocks_ms"], l["fused_steps"], l["config"]["energy_end"], l.get("trs2_wrp_check"))
except Exception as e:
    print("ERR", e); print(open("$o/bench_tile2.err").read()[-2000:])
PY
