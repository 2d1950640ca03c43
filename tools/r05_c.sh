#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05c; mkdir -p $o
for v in 0 1 2; do
NTPOLY_AMD_T2_ABLATE=$v timeout 600 python3 bench.py --steps 10 --warmup 5 --blocks 3 --no-cpu-baseline --no-wrp-check > $o/b_$v.json 2> $o/b_$v.err
python3 - <<PY
import json
try:
    l=json.loads(open("$o/b_$v.json").read().strip().splitlines()[-1])
    print("ablate $v", l["value"], l["ms_per_step"], l["roofline"]["ms_per_launch"], l["fused_steps"])
except Exception as e:
    print("ERR", e); print(open("$o/b_$v.err").read()[-1500:])
PY
done
