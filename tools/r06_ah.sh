#!/bin/bash
# eight waves per workgroup everywhere (NTPOLY_AMD_TILE_WAVES=8) against the launcher's choice, across the workloads
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06ah; mkdir -p $o
for tw in 0 8; do
  export NTPOLY_AMD_TILE_WAVES=$tw
  timeout 200 python3 bench.py --permute 42 --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/perm_$tw.json 2> $o/err; python3 -c "import json;d=json.load(open('$o/perm_$tw.json'));print('tw$tw perm',d['value'],d['roofline']['ms_per_launch'])"
  timeout 200 python3 bench.py --config 3 --steps 5 --warmup 2 --blocks 3 --no-cpu-baseline > $o/c3_$tw.json 2> $o/err; python3 -c "import json;d=json.load(open('$o/c3_$tw.json'));print('tw$tw c3',d['value'],d['ms_per_step'])"
  timeout 200 python3 bench.py --config 3 --n 65536 --halfband 50 --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline > $o/c1_$tw.json 2> $o/err; python3 -c "import json;d=json.load(open('$o/c1_$tw.json'));print('tw$tw c1',d['value'],d['ms_per_step'])"
  timeout 200 python3 bench.py --config 3 --n 262144 --halfband 30 --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline > $o/c30_$tw.json 2> $o/err; python3 -c "import json;d=json.load(open('$o/c30_$tw.json'));print('tw$tw h30',d['value'],d['ms_per_step'])"
  for s in trs4 sign isq; do
    SOLVER=$s timeout 150 python3 tools/solver_iterations.py > $o/${s}_$tw.log 2>&1; echo "tw$tw $s $(tail -1 $o/${s}_$tw.log | cut -c1-40)"
  done
  for s in sign isq; do
    CPLX=1 SOLVER=$s timeout 150 python3 tools/solver_iterations.py > $o/c${s}_$tw.log 2>&1; echo "tw$tw complex $s $(tail -1 $o/c${s}_$tw.log | cut -c1-40)"
  done
done
