#!/bin/bash
# eight waves per workgroup where fewer than four workgroups of four fit: against the library of an earlier commit
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06ai; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_fma.py tests/test_gpu_slab_algebra.py -q -m gpu -x > $o/tests.log 2>&1; echo "tests rc=$?"; tail -1 $o/tests.log
for rep in 1 2; do
  timeout 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/bench.json 2> $o/err; python3 -c "import json;d=json.load(open('$o/bench.json'));print('headline',d['value'],d['roofline']['ms_per_launch'],repr(d['config']['energy_end']))"
done
timeout 200 python3 bench.py --permute 42 --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/perm.json 2> $o/err; python3 -c "import json;d=json.load(open('$o/perm.json'));print('perm',d['value'],d['roofline']['ms_per_launch'])"
timeout 200 python3 bench.py --config 3 --steps 5 --warmup 2 --blocks 3 --no-cpu-baseline > $o/c3.json 2> $o/err; python3 -c "import json;d=json.load(open('$o/c3.json'));print('c3',d['value'],d['ms_per_step'])"
timeout 200 python3 bench.py --config 3 --n 65536 --halfband 50 --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline > $o/c1.json 2> $o/err; python3 -c "import json;d=json.load(open('$o/c1.json'));print('c1',d['value'],d['ms_per_step'])"
for s in trs4 sign isq; do
  SOLVER=$s timeout 150 python3 tools/solver_iterations.py > $o/${s}.log 2>&1; echo "$s $(tail -1 $o/${s}.log | cut -c1-40)"
done
