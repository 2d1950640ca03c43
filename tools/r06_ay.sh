#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06ay; mkdir -p $o
timeout 60 python3 bench.py --permute 42 --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/perm.json 2> $o/err; python3 -c "import json;d=json.load(open('$o/perm.json'));print('perm',d['value'],d['roofline']['ms_per_launch'],d['config']['energy_end'])"
timeout 40 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
