#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06h; mkdir -p $o
timeout 600 python3 tools/check_triple.py > $o/check.log 2>&1; echo "check rc=$?"; tail -8 $o/check.log
for rep in 1 2; do
for tri in 0 1; do
  timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check --set tile_triple=$tri > $o/bench_tri$tri.json 2> $o/bench.err; echo "bench triple=$tri rc=$?"
  python3 -c "import json;d=json.load(open('$o/bench_tri$tri.json'));print('triple=$tri',d['value'],d['roofline']['ms_per_launch'],d['config']['blocks_ms'],d['config']['energy_end'])"
done
done
