#!/bin/bash
# tile kernel: waves per workgroup (three workgroups per CU by LDS; 128 registers allow four waves per SIMD)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06v; mkdir -p $o
for rep in 1 2; do
for nw in 4 5 6 8; do
  timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check --tile-waves $nw > $o/bench_nw$nw.json 2> $o/bench.err; echo "bench nw=$nw rc=$?"
  python3 -c "import json;d=json.load(open('$o/bench_nw$nw.json'));print('nw$nw',d['value'],d['roofline']['ms_per_launch'],d['config']['blocks_ms'],d['config']['energy_end'])"
done
done
