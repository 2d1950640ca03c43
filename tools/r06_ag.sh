#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06ag; mkdir -p $o
timeout 300 rocprofv3 --kernel-trace --stats -d $o/prof -o run -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --blocks 1 --no-cpu-baseline --no-wrp-check > $o/run.log 2>&1; echo "rc=$?"
timeout 100 python3 tools/tile_sequence.py $o/prof/run_results.db > $o/sequence.txt; wc -l $o/sequence.txt; rm -rf $o/prof
for nw in 4 8; do
  timeout 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check --tile-waves $nw > $o/bench_nw$nw.json 2> $o/bench.err
  python3 -c "import json;d=json.load(open('$o/bench_nw$nw.json'));print('nw$nw',d['value'],d['roofline']['ms_per_launch'])"
done
timeout 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/bench_auto.json 2> $o/bench.err
python3 -c "import json;d=json.load(open('$o/bench_auto.json'));print('auto',d['value'],d['roofline']['ms_per_launch'])"
