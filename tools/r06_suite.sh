#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06suite; mkdir -p $o
timeout 2600 python3 -m pytest tests -q -m gpu --durations=45 > $o/suite.log 2>&1; echo "suite rc=$?"
tail -60 $o/suite.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $o/smoke.log
timeout 600 python3 bench.py --steps 20 --warmup 5 > $o/bench_with_traffic.json 2> $o/bench.err; echo "bench rc=$?"; python3 -c "import json;d=json.load(open('$o/bench_with_traffic.json'));print(d['value'],d['roofline'])"
