#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06au; mkdir -p $o
timeout 900 python3 -m pytest tests/test_gpu_multirank.py tests/test_gpu_panel_sessions.py tests/test_gpu_fma.py tests/test_gpu_thin.py tests/test_gpu_defaults.py -q -m gpu -x > $o/tests.log 2>&1; echo "tests rc=$?"; tail -2 $o/tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $o/smoke.log
timeout 300 python3 bench.py > $o/bench_default.json 2> $o/bench.err; echo "bench rc=$?"; python3 -c "import json;d=json.load(open('$o/bench_default.json'));print(d['value'],d['roofline']['frac'],d['roofline']['traffic'],d['cpu_baseline']['value'],d['trs2_wrp_check'])"
