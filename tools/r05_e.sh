#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05e; mkdir -p $o
timeout 1500 python3 -m pytest tests/test_gpu_multirank_big.py tests/test_gpu_multirank.py tests/test_gpu_defaults.py -q -x --durations=8 > $o/sel.log 2>&1; echo "rc=$?"
tail -40 $o/sel.log
NTPOLY_AMD_SHM_MB=256 timeout 900 python3 bench.py --gpus 2 --permute 42 --steps 10 --warmup 5 --blocks 3 > $o/bench2_perm.json 2> $o/bench2_perm.err; echo "bench2 perm rc=$?"; tail -c 600 $o/bench2_perm.json; tail -5 $o/bench2_perm.err
