#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06l; mkdir -p $o
for t in 16 48; do
echo "OMP_NUM_THREADS=$t"
OMP_NUM_THREADS=$t timeout 600 python3 tools/time_slab_parts.py 2>&1 | grep oracle
done
nproc; cat /proc/cpuinfo | grep "model name" | head -1
