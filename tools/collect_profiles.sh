#!/bin/bash
# Collects the judged profile artefacts of one version into gpurun_out/<tag>/ (run on the GPU box through gpurun):
#   tools/collect_profiles.sh <tag>
# 1. bench line (with cpu_baseline)   2. rocprofv3 --kernel-trace --stats of the same bench command
# 3. PMC passes FETCH_SIZE / WRITE_SIZE (separate runs, kernel trace only) of the same bench command
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
timeout 600 python3 bench.py --gpus 1 --steps 10 --warmup 2 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/$tag/stats -o run -- python3 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/$tag/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/$tag/pmc_$c -o run --output-format csv -- python3 bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/$tag/pmc_$c.log 2>&1
done
ls -R gpurun_out/$tag | head -30
