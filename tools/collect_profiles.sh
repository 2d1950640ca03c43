#!/bin/bash
# Collects the judged profile artefacts of one version into gpurun_out/<tag>/ (run on the GPU box through gpurun):
#   tools/collect_profiles.sh <tag> [extra bench.py arguments, e.g. --permute 42]
# 1. bench line (with cpu_baseline)   2. rocprofv3 --kernel-trace --stats of the same bench command
# 3. PMC passes FETCH_SIZE / WRITE_SIZE (separate runs, kernel trace only) of the same bench command
# 4. the traffic summary of the dominant kernel (tools/pmc_traffic_json.py), with the fingerprint of the kernel sources
tag=$1; shift
extra="$@"
kern=k_spgemm_tile,k_spgemm_slab,k_spgemm_ghash,k_bs_numeric   # (the dominant one of the run is picked)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
timeout 900 python3 bench.py --gpus 1 --steps 10 --warmup 3 $extra > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/$tag/stats -o run -- python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-wrp-check $extra > gpurun_out/$tag/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/$tag/pmc_$c -o run --output-format csv -- python3 bench.py --gpus 1 --steps 4 --warmup 3 --no-cpu-baseline --no-wrp-check $extra > gpurun_out/$tag/pmc_$c.log 2>&1
done
python3 tools/prof_summary.py gpurun_out/$tag/stats/run_results.db > gpurun_out/$tag/kernel_stats.csv
python3 tools/pmc_traffic_json.py gpurun_out/$tag $kern gpurun_out/$tag/pmc_traffic.json
# (gpurun merges at most 64 MiB back: the raw traces stay on the box)
rm -rf gpurun_out/$tag/stats gpurun_out/$tag/pmc_FETCH_SIZE gpurun_out/$tag/pmc_WRITE_SIZE
head -12 gpurun_out/$tag/kernel_stats.csv
