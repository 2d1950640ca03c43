#!/bin/bash
# Round-3 evidence in one GPU-box session: tools/collect_r03.sh <tag>   (everything lands under gpurun_out/<tag>*)
tag=${1:-r03_v4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/collect_profiles.sh ${tag}
bash tools/collect_profiles.sh ${tag}_unfused --arithmetic unfused
bash tools/collect_profiles.sh ${tag}_permute --permute 42
bash tools/pmc_bench_passes.sh ${tag}_pmc_tile "" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" \
  "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum"
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_tile k_spgemm_tile > gpurun_out/${tag}_pmc_tile_summary.txt
timeout 900 python3 tools/bench_configs.py --arithmetic fma > gpurun_out/${tag}_other_configs_fma.json 2> gpurun_out/${tag}_other_configs_fma.err
timeout 900 python3 tools/bench_configs.py --arithmetic unfused > gpurun_out/${tag}_other_configs_unfused.json 2> gpurun_out/${tag}_other_configs_unfused.err
timeout 600 python3 tools/rank_share.py > gpurun_out/${tag}_rank_share.json 2> gpurun_out/${tag}_rank_share.err
# the lattice workload: bench line and kernel statistics (long iterations: few of them)
mkdir -p gpurun_out/${tag}_lattice
timeout 900 python3 bench.py --lattice 64 --steps 4 --warmup 4 > gpurun_out/${tag}_lattice/bench.json 2> gpurun_out/${tag}_lattice/bench.err
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_lattice/stats -o run -- python3 bench.py --lattice 64 --steps 3 --warmup 3 --no-cpu-baseline --no-wrp-check > gpurun_out/${tag}_lattice/stats.log 2>&1
python3 tools/prof_summary.py gpurun_out/${tag}_lattice/stats/run_results.db > gpurun_out/${tag}_lattice/kernel_stats.csv
# TRS4 on the headline operand with the loop on compressed columns and in slab form: kernel statistics
for sa in 0 1; do
  NTPOLY_AMD_SLAB_ALGEBRA=$sa timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_trs4_sa$sa -o run -- python3 tools/solver_iterations.py > gpurun_out/${tag}_trs4_sa$sa.log 2>&1
  python3 tools/prof_summary.py gpurun_out/${tag}_trs4_sa$sa/run_results.db > gpurun_out/${tag}_trs4_sa${sa}_kernel_stats.csv
done
SOLVER=sign timeout 600 python3 tools/solver_iterations.py > gpurun_out/${tag}_sign_iterations.log 2>&1
timeout 900 python3 tools/solve_time.py fma > gpurun_out/${tag}_solve_time.txt 2>&1
echo "== done"; ls gpurun_out/${tag}*
