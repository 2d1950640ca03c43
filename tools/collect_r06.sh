#!/bin/bash
# Round-6 evidence: tools/collect_r06.sh <tag>  (everything under gpurun_out/<tag>*)
# bench + kernel statistics + PMC traffic of the FOUR bench workloads (headline, lattice, relabelled, the relabelled operand as it
# stands = the grouped LDS hash), BASELINE configs[3] (natural, relabelled, as it stands), the other configs, a roofline line per
# solver loop, the rank-share model, issue-side counters of the tile kernels, the L2 fragment microbenchmark.
tag=${1:-r06_v1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/collect_profiles.sh ${tag}
bash tools/collect_profiles.sh ${tag}_lattice --lattice 64
bash tools/collect_profiles.sh ${tag}_permute --permute 42
bash tools/collect_profiles.sh ${tag}_random --random 42
o=gpurun_out
timeout 900 python3 bench.py --config 3 --steps 5 --warmup 2 --blocks 3 > $o/${tag}_config3.json 2> $o/${tag}_config3.err
timeout 900 python3 bench.py --config 3 --permute 42 --steps 5 --warmup 2 --blocks 3 > $o/${tag}_config3_permute.json 2> $o/${tag}_config3_permute.err
timeout 900 python3 bench.py --config 3 --random 42 --steps 5 --warmup 2 --blocks 3 > $o/${tag}_config3_random.json 2> $o/${tag}_config3_random.err
timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $o/${tag}_bench_2ranks_shm.json 2> $o/${tag}_bench_2ranks_shm.err
timeout 1200 python3 tools/bench_configs.py --arithmetic fma > $o/${tag}_other_configs_fma.json 2> $o/${tag}_other_configs_fma.err
timeout 900 python3 tools/solver_roofline.py fma > $o/${tag}_solver_roofline.json 2> $o/${tag}_solver_roofline.err
timeout 900 python3 tools/rank_share.py > $o/${tag}_rank_share.json 2> $o/${tag}_rank_share.err
bash tools/prof_complex.sh ${tag}_complex > $o/${tag}_complex.log 2>&1
timeout 300 tools/micro/l2_fragment_bw > $o/${tag}_micro_l2_fragment_bw.txt 2>&1
echo "== done"; ls $o/${tag}*
# issue-side and cache counters: the headline's tile kernel, and the complex tile kernel of configs[4]
mkdir -p $o/${tag}_pmc
tools/pmc_bench_passes.sh ${tag}_pmc/tile "" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
  "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_WAVES GRBM_GUI_ACTIVE"
python3 tools/pmc_summary.py $o/${tag}_pmc/tile k_spgemm_tile > $o/${tag}_pmc_tile.txt
export CPLX=1 SOLVER=sign
tools/pmc_passes.sh ${tag}_pmc/tile_c "tools/solver_iterations.py" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
  "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_WAVES GRBM_GUI_ACTIVE"
python3 tools/pmc_summary.py $o/${tag}_pmc/tile_c k_spgemm_tile_c > $o/${tag}_pmc_tile_c.txt
unset CPLX SOLVER
tools/pmc_bench_passes.sh ${tag}_pmc/ghash "--random 42" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
python3 tools/pmc_summary.py $o/${tag}_pmc/ghash k_spgemm_ghash > $o/${tag}_pmc_ghash.txt
rm -rf $o/${tag}_pmc
echo "== pmc done"; cat $o/${tag}_pmc_tile.txt | head -30
