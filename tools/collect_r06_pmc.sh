#!/bin/bash
# Round-6 evidence, second part: issue-side and cache counters (tools/collect_r06_pmc.sh <tag>)
tag=${1:-r06_v1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out
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
