#!/bin/bash
# the fingerprinted profiles of the four bench workloads + what the complex kernel's change moved (tools/collect_r06_final.sh <tag>)
tag=${1:-r06_v5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/collect_r06_profiles.sh ${tag}
o=gpurun_out
timeout 900 python3 tools/solver_roofline.py fma > $o/${tag}_solver_roofline.json 2> $o/${tag}_solver_roofline.err
bash tools/prof_complex.sh ${tag}_complex > $o/${tag}_complex.log 2>&1
mkdir -p $o/${tag}_pmc
export CPLX=1 SOLVER=sign
tools/pmc_passes.sh ${tag}_pmc/tile_c "tools/solver_iterations.py" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
  "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_WAVES GRBM_GUI_ACTIVE"
python3 tools/pmc_summary.py $o/${tag}_pmc/tile_c k_spgemm_tile_c > $o/${tag}_pmc_tile_c.txt
unset CPLX SOLVER
rm -rf $o/${tag}_pmc
timeout 600 python3 tools/bench_configs.py --arithmetic fma > $o/${tag}_other_configs_fma.json 2> $o/${tag}_other_configs_fma.err
echo "== done"
