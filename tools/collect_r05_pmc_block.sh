#!/bin/bash
# issue-side and cache counters of the block kernel on the round-5 sources (bench.py --lattice 64), separate --pmc passes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05b
tools/pmc_bench_passes.sh r05b/pmc_block "--lattice 64" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
  "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_WAVES GRBM_GUI_ACTIVE"
python3 tools/pmc_summary.py gpurun_out/r05b/pmc_block k_bs_numeric > gpurun_out/r05_pmc_block_lattice.txt
rm -rf gpurun_out/r05b/pmc_block
cat gpurun_out/r05_pmc_block_lattice.txt
