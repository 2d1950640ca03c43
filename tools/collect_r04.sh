#!/bin/bash
# Round-4 evidence in one GPU-box session: tools/collect_r04.sh <tag>   (everything lands under gpurun_out/<tag>*)
tag=${1:-r04_v2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/collect_profiles.sh ${tag}
bash tools/collect_profiles.sh ${tag}_lattice --lattice 64
bash tools/collect_profiles.sh ${tag}_permute --permute 42
# PMC passes of the block kernel on the lattice bench (LDS is not used by it: wait fractions, L1 / L2 hit rates, instruction mix)
bash tools/pmc_bench_passes.sh ${tag}_pmc_block "--lattice 64" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" \
  "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_BRANCH"
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_block k_bs_numeric > gpurun_out/${tag}_pmc_block_summary.txt
rm -rf gpurun_out/${tag}_pmc_block
timeout 1200 python3 tools/bench_configs.py --arithmetic fma > gpurun_out/${tag}_other_configs_fma.json 2> gpurun_out/${tag}_other_configs_fma.err
echo "== done"; ls gpurun_out/${tag}*
