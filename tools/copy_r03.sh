#!/bin/bash
# copies the judged artefacts of tools/collect_r03.sh <tag> from gpurun_out/ (scratch) into profiles/ (tracked)
tag=${1:-r03_v4}
cd "$(dirname "$0")/.."
for v in "" _unfused _permute; do
  d=gpurun_out/${tag}${v}
  [ -s $d/bench.json ] && tail -1 $d/bench.json > profiles/${tag}${v}_bench.json
  [ -s $d/kernel_stats.csv ] && cp $d/kernel_stats.csv profiles/${tag}${v}_kernel_stats.csv
  [ -s $d/pmc_traffic.json ] && cp $d/pmc_traffic.json profiles/${tag}${v}_pmc_traffic.json && cp $d/pmc_traffic.json profiles/r03_pmc_traffic${v}.json
done
[ -s gpurun_out/${tag}_pmc_tile_summary.txt ] && cp gpurun_out/${tag}_pmc_tile_summary.txt profiles/r03_pmc_tile_summary.txt
for a in fma unfused; do [ -s gpurun_out/${tag}_other_configs_$a.json ] && cp gpurun_out/${tag}_other_configs_$a.json profiles/r03_other_configs_$a.json; done
[ -s gpurun_out/${tag}_rank_share.json ] && cp gpurun_out/${tag}_rank_share.json profiles/r03_rank_share.json
[ -s gpurun_out/${tag}_lattice/bench.json ] && tail -1 gpurun_out/${tag}_lattice/bench.json > profiles/${tag}_lattice_bench.json
[ -s gpurun_out/${tag}_lattice/kernel_stats.csv ] && cp gpurun_out/${tag}_lattice/kernel_stats.csv profiles/${tag}_lattice_kernel_stats.csv
[ -s gpurun_out/${tag}_trs4_sa0_kernel_stats.csv ] && cp gpurun_out/${tag}_trs4_sa0_kernel_stats.csv profiles/r03_trs4_kernel_stats_before.csv
[ -s gpurun_out/${tag}_trs4_sa1_kernel_stats.csv ] && cp gpurun_out/${tag}_trs4_sa1_kernel_stats.csv profiles/r03_trs4_kernel_stats_slab_algebra.csv
[ -s gpurun_out/${tag}_solve_time.txt ] && grep -v "^W2\|^$" gpurun_out/${tag}_solve_time.txt > profiles/r03_solve_time.txt
{ grep "ms per iteration" gpurun_out/${tag}_trs4_sa0.log gpurun_out/${tag}_trs4_sa1.log gpurun_out/${tag}_sign_iterations.log 2>/dev/null; } > profiles/r03_solver_iterations.txt
{ echo "== tools/micro/mfma_f64_probe.hip"; cat gpurun_out/mfma_probe.log; echo; echo "== tools/micro/mfma_layout.hip"; cat gpurun_out/mfma_layout.log; echo; echo "== tools/micro/mfma_f64_valu.hip"; cat gpurun_out/mfma_f64_valu.log; } > profiles/r03_micro_mfma_f64.txt 2>/dev/null
ls -la profiles/ | grep r03
