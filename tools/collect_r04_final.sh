#!/bin/bash
# Round-4 evidence after the complex / thin-operand kernels: tools/collect_r04_final.sh <tag>  (everything under gpurun_out/<tag>*)
# bench + kernel statistics + PMC traffic of the three bench workloads, the other BASELINE configs, a roofline line per
# solver loop (real and complex) and the kernel statistics of the complex loops.
tag=${1:-r04_v7}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/collect_profiles.sh ${tag}
bash tools/collect_profiles.sh ${tag}_lattice --lattice 64
bash tools/collect_profiles.sh ${tag}_permute --permute 42
timeout 1200 python3 tools/bench_configs.py --arithmetic fma > gpurun_out/${tag}_other_configs_fma.json 2> gpurun_out/${tag}_other_configs_fma.err
timeout 900 python3 tools/solver_roofline.py fma > gpurun_out/${tag}_solver_roofline.json 2> gpurun_out/${tag}_solver_roofline.err
bash tools/prof_complex.sh ${tag}_complex > gpurun_out/${tag}_complex.log 2>&1
echo "== done"; ls gpurun_out/${tag}*
