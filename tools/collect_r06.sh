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
echo "== done"; ls $o/${tag}*
