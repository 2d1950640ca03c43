#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06ap; mkdir -p $o
NTPOLY_AMD_SHM_MB=1024 timeout 600 python3 bench.py --gpus 2 --permute 42 --steps 10 --warmup 5 --blocks 3 --no-cpu-baseline > $o/bench_2ranks_shm_permute.json 2> $o/b2p.err; echo "2 ranks permute rc=$?"
python3 -c "import json;d=json.load(open('$o/bench_2ranks_shm_permute.json'));print(d['value'],d.get('solver_path_iters_per_s'),d.get('solver_loop_iters_per_s'),d['trs2_wrp_check'])"
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_1.json 2> $o/b1.err; python3 -c "import json;d=json.load(open('$o/bench_1.json'));print(d['value'],d['trs2_wrp_check'])"
