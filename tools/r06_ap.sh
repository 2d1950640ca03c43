#!/bin/bash
# functional lines of the several-rank bench on ONE GPU (ranks share it through the shared-memory transport), final sources
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06ap; mkdir -p $o
timeout 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $o/bench_2ranks_shm.json 2> $o/b2.err; echo "2 ranks rc=$?"
timeout 400 python3 bench.py --gpus 4 --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_4ranks_shm.json 2> $o/b4.err; echo "4 ranks rc=$?"
timeout 400 python3 bench.py --gpus 2 --permute 42 --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_2ranks_shm_permute.json 2> $o/b2p.err; echo "2 ranks permute rc=$?"
timeout 400 python3 bench.py --gpus 2 --config 3 --steps 5 --warmup 2 --blocks 3 --no-cpu-baseline > $o/config3_2ranks_shm.json 2> $o/c3.err; echo "config3 2 ranks rc=$?"
for f in bench_2ranks_shm bench_4ranks_shm bench_2ranks_shm_permute config3_2ranks_shm; do python3 -c "import json;d=json.load(open('$o/$f.json'));print('$f',d['value'],d['unit'],d['n_gpus'],d['ms_per_step'],d.get('solver_path_iters_per_s'))"; done
