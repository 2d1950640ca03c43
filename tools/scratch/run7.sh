timeout -s KILL 2400 python -m pytest tests/ -m gpu -x -q > gpurun_out/run7.log 2>&1
tail -5 gpurun_out/run7.log
bash tools/collect_profiles.sh r01_v10 > gpurun_out/collect.log 2>&1
tail -c 400 gpurun_out/r01_v10/bench.json
# complex-kernel profile: one product per variant under rocprof
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r01_v10/stats_cplx -o run -- python3 tools/bench_spgemm.py --complex 1 --n 131072 --halfband 50 --variants 400,0 --reps 3 > gpurun_out/r01_v10/cplx.log 2>&1
tail -3 gpurun_out/r01_v10/cplx.log
