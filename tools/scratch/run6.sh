timeout -s KILL 2400 python -m pytest tests/ -m gpu -x -q --durations=8 > gpurun_out/run6.log 2>&1
tail -25 gpurun_out/run6.log
timeout -s KILL 600 python bench.py > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err
tail -c 600 gpurun_out/bench_now.json
