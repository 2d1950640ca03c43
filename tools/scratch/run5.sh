timeout -s KILL 900 python -m pytest tests/test_gpu_extras.py -m gpu -x -q -k "jacobi" --durations=5 > gpurun_out/run4.log 2>&1
tail -25 gpurun_out/run4.log
