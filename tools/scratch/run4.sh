timeout -s KILL 900 python -m pytest tests/test_gpu_extras.py tests/test_gpu_parity.py -m gpu -x -q -k "remaining_solver or empty_columns" > gpurun_out/run4.log 2>&1
tail -15 gpurun_out/run4.log
