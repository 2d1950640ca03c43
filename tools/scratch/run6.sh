timeout -s KILL 1200 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_rccl_single.py -m gpu -x -q > gpurun_out/run6.log 2>&1
tail -30 gpurun_out/run6.log
