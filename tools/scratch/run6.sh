timeout -s KILL 900 python -m pytest tests/test_gpu_extras.py -m gpu -x -q -k "reference_data_fixtures" > gpurun_out/run6.log 2>&1
tail -30 gpurun_out/run6.log
