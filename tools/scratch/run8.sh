timeout -s KILL 900 python tools/bench_configs.py > gpurun_out/configs.json 2> gpurun_out/configs.err
tail -c 3500 gpurun_out/configs.json; tail -3 gpurun_out/configs.err
