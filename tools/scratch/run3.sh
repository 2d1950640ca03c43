timeout 900 python tools/bench_spgemm.py --variants 400,411,412,413,414 --reps 5 2>&1 | tail -7
