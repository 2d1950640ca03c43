timeout 600 python tools/bench_spgemm.py --complex 1 --n 131072 --halfband 50 --variants 400,0 2>&1 | tail -6
timeout 600 python tools/bench_spgemm.py --complex 1 --n 131072 --halfband 100 --variants 400,0 --iters 4 2>&1 | tail -4
