timeout 900 python tools/bench_spgemm.py --variants 400,408,409 --reps 4 2>&1 | tail -4
