#!/bin/bash
# tile kernel: multiplier tile as pairs of rows through a buffer resource (option tile_bbuf = 2 / 1 / 0)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06y; mkdir -p $o
timeout 1200 python3 -m pytest tests/test_gpu_fma.py tests/test_gpu_slab_algebra.py tests/test_gpu_panel_sessions.py tests/test_gpu_thin.py -q -m gpu -x > $o/tests.log 2>&1; echo "tests rc=$?"; tail -3 $o/tests.log
for rep in 1; do
for v in 2 1 0; do
  timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check --set tile_bbuf=$v > $o/bench_$v.json 2> $o/bench.err; echo "bench bbuf=$v rc=$?"
  python3 -c "import json;d=json.load(open('$o/bench_$v.json'));print('bbuf$v',d['value'],d['roofline']['ms_per_launch'],d['config']['blocks_ms'],d['config']['energy_end'],d.get('spgemm_products_per_s'))"
done
done
v=st_bbuf
NTPOLY_AMD_LIB=$PWD/ntpoly_amd/libntpoly_amd_$v.so NTP_TILE_STAMPS_FILE=$PWD/$o/stamps.bin timeout 300 python3 bench.py --steps 12 --warmup 5 --blocks 1 --no-cpu-baseline --no-wrp-check > $o/bench_$v.json 2> $o/bench_$v.err; echo "$v rc=$?"
python3 tools/tile_stamps.py $o/stamps.bin 4 > $o/stamps_$v.txt 2>&1; tail -5 $o/stamps_$v.txt
rm -f $o/stamps.bin $o/stamps.bin.blocks
