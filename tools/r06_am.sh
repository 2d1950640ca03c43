#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06am; mkdir -p $o
timeout 300 python3 -m pytest tests/test_gpu_thin.py -q -m gpu -x > $o/tests.log 2>&1; echo "tests rc=$?"; tail -2 $o/tests.log
v=st
for steps in 20 21; do
NTPOLY_AMD_LIB=$PWD/ntpoly_amd/libntpoly_amd_$v.so NTP_TILE_STAMPS_FILE=$PWD/$o/stamps.bin timeout 300 python3 bench.py --steps $steps --warmup 5 --blocks 1 --no-cpu-baseline --no-wrp-check > $o/bench_$v.json 2> $o/bench_$v.err; echo "$v rc=$?"
python3 tools/tile_stamps.py $o/stamps.bin 8 > $o/stamps_$steps.txt 2>&1; tail -5 $o/stamps_$steps.txt
python3 tools/tile_blocks.py $o/stamps.bin.blocks > $o/blocks_$steps.txt 2>&1; head -1 $o/blocks_$steps.txt
rm -f $o/stamps.bin $o/stamps.bin.blocks
done
