#!/bin/bash
# stamps of the tile kernel's prologue: base / records first / records first and no loads for the multiplier tile (ablation)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06x; mkdir -p $o
for v in st_nob; do
NTPOLY_AMD_LIB=$PWD/ntpoly_amd/libntpoly_amd_$v.so NTP_TILE_STAMPS_FILE=$PWD/$o/stamps.bin timeout 300 python3 bench.py --config 3 --n 262144 --halfband 157 --steps 5 --warmup 2 --blocks 1 --no-cpu-baseline > $o/bench_$v.json 2> $o/bench_$v.err; echo "$v rc=$?"
python3 tools/tile_stamps.py $o/stamps.bin 4 > $o/stamps_$v.txt 2>&1; tail -5 $o/stamps_$v.txt
python3 tools/tile_blocks.py $o/stamps.bin.blocks > $o/blocks_$v.txt 2>&1; tail -2 $o/blocks_$v.txt
rm -f $o/stamps.bin $o/stamps.bin.blocks
done
