#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06p; mkdir -p $o
NTPOLY_AMD_LIB=$PWD/ntpoly_amd/libntpoly_amd_vstamps.so NTP_TILE_STAMPS_FILE=$PWD/$o/stamps.bin timeout 300 python3 bench.py --steps 12 --warmup 5 --blocks 1 --no-cpu-baseline --no-wrp-check > $o/bench.json 2> $o/bench.err; echo "rc=$?"
python3 tools/tile_stamps.py $o/stamps.bin 4 > $o/stamps.txt 2>&1; tail -6 $o/stamps.txt; head -6 $o/stamps.txt | cut -c1-400
python3 tools/tile_blocks.py $o/stamps.bin.blocks > $o/blocks.txt 2>&1; head -3 $o/blocks.txt; tail -2 $o/blocks.txt
rm -f $o/stamps.bin $o/stamps.bin.blocks
